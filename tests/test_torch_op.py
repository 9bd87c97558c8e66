"""The layer call registered as torch.library operators (egc_amd/ops.py): schema, HIP-only dispatch
(no CPU kernel: a CPU tensor fails loudly), and -- on the GPU -- equality with the direct path, autograd through the
registered backward, torch.library.opcheck and torch.compile."""
import numpy as np
import pytest
import torch

import egc_amd
from egc_amd import ops


def test_operators_are_registered_with_their_schemas():
    assert hasattr(torch.ops.egc_amd, "layer_forward") and hasattr(torch.ops.egc_amd, "layer_forward_train")
    assert hasattr(torch.ops.egc_amd, "layer_backward")
    s = str(torch.ops.egc_amd.layer_forward.default._schema)
    assert "Tensor x" in s and "Int graph_handle" in s and "-> Tensor" in s


def test_no_cpu_kernel():
    conv = egc_amd.EGConv(8, 8, aggrs=["sum"], num_heads=2, num_bases=2)
    wcat, bcat = conv._packed_weights()
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.egc_amd.layer_forward(torch.randn(4, 8), wcat, bcat, conv.bias, 0, ops.handle_of(conv._spec_coo))


def test_stale_handle_is_an_error():
    with pytest.raises(RuntimeError, match="stale"):
        ops._get(123456789)


def test_handles_are_never_reused():
    """A handle comes from a counter, not from id(): a freed object's handle stays stale even when a new object
    takes over its address."""
    import gc
    seen = set()
    for _ in range(50):
        conv = egc_amd.EGConv(8, 8, aggrs=["sum"], num_heads=2, num_bases=2)
        h = ops.handle_of(conv._spec_coo)
        assert h == ops.handle_of(conv._spec_coo)       # one handle per live object
        assert h not in seen
        seen.add(h)
        del conv
        gc.collect()
    for h in seen:
        with pytest.raises(RuntimeError, match="stale"):
            ops._get(h)


@pytest.mark.gpu
def test_op_equals_direct_path_forward_and_backward(monkeypatch):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n, e = 700, 6000
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e))).to(dev)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(64, 64, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev)
    x = torch.randn(n, 64, device=dev, requires_grad=True)
    gout = torch.randn(n, 64, device=dev)
    out_a = conv(x, ei)
    out_a.backward(gout)
    ga = [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    x.grad = None
    conv.zero_grad()
    monkeypatch.setenv("EGC_USE_TORCH_OP", "1")
    out_b = conv(x, ei)
    out_b.backward(gout)
    gb = [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    assert torch.equal(out_a, out_b)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=0, atol=1e-5 * max(1.0, float(a.abs().max())))   # float atomics in the source kernel
    with torch.no_grad():
        assert torch.equal(conv(x, ei), out_a.detach())


@pytest.mark.gpu
def test_opcheck_and_compile():
    dev = torch.device("cuda:0")
    n, e = 300, 2500
    ei = torch.randint(0, n, (2, e), device=dev)
    conv = egc_amd.EGConv(32, 32, aggrs=["symnorm", "max"], num_heads=4, num_bases=4).to(dev).eval()
    x = torch.randn(n, 32, device=dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    wcat, bcat = conv._packed_weights()
    args = (x, wcat, bcat, conv.bias.detach(), ops.handle_of(g), ops.handle_of(conv._spec_coo))
    torch.library.opcheck(torch.ops.egc_amd.layer_forward.default, args, test_utils=("test_schema", "test_faketensor"))
    with torch.no_grad():
        want = conv(x, g)

        def f(x):
            return torch.ops.egc_amd.layer_forward(x * 1.0, wcat, bcat, conv.bias, args[4], args[5]) + 0.0
        got = torch.compile(f, backend="eager")(x)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("aggrs", [["symnorm", "max"], ["sum", "mean"], ["min", "std"]])
def test_opcheck_training_operators_and_compiled_training_step(aggrs):
    """The fake (meta) implementations of the training operators return the shapes of the real ones -- the statistics
    width of the library, arg tables only for layers with max / min -- so AOT autograd / inductor plan the saved
    tensors correctly; a training step compiled with the inductor-free AOT backend equals the eager one."""
    dev = torch.device("cuda:0")
    n, e = 300, 2500
    torch.manual_seed(0)
    ei = torch.randint(0, n, (2, e), device=dev)
    conv = egc_amd.EGConv(32, 32, aggrs=aggrs, num_heads=4, num_bases=4).to(dev)
    x = torch.randn(n, 32, device=dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    wcat, bcat = conv._packed_weights()
    wcat, bcat, bias = wcat.detach(), bcat.detach(), conv.bias.detach()
    gh, sh = ops.handle_of(g), ops.handle_of(conv._spec_coo)
    args = (x, wcat, bcat, bias, gh, sh)
    torch.library.opcheck(torch.ops.egc_amd.layer_forward_train.default, args,
                          test_utils=("test_schema", "test_faketensor"))
    out, bases, wts, stats, cnt, amax, amin = torch.ops.egc_amd.layer_forward_train(*args)
    gout = torch.randn_like(out)
    torch.library.opcheck(torch.ops.egc_amd.layer_backward.default,
                          (gout, x, wcat, bases, wts, stats, cnt, amax, amin, gh, sh, True, True),
                          test_utils=("test_schema", "test_faketensor"))

    def step(x, wcat, bcat, bias):
        return torch.ops.egc_amd.layer_forward_train(x, wcat, bcat, bias, gh, sh)[0]
    leaves = [t.clone().requires_grad_(True) for t in (x, wcat, bcat, bias)]
    step(*leaves).backward(gout)
    want = [t.grad.clone() for t in leaves]
    leaves2 = [t.clone().requires_grad_(True) for t in (x, wcat, bcat, bias)]
    torch.compile(step, backend="aot_eager")(*leaves2).backward(gout)
    for a, b in zip(want, [t.grad for t in leaves2]):
        assert torch.allclose(a, b, rtol=0, atol=1e-5 * max(1.0, float(a.abs().max())))
