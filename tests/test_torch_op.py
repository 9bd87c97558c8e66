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
