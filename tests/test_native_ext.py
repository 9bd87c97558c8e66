"""The compiled PyTorch binding over the C ABI (egc_amd/csrc_ext/egc_torch_ext.cpp; north_star: "exposed as a PyTorch-ROCm
C++/HIP extension"): TORCH_LIBRARY operators for HIP devices only, used by the layer modules' inference forward."""
import numpy as np
import pytest
import torch

import egc_amd
from egc_amd import _native


def test_extension_is_built_and_registers_its_operators():
    nat = _native.ops()
    assert nat is not None, "egc_amd/lib/libegc_torch_ext.so is missing: run __graft_entry__.build()"
    s = str(torch.ops.egc_amd_native.layer_forward.default._schema)
    assert "Tensor x" in s and "int graph" in s and "-> Tensor" in s
    assert hasattr(torch.ops.egc_amd_native, "layer_forward_post")


def test_no_cpu_kernel_in_the_extension():
    _native.ops()
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.egc_amd_native.layer_forward(torch.randn(4, 8), torch.zeros(16, dtype=torch.uint8), None, None, 0, 0,
                                               torch.zeros(1, dtype=torch.uint8), 0, 8, 8, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,aggrs", [("opt", ["sum", "mean", "max", "symnorm"]), ("lay", ["symadd", "std", "max"])])
def test_native_binding_equals_the_ctypes_path_bit_for_bit(kind, aggrs, monkeypatch):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n, e = 900, 9000
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e))).to(dev)
    ei[1, :300] = 7
    torch.manual_seed(0)
    if kind == "opt":
        conv = egc_amd.EGConv(128, 128, aggrs=aggrs, num_heads=8, num_bases=4).to(dev).eval()
    else:
        conv = egc_amd.EfficientGraphConv(128, 128, 4, 4, False, aggrs=aggrs).to(dev).eval()
    bn = torch.nn.BatchNorm1d(128).to(dev).eval()
    with torch.no_grad():
        conv.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(n, 128, device=dev)
    block = egc_amd.FusedEGCBlock(conv, bn).eval()
    with torch.no_grad():
        a = conv(x, ei) if kind == "opt" else conv(x=x, edge_index=ei)
        ap = block(x, ei)
    monkeypatch.setenv("EGC_NO_NATIVE_EXT", "1")
    monkeypatch.setattr(_native, "_TRIED", False)
    monkeypatch.setattr(_native, "_OPS", None)
    with torch.no_grad():
        b = conv(x, ei) if kind == "opt" else conv(x=x, edge_index=ei)
        bp = block(x, ei)
    assert _native.ops() is None
    assert torch.equal(a, b) and torch.equal(ap, bp)

def _same_gradient(name, a, b, go):
    """Bit for bit -- except the bias of a conv in front of a BatchNorm on batch statistics: its gradient is the column sum of the
    BatchNorm's dh, zero but for rounding.  The Python Functions add dh up in float32 (as autograd does); the block node takes
    coef_g sum g + coef_h sum h + n coef_1 from the sums its BatchNorm step already holds (egc_bn_backward_stats_sums_f32: held
    against the float64 sum of the actual dh in tests/test_callers.py).  Both are noise around zero: the same to within it."""
    if name.endswith("conv.bias"):
        noise = 4e-6 * go.size(0) ** 0.5 * float(go.abs().max())
        assert float(a.abs().max()) <= noise and float(b.abs().max()) <= noise, (name, float(a.abs().max()), float(b.abs().max()), noise)
    else:
        assert torch.equal(a, b), name


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["opt", "lay"])
@pytest.mark.parametrize("residual", [True, False])
def test_batch_block_train_node_equals_the_python_functions(kind, residual, monkeypatch):
    """The reference's block x -> x + relu(bn(conv(x))) on a GraphBatch in training as ONE autograd node of the compiled binding
    (csrc_ext: batch_block_train) against the Python Functions (EGC_NO_NATIVE_TRAIN=1): the same library calls in the same
    order on deterministic kernels -- outputs, every gradient and BatchNorm's running statistics bit for bit."""
    import egc_amd
    from egc_amd import _native
    from test_batch_tile_gpu import _messy_batch
    assert _native.ops() is not None and hasattr(_native.ops(), "batch_block_train")
    dev = torch.device("cuda:0")
    ei, n, ptr = _messy_batch(21, n_graphs=200, max_size=70)
    x0, go = torch.randn(n, 128), torch.randn(n, 128)

    def build():
        torch.manual_seed(5)
        if kind == "opt":
            conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4)
        else:
            conv = egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=["symadd", "max", "mean"])
        blocks = torch.nn.ModuleList([egc_amd.FusedEGCBlock(conv, torch.nn.BatchNorm1d(128), residual=residual),
                                      egc_amd.FusedEGCBlock(egc_amd.EGConv(128, 128, aggrs=["sum", "max"], num_heads=8, num_bases=4),
                                                            torch.nn.BatchNorm1d(128, momentum=None), residual=residual)])
        return blocks.to(dev).train()
    res = {}
    for mode in ("native", "python"):
        if mode == "python":
            monkeypatch.setenv("EGC_NO_NATIVE_TRAIN", "1")
        blocks = build()
        for step in range(2):           # (two steps: the running statistics and num_batches_tracked move twice)
            for p in blocks.parameters():
                p.grad = None
            x = x0.to(dev).requires_grad_(True)
            gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=70)
            h = x
            for b in blocks:
                h = b(h, gb)
            h.backward(go.to(dev))
            gb.check()
        node = h.grad_fn.name() if h.grad_fn is not None else ""
        res[mode] = (h.detach().clone(), x.grad.clone(), [(k, p.grad.clone()) for k, p in blocks.named_parameters()],
                     [b.clone() for b in blocks.buffers() if b.dtype != torch.int32], node)
    assert "BatchBlockTrainFn" in res["native"][4] and "BatchBlockTrainFn" not in res["python"][4], (res["native"][4], res["python"][4])
    assert torch.equal(res["native"][0], res["python"][0])
    assert torch.equal(res["native"][1], res["python"][1])
    for (k, a), (_, b) in zip(res["native"][2], res["python"][2]):
        _same_gradient(k, a, b, go)
    for a, b in zip(res["native"][3], res["python"][3]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_batch_conv_train_node_outside_a_block(monkeypatch):
    """conv(x, GraphBatch) in training outside a FusedEGCBlock: the same node without the tail."""
    import egc_amd
    from test_batch_tile_gpu import _messy_batch
    dev = torch.device("cuda:0")
    ei, n, ptr = _messy_batch(22, n_graphs=150, max_size=70)
    torch.manual_seed(7)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).train()
    x0, go = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
    res = {}
    for mode in ("native", "python"):
        if mode == "python":
            monkeypatch.setenv("EGC_NO_NATIVE_TRAIN", "1")
        conv.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=70)
        out = conv(x, gb)
        out.backward(go)
        gb.check()
        res[mode] = (out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in conv.parameters()], out.grad_fn.name())
    assert "BatchBlockTrainFn" in res["native"][3] and "BatchBlockTrainFn" not in res["python"][3]
    assert torch.equal(res["native"][0], res["python"][0]) and torch.equal(res["native"][1], res["python"][1])
    for a, b in zip(res["native"][2], res["python"][2]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs", [(168, 8, 4, ["symadd"]), (224, 4, 4, ["add", "mean", "max"]), (296, 8, 4, ["symadd"]),
                                             (300, 4, 4, ["symadd", "min", "max"]), (128, 8, 4, ["symadd", "max", "mean"])])
def test_csr_path_training_through_the_binding_at_the_reference_widths(hidden, H, B, aggrs, monkeypatch):
    """The reference's own batched nets (run_pretrained.sh:7,12,23,24,48: 168 / 224 / 296 / 300 wide EfficientGraphConv layers) train on the
    CSR path; since round 6 its compiled calls (csrc_ext: train_forward / train_backward) take those shapes too -- the dense
    gradients through the general sequence instead of the one-pass kernel.  Same library calls in the same order as the Python
    path (EGC_NO_NATIVE_TRAIN=1): outputs and every gradient bit for bit."""
    from egc_amd import functional as F
    from test_batch_tile_gpu import _messy_batch
    dev = torch.device("cuda:0")
    ei, n, ptr = _messy_batch(hidden, n_graphs=120, max_size=60)
    torch.manual_seed(hidden)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs).to(dev).train()
    x0, go = torch.randn(n, hidden, device=dev), torch.randn(n, hidden, device=dev)
    seen = []
    real = F._native_train_ops
    monkeypatch.setattr(F, "_native_train_ops", lambda *a, **k: (seen.append(real(*a, **k) is not None), real(*a, **k))[1])
    res = {}
    for mode in ("native", "python"):
        if mode == "python":
            monkeypatch.setenv("EGC_NO_NATIVE_TRAIN", "1")
        conv.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = conv(x=x, edge_index=ei.to(dev))
        out.backward(go)
        res[mode] = (out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in conv.parameters()])
    assert seen == [True, False], seen
    assert torch.equal(res["native"][0], res["python"][0]) and torch.equal(res["native"][1], res["python"][1])
    for a, b in zip(res["native"][2], res["python"][2]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs,as_batch", [(168, 8, 4, ["symadd"], True), (224, 4, 4, ["add", "mean", "max"], True),
                                                      (296, 8, 4, ["symadd"], False), (128, 8, 4, ["symadd", "max", "mean"], False)])
def test_csr_block_train_node_equals_the_python_functions(hidden, H, B, aggrs, as_batch, monkeypatch):
    """x -> x + relu(bn(conv(x))) in training OUTSIDE the one-launch envelope (the reference's wide batched nets handed over as a
    GraphBatch; a plain edge_index at any width) as ONE autograd node of the compiled binding (csr_block_train) against the Python
    Functions: outputs, every gradient and BatchNorm's running statistics bit for bit, over two steps."""
    from test_batch_tile_gpu import _messy_batch
    dev = torch.device("cuda:0")
    ei, n, ptr = _messy_batch(hidden + 1, n_graphs=150, max_size=60)
    x0, go = torch.randn(n, hidden), torch.randn(n, hidden)

    def build():
        torch.manual_seed(11)
        return torch.nn.ModuleList([egc_amd.FusedEGCBlock(egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs),
                                                          torch.nn.BatchNorm1d(hidden)) for _ in range(2)]).to(dev).train()
    res = {}
    for mode in ("native", "python"):
        if mode == "python":
            monkeypatch.setenv("EGC_NO_NATIVE_TRAIN", "1")
        blocks = build()
        for step in range(2):
            for p in blocks.parameters():
                p.grad = None
            x = x0.to(dev).requires_grad_(True)
            g = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=60) if as_batch else ei.to(dev)
            h = x
            for b in blocks:
                h = b(h, g)
            h.backward(go.to(dev))
        res[mode] = (h.detach().clone(), x.grad.clone(), [(k, p.grad.clone()) for k, p in blocks.named_parameters()],
                     [b.clone() for b in blocks.buffers() if b.dtype != torch.int32], h.grad_fn.name())
    assert "CsrBlockTrainFn" in res["native"][4] and "BlockTrainFn" not in res["python"][4], (res["native"][4], res["python"][4])
    assert torch.equal(res["native"][0], res["python"][0]) and torch.equal(res["native"][1], res["python"][1])
    for (k, a), (_, b) in zip(res["native"][2], res["python"][2]):
        _same_gradient(k, a, b, go)
    for a, b in zip(res["native"][3], res["python"][3]):
        assert torch.equal(a, b)
