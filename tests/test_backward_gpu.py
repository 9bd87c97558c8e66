"""GPU tests of the backward of the fused layer: gradients from the HIP kernels (through autograd and the
C ABI) against float64 autograd through the differentiable CPU restatement (oracle/egc_torch_ref.py)."""
import os
import numpy as np
import pytest
import torch

from oracle import egc_torch_ref as tref

pytestmark = pytest.mark.gpu

def gtol(aggrs):
    """Bound on |HIP gradient - float64 gradient|, scale-relative: north_star's 1e-5 for every layer without std / var.
    With std / var the gradient carries 1 / (2 std) (158 at zero variance): the REFERENCE's own float32 run is 4.5e-5 to
    7.6e-5 away from its float64 run on the committed fixtures (tests/golden/MANIFEST_GRAD.json, f32_vs_f64_grad_x),
    so nothing evaluated in float32 can be held to 1e-5 there (tests/test_backward_golden.py applies a per-fixture
    bound derived from that distance)."""
    return 2e-4 if {"std", "var"} & set(aggrs) else 1e-5


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def _graph(rng, n, e, hub=None, self_loops=0):
    ei = rng.integers(0, n, size=(2, e))
    if hub is not None:
        ei[1, :hub] = 1
    if self_loops:
        s = rng.integers(0, n, size=self_loops)
        ei = np.concatenate([ei, np.stack([s, s])], axis=1)
    return ei[:, rng.permutation(ei.shape[1])].astype(np.int64)


@pytest.mark.parametrize("aggrs,kw", [
    (["sum", "mean", "max", "symnorm"], {}),
    (["symnorm"], {}),
    (["min", "std", "var"], {}),
    (["sum", "max", "std"], {"add_self_loops": False}),
    (["symnorm", "mean"], {"sigmoid": True}),
])
def test_egconv_gradients(aggrs, kw):
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(len(aggrs) * 7 + len(kw))
    n, fin, fout, H, B = 300, 48, 64, 8, 4
    ei = _graph(rng, n, 2500, hub=200, self_loops=12)
    torch.manual_seed(1)
    conv = egc_amd.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B, **kw).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, fin, device=dev, requires_grad=True)
    gout = torch.randn(n, fout, device=dev)
    out = conv(x, torch.from_numpy(ei).to(dev))
    out.backward(gout)
    # float64 reference
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    ref = tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"],
                              p64["bias"], H, B, aggrs, add_self_loops=kw.get("add_self_loops", True),
                              sigmoid=kw.get("sigmoid", False))
    ref.backward(gout.double().cpu())
    assert _rel(out, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


@pytest.mark.parametrize("aggrs", [["sum", "std", "max"], ["mean", "var"]])
def test_std_gradient_on_tied_neighbourhoods(aggrs):
    """Neighbourhoods whose rows are identical (|x| ~ 10): the forward's variance is 0 (or -tiny -> relu -> 0) there, and the
    backward must use THAT number -- its relu mask, its std -- not one re-derived from a sum of squares with the cancelling
    formula (the training record now keeps the forward's variance).  Gradients against float64."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    n, fin, fout, H, B = 240, 32, 64, 8, 4
    # destinations 0..79 gather from sources inside one of 8 groups of identical rows; the rest of the graph is random
    src, dst = [], []
    for i in range(80):
        grp = 80 + 10 * (i % 8)
        for j in rng.integers(0, 10, size=int(rng.integers(2, 9))):
            src.append(grp + int(j)); dst.append(i)
    rnd = _graph(rng, n, 1200, hub=60)
    rnd = rnd[:, rnd[1] >= 80]
    ei = np.concatenate([np.array([src, dst], dtype=np.int64), rnd], axis=1)
    torch.manual_seed(4)
    conv = egc_amd.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=False).to(dev)
    x0 = torch.randn(n, fin)
    for k in range(8):
        x0[80 + 10 * k: 90 + 10 * k] = 10.0 * torch.randn(1, fin)      # ten identical rows of magnitude ~ 10
    x = x0.to(dev).requires_grad_(True)
    gout = torch.randn(n, fout, device=dev)
    out = conv(x, torch.from_numpy(ei).to(dev))
    out.backward(gout)
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x0.double().requires_grad_(True)
    ref = tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"], p64["bias"], H, B,
                              aggrs, add_self_loops=False, sigmoid=False)
    ref.backward(gout.double().cpu())
    assert torch.isfinite(x.grad).all()
    assert _rel(out, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


@pytest.mark.parametrize("aggrs,kw", [
    (["symadd", "max", "mean"], {}),
    (["add", "std", "max"], {}),
    (["symadd", "min", "var"], {"softmax": True}),
    (["mean"], {"hardtanh": True}),
    (["symadd"], {"add_self_loops": False}),
])
def test_efficient_graph_conv_gradients(aggrs, kw):
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11 + len(aggrs))
    n, hidden, H, B = 250, 42, 6, 3   # L = 7: non power of two -> generic forward kernels
    ei = _graph(rng, n, 1800, hub=150, self_loops=9)
    torch.manual_seed(2)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, kw.get("softmax", False), aggrs=aggrs,
                                      add_self_loops=kw.get("add_self_loops", True),
                                      hardtanh_weights=kw.get("hardtanh", False)).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, hidden, device=dev, requires_grad=True)
    gout = torch.randn(n, hidden, device=dev)
    out = conv(x=x, edge_index=torch.from_numpy(ei).to(dev))
    out.backward(gout)
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    ref = tref.efficient_graph_conv_forward(
        x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)], p64["comb_weights.weight"], p64["comb_weights.bias"],
        p64["bias"], H, aggrs, softmax=kw.get("softmax", False), hardtanh=kw.get("hardtanh", False),
        add_self_loops=kw.get("add_self_loops", True))
    ref.backward(gout.double().cpu())
    assert _rel(out, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


def test_max_gradient_goes_to_first_maximal_edge():
    """torch_scatter arg semantics: with ties, the FIRST edge in input order gets the whole gradient."""
    import egc_amd
    dev = torch.device("cuda:0")
    # node 0 receives edges from 1, 2, 3 (in this order); nodes 2 and 3 carry the same maximal feature
    ei = torch.tensor([[1, 2, 3], [0, 0, 0]], device=dev)
    conv = egc_amd.EGConv(1, 1, aggrs=["max"], num_heads=1, num_bases=1, add_self_loops=False, bias=False).to(dev)
    with torch.no_grad():
        conv.bases_weight.fill_(1.0)
        conv.comb_weight.weight.zero_()
        conv.comb_weight.bias.fill_(1.0)
    x = torch.tensor([[0.0], [1.0], [5.0], [5.0]], device=dev, requires_grad=True)
    out = conv(x, ei)
    out[0, 0].backward()
    assert out[0, 0].item() == 5.0
    assert x.grad.flatten().tolist() == [0.0, 0.0, 1.0, 0.0]


def test_training_step_reduces_loss():
    """End to end: a few Adam steps through two stacked layers lower a regression loss."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n = 2000
    ei = torch.from_numpy(_graph(rng, n, 16000)).to(dev)
    torch.manual_seed(0)
    l1 = egc_amd.EGConv(32, 64, aggrs=["symnorm", "max", "mean"], num_heads=4, num_bases=4).to(dev)
    l2 = egc_amd.EGConv(64, 8, aggrs=["sum"], num_heads=2, num_bases=2).to(dev)
    opt = torch.optim.Adam(list(l1.parameters()) + list(l2.parameters()), lr=1e-2)
    x = torch.randn(n, 32, device=dev)
    y = torch.randn(n, 8, device=dev)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        loss = ((l2(torch.relu(l1(x, ei)), ei) - y) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses[::6]


@pytest.mark.parametrize("hidden,H,B", [(64, 8, 4), (42, 6, 3), (88, 2, 2)])  # register (2^k) / generic / register (44 slots)
def test_arg_extrema_match_scatter_arg_bit_exact(hidden, H, B):
    """arg_max / arg_min of the training forward == the `arg` of torch_scatter's scatter_max / scatter_min as
    restated by the oracle (first entry in input order attaining the extremum; E for an empty row there, -1
    here), including exact ties (duplicated source features) and a hub row cut into chunks."""
    import egc_amd
    from egc_amd.functional import egc_aggregate_combine_train
    from oracle import egc_oracle as orc
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n = 400
    ei = _graph(rng, n, 3000, hub=300, self_loops=10)
    ei = ei[:, ei[1] < n - 5]                       # the last rows receive nothing
    f_g = B * (hidden // H)
    bases_np = rng.standard_normal((n, f_g)).astype(np.float32)
    bases_np[rng.integers(0, n, size=150)] = bases_np[7]      # exact ties between many sources
    conv = egc_amd.EfficientGraphConv(hidden, hidden, num_heads=H, num_bases=B, softmax_weights=False, aggrs=["max", "min"])
    spec = conv._spec
    ldb, L, Ls = spec.ldb, spec.basis_len, spec.basis_stride   # each basis may be padded to whole 16-byte slots
    bases = torch.zeros(n, ldb)
    bases[:, :B * Ls].view(n, B, Ls)[:, :, :L] = torch.from_numpy(bases_np).view(n, B, L)
    g = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(ei).to(dev), n)
    wt = torch.randn(n, spec.w_cols, device=dev)
    _, (stats, cnt, arg_max, arg_min) = egc_aggregate_combine_train(g, spec, bases.to(dev), wt, None)
    edge_id = g.edge_id.cpu().numpy()
    e = ei.shape[1]
    for name, arg in (("max", arg_max), ("min", arg_min)):
        _, ref = orc.scatter(bases_np[ei[0]], ei[1], n, name)
        got = arg.cpu()[:, :B * Ls].view(n, B, Ls)[:, :, :L].reshape(n, f_g).numpy()
        got_edges = np.where(got >= 0, edge_id[np.clip(got, 0, e - 1)], e)   # CSR position -> input edge; empty -> E
        assert np.array_equal(got_edges, ref), name
    deg = np.bincount(ei[1], minlength=n)
    assert np.array_equal(cnt.cpu().numpy()[:n], deg)


@pytest.mark.parametrize("generic", [False, True])
def test_arg_extrema_with_appended_self_loops_bit_exact(generic, monkeypatch):
    """EGConv's edge set (gcn_norm: existing self-loops dropped, one appended per node at the END of the edge
    list): the arg of a column whose maximum is the node's own feature is n_edges, pre-existing self entries are
    never named, ties go to the earliest remaining entry.  Register-resident kernels (arg tracked inside the
    aggregation) and generic kernels (separate arg pass) must agree with a numpy restatement bit for bit."""
    import egc_amd
    from egc_amd.functional import egc_aggregate_combine_train
    if generic:
        monkeypatch.setenv("EGC_FORCE_GENERIC", "1")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    n, hidden, H, B = 300, 64, 8, 4
    ei = _graph(rng, n, 2400, hub=180, self_loops=40)
    conv = egc_amd.EGConv(hidden, hidden, aggrs=["symnorm", "max", "min"], num_heads=H, num_bases=B)
    spec = conv._spec_coo
    f_g = spec.f_g
    bases_np = rng.standard_normal((n, f_g)).astype(np.float32)
    bases_np[rng.integers(0, n, size=100)] = bases_np[11]          # exact ties between sources
    bases_np[::7] += 4.0                                            # nodes whose own feature wins (self-loop)
    g = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(ei).to(dev), n)
    wt = torch.randn(n, spec.w_cols, device=dev)
    _, (stats, cnt, arg_max, arg_min) = egc_aggregate_combine_train(g, spec, torch.from_numpy(bases_np).to(dev), wt, None)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    e = ei.shape[1]
    want_max = np.empty((n, f_g), dtype=np.int64)
    want_min = np.empty((n, f_g), dtype=np.int64)
    for i in range(n):
        pos = np.array([p for p in range(rowptr[i], rowptr[i + 1]) if col[p] != i] + [e])
        src = np.array([col[p] for p in pos[:-1]] + [i])
        vals = bases_np[src]                                        # [k, f_g] in edge order, self-loop last
        want_max[i] = pos[np.argmax(vals, axis=0)]                  # argmax / argmin return the FIRST extremum
        want_min[i] = pos[np.argmin(vals, axis=0)]
    assert np.array_equal(arg_max.cpu().numpy()[:, :f_g], want_max)
    assert np.array_equal(arg_min.cpu().numpy()[:, :f_g], want_min)
    assert int((want_max == e).sum()) > 100                         # the self-loop case is really exercised


@pytest.mark.parametrize("fin,aggrs", [(192, ["symnorm"]), (192, ["sum", "max"]), (128, ["sum", "mean", "max", "symnorm"])])
def test_input_gradient_through_the_packed_gemm(fin, aggrs):
    """d x = [d bases | d weightings] @ [bases_weight | comb_weight^T]^T runs on the forward's split-precision GEMM
    with no weightings block: K = F_g + W = 128 (the fp16x2 kernel, 192 output columns), 160 and 192 (bf16x3)."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(fin + len(aggrs))
    n, H, B = 1500, 8, 4
    ei = _graph(rng, n, 12000, hub=300, self_loops=20)
    torch.manual_seed(3)
    conv = egc_amd.EGConv(fin, fin, aggrs=aggrs, num_heads=H, num_bases=B).to(dev)
    x = torch.randn(n, fin, device=dev, requires_grad=True)
    gout = torch.randn(n, fin, device=dev)
    out = conv(x, torch.from_numpy(ei).to(dev))
    out.backward(gout)
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    ref = tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"],
                              p64["bias"], H, B, aggrs)
    ref.backward(gout.double().cpu())
    assert _rel(out, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


@pytest.mark.parametrize("generic", [False, True])
def test_d_bases_needs_no_fill_on_square_graphs(generic, monkeypatch):
    """egc_aggregate_combine_backward_f32 on a square graph WRITES every row of d_bases (short source rows are stored,
    hub rows zeroed by the destination kernel before their chunks add): the array is handed over full of NaN here
    (C ABI directly, joint [d_bases | d_weightings] layout) and must come back equal to the zero-filled run."""
    import ctypes as C
    import egc_amd
    from egc_amd import _C, functional as F
    from egc_amd.workloads import heavy_tailed_graph
    if generic:
        monkeypatch.setenv("EGC_BWD_GENERIC", "1")
    dev = torch.device("cuda:0")
    lib = _C.load()
    n = 12000
    ei = heavy_tailed_graph(n, 150000, seed=3).to(dev)          # hub rows on both sides of the transposition
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev)
    spec = conv._spec_coo
    with torch.no_grad():
        wcat, bcat = conv._packed_weights()
        x = torch.randn(n, 128, device=dev)
        go = torch.randn(n, 128, device=dev)
        bases, wt = F.egc_basis_transform(g, spec, x, wcat, bcat, None)
        _, (stats, cnt, arg_max, arg_min) = F.egc_aggregate_combine_train(g, spec, bases, wt, conv.bias)
        tg = g.transposed()
        cg, ct = g.c_struct(), tg.c_struct()
        nbytes = lib.egc_backward_workspace_bytes(C.byref(spec.c), n)
        outs = []
        for fill in (0.0, float("nan")):
            d_cat = torch.full((n, spec.ldb + spec.w_cols), fill, device=dev)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _C.check(lib.egc_aggregate_combine_backward_f32(
                C.byref(cg), C.byref(ct), C.byref(spec.c), bases.data_ptr(), spec.ldb, wt.data_ptr(), go.data_ptr(),
                stats.data_ptr(), cnt.data_ptr(), arg_max.data_ptr(), None, d_cat.data_ptr(), d_cat.stride(0),
                d_cat[:, spec.ldb:].data_ptr(), d_cat.stride(0), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream().cuda_stream), "egc_aggregate_combine_backward_f32")
            torch.cuda.synchronize()
            outs.append(d_cat)
    assert bool(torch.isfinite(outs[1]).all())
    # hub rows: float atomics in arrival order -> equal up to summation order, everything else bit for bit
    deg_t = torch.bincount(ei[0], minlength=n)
    short = deg_t <= 64
    assert torch.equal(outs[0][short], outs[1][short])
    assert _rel(outs[1], outs[0]) <= 1e-6


@pytest.mark.parametrize("hidden,H,B,aggrs", [
    (124, 4, 4, ["add", "std", "max"]),        # zinc EGC-M: padded bases (L = 31), 32 slots, H < slots per basis
    (128, 4, 4, ["symadd", "std", "max"]),     # CIFAR EGC-M
    (136, 4, 4, ["symadd", "max", "mean"]),    # arxiv EGC-M: 36 slots -> generic destination kernel
    (168, 8, 4, ["symadd"]),                   # zinc / CIFAR EGC-S
    (224, 4, 4, ["add", "mean", "max"]),       # molhiv EGC-M
    (296, 8, 4, ["symadd"]),                   # molhiv EGC-S: 37 slots
])
def test_trained_reference_shapes_gradients(hidden, H, B, aggrs):
    """Gradients of EfficientGraphConv at the layer shapes of the reference's trained nets (hyperparameters.md) -- the
    register-resident and the generic backward kernels both occur -- against float64 autograd through the restatement."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(hidden)
    n = 300
    ei = _graph(rng, n, 2400, hub=170, self_loops=7)
    torch.manual_seed(hidden)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, hidden, device=dev, requires_grad=True)
    gout = torch.randn(n, hidden, device=dev)
    out = conv(x=x, edge_index=torch.from_numpy(ei).to(dev))
    out.backward(gout)
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    ref = tref.efficient_graph_conv_forward(
        x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)], p64["comb_weights.weight"], p64["comb_weights.bias"],
        p64["bias"], H, aggrs)
    ref.backward(gout.double().cpu())
    assert _rel(out, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


@pytest.mark.parametrize("hidden,H,B,aggrs", [(296, 8, 4, ["symadd"]), (224, 4, 4, ["add", "mean", "max"]), (136, 4, 4, ["symadd", "max", "mean"])])
def test_one_row_groups_on_a_graph_of_33k_rows(hidden, H, B, aggrs):
    """33 - 64-slot layers on >= 32,768 rows: the forward aggregate runs four row groups per wavefront there (the next group's
    bounds and indices requested ahead; round 6) -- a ragged row count (the last wavefront's groups run past the end), a hub row
    beyond the long-row threshold in the middle of a wavefront's groups, empty rows; forward and every gradient against float64
    autograd through the restatement."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(hidden)
    n = 32768 + 1003
    ei = _graph(rng, n, 3 * n, hub=900, self_loops=11)
    ei[1][ei[1] % 7 == 3] = 5                       # rows without entries; row 5 becomes a second long row
    torch.manual_seed(hidden)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, hidden, device=dev, requires_grad=True)
    gout = torch.randn(n, hidden, device=dev)
    out = conv(x=x, edge_index=torch.from_numpy(ei).to(dev))
    out.backward(gout)
    conv.eval()
    with torch.no_grad():
        out_eval = conv(x=x.detach(), edge_index=torch.from_numpy(ei).to(dev))
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in conv.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    ref = tref.efficient_graph_conv_forward(
        x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)], p64["comb_weights.weight"], p64["comb_weights.bias"],
        p64["bias"], H, aggrs)
    ref.backward(gout.double().cpu())
    assert _rel(out, ref) <= 1e-5 and _rel(out_eval, ref) <= 1e-5
    assert _rel(x.grad, x64.grad) <= gtol(aggrs)
    for k, v in conv.named_parameters():
        assert _rel(v.grad, p64[k].grad) <= gtol(aggrs), k


@pytest.mark.parametrize("kind", ["opt", "lay"])
def test_frozen_parameters_take_the_operand_level_node_and_agree(kind):
    """With trainable parameters the modules run pack + layer + unpack as one autograd node (_EGCLayerParamsFunction);
    with frozen parameters (fine-tuning a head, saliency maps) the cached operand goes through _EGCLayerFunction.
    Same d out / d x either way; the frozen layer produces no parameter gradients."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n, hidden = 700, 64
    ei = torch.from_numpy(rng.integers(0, n, size=(2, 6000))).to(dev)
    torch.manual_seed(1)
    if kind == "opt":
        layer = egc_amd.EGConv(hidden, hidden, aggrs=["sum", "max", "symnorm"], num_heads=4, num_bases=4).to(dev)
    else:
        layer = egc_amd.EfficientGraphConv(hidden, hidden, 4, 4, False, aggrs=["add", "max", "symadd"]).to(dev)
    x = torch.randn(n, hidden, device=dev)
    gout = torch.randn(n, hidden, device=dev)
    call = (lambda t: layer(t, ei)) if kind == "opt" else (lambda t: layer(x=t, edge_index=ei))
    xa = x.clone().requires_grad_(True)
    out_a = call(xa)
    out_a.backward(gout)
    assert all(p.grad is not None for p in layer.parameters())
    for p in layer.parameters():
        p.requires_grad_(False)
        p.grad = None
    xb = x.clone().requires_grad_(True)
    out_b = call(xb)
    out_b.backward(gout)
    assert all(p.grad is None for p in layer.parameters())
    assert torch.equal(out_a.detach(), out_b.detach())
    scale = max(1.0, float(xa.grad.abs().max()))
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("kind,hidden,H,B,aggrs", [
    ("opt", 64, 8, 4, ["sum", "mean", "max", "symnorm"]),      # north star: records built inside bwd_dst_fast_kernel
    ("opt", 64, 8, 4, ["min", "max"]),
    ("lay", 128, 4, 4, ["symadd", "std", "max"]),              # 32 slots
    ("lay", 224, 4, 4, ["add", "mean", "max"]),                # 56 slots (64-lane groups)
    ("lay", 136, 4, 4, ["symadd", "max", "min"]),              # 36 slots, two record arrays
    ("opt", 42, 6, 3, ["max", "min", "mean"]),                 # LDS-based destination kernel -> bwd_records_kernel's short rows
    ("lay", 256, 4, 4, ["max"]),                               # 64 slots
])
def test_extremum_gradient_records_equal_the_arg_byte_path(kind, hidden, H, B, aggrs, monkeypatch):
    """The gradients of max / min reach the source kernel as one 64-byte record per entry (egc_backward.hip,
    bwd_records_kernel / the fused builder): same gradients as the arg-byte path (EGC_BWD_NO_REC=1) up to the order of
    float additions, on a graph with rows of 0..5 entries (records overflow: more than 12 columns go to one entry), hub
    rows cut into chunks (both sides), exact ties between sources, explicit self loops; float64 autograd bounds both."""
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(hidden + len(aggrs))
    n = 1500
    parts = [rng.integers(0, n, size=(2, 9000))]
    for node, deg in [(0, 2100), (1, 300), (77, 65), (78, 64), (400, 257)]:          # destination hubs
        parts.append(np.stack([rng.integers(0, n, size=deg), np.full(deg, node)]))
    for node, deg in [(5, 1500), (6, 70)]:                                            # source hubs (long transposed rows)
        parts.append(np.stack([np.full(deg, node), rng.integers(0, n, size=deg)]))
    s = rng.integers(0, n, size=40)
    parts.append(np.stack([s, s]))
    ei = np.concatenate(parts, axis=1)
    ei = ei[:, (ei[1] < n - 300) | (rng.random(ei.shape[1]) < 0.15)]                  # the last rows: 0..5 entries each
    ei = torch.from_numpy(ei[:, rng.permutation(ei.shape[1])].astype(np.int64)).to(dev)
    torch.manual_seed(3)
    if kind == "opt":
        conv = egc_amd.EGConv(48, hidden, aggrs=aggrs, num_heads=H, num_bases=B).to(dev)
    else:
        conv = egc_amd.EfficientGraphConv(48, hidden, H, B, False, aggrs=aggrs).to(dev)
    x0 = torch.randn(n, 48, device=dev)
    x0[torch.from_numpy(rng.integers(12, n, size=200)).to(dev)] = x0[11].clone()   # exact ties
    gout = torch.randn(n, hidden, device=dev)

    def grads():
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        out = conv(x, ei) if kind == "opt" else conv(x=x, edge_index=ei)
        out.backward(gout)
        return [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]

    rec = grads()
    monkeypatch.setenv("EGC_BWD_NO_REC", "1")
    ref = grads()
    monkeypatch.delenv("EGC_BWD_NO_REC")
    monkeypatch.setenv("EGC_BWD_REC_SEPARATE", "1")       # every record by bwd_records_kernel
    sep = grads()
    for a, b, c in zip(rec, ref, sep):
        assert _rel(a, b) <= 2e-6 and _rel(c, b) <= 2e-6
    assert float(rec[0].abs().max()) > 0


def test_backward_workspace_sizes():
    """egc_backward_workspace_bytes_for = the tables + 64 bytes per entry and max / min aggregator; the smaller size of
    egc_backward_workspace_bytes is still accepted (arg-byte path)."""
    import ctypes as C
    from egc_amd import _C
    from egc_amd import functional as F
    import egc_amd
    dev = torch.device("cuda:0")
    lib = _C.load()
    rng = np.random.default_rng(0)
    n = 500
    ei = torch.from_numpy(_graph(rng, n, 4000, hub=100)).to(dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    for aggrs, extrema in [(["sum", "max"], 1), (["min", "max", "mean"], 2), (["symnorm"], 0)]:
        conv = egc_amd.EGConv(16, 64, aggrs=aggrs, num_heads=8, num_bases=4)
        spec = conv._spec_coo
        gs = g.c_struct()
        small = lib.egc_backward_workspace_bytes(C.byref(spec.c), n)
        big = lib.egc_backward_workspace_bytes_for(C.byref(spec.c), C.byref(gs))
        assert big == small + extrema * g.n_edges * 64 and small % 256 == 0


def test_records_only_where_an_entry_receives_few_columns():
    """The per-entry extremum records hold 12 (value, column) pairs; an entry receives ldb N / E columns on average.  Round 6: they
    are built only where that is at most 10 (every record of a molecule batch at 224 columns overflowed: 282 MB fetched for 110 MB
    of payload) -- the workspace query and the backward agree on it, and both forms give the same gradients."""
    import ctypes as C
    from egc_amd import _C
    import egc_amd
    dev = torch.device("cuda:0")
    lib = _C.load()
    rng = np.random.default_rng(3)
    n = 2000
    for e, with_records in ((2 * n, False), (40 * n, True)):          # 32 basis columns: 16 / 0.8 columns per entry
        ei = torch.from_numpy(_graph(rng, n, e)).to(dev)
        g = egc_amd.CSRGraph.from_edge_index(ei, n)
        conv = egc_amd.EGConv(32, 64, aggrs=["sum", "max"], num_heads=8, num_bases=4).to(dev)
        gs = g.c_struct()
        small = lib.egc_backward_workspace_bytes(C.byref(conv._spec_coo.c), n)
        big = lib.egc_backward_workspace_bytes_for(C.byref(conv._spec_coo.c), C.byref(gs))
        assert (big > small) == with_records, (e, small, big)
        grads = []
        for env in (None, "1"):
            if env: os.environ["EGC_BWD_NO_REC"] = env
            try:
                x = torch.randn(n, 32, device=dev, requires_grad=True)
                torch.manual_seed(1)
                x.data.copy_(torch.randn(n, 32, device=dev))
                conv.zero_grad()
                conv(x, ei).square().sum().backward()
                grads.append([x.grad.clone()] + [p.grad.clone() for p in conv.parameters()])
            finally:
                os.environ.pop("EGC_BWD_NO_REC", None)
        for a, b in zip(*grads):
            assert _rel(a, b) <= 2e-6
