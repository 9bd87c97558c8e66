"""Batches of whole graphs, the layer in ONE launch (egc_fused_tile.hip; egc_layer_forward_batch_fused_f32): plan, basis
transform + weightings on the matrix cores, per-tile CSR, aggregation and combine inside one kernel -- no `bases` /
`weightings` in memory.  Against the numpy oracle of the reference's layers (layers.py:89-225, optimized_layers.py:124-278)
on PyG-shaped batches (molecules, superpixel k-NN graphs, hub rows, isolated nodes, self loops, duplicate edges, empty
graphs), both layers' edge-set conventions, the fused BatchNorm(eval) / ReLU / residual tail, the full size of BASELINE
configs 1 / 3 / 4, with and without the graphs' edge offsets; its envelope and its error reporting."""
import numpy as np
import pytest
import torch

from golden_util import elementwise_excess, rel_err
from test_batch_tile_gpu import _layer, _messy_batch, _oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _ran_fused(gb):
    return any(k[-1] == "fused" and v for k, v in gb._setups.items() if isinstance(k, tuple))


def _edge_ptr(ei, ptr):
    """The graphs' edge offsets as PyG's collation keeps them (edges grouped by graph)."""
    d = ei[1].numpy()
    return torch.from_numpy(np.searchsorted(d, ptr.numpy(), side="left").astype(np.int64)) if np.all(np.diff(d) >= 0) else None


@pytest.mark.parametrize("kind,hidden,H,B,aggrs,asl", [
    ("opt", 128, 8, 4, ["sum", "mean", "max", "symnorm"], True),       # north star (static configuration)
    ("lay", 128, 8, 4, ["symadd", "max", "mean"], True),               # EfficientGraphConv EGC-M (static)
    ("lay", 128, 8, 4, ["symadd"], True),                              # EGC-S: A = 1 (weightings rows padded to [h][b][4])
    ("opt", 128, 1, 1, ["sum"], True),                                 # BASELINE config 1's plumbing shape: ldb = 128, W = 1
    ("opt", 64, 4, 4, ["min", "max", "mean"], True),                   # run-time configuration, NEED_MN, F_in = 64
    ("opt", 96, 4, 2, ["sum", "max"], True),                           # no symnorm: loops from add_remaining_self_loops
    ("opt", 128, 8, 4, ["symnorm", "mean"], False),                    # RAW sets
    ("lay", 124, 4, 4, ["add", "mean", "max"], True),                  # L = 31: padded bases, 32 slots
    ("opt", 128, 8, 2, ["sum", "symnorm", "max"], True),               # 8 lanes per basis, A = 3
    ("lay", 124, 4, 4, ["add", "std", "max"], True),                   # the reference's ZINC EGC-M layer (std: shifted variance)
    ("lay", 128, 4, 4, ["symadd", "std", "max"], True),                # the reference's CIFAR EGC-M layer
    ("opt", 128, 8, 4, ["sum", "var", "min"], True),                   # var + min
    # ---- the WIDE form (round 5): the reference's own batched layer shapes (run_pretrained.sh:7,12,23,24) and their neighbours ----
    ("lay", 168, 8, 4, ["symadd"], True),                              # zinc / cifar EGC-S: L = 21 padded to 24, 24 slots, two k-slabs
    ("lay", 296, 8, 4, ["symadd"], True),                              # molhiv EGC-S: L = 37 padded to 40, 40 slots, three k-slabs
    ("lay", 224, 4, 4, ["add", "mean", "max"], True),                  # molhiv EGC-M: 56 slots, 224 + 48 columns (9 tiles of 32)
    ("opt", 256, 8, 4, ["sum", "mean", "max", "symnorm"], True),       # F_in a multiple of 128, 32 slots, A = 4
    ("lay", 136, 4, 4, ["symadd", "max", "mean"], True),               # arxiv EGC-M's width: a partial second slab of 8 columns
    ("lay", 184, 8, 4, ["symadd", "std", "max"], True),                # std in the wide form (NEED_SQ), L = 23 padded to 24
    ("opt", 200, 4, 2, ["min", "max"], False),                         # RAW sets, min, B = 2, no symnorm, F_in % 32 != 0
    ("opt", 160, 16, 4, ["sum", "symnorm"], True),                     # H / B = 4 heads per basis, A = 2 (two floats per (h, b) block)
    ("lay", 300, 4, 4, ["symadd", "min", "max"], True),                # ogbg-code EGC-M (run_pretrained.sh:48): 76 slots -> two passes of 40 / 36 lanes
    ("lay", 304, 8, 8, ["symadd"], True),                              # ogbg-code EGC-S (run_pretrained.sh:47): 80 slots, B = 8, 384 columns
])
@pytest.mark.parametrize("with_edge_ptr", [False, True])
def test_one_launch_layer_matches_the_oracle_on_a_messy_batch(kind, hidden, H, B, aggrs, asl, with_edge_ptr):
    import egc_amd
    dev = _dev()
    # (the 300- / 304-wide layers keep 64-row tiles: their messy batch has graphs of at most 63 nodes)
    ei, n, ptr = _messy_batch(hidden + len(aggrs), max_size=90 if hidden < 300 else 64)
    torch.manual_seed(1)
    conv = _layer(kind, hidden, H, B, aggrs, asl)
    x = torch.randn(n, hidden)
    ref = _oracle(conv, kind, x, ei, H, B, aggrs, asl)
    conv = conv.to(dev).eval()
    # the messy batch's edges are grouped by graph but not sorted by destination: offsets from the generator's own layout
    eptr = None
    if with_edge_ptr:
        g_of_edge = np.searchsorted(ptr.numpy(), ei[1].numpy(), side="right") - 1
        assert np.all(np.diff(g_of_edge) >= 0)
        eptr = torch.from_numpy(np.searchsorted(g_of_edge, np.arange(ptr.numel()), side="left").astype(np.int64)).to(dev)
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90 if hidden < 300 else 64, edge_ptr=eptr)
    with torch.no_grad():
        out = conv(x.to(dev), gb) if kind == "opt" else conv(x=x.to(dev), edge_index=gb)
    gb.check()
    assert _ran_fused(gb) and not gb._plans, "the one-launch path did not run"
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    if any(a in ("std", "var") for a in aggrs):
        # the float32 formula E[x^2] - E[x]^2 of the oracle is itself up to 1e-4 from the true value on (nearly) tied
        # neighbourhoods (tests/test_fuzz_gpu.py); the kernels' variance about the row's first entry is not: against float64
        truth = _truth64(conv.cpu(), kind, x, ei, H, B, aggrs, asl)
        assert rel_err(got, truth) <= TOL, rel_err(got, truth)
        assert elementwise_excess(got, truth, TOL) <= 1.0, elementwise_excess(got, truth, TOL)      # (every element, on its row's scale)
        assert rel_err(got, ref) <= 1e-4
        return
    assert rel_err(got, ref) <= TOL, rel_err(got, ref)
    assert elementwise_excess(got, ref, TOL) <= 1.0


def _truth64(conv, kind, x, ei, H, B, aggrs, asl):
    """The layer in float64 through the torch restatement of the reference (oracle/egc_torch_ref.py)."""
    from oracle import egc_torch_ref as tref
    p = {k: v.detach().double().cpu() for k, v in conv.named_parameters()}
    x64 = x.double()
    e = ei.numpy()
    with torch.no_grad():
        if kind == "opt":
            out = tref.egconv_forward(x64, e, p["bases_weight"], p["comb_weight.weight"], p["comb_weight.bias"], p["bias"], H, B, aggrs,
                                      add_self_loops=asl, sigmoid=False)
        else:
            out = tref.efficient_graph_conv_forward(x64, e, [p[f"bases_weight.{b}"] for b in range(B)], p["comb_weights.weight"],
                                                    p["comb_weights.bias"], p["bias"], H, aggrs, softmax=False, hardtanh=False,
                                                    sigmoid=False, add_self_loops=asl)
    return out.numpy()


def test_without_edge_offsets_the_edges_must_be_sorted_by_graph_not_by_destination():
    """edge_ptr = None: the tile's edge range is found by searching the destination row for the tile's first node, which only
    needs the edges GROUPED by graph in graph order (PyG's collation) -- the messy batch's rows are unsorted inside a graph."""
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(3, n_graphs=120)
    aggrs = ["sum", "mean", "max", "symnorm"]
    torch.manual_seed(5)
    conv = _layer("opt", 128, 8, 4, aggrs)
    x = torch.randn(n, 128)
    ref = _oracle(conv, "opt", x, ei, 8, 4, aggrs)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        out = conv(x.to(dev), gb)
    gb.check()
    assert _ran_fused(gb)
    assert rel_err(out.cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("workload", ["molhiv", "cifar", "zinc"])
@pytest.mark.parametrize("with_edge_ptr", [False, True])
def test_one_launch_layer_at_full_batch_sizes(workload, with_edge_ptr):
    """BASELINE configs 3 / 4 (2048 molecules; 2048 superpixel 8-NN graphs, 1.93 M edges) and a ZINC batch of 128 (config 1's
    batch), north-star layer through EGConv; whole output against the oracle, scale-relative AND element-wise."""
    import egc_amd
    from egc_amd.workloads import knn_superpixel_batch, molecule_batch, zinc_like_batch
    dev = _dev()
    if workload == "molhiv":
        ei, n, batch = molecule_batch(2048, seed=0); G = 2048
    elif workload == "cifar":
        ei, n, batch = knn_superpixel_batch(2048, seed=0); G = 2048
    else:
        _, ei, n, batch = zinc_like_batch(128, seed=0); G = 128
    sizes = torch.bincount(batch, minlength=G)
    mx = int(sizes.max())
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
    eptr = None
    if with_edge_ptr:
        eptr = torch.searchsorted(batch[ei[1]].contiguous(), torch.arange(G + 1)).to(dev)
    torch.manual_seed(3)
    aggrs = ["sum", "mean", "max", "symnorm"]
    conv = _layer("opt", 128, 8, 4, aggrs)
    x = torch.randn(n, 128)
    ref = _oracle(conv, "opt", x, ei, 8, 4, aggrs)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=mx, edge_ptr=eptr)
    with torch.no_grad():
        out = conv(x.to(dev), gb)
    gb.check()
    assert _ran_fused(gb) and not gb._plans
    got = out.cpu().numpy()
    assert rel_err(got, ref) <= TOL, rel_err(got, ref)
    assert elementwise_excess(got, ref, TOL) <= 1.0
    # and the two-launch tile path on the same batch: same results up to summation order
    import os
    os.environ["EGC_NO_FUSED_TILE"] = "1"
    try:
        gb2 = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=mx, edge_ptr=eptr)
        with torch.no_grad():
            out2 = conv(x.to(dev), gb2)
        gb2.check()
    finally:
        del os.environ["EGC_NO_FUSED_TILE"]
    assert rel_err(got, out2.cpu().numpy()) <= TOL


@pytest.mark.parametrize("workload,hidden,H,B,aggrs", [
    ("zinc", 168, 8, 4, ["symadd"]),                 # run_pretrained.sh:7   zinc EGC-S
    ("cifar", 168, 8, 4, ["symadd"]),                # run_pretrained.sh:12  cifar EGC-S (graphs of up to 150 nodes: 160-row tiles)
    ("molhiv", 296, 8, 4, ["symadd"]),               # run_pretrained.sh:23  molhiv EGC-S
    ("molhiv", 224, 4, 4, ["add", "mean", "max"]),   # run_pretrained.sh:24  molhiv EGC-M (BASELINE config 3's own net)
    ("zinc", 300, 4, 4, ["symadd", "min", "max"]),   # run_pretrained.sh:48  code EGC-M's layer (no code-shaped batch in BASELINE: the ZINC batch)
    ("zinc", 304, 8, 8, ["symadd"]),                 # run_pretrained.sh:47  code EGC-S's layer
])
def test_wide_one_launch_layer_at_full_batch_sizes(workload, hidden, H, B, aggrs):
    """The reference's own batched nets at the full batch of their dataset's shape through the WIDE one-launch form
    (EfficientGraphConv, layers.py:89-140): whole output against the oracle, scale-relative AND element-wise, and against the CSR path."""
    import egc_amd
    from egc_amd.workloads import knn_superpixel_batch, molecule_batch, zinc_like_batch
    dev = _dev()
    if workload == "molhiv":
        ei, n, batch = molecule_batch(2048, seed=0); G = 2048
    elif workload == "cifar":
        ei, n, batch = knn_superpixel_batch(512, seed=0); G = 512        # (512 graphs: the float32 oracle of 2048 takes minutes)
    else:
        _, ei, n, batch = zinc_like_batch(128, seed=0); G = 128
    sizes = torch.bincount(batch, minlength=G)
    mx = int(sizes.max())
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
    torch.manual_seed(3)
    conv = _layer("lay", hidden, H, B, aggrs)
    x = torch.randn(n, hidden)
    ref = _oracle(conv, "lay", x, ei, H, B, aggrs)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=mx)
    with torch.no_grad():
        out = conv(x=x.to(dev), edge_index=gb)
        csr = conv(x=x.to(dev), edge_index=ei.to(dev))
    gb.check()
    assert _ran_fused(gb) and not gb._plans, "the one-launch path did not run"
    got = out.cpu().numpy()
    assert rel_err(got, ref) <= TOL, rel_err(got, ref)
    assert elementwise_excess(got, ref, TOL) <= 1.0
    assert rel_err(got, csr.cpu().numpy()) <= TOL


def test_fused_block_tail_in_the_one_launch_kernel():
    """FusedEGCBlock (eval): BatchNorm affine + ReLU + residual in the kernel's store == the plain composition."""
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(5)
    torch.manual_seed(2)
    conv = _layer("lay", 128, 8, 4, ["symadd", "max", "mean"]).to(dev)
    bn = torch.nn.BatchNorm1d(128).to(dev)
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
    x = torch.randn(n, 128, device=dev)
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
    block = egc_amd.FusedEGCBlock(conv, bn).eval()
    with torch.no_grad():
        got = block(x, gb)
        ref = block._plain(x, ei.to(dev))
    gb.check()
    assert _ran_fused(gb)
    assert float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())) <= 1e-5


def test_weight_nonlinearities_in_the_gemm_epilogue():
    """sigmoid / hardtanh on the weightings (layers.py:123-125) are applied where the GEMM writes them to the LDS image."""
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(17, n_graphs=80)
    for kw in ({"sigmoid_weights": True}, {"hardtanh_weights": True}):
        torch.manual_seed(6)
        conv = egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=["symadd", "max", "mean"], **kw)
        with torch.no_grad():
            conv.bias.normal_()
        conv = conv.to(dev).eval()
        x = torch.randn(n, 128, device=dev)
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
        with torch.no_grad():
            got = conv(x=x, edge_index=gb)
            ref = conv(x=x, edge_index=ei.to(dev))
        gb.check()
        assert _ran_fused(gb)
        assert float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())) <= 1e-5


def test_envelope_and_fallbacks(monkeypatch):
    """Outside the envelope the batch takes the two-launch tile path / the CSR path with the same results: a declared graph
    size beyond the LDS image, F_in > 128, a layer that asks for the 24-bit-operand GEMM, training."""
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(23, n_graphs=100)
    aggrs = ["sum", "mean", "max", "symnorm"]
    torch.manual_seed(7)
    conv = _layer("opt", 128, 8, 4, aggrs).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    with torch.no_grad():
        ref = conv(x, ei.to(dev))
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=600)      # beyond the image: two-launch tile path
        out = conv(x, gb)
        gb.check()
        assert not _ran_fused(gb) and gb._plans
        assert float((out - ref).abs().max()) / max(1.0, float(ref.abs().max())) <= 1e-5
        sv = _layer("opt", 128, 8, 4, ["sum", "std", "max"]).to(dev).eval()     # std: one launch like every other layer ...
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
        sv(x, gb)
        gb.check()
        assert _ran_fused(gb)
        monkeypatch.setenv("EGC_GEMM_STDVAR_24BIT", "1")                        # ... unless the 24-bit-operand GEMM is asked for
        sv = _layer("opt", 128, 8, 4, ["sum", "std", "max"]).to(dev).eval()
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
        sv(x, gb)
        gb.check()
        assert not _ran_fused(gb)
        monkeypatch.delenv("EGC_GEMM_STDVAR_24BIT")
        wide = _layer("opt", 352, 8, 4, ["symnorm"]).to(dev).eval()             # F_in = 352 (the ogbn-mag layer): beyond the wide form's 320
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
        xw = torch.randn(n, 352, device=dev)
        outw = wide(xw, gb)
        gb.check()
        assert not _ran_fused(gb)
        refw = wide(xw, ei.to(dev))
        assert float((outw - refw).abs().max()) / max(1.0, float(refw.abs().max())) <= 1e-5


def test_malformed_batches_are_reported_by_the_one_launch_kernel():
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(12, n_graphs=60)
    conv = _layer("opt", 128, 8, 4, ["sum", "mean", "max", "symnorm"]).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    # (1) an edge between two graphs of different tiles
    bad = ei.clone()
    bad[0, 5] = n - 1
    gb = egc_amd.GraphBatch(bad.to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        conv(x, gb)
    assert _ran_fused(gb)
    with pytest.raises(RuntimeError, match="not grouped by graph"):
        gb.check()
    # (2) edge list shuffled across graphs
    perm = torch.randperm(ei.size(1))
    gb = egc_amd.GraphBatch(ei[:, perm].to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        conv(x, gb)
    with pytest.raises(RuntimeError, match="not grouped by graph"):
        gb.check()
    # (3) a graph larger than max_nodes promises -> a tile beyond the LDS image
    big_ptr = torch.tensor([0, n])
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=big_ptr.to(dev), max_nodes=16)
    with torch.no_grad():
        conv(x, gb)
    with pytest.raises(RuntimeError, match="exceeds the"):
        gb.check()
    # (4) graph offsets that are not monotone (such a ptr made the plan launch write out of bounds; this kernel
    # keeps no tile list): reported, nothing outside `out` is written -- a guard array behind `out` stays intact
    ptr_bad = ptr.clone()
    ptr_bad[1::2] = ptr[-1]
    ptr_bad[2::2] = 0
    ptr_bad[-1] = n
    eptr = torch.zeros_like(ptr_bad)
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr_bad.to(dev), max_nodes=90, edge_ptr=eptr.to(dev))
    with torch.no_grad():
        conv(x, gb)
    with pytest.raises(RuntimeError):
        gb.check()
    with torch.no_grad():
        ok = conv(x, ei.to(dev))      # the device is still healthy
    assert torch.isfinite(ok).all()
