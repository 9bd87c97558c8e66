"""Batches of small graphs on the tile kernels (egc_aggregate_tile.hip; egc_amd.GraphBatch): the CSR of each tile of whole
graphs is built in LDS by the workgroup that aggregates it.  Against the numpy oracle of the reference's layers
(layers.py:89-225, optimized_layers.py:124-278) on PyG-shaped batches -- molecules, superpixel k-NN graphs, hub rows
inside a graph, isolated nodes, self loops, duplicate edges, empty graphs -- at both layers' edge-set conventions, with
the fused BatchNorm(eval) / ReLU / residual tail, at the full size of BASELINE configs 3 and 4; and its error reporting
(edge list not grouped by graph, ids out of range, tiles beyond the LDS areas)."""
import numpy as np
import pytest
import torch

from golden_util import elementwise_excess, rel_err
from oracle import egc_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(autouse=True)
def _two_launch_tile_path(monkeypatch):
    """This file pins the TWO-launch tile path (GEMM + agg_tile_kernel); batches that qualify take the one-launch kernel of
    egc_fused_tile.hip by default (tests/test_fused_tile_gpu.py)."""
    monkeypatch.setenv("EGC_NO_FUSED_TILE", "1")


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _oracle(conv, kind, x, ei, H, B, aggrs, asl=True):
    sd = {k: v.detach().cpu().numpy() for k, v in conv.state_dict().items()}
    if kind == "opt":
        return orc.egconv_forward(x.numpy(), ei.numpy(), sd["bases_weight"], sd["comb_weight.weight"], sd["comb_weight.bias"],
                                  sd["bias"], H, B, aggrs, add_self_loops=asl)
    return orc.efficient_graph_conv_forward(x.numpy(), ei.numpy(), [sd[f"bases_weight.{b}"] for b in range(B)],
                                            sd["comb_weights.weight"], sd["comb_weights.bias"], sd["bias"], H, aggrs,
                                            add_self_loops=asl)


def _messy_batch(seed, n_graphs=300, max_size=90):
    """Graph sizes 1..max_size - 1 incl. empty graphs' neighbours, self loops, duplicates, a hub row, isolated nodes."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, max_size, size=n_graphs)
    sizes[rng.integers(0, n_graphs, size=10)] = 1
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    srcs, dsts = [], []
    for g in range(n_graphs):
        n, o = int(sizes[g]), int(ptr[g])
        e = int(rng.integers(0, 4 * n + 1))
        if g % 37 == 0:
            e = 0                                            # a graph without edges
        s, d = rng.integers(0, n, size=e), rng.integers(0, n, size=e)
        if g % 11 == 0 and e > 5:
            d[: e // 2] = 0                                  # a hub row inside the graph
        if e > 3:
            s[-2:] = d[-2:]                                  # self loops
            s[:2], d[:2] = s[2:4], d[2:4]                    # duplicates
        srcs.append(s + o)
        dsts.append(d + o)
    ei = torch.from_numpy(np.stack([np.concatenate(srcs), np.concatenate(dsts)]).astype(np.int64))
    return ei, int(ptr[-1]), torch.from_numpy(ptr.astype(np.int64))


def _layer(kind, hidden, H, B, aggrs, asl=True):
    import egc_amd
    if kind == "opt":
        conv = egc_amd.EGConv(hidden, hidden, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=asl)
    else:
        conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs, add_self_loops=asl)
    with torch.no_grad():
        conv.bias.normal_()
    return conv


@pytest.mark.parametrize("kind,hidden,H,B,aggrs,asl", [
    ("opt", 128, 8, 4, ["sum", "mean", "max", "symnorm"], True),       # north star (static configuration)
    ("lay", 128, 8, 4, ["symadd", "max", "mean"], True),               # EfficientGraphConv EGC-M (static)
    ("lay", 128, 8, 4, ["symadd"], True),
    ("opt", 64, 4, 4, ["min", "std", "var"], True),                    # run-time configuration, NEED_SQ | NEED_MN
    ("opt", 96, 4, 2, ["sum", "max"], True),                           # no symnorm: loops from add_remaining_self_loops (max index)
    ("opt", 128, 8, 4, ["symnorm", "mean"], False),                    # RAW sets
    ("lay", 168, 8, 4, ["symadd"], True),                              # padded bases (L = 21)
    ("lay", 124, 4, 4, ["add", "std", "max"], True),                   # L = 31, 32 slots
])
def test_tile_path_matches_the_oracle_on_a_messy_batch(kind, hidden, H, B, aggrs, asl):
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(hidden + len(aggrs))
    torch.manual_seed(1)
    conv = _layer(kind, hidden, H, B, aggrs, asl)
    x = torch.randn(n, hidden)
    ref = _oracle(conv, kind, x, ei, H, B, aggrs, asl)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        out = conv(x.to(dev), gb) if kind == "opt" else conv(x=x.to(dev), edge_index=gb)
        plain = conv(x.to(dev), ei.to(dev)) if kind == "opt" else conv(x=x.to(dev), edge_index=ei.to(dev))
    gb.check()
    assert gb._plans, "the tile path did not run"
    tol = 1e-4 if any(a in ("std", "var") for a in aggrs) else TOL
    assert rel_err(out.cpu().numpy(), ref) <= tol, rel_err(out.cpu().numpy(), ref)
    assert elementwise_excess(out.cpu().numpy(), ref, tol) <= 1.0
    assert rel_err(out.cpu().numpy(), plain.cpu().numpy()) <= tol


@pytest.mark.parametrize("workload", ["molhiv", "cifar", "zinc"])
def test_tile_path_at_full_batch_sizes(workload):
    """BASELINE configs 3 / 4 (2048 molecules; 2048 superpixel 8-NN graphs, 1.93 M edges) and a ZINC batch of 128, north-star
    layer through EGConv, batch given by its `batch` vector; whole output against the oracle."""
    import egc_amd
    from egc_amd.workloads import knn_superpixel_batch, molecule_batch, zinc_like_batch
    dev = _dev()
    if workload == "molhiv":
        ei, n, batch = molecule_batch(2048, seed=0); G, mx = 2048, 222
    elif workload == "cifar":
        ei, n, batch = knn_superpixel_batch(2048, seed=0); G, mx = 2048, 150
    else:
        _, ei, n, batch = zinc_like_batch(128, seed=0); G, mx = 128, 37
    torch.manual_seed(3)
    aggrs = ["sum", "mean", "max", "symnorm"]
    conv = _layer("opt", 128, 8, 4, aggrs)
    x = torch.randn(n, 128)
    ref = _oracle(conv, "opt", x, ei, 8, 4, aggrs)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), batch=batch.to(dev), num_graphs=G, max_nodes=mx)
    with torch.no_grad():
        out = conv(x.to(dev), gb)
    gb.check()
    assert rel_err(out.cpu().numpy(), ref) <= TOL, rel_err(out.cpu().numpy(), ref)
    assert elementwise_excess(out.cpu().numpy(), ref, TOL) <= 1.0


def test_fused_block_tail_on_the_tile_path():
    """FusedEGCBlock (eval): BatchNorm affine + ReLU + residual in the tile kernel's store == the plain composition."""
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
    assert float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())) <= 1e-5


def test_tiles_larger_than_the_lds_area_gather_from_memory():
    """A batch whose graphs are much larger than its average promises (one 900-node graph among small ones): its tile does
    not fit the LDS area and takes the memory-gather form of the same kernel -- same results."""
    import egc_amd
    dev = _dev()
    rng = np.random.default_rng(21)
    sizes = np.array([20] * 150 + [900] + [20] * 150)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    srcs, dsts = [], []
    for g, n in enumerate(sizes):
        e = 5 * n
        srcs.append(rng.integers(0, n, size=e) + ptr[g])
        dsts.append(rng.integers(0, n, size=e) + ptr[g])
    ei = torch.from_numpy(np.stack([np.concatenate(srcs), np.concatenate(dsts)]).astype(np.int64))
    n = int(ptr[-1])
    aggrs = ["sum", "mean", "max", "symnorm"]
    torch.manual_seed(8)
    conv = _layer("opt", 128, 8, 4, aggrs)
    x = torch.randn(n, 128)
    ref = _oracle(conv, "opt", x, ei, 8, 4, aggrs)
    conv = conv.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=torch.from_numpy(ptr.astype(np.int64)).to(dev), max_nodes=900)
    with torch.no_grad():
        out = conv(x.to(dev), gb)
    gb.check()
    lds_nodes = next(iter(gb._setups.values()))[1]
    assert lds_nodes < 900
    assert rel_err(out.cpu().numpy(), ref) <= TOL


def test_training_through_a_graph_batch_uses_the_csr_path():
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(9, n_graphs=40)
    torch.manual_seed(4)
    conv = _layer("opt", 64, 8, 4, ["sum", "mean", "max", "symnorm"]).to(dev).train()
    x = torch.randn(n, 64, device=dev, requires_grad=True)
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=90)
    out = conv(x, gb)
    out.sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    conv.zero_grad()
    conv(x, ei.to(dev)).sum().backward()
    assert torch.allclose(g1, x.grad, rtol=0, atol=1e-5 * max(1.0, float(g1.abs().max())))


def test_malformed_batches_are_reported():
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
    with pytest.raises(RuntimeError, match="not grouped by graph"):
        gb.check()
    # (2) edge list shuffled across graphs
    perm = torch.randperm(ei.size(1))
    gb = egc_amd.GraphBatch(ei[:, perm].to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        conv(x, gb)
    with pytest.raises(RuntimeError, match="not grouped by graph"):
        gb.check()
    # (3) a graph larger than max_nodes promises -> a tile beyond the per-tile CSR areas
    big_ptr = torch.tensor([0, n])
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=big_ptr.to(dev), max_nodes=16)
    with torch.no_grad():
        conv(x, gb)
    with pytest.raises(RuntimeError, match="exceeds the"):
        gb.check()
    # (4) without an explicit check the error surfaces at the next call into the package
    gb = egc_amd.GraphBatch(bad.to(dev), ptr=ptr.to(dev), max_nodes=90)
    with torch.no_grad():
        conv(x, gb)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="out of range"), torch.no_grad():
        conv(x, ei.to(dev))
    with torch.no_grad():
        conv(x, ei.to(dev))


def test_non_monotone_graph_offsets_cannot_overrun_the_tile_list():
    """with edge_ptr the plan writes one record per tile HEAD; graph offsets that jump back and forth make nearly
    every graph a head -- more heads than the list has records.  The plan stops at the list's capacity, the tile kernel
    reports the batch, and nothing outside the list is written (a canary behind it stays intact)."""
    import ctypes as C
    import egc_amd
    from egc_amd import _C
    dev = _dev()
    lib = _C.load()
    n, n_graphs, slot = 4000, 200, 100
    ptr = torch.zeros(n_graphs + 1, dtype=torch.int64)
    ptr[1::2] = 3000                                      # 0, 3000, 0, 3000, ...: every graph starts in another slot than its predecessor
    ptr[-1] = n
    eptr = torch.zeros(n_graphs + 1, dtype=torch.int64)
    n_slots = (n + slot - 1) // slot
    buf = torch.full((4 * n_slots + 4 + 64,), -7, dtype=torch.int32, device=dev)
    tiles, count, canary = buf[:4 * n_slots], buf[4 * n_slots:4 * n_slots + 4], buf[4 * n_slots + 4:]
    dst = torch.zeros(1, dtype=torch.int64, device=dev)
    ptr_d, eptr_d = ptr.to(dev), eptr.to(dev)              # (kept alive across the call: the C ABI takes raw pointers)
    _C.check(lib.egc_batch_plan(ptr_d.data_ptr(), eptr_d.data_ptr(), n_graphs, dst.data_ptr(), 0, n, slot,
                                tiles.data_ptr(), n_slots, count.data_ptr(), torch.cuda.current_stream().cuda_stream), "egc_batch_plan")
    torch.cuda.synchronize()
    assert int(count[0]) > n_slots                         # more heads than records ...
    assert bool((canary == -7).all()) and bool((count[1:] == -7).all())   # ... and none written past the list
    # through the layer: reported, not silently wrong
    ei, _, _ = _messy_batch(12, n_graphs=60)
    gb = egc_amd.GraphBatch(torch.zeros((2, 0), dtype=torch.int64, device=dev), ptr=ptr.to(dev), num_nodes=n, max_nodes=90,
                            edge_ptr=eptr.to(dev))
    conv = _layer("opt", 128, 8, 4, ["sum", "mean", "max", "symnorm"]).to(dev).eval()
    with torch.no_grad():
        conv(torch.randn(n, 128, device=dev), gb)
    with pytest.raises(RuntimeError):
        gb.check()


def test_rows_of_a_reported_tile_are_zero_not_garbage():
    import egc_amd
    dev = _dev()
    ei, n, ptr = _messy_batch(12, n_graphs=60)
    conv = _layer("opt", 128, 8, 4, ["sum", "mean", "max", "symnorm"]).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    big_ptr = torch.tensor([0, n])                          # one "graph" far beyond max_nodes
    for env in ("0", "1"):
        import os
        os.environ["EGC_NO_FUSED_TILE"] = env
        try:
            gb = egc_amd.GraphBatch(ei.to(dev), ptr=big_ptr.to(dev), max_nodes=16)
            junk = torch.full((n, 128), float("nan"), device=dev)       # make sure the allocator hands out non-zero memory
            del junk
            with torch.no_grad():
                out = conv(x, gb)
            with pytest.raises(RuntimeError, match="exceeds the"):
                gb.check()
            assert bool((out == 0).all())
        finally:
            os.environ.pop("EGC_NO_FUSED_TILE", None)
