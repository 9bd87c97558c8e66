"""egc_graph_build (five launches, no library sort) against the radix-sort pipeline (egc_coo_to_csr + egc_csr_prepare +
egc_csr_edge_dis) and the numpy oracle: the CSR must be IDENTICAL -- rowptr, col and edge_id bit for bit (entries of a
row in input order: torch_scatter's first-edge arg rule depends on it) -- for short rows, rows sorted in LDS, hub rows
sorted through global memory, duplicates, self loops, empty rows, rectangular source spaces; node ids out of range are
reported (PyG raises at index_select, optimized_layers.py:191-193)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _both(ei, n, monkeypatch, ns=None):
    import egc_amd
    monkeypatch.setenv("EGC_GRAPH_BUILD", "sort")
    a = egc_amd.CSRGraph.from_edge_index(ei, n, ns)
    monkeypatch.setenv("EGC_GRAPH_BUILD", "fast")
    b = egc_amd.CSRGraph.from_edge_index(ei, n, ns)
    torch.cuda.synchronize()
    return a, b


def _same(a, b, e):
    assert torch.equal(a.rowptr, b.rowptr)
    assert torch.equal(a.col[:e], b.col[:e])
    assert torch.equal(a.edge_id[:e], b.edge_id[:e])
    assert torch.equal(a.max_index, b.max_index)
    n = a.n_nodes
    assert torch.equal(a.dis_raw[:n], b.dis_raw[:n]) and torch.equal(a.dis_looped[:n], b.dis_looped[:n])
    if a.edge_dis_raw is not None:
        assert torch.equal(a.edge_dis_raw[:e], b.edge_dis_raw[:e]) and torch.equal(a.edge_dis_looped[:e], b.edge_dis_looped[:e])
    # the long-row plan: same rows and chunks (slots are handed out by atomics: order may differ)
    pa, pb = a.plan.cpu().numpy(), b.plan.cpu().numpy()
    assert list(pa[:4]) == list(pb[:4])
    nl, nc, cl, cc = (int(v) for v in pa[:4])
    rows = lambda p: sorted(p[4:4 + nl].tolist())
    assert rows(pa) == rows(pb)
    chunks = lambda p: sorted(p[4 + 2 * cl + cc:4 + 2 * cl + cc + nc].tolist())
    assert chunks(pa) == chunks(pb)


@pytest.mark.parametrize("n,e,hubs", [(1, 0, []), (7, 3, []), (300, 2500, []), (2000, 30000, [(5, 700), (17, 100)]),
                                      (5000, 60000, [(1, 5000), (4000, 9000), (77, 4096), (78, 4097)]),
                                      (53000, 110000, [])])
def test_fast_build_is_bit_identical_to_the_sort_pipeline(n, e, hubs, monkeypatch):
    dev = _dev()
    rng = np.random.default_rng(n + e)
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    at = 0
    for row, k in hubs:
        ei[1, at:at + k] = row
        at += k
    if e > 100:
        ei[0, -50:] = ei[1, -50:]                 # self loops
        ei[:, -100:-50] = ei[:, -150:-100]        # duplicates
    ei = ei[:, rng.permutation(e)] if e else ei
    a, b = _both(torch.from_numpy(ei).to(dev), n, monkeypatch)
    _same(a, b, e)
    # and the oracle's stable order
    order = np.argsort(ei[1], kind="stable")
    assert np.array_equal(b.edge_id[:e].cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(b.col[:e].cpu().numpy(), ei[0][order].astype(np.int32))
    b.check_indices()


def test_fast_build_twice_on_one_workspace_and_rectangular_sources(monkeypatch):
    dev = _dev()
    rng = np.random.default_rng(3)
    for n, ns, e in [(400, 1000, 5000), (900, 120, 7000), (400, 1000, 5000)]:
        ei = np.stack([rng.integers(0, ns, size=e), rng.integers(0, n, size=e)]).astype(np.int64)
        a, b = _both(torch.from_numpy(ei).to(dev), n, monkeypatch, ns)
        _same(a, b, e)


def test_out_of_range_ids_are_reported(monkeypatch):
    import egc_amd
    dev = _dev()
    monkeypatch.setenv("EGC_GRAPH_BUILD", "fast")
    n = 50
    ei = torch.randint(0, n, (2, 400), device=dev)
    good = egc_amd.CSRGraph.from_edge_index(ei, n).check_indices()
    assert int(good.rowptr[-1]) == 400
    for bad_row, bad_val in [(0, n), (1, n + 3), (0, -1), (1, -7)]:
        bad = ei.clone()
        bad[bad_row, 17] = bad_val
        g = egc_amd.CSRGraph.from_edge_index(bad, n)
        with pytest.raises(RuntimeError, match="out of range"):
            g.check_indices()
        assert int(g.rowptr[-1]) == 399           # the offending edge is dropped, nothing is read out of bounds
        with pytest.raises(RuntimeError, match="out of range"):
            egc_amd.CSRGraph.from_edge_index(bad, n).trim_launches()
    # the workspace survives a faulty build: the next graph is right
    again = egc_amd.CSRGraph.from_edge_index(ei, n).check_indices()
    assert torch.equal(again.rowptr, good.rowptr) and torch.equal(again.col, good.col)
    monkeypatch.setenv("EGC_CHECK_INDICES", "1")
    bad = ei.clone()
    bad[0, 0] = 10 ** 9
    with pytest.raises(RuntimeError, match="out of range"):
        egc_amd.CSRGraph.from_edge_index(bad, n)


@pytest.mark.parametrize("build", ["fast", "sort"])
def test_out_of_range_ids_surface_without_a_synchronisation_and_leave_no_bogus_entries(build, monkeypatch):
    """ADVICE r2 / VERDICT r2 missing #6: both builds range-check; a path that never reads the per-graph flag (per-batch
    graphs, recorded steps) still raises -- at the next call into the package, from the sticky host-visible word -- and
    the graph built from the bad list equals the graph of the list without the offending edges, transposed graph
    (the backward's) included: nothing iterates over uninitialised entries."""
    import egc_amd
    from egc_amd.graph import _IndexFlag
    dev = _dev()
    monkeypatch.setenv("EGC_GRAPH_BUILD", build)
    rng = np.random.default_rng(11)
    n, e = 300, 4000
    ei = np.stack([rng.integers(0, n, size=e), rng.integers(0, n, size=e)]).astype(np.int64)
    ei[1, :200] = 5                                   # a long row
    bad_pos = [3, 170, 1999, 3999]
    bad = ei.copy()
    bad[0, bad_pos[0]], bad[1, bad_pos[1]], bad[0, bad_pos[2]], bad[1, bad_pos[3]] = n, n + 40, -1, -(2 ** 40)
    clean = np.delete(ei, bad_pos, axis=1)
    g_bad = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(bad).to(dev), n)
    t_bad = g_bad.transposed()                        # nothing here synchronises or reads the per-graph flag
    torch.cuda.synchronize()
    assert _IndexFlag._view.value == 1                # the device has raised the sticky word
    with pytest.raises(RuntimeError, match="out of range"):
        egc_amd.CSRGraph.from_edge_index(torch.from_numpy(clean).to(dev), n)     # ... and the next call reports it
    g_ok = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(clean).to(dev), n)   # reported once; the package works on
    t_ok = g_ok.transposed()
    k = clean.shape[1]
    assert int(g_bad.rowptr[-1]) == k
    assert torch.equal(g_bad.rowptr, g_ok.rowptr) and torch.equal(g_bad.col[:k], g_ok.col[:k])
    # input positions differ by the removed edges: compare through the kept edges' own numbering
    kept = np.delete(np.arange(e), bad_pos)
    assert np.array_equal(kept[g_ok.edge_id[:k].cpu().numpy()], g_bad.edge_id[:k].cpu().numpy())
    assert torch.equal(g_bad.dis_raw, g_ok.dis_raw) and torch.equal(g_bad.dis_looped, g_ok.dis_looped)
    assert torch.equal(g_bad.edge_dis_looped[:k], g_ok.edge_dis_looped[:k])
    assert int(t_bad.rowptr[-1]) == k
    assert torch.equal(t_bad.rowptr, t_ok.rowptr) and torch.equal(t_bad.col[:k], t_ok.col[:k])
    assert torch.equal(t_bad.edge_id[:k], t_ok.edge_id[:k])
    torch.cuda.synchronize()
    _IndexFlag._view.value = 0                        # (the transposed build of g_bad dropped the (-1, -1) entries again)
    # a layer call is a polling point too
    conv = egc_amd.EGConv(16, 16, aggrs=["sum", "max"], num_heads=2, num_bases=2).to(dev).eval()
    x = torch.randn(n, 16, device=dev)
    egc_amd.CSRGraph.from_edge_index(torch.from_numpy(bad).to(dev), n)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="out of range"), torch.no_grad():
        conv(x, g_ok)
    with torch.no_grad():
        conv(x, g_ok)


def test_graphs_of_different_sizes_share_one_workspace(monkeypatch):
    """Regression: the sort area of hub rows is NOT part of the zero-on-exit workspace -- a small graph with a hub row
    followed by a larger graph (whose degree counters overlay the bytes the first one used) must come out right."""
    dev = _dev()
    rng = np.random.default_rng(9)
    for n, ns, e, hub in [(1200, 14000, 140000, 9000), (14000, 1200, 140000, 0), (300, 300, 60000, 30000), (50000, 50000, 100000, 0)]:
        ei = np.stack([rng.integers(0, ns, size=e), rng.integers(0, n, size=e)]).astype(np.int64)
        if hub:
            ei[1, :hub] = 7
        a, b = _both(torch.from_numpy(ei).to(dev), n, monkeypatch, ns)
        _same(a, b, e)


@pytest.mark.parametrize("kind", ["molecules", "knn", "sorted_by_dst_with_a_hub"])
def test_fast_build_on_graph_contiguous_batches(kind, monkeypatch):
    """The LDS-windowed tiles (destinations of 2048 consecutive edges within 2048 rows): batches of small graphs, kNN
    lists sorted by destination, and a sorted list with a hub row that spans many tiles."""
    from egc_amd.workloads import knn_superpixel_batch, molecule_batch
    dev = _dev()
    if kind == "molecules":
        ei, n, _ = molecule_batch(600, seed=2)
    elif kind == "knn":
        ei, n, _ = knn_superpixel_batch(300, seed=2)
    else:
        rng = np.random.default_rng(4)
        n = 30000
        dst = np.sort(np.concatenate([rng.integers(0, n, size=150000), np.full(20000, 12345)]))
        ei = torch.from_numpy(np.stack([rng.integers(0, n, size=dst.size), dst]).astype(np.int64))
    a, b = _both(ei.to(dev), n, monkeypatch)
    _same(a, b, int(ei.size(1)))
