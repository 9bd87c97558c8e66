"""GPU parity tests: the HIP path (through the C ABI, via the drop-in modules) against
  (1) the committed golden vectors produced by the reference's own layer code, and
  (2) the numpy oracle on seeded inputs (edge cases, long rows, wide shapes, full config-2 size).

Tolerance (north_star): fp32 results within 1e-5 of the reference, measured scale-relative
(max |diff| / max(1, max |ref|)); integer outputs (CSR, argmax) bit-exact.
"""
import numpy as np
import pytest
import torch

from golden_util import elementwise_excess, float64_forward, golden_names, load_golden, oracle_forward, rel_err
from oracle import egc_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def build_layer(meta, params, dev):
    import egc_amd
    if meta["kind"] == "lay":
        layer = egc_amd.EfficientGraphConv(
            meta["fin"], meta["fout"], num_heads=meta["H"], num_bases=meta["B"], softmax_weights=meta["softmax"],
            add_self_loops=meta["add_self_loops"], bias=meta["bias"], aggrs=meta["aggrs"],
            sigmoid_weights=meta["sigmoid"], hardtanh_weights=meta["hardtanh"])
    else:
        layer = egc_amd.EGConv(meta["fin"], meta["fout"], aggrs=meta["aggrs"], num_heads=meta["H"],
                               num_bases=meta["B"], add_self_loops=meta["add_self_loops"], bias=meta["bias"],
                               sigmoid=meta["sigmoid"])
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return layer.to(dev).eval()


def run_layer(layer, g, dev):
    import egc_amd
    x = torch.from_numpy(g["x"]).to(dev)
    ei = torch.from_numpy(g["edge_index"]).to(dev)
    n = g["x"].shape[0]
    if g["meta"]["sparse"]:
        arg = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(n, n))
    else:
        arg = ei
    with torch.no_grad():
        out = layer(x, arg) if g["meta"]["kind"] == "opt" else layer(x=x, edge_index=arg)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", golden_names())
def test_golden(name):
    """HIP layer == output of the reference's own layer code on the committed fixture."""
    dev = _dev()
    g = load_golden(name)
    layer = build_layer(g["meta"], g["params"], dev)
    out = run_layer(layer, g, dev)
    assert out.shape == g["out"].shape
    assert rel_err(out, g["out"]) <= TOL, f"{name}: rel err {rel_err(out, g['out']):.3e}"
    # element by element against the element's own row scale (small rows next to large ones) -- std / var layers included
    # since round 6 (tools/stdvar_modes.py: largest excess 0.32 at 1e-5 over the 16 std / var fixtures, with the default
    # formula and with EGC_STDVAR_REFERENCE=1 alike; rounds 4-5 allowed 1e-4 there)
    assert elementwise_excess(out, g["out"], TOL) <= 1.0, f"{name}: element-wise excess {elementwise_excess(out, g['out'], TOL):.3f}"
    # and against the oracle (pins the oracle and the HIP path to each other as well)
    assert rel_err(out, oracle_forward(g, orc)) <= TOL


def _stdvar_goldens():
    return [n for n in golden_names() if any(a in ("std", "var") for a in load_golden(n)["meta"]["aggrs"])]


@pytest.mark.parametrize("name", _stdvar_goldens())
def test_golden_stdvar_reference_formula(name, monkeypatch):
    """EGC_STDVAR_REFERENCE=1: std / var by the reference's own float32 formula, mean(x^2) - mean(x)^2 (layers.py:203-214,
    optimized_layers.py:237-244; SURVEY.md 8a note 5: "match the formula, not a better one") -- the general kernels, one row per
    wavefront, squares about zero, 24-bit GEMM operands -- against the fixture the reference's own code produced: 1e-5 on the
    array's scale AND element by element (the default formula, about the row's first entry, is held to 1e-4 element-wise there:
    it is nearer float64 than the fixture is)."""
    monkeypatch.setenv("EGC_STDVAR_REFERENCE", "1")
    dev = _dev()
    g = load_golden(name)
    layer = build_layer(g["meta"], g["params"], dev)
    out = run_layer(layer, g, dev)
    assert rel_err(out, g["out"]) <= TOL, f"{name}: rel err {rel_err(out, g['out']):.3e}"
    ex = elementwise_excess(out, g["out"], TOL)
    assert ex <= 1.0, f"{name}: element-wise excess {ex:.3f} at 1e-5 with the reference's formula"


def test_repr_matches_reference_goldens():
    for name in golden_names():
        g = load_golden(name)
        layer = build_layer(g["meta"], g["params"], torch.device("cpu"))
        if g["meta"]["kind"] == "opt":
            assert repr(layer) == g["meta"]["repr"]


def _rand_graph(rng, n, e):
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    return ei


@pytest.mark.parametrize("n,e", [(1, 0), (7, 0), (64, 500), (1000, 20000), (50000, 400000)])
def test_coo_to_csr_bit_exact(n, e):
    """CSR build is integer work: rowptr / col / edge_id must equal the stable-sort oracle exactly."""
    import egc_amd
    dev = _dev()
    rng = np.random.default_rng(n + e)
    ei = _rand_graph(rng, n, e)
    g = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(ei).to(dev), n)
    torch.cuda.synchronize()
    rowptr, col, eid = orc.csr_from_coo(ei, n)
    assert np.array_equal(g.rowptr.cpu().numpy().astype(np.int64), rowptr)
    if e:
        assert np.array_equal(g.col.cpu().numpy()[:e].astype(np.int64), col)
        assert np.array_equal(g.edge_id.cpu().numpy()[:e].astype(np.int64), eid)
        assert int(g.max_index.cpu()) == int(ei.max())
    else:
        assert int(g.max_index.cpu()) == -1
    # degree statistics (gcn_norm's deg^-1/2), raw and self-looped conventions
    deg = np.diff(rowptr).astype(np.float32)
    with np.errstate(divide="ignore"):
        dis_raw = np.where(deg > 0, 1.0 / np.sqrt(deg), 0).astype(np.float32)
    nonself = np.bincount(ei[1][ei[0] != ei[1]], minlength=n).astype(np.float32)
    dis_looped = (1.0 / np.sqrt(nonself + 1)).astype(np.float32)
    np.testing.assert_allclose(g.dis_raw.cpu().numpy()[:n], dis_raw, rtol=2e-7)
    np.testing.assert_allclose(g.dis_looped.cpu().numpy()[:n], dis_looped, rtol=2e-7)


def _oracle_case(kind, rng, n, ei, fin, fout, H, B, aggrs, dev, **flags):
    """Random parameters -> (HIP output, oracle output)."""
    import egc_amd
    torch.manual_seed(int(rng.integers(1 << 30)))
    if kind == "lay":
        layer = egc_amd.EfficientGraphConv(fin, fout, H, B, flags.get("softmax", False), aggrs=aggrs,
                                           add_self_loops=flags.get("add_self_loops", True),
                                           sigmoid_weights=flags.get("sigmoid", False),
                                           hardtanh_weights=flags.get("hardtanh", False))
    else:
        layer = egc_amd.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B,
                               add_self_loops=flags.get("add_self_loops", True), sigmoid=flags.get("sigmoid", False))
    with torch.no_grad():
        layer.bias.normal_()
    x = rng.standard_normal((n, fin)).astype(np.float32)
    sd = {k: v.numpy() for k, v in layer.state_dict().items()}
    meta = dict(kind=kind, fin=fin, fout=fout, H=H, B=B, aggrs=aggrs, softmax=flags.get("softmax", False),
                sigmoid=flags.get("sigmoid", False), hardtanh=flags.get("hardtanh", False),
                add_self_loops=flags.get("add_self_loops", True), bias=True, sparse=False)
    g = dict(meta=meta, params=sd, x=x, edge_index=ei)
    out = run_layer(layer.to(dev).eval(), g, dev)
    if flags.get("check64", True):          # (every sweep, round 6; pass check64=False to opt out)
        # every element against a float64 evaluation of the reference's formula, on the scale of the element's OWN row (rel_err
        # alone measures against the largest output of the whole array and says nothing about small rows next to large ones)
        ex = elementwise_excess(out, float64_forward(g), TOL)
        assert ex <= 1.0, f"element-wise excess {ex:.3f} against float64 ({kind} {fin} H{H} B{B} {aggrs})"
    return out, oracle_forward(g, orc)


def _hub_graph(rng, n, e, hubs):
    """Random graph plus a few destination hubs with in-degree far above the long-row threshold."""
    parts = [rng.integers(0, n, size=(2, e))]
    for node, deg in hubs:
        parts.append(np.stack([rng.integers(0, n, size=deg), np.full(deg, node)]))
    ei = np.concatenate(parts, axis=1).astype(np.int64)
    return ei[:, rng.permutation(ei.shape[1])]


@pytest.mark.parametrize("kind,aggrs", [
    ("opt", ["sum", "mean", "max", "symnorm"]), ("opt", ["min", "std", "var"]),
    ("lay", ["symadd", "std", "max"]), ("lay", ["add", "mean", "min", "var"]),
])
def test_long_rows_chunk_merge_path(kind, aggrs):
    """Rows above EGC_LONG_ROW_THRESHOLD go through the chunk + merge kernels (hubs around every boundary of
    the threshold and the chunk length, up to 5000 entries)."""
    dev = _dev()
    rng = np.random.default_rng(7)
    n = 3000
    from egc_amd import _C
    T, K = _C.LONG_ROW_THRESHOLD, _C.LONG_ROW_CHUNK
    hubs = [(0, 5000), (17, K + 1), (18, K), (19, T), (20, T + 1), (21, 2 * K), (22, 2 * K + 1), (999, 1300), (n - 1, 257), (5, 4096)]
    ei = _hub_graph(rng, n, 12000, hubs)
    out, ref = _oracle_case(kind, rng, n, ei, 64, 64, 8, 4, aggrs, dev, check64=True)
    assert rel_err(out, ref) <= TOL


@pytest.mark.parametrize("hidden,H,B", [(304, 8, 8), (300, 4, 4), (352, 8, 4), (168, 8, 4), (124, 4, 4),
                                        (296, 8, 4), (224, 4, 4), (136, 4, 4), (21, 1, 1), (16, 16, 16), (6, 2, 1),
                                        # power-of-two shapes -> register-resident kernel family
                                        (128, 8, 4), (128, 4, 4), (256, 4, 4), (64, 4, 4), (128, 16, 8), (64, 2, 8),
                                        (64, 1, 1), (512, 8, 2), (32, 2, 4)])
def test_shipped_and_odd_shapes(hidden, H, B):
    """(hidden, H, B) from run_pretrained.sh / train_main_table.sh incl. basis widths > 256 floats
    (multi-slot lanes) and widths that are not multiples of 4 (padded leading dimension)."""
    dev = _dev()
    rng = np.random.default_rng(hidden * 31 + H)
    n = 400
    ei = _hub_graph(rng, n, 3000, [(3, 300)])
    out, ref = _oracle_case("opt", rng, n, ei, hidden, hidden, H, B, ["symnorm", "max", "std"], dev, check64=True)
    assert rel_err(out, ref) <= TOL
    out, ref = _oracle_case("lay", rng, n, ei, hidden, hidden, H, B, ["symadd", "min", "mean"], dev, softmax=True, check64=True)
    assert rel_err(out, ref) <= TOL


@pytest.mark.parametrize("hidden,H,B", [(352, 8, 4), (224, 4, 4), (96, 8, 4), (48, 2, 2), (40, 2, 1), (480, 8, 4),
                                        # L % 4 != 0: per-basis padded rows, ragged head tails
                                        (184, 8, 4), (168, 8, 4), (124, 4, 4), (136, 4, 4), (21, 1, 1), (6, 2, 1)])
@pytest.mark.parametrize("kind,aggrs", [("opt", ["sum", "mean", "max", "symnorm"]), ("opt", ["min", "std"]),
                                        ("lay", ["symadd", "max", "mean"]), ("lay", ["add"])])
def test_register_kernels_on_non_power_of_two_rows(hidden, H, B, kind, aggrs):
    """Slot counts that are not powers of two (ogbn-mag 352/H8/B4: 44 slots, molhiv 224/H4/B4: 56) and basis
    lengths that are not multiples of 4 (arxiv 184/H8/B4: L = 23, zinc 168/H8/B4: L = 21, ...): idle lanes,
    division-based lane mapping, rotation butterfly, padded bases with ragged head tails -- short rows, chunked
    hub rows, and the result must equal the LDS-based generic kernels' apart from summation order."""
    import os
    dev = _dev()
    rng = np.random.default_rng(hidden + 13 * len(aggrs))
    n = 1500
    ei = _hub_graph(rng, n, 9000, [(0, 2100), (77, 129), (n - 1, 40), (400, 33)])
    out, ref = _oracle_case(kind, rng, n, ei, hidden, hidden, H, B, aggrs, dev, check64=True)
    assert rel_err(out, ref) <= TOL
    os.environ["EGC_FORCE_GENERIC"] = "1"
    try:
        rng = np.random.default_rng(hidden + 13 * len(aggrs))
        ei = _hub_graph(rng, n, 9000, [(0, 2100), (77, 129), (n - 1, 40), (400, 33)])
        out_g, _ = _oracle_case(kind, rng, n, ei, hidden, hidden, H, B, aggrs, dev)
    finally:
        del os.environ["EGC_FORCE_GENERIC"]
    assert rel_err(out, out_g) <= 2e-6


@pytest.mark.parametrize("hidden,H,B,aggrs", [(300, 4, 4, ["symadd", "min", "max"]), (304, 8, 8, ["symadd"]),   # compile-time forms
                                             (300, 4, 4, ["add", "std", "max"]), (304, 8, 8, ["mean", "var", "min", "symadd"]),
                                             (260, 4, 4, ["symadd", "max"]), (400, 4, 4, ["symadd", "mean"]), (1000, 8, 4, ["max"]),
                                             (600, 4, 2, ["symadd", "min", "max"]), (272, 2, 2, ["add"]),
                                             (1056, 8, 2, ["symadd", "max"])])
def test_two_slots_per_lane_kernel(hidden, H, B, aggrs):
    """Rows of 65..128 16-byte slots (the ogbg-code nets of run_pretrained.sh:47-48: 300/H4/B4 symadd,min,max and
    304/H8/B8 symadd, and other widths) on agg_wide_kernel: two slots per lane, both sets' epilogues, short rows and
    hub rows cut into chunks (boundaries of the threshold and the chunk length) in ONE launch; against the numpy oracle,
    and against the LDS-based kernels and the run-time-configured form of the same kernel."""
    import os
    from egc_amd import _C
    dev = _dev()
    T, K = _C.LONG_ROW_THRESHOLD, _C.LONG_ROW_CHUNK
    n = 1200
    hubs = [(0, 3 * K + 7), (17, K + 1), (18, K), (19, T), (20, T + 1), (21, 2 * K), (n - 1, 2 * K + 1), (400, 63), (401, 64), (402, 65)]

    def run():
        rng = np.random.default_rng(hidden * 7 + len(aggrs))
        ei = _hub_graph(rng, n, 7000, hubs)
        return _oracle_case("lay", rng, n, ei, 48, hidden, H, B, aggrs, dev)

    out, ref = run()
    assert rel_err(out, ref) <= TOL
    for flag in ("EGC_NO_WIDE", "EGC_NO_STATIC_CFG"):
        os.environ[flag] = "1"
        try:
            other, _ = run()
        finally:
            del os.environ[flag]
        assert rel_err(out, other) <= 2e-6, flag
    out2, _ = run()
    assert np.array_equal(out, out2)       # chunk-order merge: run-to-run identical


def test_fin_not_multiple_of_4_and_fin_ne_fout():
    dev = _dev()
    rng = np.random.default_rng(3)
    n = 257
    ei = _rand_graph(rng, n, 2000)
    for fin, fout in [(5, 32), (37, 64), (128, 352), (130, 16), (64, 8), (32, 16), (96, 24)]:
        out, ref = _oracle_case("opt", rng, n, ei, fin, fout, 8, 4, ["sum", "max"], dev)
        assert rel_err(out, ref) <= TOL


def test_intermediates_match_oracle():
    """bases / weightings written by the MFMA GEMM == oracle's fp32 matmuls (within GEMM reordering)."""
    import egc_amd
    from egc_amd.functional import egc_layer_forward
    dev = _dev()
    g = load_golden("opt_northstar_small")
    layer = build_layer(g["meta"], g["params"], dev)
    x = torch.from_numpy(g["x"]).to(dev)
    graph = egc_amd.CSRGraph.from_edge_index(torch.from_numpy(g["edge_index"]).to(dev), x.size(0))
    with torch.no_grad():
        wcat, bcat = layer._packed_weights()
        out, bases, weightings = egc_layer_forward(graph, layer._spec_coo, x, wcat, bcat, layer.bias,
                                                   return_intermediates=True)
    _, inter = oracle_forward(g, orc, return_intermediates=True)
    assert rel_err(bases.cpu().numpy()[:, :inter["bases"].shape[1]], inter["bases"]) <= TOL
    # internal weightings order is [h][b][a]; the reference's is [h][a][b] (optimized_layers.py:195-202)
    m = g["meta"]
    H, B, A = m["H"], m["B"], len(m["aggrs"])
    w_ref = inter["weightings"].reshape(-1, H, A, B).transpose(0, 1, 3, 2).reshape(-1, H * B * A)
    assert rel_err(weightings.cpu().numpy(), w_ref) <= TOL
    assert rel_err(out.cpu().numpy(), g["out"]) <= TOL


def test_cached_graph_and_shared_graph_cache():
    """cached=True pins the first graph (optimized_layers.py:138-139); un-cached layers share one
    CSR per edge_index tensor through the global cache."""
    import egc_amd
    dev = _dev()
    rng = np.random.default_rng(11)
    n = 100
    ei1 = torch.from_numpy(_rand_graph(rng, n, 700)).to(dev)
    ei2 = torch.from_numpy(_rand_graph(rng, n, 700)).to(dev)
    x = torch.randn(n, 32, device=dev)
    conv = egc_amd.EGConv(32, 32, aggrs=["symnorm", "max"], num_heads=4, num_bases=4, cached=True).to(dev)
    plain = egc_amd.EGConv(32, 32, aggrs=["symnorm", "max"], num_heads=4, num_bases=4).to(dev)
    plain.load_state_dict(conv.state_dict())
    with torch.no_grad():
        a1, a2 = conv(x, ei1), conv(x, ei2)      # second call must reuse graph 1
        b1, b2 = plain(x, ei1), plain(x, ei2)
    assert torch.equal(a1, a2) and torch.equal(a1, b1) and not torch.equal(b1, b2)
    conv.reset_parameters()
    assert conv._cached_graph is None
    g_a = egc_amd.graph.GLOBAL_GRAPH_CACHE.get(ei1, n)
    g_b = egc_amd.graph.GLOBAL_GRAPH_CACHE.get(ei1, n)
    assert g_a is g_b


def test_two_reference_layers_agree_through_hip():
    """SURVEY.md 8a notes 1-2: with W_opt = W_lay.view(H,B,A,F).permute(0,2,1,3) and pre-self-looped edges
    the two layer classes compute the same function."""
    import egc_amd
    dev = _dev()
    rng = np.random.default_rng(5)
    n, f, H, B = 300, 64, 8, 4
    ei = _rand_graph(rng, n, 3000)
    ei = ei[:, ei[0] != ei[1]]
    looped = np.concatenate([ei, np.stack([np.arange(n), np.arange(n)])], axis=1)
    lay = egc_amd.EfficientGraphConv(f, f, H, B, False, aggrs=["symadd", "max", "mean"]).to(dev)
    opt = egc_amd.EGConv(f, f, aggrs=["symnorm", "max", "mean"], num_heads=H, num_bases=B).to(dev)
    A = 3
    with torch.no_grad():
        opt.bases_weight.copy_(torch.cat(list(lay.bases_weight), dim=1))
        opt.comb_weight.weight.copy_(lay.comb_weights.weight.view(H, B, A, f).permute(0, 2, 1, 3).reshape(H * A * B, f))
        opt.comb_weight.bias.copy_(lay.comb_weights.bias.view(H, B, A).permute(0, 2, 1).reshape(-1))
        x = torch.randn(n, f, device=dev)
        o_lay = lay(x, torch.from_numpy(looped).to(dev))   # raw aggregators see the explicit loops
        o_opt = opt(x, torch.from_numpy(ei).to(dev))       # EGConv adds them itself
    assert rel_err(o_lay.cpu().numpy(), o_opt.cpu().numpy()) <= TOL


def test_config2_full_size_against_oracle():
    """BASELINE config 2 at full size (N = 169,343, ~2.33 M edges + self loops, d=128, H=8, B=4,
    sum+mean+max+symnorm): HIP vs numpy oracle, plus size-independent properties."""
    import egc_amd
    from egc_amd.workloads import arxiv_like
    dev = _dev()
    ei, n = arxiv_like(seed=0)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 128)
    sd = {k: v.numpy() for k, v in conv.state_dict().items()}
    ref = orc.egconv_forward(x.numpy(), ei.numpy(), sd["bases_weight"], sd["comb_weight.weight"],
                             sd["comb_weight.bias"], sd["bias"], 8, 4, conv.aggregators)
    conv = conv.to(dev).eval()
    with torch.no_grad():
        out = conv(x.to(dev), ei.to(dev))
        # property: permuting the edge list changes nothing for max, and sums only within rounding
        perm = torch.randperm(ei.size(1))
        out_p = conv(x.to(dev), ei[:, perm].to(dev))
    out, out_p = out.cpu().numpy(), out_p.cpu().numpy()
    assert rel_err(out, ref) <= TOL, rel_err(out, ref)
    assert rel_err(out_p, out) <= TOL


def test_linearity_and_max_idempotence_properties():
    """Size-independent properties on a config-2-shaped layer: sum/mean/symnorm are linear in x
    through the bases; max-aggregation of a constant feature field returns the constant."""
    import egc_amd
    from egc_amd.workloads import heavy_tailed_graph
    dev = _dev()
    n = 20000
    ei = heavy_tailed_graph(n, 150000, seed=1).to(dev)
    conv = egc_amd.EGConv(64, 64, aggrs=["sum", "mean", "symnorm"], num_heads=4, num_bases=4, bias=False).to(dev)
    with torch.no_grad():
        # make the weightings independent of x so the whole layer is linear in x
        conv.comb_weight.weight.zero_()
        conv.comb_weight.bias.normal_()
        x1, x2 = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
        lhs = conv(2.0 * x1 + x2, ei)
        rhs = 2.0 * conv(x1, ei) + conv(x2, ei)
    assert rel_err(lhs.cpu().numpy(), rhs.cpu().numpy()) <= 5e-5
    mx = egc_amd.EGConv(8, 8, aggrs=["max", "min"], num_heads=1, num_bases=1, bias=False).to(dev)
    with torch.no_grad():
        mx.bases_weight.copy_(torch.eye(8, device=dev))
        mx.comb_weight.weight.zero_()
        mx.comb_weight.bias.copy_(torch.tensor([1.0, 0.0], device=dev))  # picks 'max'
        const = torch.full((n, 8), 3.25, device=dev)
        assert torch.equal(mx(const, ei), const)


@pytest.mark.parametrize("world", [2, 8])
def test_vertex_partition_equals_single_gpu(world):
    """per-rank CSR rows + [owned | halo] basis rows reproduce the single-GPU output
    (the all-to-all-v is simulated inside one process; the collective itself is covered by the gloo test)."""
    import egc_amd
    from egc_amd import partition as P
    from egc_amd.functional import egc_aggregate_combine, egc_basis_transform
    from egc_amd.workloads import heavy_tailed_graph
    dev = _dev()
    n = 6000
    ei = heavy_tailed_graph(n, 40000, seed=2)
    torch.manual_seed(1)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    with torch.no_grad():
        ref = conv(x, ei.to(dev))
        wcat, bcat = conv._packed_weights()
        parts = P.build_local_simulation(ei, n, world)
        graphs = [egc_amd.CSRGraph.from_partition(e.to(dev), plan, global_max_index=int(ei.max()), exchange_dis=False)
                  for e, plan in parts]
        plans = [p for _, p in parts]
        for key in ("dis_raw", "dis_looped"):
            P.simulate_exchange([getattr(g, key) for g in graphs], plans_on(plans, dev))
        stage = [egc_basis_transform(g, conv._spec_coo, x[pl.lo:pl.hi], wcat, bcat) for g, pl in zip(graphs, plans)]
        P.simulate_exchange([b for b, _ in stage], plans_on(plans, dev))
        outs = [egc_aggregate_combine(g, conv._spec_coo, b, w, conv.bias) for g, (b, w) in zip(graphs, stage)]
    got = torch.cat(outs)
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= TOL


@pytest.mark.parametrize("world", [2, 5])
def test_interior_first_partition_and_row_ranges(world):
    """The overlap form of the partitioned forward: owned vertices renumbered interior-first, interior rows
    finished from the owned basis rows ALONE (the halo region still holds NaN), boundary rows after the exchange;
    un-permuted and concatenated it equals the single-GPU output."""
    import egc_amd
    from egc_amd import partition as P
    from egc_amd.functional import egc_aggregate_combine, egc_basis_transform
    from egc_amd.workloads import heavy_tailed_graph
    dev = _dev()
    n = 6000
    ei = heavy_tailed_graph(n, 40000, seed=3)
    torch.manual_seed(2)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    with torch.no_grad():
        ref = conv(x, ei.to(dev))
        wcat, bcat = conv._packed_weights()
        parts = P.build_local_simulation(ei, n, world, interior_first=True)
        plans = [p for _, p in parts]
        for p in plans:
            p.order, p.new_of_old = p.order.to(dev), p.new_of_old.to(dev)
        graphs = [egc_amd.CSRGraph.from_partition(e.to(dev), plan, global_max_index=int(ei.max()), exchange_dis=False)
                  for e, plan in parts]
        for key in ("dis_raw", "dis_looped"):
            P.simulate_exchange([getattr(g, key) for g in graphs], plans_on(plans, dev))
        for g in graphs:
            g.refresh_edge_dis()   # per-entry deg^-1/2 copies follow the exchanged tables
        stage = [egc_basis_transform(g, conv._spec_coo, x[pl.lo:pl.hi][pl.order], wcat, bcat) for g, pl in zip(graphs, plans)]
        outs = []
        for g, pl, (b, w) in zip(graphs, plans, stage):
            assert 0 <= pl.n_interior <= pl.n_local
            b[pl.n_local:] = float("nan")                      # halo rows have not arrived yet
            out = torch.full((pl.n_local, 128), float("nan"), device=dev)
            egc_aggregate_combine(g, conv._spec_coo, b, w, conv.bias, rows=(0, pl.n_interior), out=out)
            assert bool(torch.isfinite(out[:pl.n_interior]).all()) and bool(torch.isnan(out[pl.n_interior:]).all())
            outs.append(out)
        P.simulate_exchange([b for b, _ in stage], plans)
        got = []
        for g, pl, (b, w), out in zip(graphs, plans, stage, outs):
            egc_aggregate_combine(g, conv._spec_coo, b, w, conv.bias, rows=(pl.n_interior, pl.n_local), out=out)
            back = torch.empty_like(out)
            back[pl.order] = out
            got.append(back)
    assert rel_err(torch.cat(got).cpu().numpy(), ref.cpu().numpy()) <= TOL


def plans_on(plans, dev):
    for p in plans:
        p.halo_global_ids = p.halo_global_ids.to(dev)
    return plans


def test_layer_forward_is_graph_capturable():
    """The library never allocates or synchronises and launches on the caller's stream, so a whole layer forward
    (GEMM + fused aggregate, incl. the long-row handshake) can be captured in a HIP graph and replayed."""
    import ctypes as C
    import egc_amd
    from egc_amd import _C
    from egc_amd.functional import pack_weights
    from egc_amd.workloads import heavy_tailed_graph
    dev = _dev()
    lib = _C.load()
    n = 30000
    ei = heavy_tailed_graph(n, 250000, seed=5).to(dev)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev).eval()
    graph = egc_amd.CSRGraph.from_edge_index(ei, n)
    spec = conv._spec_coo
    with torch.no_grad():
        wcat, bcat = conv._packed_weights()
        planes = pack_weights(spec, wcat)
        x = torch.randn(n, 128, device=dev)
        ref = conv(x, graph)
    bases = torch.empty((n, spec.ldb), device=dev)
    wt = torch.empty((n, spec.w_cols), device=dev)
    out = torch.zeros((n, 128), device=dev)
    ws = torch.zeros(max(lib.egc_aggregate_workspace_bytes(C.byref(spec.c), n, ei.size(1)), 1), dtype=torch.uint8, device=dev)
    g = graph.c_struct()

    def step(stream):
        _C.check(lib.egc_layer_forward_packed(C.byref(g), C.byref(spec.c), x.data_ptr(), planes.data_ptr(), bcat.data_ptr(),
                                              conv.bias.data_ptr(), bases.data_ptr(), spec.ldb, wt.data_ptr(), out.data_ptr(),
                                              ws.data_ptr(), ws.numel(), stream), "egc_layer_forward_packed")

    step(torch.cuda.current_stream().cuda_stream)   # one eager call first (one-time function attributes)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        step(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        out.zero_()
        x.mul_(1.0)          # same values, new launch order
        cg.replay()
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) <= TOL
    x.copy_(torch.randn(n, 128, device=dev))   # new inputs through the same captured graph
    cg.replay()
    with torch.no_grad():
        ref2 = conv(x, graph)
    assert rel_err(out.cpu().numpy(), ref2.cpu().numpy()) <= TOL


def test_workspace_needs_only_its_zero_prefix():
    """egc_aggregate_workspace_zero_bytes: only the arrival counters at the front of the workspace have to be zero
    before the first use; the chunk records behind them may hold anything (a per-batch graph's workspace is not
    filled).  Hub rows present: long-row chunks are what the workspace is for."""
    import ctypes as C
    import egc_amd
    from egc_amd import _C
    from egc_amd import functional as F
    from egc_amd.workloads import heavy_tailed_graph
    dev = _dev()
    lib = _C.load()
    n = 20000
    ei = heavy_tailed_graph(n, 200000, seed=11).to(dev)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev).eval()
    graph = egc_amd.CSRGraph.from_edge_index(ei, n)
    spec = conv._spec_coo
    total = int(lib.egc_aggregate_workspace_bytes(C.byref(spec.c), n, ei.size(1)))
    zero = int(lib.egc_aggregate_workspace_zero_bytes(C.byref(spec.c), n, ei.size(1)))
    assert 0 < zero < total and zero % 16 == 0
    with torch.no_grad():
        wcat, bcat = conv._packed_weights()
        x = torch.randn(n, 128, device=dev)
        bases, wt = F.egc_basis_transform(graph, spec, x, wcat, bcat, None)
        ref = F.egc_aggregate_combine(graph, spec, bases, wt, conv.bias)       # the graph's own (zero-prefixed) workspace
        g = graph.c_struct()
        for fill in (0xFF, 0x7F, 0x01):
            ws = torch.full((total,), fill, dtype=torch.uint8, device=dev)
            ws[:zero].zero_()
            out = torch.empty((n, 128), device=dev)
            for _ in range(2):   # second call: the workspace is left ready by the first
                _C.check(lib.egc_aggregate_combine_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb, wt.data_ptr(),
                                                       conv.bias.data_ptr(), out.data_ptr(), None, None, ws.data_ptr(),
                                                       ws.numel(), torch.cuda.current_stream().cuda_stream),
                         "egc_aggregate_combine_f32")
                torch.cuda.synchronize()
                assert torch.equal(out, ref), fill
            assert bool((ws[:zero] == 0).all())


def test_arrays_beyond_two_gib_sampled_rows_against_float64():
    """N = 4.3 M nodes: x, weightings and out exceed 2 GiB each (the GEMM runs in row ranges, the aggregate's
    32-bit buffer offsets pass 2^31), a hub row of ~10^5 entries.  Sampled rows -- the first, the last, the hub,
    random ones -- are recomputed in float64."""
    import egc_amd
    dev = _dev()
    n, e = 4_300_000, 20_000_000
    g = torch.Generator(device="cpu").manual_seed(1)
    src = torch.randint(0, n, (e,), generator=g)
    dst = (n * torch.rand(e, generator=g) ** 3).long().clamp_(max=n - 1)
    ei = torch.stack([src, dst]).to(dev)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    H, B, A = 8, 4, 4
    with torch.no_grad():
        out = conv(x, ei)
        rows = torch.cat([torch.tensor([0, 1, n - 1, n - 2]), torch.randint(0, n, (200,), generator=g)]).to(dev)
        nonself = torch.bincount(ei[1][ei[0] != ei[1]], minlength=n).double() + 1   # gcn_norm: self loops replaced
        dis = nonself.pow(-0.5)
        sel = torch.isin(ei[1], rows) & (ei[0] != ei[1])
        s_sel, d_sel = ei[0][sel], ei[1][sel]
        W = conv.bases_weight.double()
        worst = 0.0
        for r in rows.tolist():
            nb = torch.cat([s_sel[d_sel == r], torch.tensor([r], device=dev)])
            bj = x[nb].double() @ W
            aggs = torch.stack([bj.sum(0), bj.mean(0), bj.max(0).values, (bj * (dis[nb] * dis[r])[:, None]).sum(0)])
            wt = (x[r].double() @ conv.comb_weight.weight.double().t() + conv.comb_weight.bias.double()).view(H, A, B)
            ref = torch.einsum("hab,abl->hl", wt, aggs.view(A, B, 16)).reshape(-1) + conv.bias.double()
            worst = max(worst, float((out[r].double() - ref).abs().max() / ref.abs().max().clamp(min=1)))
    assert worst <= TOL, worst


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("hidden,H,B,aggrs", [(64, 8, 4, ["mean", "max"]), (42, 6, 3, ["sum", "std", "min"])])
def test_weightings_read_as_a_column_block(hidden, H, B, aggrs, generic, monkeypatch):
    """egc_aggregate_combine_strided_f32: the weightings of a term are a column block of a wider array (several
    terms from one GEMM, relational EGC) -- same bits as with a dense copy, on both kernel families."""
    import egc_amd
    from egc_amd.functional import PostOp, egc_aggregate_combine
    if generic:
        monkeypatch.setenv("EGC_FORCE_GENERIC", "1")
    dev = _dev()
    rng = np.random.default_rng(hidden)
    n = 500
    ei = torch.from_numpy(_hub_graph(rng, n, 4000, [(5, 400)])).to(dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    conv = egc_amd.EGConv(hidden, hidden, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=False)
    spec = conv._spec_coo
    torch.manual_seed(0)
    bases = torch.zeros(n, spec.ldb, device=dev)
    bases[:, :spec.f_g] = torch.randn(n, spec.f_g, device=dev)
    if spec.basis_stride != spec.basis_len:    # padded layout: pad columns must be zero
        bases.view(n, -1)[:, :B * spec.basis_stride].view(n, B, spec.basis_stride)[:, :, spec.basis_len:] = 0
    W = spec.w_cols
    pad = (-W) % 4
    wide = torch.randn(n, 8 + W + pad + 12, device=dev)          # the block starts at column 8 (32-byte offset)
    block = wide[:, 8:8 + W]
    res = torch.randn(n, hidden, device=dev)
    if W % 4 == 0:
        assert block.stride(0) != W and block.data_ptr() % 16 == 0
    dense = egc_aggregate_combine(g, spec, bases, block.contiguous(), None)
    strided = egc_aggregate_combine(g, spec, bases, block, None)
    assert torch.equal(dense, strided)
    dense_p = egc_aggregate_combine(g, spec, bases, block.contiguous(), None, post=PostOp(residual=res, relu=True))
    strided_p = egc_aggregate_combine(g, spec, bases, block, None, post=PostOp(residual=res, relu=True))
    assert torch.equal(dense_p, strided_p)


@pytest.mark.parametrize("hidden,H,B", [(4, 1, 1), (8, 1, 1), (16, 4, 1), (8, 2, 2), (24, 2, 1), (48, 2, 1), (80, 2, 1)])
@pytest.mark.parametrize("kind,aggrs", [("lay", ["mean"]), ("lay", ["var", "symadd", "min"]), ("opt", ["std", "max", "sum"])])
def test_multi_chunk_hub_rows_with_few_slots_on_small_graphs(hidden, H, B, kind, aggrs):
    """Rows cut into several chunks publish one record per chunk, laid out by the kernels' lane-group size (16 / 32 /
    64 lanes) -- more than the row's slot count when that is 1, 2, 6, 12 or 20: on a SMALL graph (little spare
    capacity in the workspace) an undersized record area was overrun (found by the randomised sweep)."""
    dev = _dev()
    rng = np.random.default_rng(hidden * 7 + len(aggrs))
    n = 200
    ei = _hub_graph(rng, n, 500, [(3, 300), (77, 700)])
    out, ref = _oracle_case(kind, rng, n, ei, hidden, hidden, H, B, aggrs, dev, check64=True)    # (every element against float64 at 1e-5)
    # (against the float32 restatement: its own E[x^2] - E[x]^2 is up to 1e-4 off on hub rows of nearly tied entries)
    assert rel_err(out, ref) <= (1e-4 if any(a in ("std", "var") for a in aggrs) else TOL)


@pytest.mark.parametrize("kind,aggrs,near_constant", [
    ("opt", ["std", "max", "sum"], False), ("opt", ["var", "mean"], False), ("opt", ["std"], True),
    ("opt", ["var", "symnorm", "std", "min"], True), ("lay", ["std", "add", "max"], False), ("lay", ["var", "std"], True)])
def test_std_var_layers_are_as_accurate_as_the_fp32_restatement(kind, aggrs, near_constant):
    """Where the parity tests allow 1e-4 for `std` / `var` layers (sqrt(relu(E[x^2] - E[x]^2) + 1e-5) amplifies last-bit
    differences between two correct fp32 evaluations), this pins the HIP path the other way round: against the SAME layer
    evaluated in float64, its error on these inputs is no larger than that of the fp32 restatement of the reference's
    arithmetic (oracle/egc_oracle.py; x2 + the 1e-5 of north_star as slack) -- on random inputs and on nearly constant
    neighbourhoods, the worst case of the cancellation.  Not a universal bound: the randomised sweep finds small
    graphs where the split-precision GEMM's 2^-22 operand rounding, amplified the same way, leaves the HIP result a few
    times further from float64 than the restatement (7e-5 against 2e-5); those stay inside the 1e-4 of the sweep."""
    import egc_amd
    from oracle import egc_torch_ref as tref
    dev = _dev()
    rng = np.random.default_rng(len(aggrs) * 31 + near_constant)
    n, hidden, H, B = 1500, 64, 4, 4
    ei = _hub_graph(rng, n, 9000, [(5, 400)])
    torch.manual_seed(3)
    if kind == "opt":
        layer = egc_amd.EGConv(hidden, hidden, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=True)
    else:
        layer = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs, add_self_loops=True)
    with torch.no_grad():
        layer.bias.normal_()
    x = rng.standard_normal((n, hidden)).astype(np.float32)
    if near_constant:   # every node a tiny perturbation of one feature vector: var ~ 1e-8 under the 1e-5 epsilon
        x = (x[:1] + 1e-4 * x).astype(np.float32)
    sd = {k: v.numpy() for k, v in layer.state_dict().items()}
    meta = dict(kind=kind, fin=hidden, fout=hidden, H=H, B=B, aggrs=aggrs, softmax=False, sigmoid=False, hardtanh=False,
                add_self_loops=True, bias=True, sparse=False)
    ref32 = oracle_forward(dict(meta=meta, params=sd, x=x, edge_index=ei), orc)
    p64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    x64 = torch.from_numpy(x).double()
    if kind == "opt":
        truth = tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"],
                                    p64["bias"], H, B, aggrs, add_self_loops=True)
    else:
        truth = tref.efficient_graph_conv_forward(x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)],
                                                  p64["comb_weights.weight"], p64["comb_weights.bias"], p64["bias"], H,
                                                  aggrs, add_self_loops=True)
    truth = truth.numpy()
    layer = layer.to(dev)
    with torch.no_grad():
        xt, eit = torch.from_numpy(x).to(dev), torch.from_numpy(ei).to(dev)
        out = (layer(xt, eit) if kind == "opt" else layer(x=xt, edge_index=eit)).cpu().numpy()
    err_hip, err_ref = rel_err(out, truth), rel_err(ref32, truth)
    assert err_hip <= 2.0 * err_ref + 1e-5, (err_hip, err_ref)
    assert rel_err(out, ref32) <= max(1e-4, 2.0 * err_ref)   # (two fp32 evaluations are at most their two errors apart)


def test_foreign_sparse_tensor_adj_t_equals_own_sparse_tensor():
    """An object with torch_sparse.SparseTensor's API (`.csr()`, `.sparse_sizes()`: what `ToSparseTensor` hands to the mag net,
    mag/configs.py:84-85) through EGConv == egc_amd.SparseTensor on the same graph, bit for bit."""
    import egc_amd
    from test_host_cpu import _TorchSparseLikeAdjT
    dev = _dev()
    rng = np.random.default_rng(31)
    n = 700
    ei = torch.from_numpy(rng.integers(0, n, size=(2, 5000)).astype(np.int64)).to(dev)
    torch.manual_seed(2)
    conv = egc_amd.EGConv(64, 64, aggrs=["symnorm", "mean", "max"], num_heads=8, num_bases=4).to(dev).eval()
    x = torch.randn(n, 64, device=dev)
    own = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(n, n))
    g = own.graph
    foreign = _TorchSparseLikeAdjT(g.rowptr.long(), g.col[:g.n_edges].long(), n, n)
    with torch.no_grad():
        a = conv(x, own)
        b = conv(x, foreign)
    assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden,H,B,aggrs", [
    ("opt", 64, 4, 4, ["min", "std", "var"]),          # register kernels, squares + min
    ("opt", 128, 8, 4, ["sum", "std", "max", "symnorm"]),   # squares only
    ("lay", 124, 4, 4, ["add", "std", "max"]),        # the reference's ZINC EGC-M layer (static configuration)
    ("opt", 304, 8, 8, ["std", "mean"]),              # two slots per lane
])
def test_std_var_on_tied_neighbourhoods_is_closer_to_float64_than_the_float32_formula(kind, hidden, H, B, aggrs):
    """Graphs of one or two nodes with twenty edges per node, self loops added: every neighbourhood is a multiset over at most
    two distinct rows, half of the entries self-entries that the LOOPED x-part skips -- the worst case of the float32 formula
    E[x^2] - E[x]^2 (1e-5 .. 2e-5 from float64 here) and the case that caught a variance shift taken from a SKIPPED first entry
    (tools/tile_fuzz.py seed 77).  The kernels' variance about the row's first entry holds 5e-6 of the output scale against
    float64 and is closer to it than the float32 restatement."""
    import egc_amd
    from oracle import egc_torch_ref as tref
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 3, size=1500)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    srcs, dsts = [], []
    for g in range(sizes.size):
        n, o = int(sizes[g]), int(ptr[g])
        e = int(rng.poisson(20.0 * n))
        srcs.append(rng.integers(0, n, size=e) + o); dsts.append(rng.integers(0, n, size=e) + o)
    ei = np.stack([np.concatenate(srcs), np.concatenate(dsts)]).astype(np.int64)
    N = int(ptr[-1])
    torch.manual_seed(3)
    if kind == "opt":
        conv = egc_amd.EGConv(hidden, hidden, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=True)
    else:
        conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs, add_self_loops=True)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(N, hidden)
    p = {k: v.detach().double() for k, v in conv.named_parameters()}
    with torch.no_grad():
        if kind == "opt":
            args = ("bases_weight", "comb_weight.weight", "comb_weight.bias", "bias")
            t64 = tref.egconv_forward(x.double(), ei, *[p[k] for k in args], H, B, aggrs, add_self_loops=True, sigmoid=False)
            t32 = tref.egconv_forward(x, ei, *[p[k].float() for k in args], H, B, aggrs, add_self_loops=True, sigmoid=False)
        else:
            bw = [f"bases_weight.{b}" for b in range(B)]
            t64 = tref.efficient_graph_conv_forward(x.double(), ei, [p[k] for k in bw], p["comb_weights.weight"], p["comb_weights.bias"],
                                                    p["bias"], H, aggrs, softmax=False, hardtanh=False, sigmoid=False, add_self_loops=True)
            t32 = tref.efficient_graph_conv_forward(x, ei, [p[k].float() for k in bw], p["comb_weights.weight"].float(),
                                                    p["comb_weights.bias"].float(), p["bias"].float(), H, aggrs, softmax=False,
                                                    hardtanh=False, sigmoid=False, add_self_loops=True)
        conv = conv.to(dev).eval()
        eit = torch.from_numpy(ei).to(dev)
        out = (conv(x.to(dev), eit) if kind == "opt" else conv(x=x.to(dev), edge_index=eit)).double().cpu()
    scale = max(1.0, float(t64.abs().max()))
    e_hip, e_ref = float((out - t64).abs().max()) / scale, float((t32.double() - t64).abs().max()) / scale
    assert e_hip <= 5e-6, (e_hip, e_ref)
    assert e_hip <= max(e_ref, 1e-6), (e_hip, e_ref)
