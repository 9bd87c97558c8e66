"""CPU tests of the host side: the C-ABI library loads and exports every symbol of
include/egc_hip.h (no compute calls without a GPU), and the drop-in modules mirror the reference's
constructor contract, parameter names/shapes, repr and error behaviour."""
import json
import os
import re

import pytest
import torch

import egc_amd
from egc_amd import _C
from golden_util import golden_names, load_golden
from oracle import egc_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "egc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(egc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_C.SYMBOLS), (declared ^ set(_C.SYMBOLS))
    lib = _C.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.egc_version().decode().endswith("gfx950")
    # pure host-side size queries (no GPU touched)
    assert lib.egc_plan_ints(10, 0) == 4 + 2 * 1 + 2 * 1
    assert lib.egc_plan_ints(-1, 0) == -1


def test_header_constants_match_python_binding():
    hdr = open(os.path.join(ROOT, "include", "egc_hip.h")).read()
    assert int(re.search(r"#define EGC_MAX_AGGRS (\d+)", hdr).group(1)) == _C.EGC_MAX_AGGRS
    assert int(re.search(r"#define EGC_LONG_ROW_THRESHOLD (\d+)", hdr).group(1)) == _C.LONG_ROW_THRESHOLD
    assert int(re.search(r"#define EGC_LONG_ROW_CHUNK (\d+)", hdr).group(1)) == _C.LONG_ROW_CHUNK


def test_no_cpu_fallback():
    conv = egc_amd.EGConv(8, 8, aggrs=["sum"], num_heads=2, num_bases=2)
    x = torch.randn(4, 8)
    ei = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv(x, ei)
    lay = egc_amd.EfficientGraphConv(8, 8, 2, 2, False, aggrs=["add"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lay(x=x, edge_index=ei)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "_LIB_PATH", "/nonexistent/libegc_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        _C.load()


def test_product_package_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+\S*oracle", re.M)
    for fn in os.listdir(os.path.join(ROOT, "egc_amd")):
        if fn.endswith(".py"):
            assert not pat.search(open(os.path.join(ROOT, "egc_amd", fn)).read()), fn


# ---- EfficientGraphConv contract (experiments/layers.py:13-80, 142-147) ----------------------

def test_efficient_graph_conv_state_dict_and_repr():
    conv = egc_amd.EfficientGraphConv(168, 168, num_heads=8, num_bases=4, softmax_weights=False, aggrs=["symadd"])
    sd = conv.state_dict()
    assert list(sd.keys()) == ["bias", "comb_weights.weight", "comb_weights.bias", "bases_weight.0",
                               "bases_weight.1", "bases_weight.2", "bases_weight.3"]
    assert tuple(sd["comb_weights.weight"].shape) == (32, 168)
    assert all(tuple(sd[f"bases_weight.{b}"].shape) == (168, 21) for b in range(4))
    assert sum(p.numel() for p in conv.parameters()) == orc.layer_param_count(168, 168, 8, 4, 1)
    # format of output/pretrained.txt:47-60
    assert conv.extra_repr() == "(In=168, Out=168, H=8, B=4, SL=True, SM=False, Bias=True)"
    assert "(0): _AggLayer(symadd)" in repr(conv)
    assert float(conv.bias.abs().sum()) == 0.0
    bound = orc.glorot_bound(168, 21)
    assert all(float(w.abs().max()) <= bound for w in conv.bases_weight)


def test_efficient_graph_conv_ctor_errors():
    with pytest.raises(AssertionError):
        egc_amd.EfficientGraphConv(8, 8, 2, 2, False)  # aggrs is None (layers.py:29)
    with pytest.raises(AssertionError):
        egc_amd.EfficientGraphConv(8, 8, 2, 2, True, aggrs=["add"], sigmoid_weights=True)
    with pytest.raises(AssertionError):
        egc_amd.EfficientGraphConv(8, 8, 2, 2, False, aggrs=["add"], sigmoid_weights=True, hardtanh_weights=True)
    with pytest.raises(AssertionError):
        egc_amd.EfficientGraphConv(8, 9, 2, 2, False, aggrs=["add"])  # out % heads
    conv = egc_amd.EfficientGraphConv(8, 8, 2, 2, False, aggrs=["add"], bias=False, some_future_kwarg=1)
    assert conv.bias is None and "bias" not in conv.state_dict()


def test_import_paths_of_the_reference_resolve():
    from experiments.layers import EfficientGraphConv
    from experiments.optimized_layers import EGConv
    assert EfficientGraphConv is egc_amd.EfficientGraphConv and EGConv is egc_amd.EGConv


# ---- EGConv contract (experiments/optimized_layers.py:74-122, 280-286) -----------------------

def test_egconv_state_dict_repr_and_defaults():
    conv = egc_amd.EGConv(128, 352, aggrs=["mean"], num_heads=8, num_bases=4, cached=True)
    sd = conv.state_dict()
    assert list(sd.keys()) == ["bases_weight", "bias", "comb_weight.weight", "comb_weight.bias"]
    assert tuple(sd["bases_weight"].shape) == (128, 176) and tuple(sd["comb_weight.weight"].shape) == (32, 128)
    assert repr(conv) == "EGConv(128, 352, ['mean'])"
    d = egc_amd.EGConv(16, 16)
    assert d.aggregators == ["symnorm"] and d.num_heads == 8 and d.num_bases == 4 and not d.cached
    assert d.add_self_loops and d.bias is not None and not d.sigmoid


def test_egconv_ctor_errors():
    with pytest.raises(ValueError, match="divisible"):
        egc_amd.EGConv(8, 9, num_heads=2)
    with pytest.raises(ValueError, match="Unsupported aggregator"):
        egc_amd.EGConv(8, 8, aggrs=["add"], num_heads=2)  # 'add' is the layers.py name, not EGConv's


@pytest.mark.parametrize("name", golden_names())
def test_golden_state_dicts_load_strictly(name):
    """Parameter names/shapes saved from the REFERENCE modules load into the drop-ins unchanged."""
    g = load_golden(name)
    m = g["meta"]
    if m["kind"] == "lay":
        layer = egc_amd.EfficientGraphConv(m["fin"], m["fout"], m["H"], m["B"], m["softmax"], aggrs=m["aggrs"],
                                           add_self_loops=m["add_self_loops"], bias=m["bias"],
                                           sigmoid_weights=m["sigmoid"], hardtanh_weights=m["hardtanh"])
        ref_repr = m["repr"]
        # the reference prints '_AggLayer(<name>)' children and the same extra_repr line
        assert layer.extra_repr() in ref_repr
        for a in m["aggrs"]:
            assert f"_AggLayer({a})" in ref_repr and f"_AggLayer({a})" in repr(layer)
    else:
        layer = egc_amd.EGConv(m["fin"], m["fout"], aggrs=m["aggrs"], num_heads=m["H"], num_bases=m["B"],
                               add_self_loops=m["add_self_loops"], bias=m["bias"], sigmoid=m["sigmoid"])
        assert repr(layer) == m["repr"]
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()}, strict=True)


def test_packed_weight_layout_matches_oracle_bases():
    """[bases_weight | comb.weight^T] packing: column b*L + l of the bases block is basis b, channel l."""
    conv = egc_amd.EfficientGraphConv(12, 8, 2, 3, False, aggrs=["add", "max"])
    w = conv._packed_weights()
    assert tuple(w.shape) == (12, 3 * 4 + 2 * 3 * 2)
    for b in range(3):
        assert torch.equal(w[:, b * 4:(b + 1) * 4], conv.bases_weight[b].detach())
    assert torch.equal(w[:, 12:], conv.comb_weights.weight.detach().t())
    # cache invalidates on in-place parameter updates
    with torch.no_grad():
        conv.bases_weight[0].add_(1.0)
    assert torch.equal(conv._packed_weights()[:, :4], conv.bases_weight[0].detach())


def test_workload_byte_model_matches_survey_config2():
    """SURVEY.md 8(d) worked example: 864.2 MB at E_eff = 2,484,941."""
    from egc_amd.workloads import algorithmic_bytes
    t = algorithmic_bytes(169343, 2484941, 128, 64, 128, 128, True)
    assert abs(t["layer"] / 1e6 - 864.2) < 0.2


def test_rmag_like_workload_has_the_reference_relations():
    """The heterogeneous ogbn-mag-shaped workload carries exactly the seven relations REGConv iterates
    (rmag/models.py:18-26), with ids inside the node counts and each reverse relation the transpose."""
    from egc_amd.relational import EDGE_TYPES
    from egc_amd.workloads import rmag_like
    nodes, rel = rmag_like(seed=1, scale=0.001)
    assert set(rel) == set(EDGE_TYPES)
    for (s, _, d), ei in rel.items():
        assert ei.dtype == torch.int64 and ei.shape[0] == 2 and ei.shape[1] > 0
        assert int(ei[0].max()) < nodes[s] and int(ei[1].max()) < nodes[d] and int(ei.min()) >= 0
    fwd, rev = rel[("author", "writes", "paper")], rel[("paper", "to", "author")]
    assert torch.equal(fwd[0], rev[1]) and torch.equal(fwd[1], rev[0])
    c = rel[("paper", "cites", "paper")]
    key = c[0] * nodes["paper"] + c[1]
    assert torch.equal(torch.sort(key).values, torch.sort(c[1] * nodes["paper"] + c[0]).values)   # symmetric


def test_graphed_step_and_fused_block_host_logic():
    """No GPU: GraphedStep refuses loudly (there is nothing to record on); FusedEGCBlock validates its dropout
    probability; the graph cache's recording scope nests and clears what it built."""
    import egc_amd
    from egc_amd import graph as G
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            egc_amd.GraphedStep(lambda: None)
    conv = egc_amd.EGConv(16, 16, aggrs=["sum"], num_heads=4, num_bases=2)
    with pytest.raises(ValueError):
        egc_amd.FusedEGCBlock(conv, torch.nn.BatchNorm1d(16), dropout=1.0)
    blk = egc_amd.FusedEGCBlock(conv, torch.nn.BatchNorm1d(16), dropout=0.25)
    assert blk.dropout == 0.25 and blk._dropping() and not blk.eval()._dropping()
    assert G._RECORDING[0] is None
    with G.recording_scope() as outer:
        tok = G._RECORDING[0]
        assert tok is not None
        with G.recording_scope():
            assert G._RECORDING[0] is not tok
        assert G._RECORDING[0] is tok
    assert G._RECORDING[0] is None


class _TorchSparseLikeAdjT:
    """The public API of ``torch_sparse.SparseTensor`` the layers touch, as mag/configs.py:84-85 produces it (``ToSparseTensor``:
    adj_t with rows = destinations): ``csr()``, ``sparse_sizes()``, ``storage`` -- no egc_amd type involved."""

    def __init__(self, rowptr, col, n_dst, n_src):
        self._rowptr, self._col, self._sizes = rowptr, col, (n_dst, n_src)
        self.storage = self

    def csr(self):
        return self._rowptr, self._col, None

    def sparse_sizes(self):
        return self._sizes

    def set_value(self, value, layout=None):
        return self


def test_a_foreign_sparse_tensor_is_recognised_by_its_csr_method(monkeypatch):
    """VERDICT r3 missing #6: `graph_from_input` has only ever met egc_amd.SparseTensor.  A stand-in with torch_sparse's API
    (`.csr()` / `.sparse_sizes()`) must take the adj_t route: CSRGraph.from_csr on its arrays, the `_spec_adj` of EGConv,
    NotImplementedError for var / std in EfficientGraphConv (layers.py:221-224)."""
    import egc_amd.graph as G
    seen = {}

    class _FakeGraph:
        def trim_launches(self):
            return self

    def fake_from_csr(rowptr, col, num_nodes=None, num_src_rows=None):
        seen["args"] = (rowptr, col, num_nodes, num_src_rows)
        return _FakeGraph()
    monkeypatch.setattr(G.CSRGraph, "from_csr", staticmethod(fake_from_csr))
    rowptr, col = torch.tensor([0, 1, 3]), torch.tensor([1, 0, 1])
    adj = _TorchSparseLikeAdjT(rowptr, col, 2, 2)
    g = G.graph_from_input(adj, 2)
    assert isinstance(g, _FakeGraph) and seen["args"][0] is rowptr and seen["args"][2:] == (2, 2)
    assert G.graph_from_input(adj, 2) is g                      # converted once, cached on the object
    with pytest.raises(RuntimeError, match="rows"):
        G.graph_from_input(_TorchSparseLikeAdjT(rowptr, col, 2, 2), 3)
    lay = egc_amd.EfficientGraphConv(8, 8, 2, 2, False, aggrs=["add", "std"])
    with pytest.raises(NotImplementedError):
        lay(torch.randn(2, 8), adj)


def test_own_graph_objects_are_not_taken_for_an_adj_t():
    """CSRGraph and GraphBatch also have a csr(): they are this package's graphs and take var / std (only an adjacency object in
    the reference's sense raises NotImplementedError there, layers.py:221-224)."""
    from egc_amd.layers import _is_adj_t
    import egc_amd.graph as G
    assert _is_adj_t(_TorchSparseLikeAdjT(torch.tensor([0, 1]), torch.tensor([0]), 1, 1))
    assert not _is_adj_t(torch.zeros(2, 3, dtype=torch.long))
    assert not _is_adj_t(object.__new__(G.GraphBatch)) and not _is_adj_t(object.__new__(G.CSRGraph))
    assert _is_adj_t(object.__new__(G.SparseTensor))


def test_weight_gradient_plan_keeps_every_xcd_at_one_round(monkeypatch):
    """egc_weight_grad_plan (host only): the tile grid covers the output, the row ranges cover the rows, the workspace holds one
    record per range, and -- the round-6 fix -- the tiles of the ranges that land on one XCD fit its 32 CUs whenever the output has
    at most 32 tiles (86 ranges of three tiles used to put 33 workgroups on some XCDs: two rounds)."""
    import ctypes as C
    lib = _C.load()
    monkeypatch.delenv("EGC_XT_TILE", raising=False)
    monkeypatch.delenv("EGC_GEMM_EXACT", raising=False)
    plan = (C.c_int32 * 8)()
    for n in (0, 1, 700, 2998, 52771, 169343, 736389):
        for f, k in ((128, 192), (168, 116), (224, 272), (296, 180), (136, 184), (304, 368), (352, 208), (4, 4), (384, 384)):
            assert lib.egc_weight_grad_plan(n, f, k, C.cast(plan, C.c_void_p)) == 0
            x3, tm, tn, mt, nt, ranges, rows, threads = list(plan)
            assert x3 == int(f <= 128 and k <= 192)
            assert mt * tm >= f and nt * tn >= k and ranges * rows >= n and rows % 32 == 0 and threads in (384, 512, 768)
            assert lib.egc_weight_grad_workspace_bytes(n, f, k) == ranges * (f * k + k) * 4
            if not x3 and mt * nt <= 32:
                assert -(-ranges // 8) * mt * nt <= 32, (n, f, k, list(plan))
    monkeypatch.setenv("EGC_XT_TILE", "3,3,3")          # not a compiled tile
    assert lib.egc_weight_grad_plan(1000, 224, 272, C.cast(plan, C.c_void_p)) == 4
    assert lib.egc_weight_grad_plan(1000, 0, 272, C.cast(plan, C.c_void_p)) == 1
