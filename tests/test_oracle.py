"""CPU tests of the oracle itself: it must reproduce every golden vector (produced by the reference's
own layer code), agree with independent torch implementations of the restated third-party ops, and
satisfy the reference's known answers (parameter counts from output/pretrained.txt)."""
import numpy as np
import pytest
import torch

from golden_util import elementwise_excess, golden_names, load_golden, oracle_forward, rel_err
from oracle import egc_oracle as orc


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_golden(name):
    g = load_golden(name)
    out = oracle_forward(g, orc)
    assert out.shape == g["out"].shape and out.dtype == np.float32
    # differences come only from numpy-vs-torch fp32 GEMM summation order
    assert rel_err(out, g["out"]) <= 2e-6, rel_err(out, g["out"])
    assert elementwise_excess(out, g["out"], 1e-5) <= 1.0     # element-wise, against each element's own row scale


def test_golden_set_covers_the_survey_list():
    names = set(golden_names())
    assert len(names) >= 45
    for a in orc.AGGRS_LAYERS:
        assert f"lay_single_{a}" in names
    for a in orc.AGGRS_OPT:
        assert f"opt_single_{a}" in names
    for must in ("lay_softmax", "lay_sigmoid", "lay_hardtanh", "opt_noedges", "opt_ties", "opt_sparse_mean",
                 "opt_selfloops_inferredN", "opt_northstar_small", "lay_L21", "lay_L31", "opt_mid"):
        assert must in names


@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
def test_scatter_against_independent_torch_ops(reduce):
    """oracle.scatter vs torch's own index_add_ / scatter_reduce_ (an independent implementation
    of the torch_scatter semantics: mean = sum / clamp(count, 1); empty min/max rows -> 0)."""
    rng = np.random.default_rng(0)
    n, e, f = 37, 400, 5
    src = rng.standard_normal((e, f)).astype(np.float32)
    idx = rng.integers(0, n - 4, size=e)  # last rows stay empty
    out, arg = orc.scatter(src, idx, n, reduce)
    ts, ti = torch.from_numpy(src), torch.from_numpy(idx)
    if reduce in ("sum", "mean"):
        ref = torch.zeros(n, f).index_add_(0, ti, ts)
        if reduce == "mean":
            cnt = torch.zeros(n).index_add_(0, ti, torch.ones(e)).clamp_(min=1)
            ref = ref / cnt[:, None]
        np.testing.assert_allclose(out, ref.numpy(), rtol=1e-6, atol=1e-6)
    else:
        ref = torch.zeros(n, f).scatter_reduce_(0, ti[:, None].expand(e, f), ts,
                                                "amax" if reduce == "max" else "amin", include_self=False)
        assert np.array_equal(out, ref.numpy())
        # argument = FIRST edge (input order) attaining the extremum; empty rows -> E
        for r in range(n):
            rows = np.flatnonzero(idx == r)
            for c in range(f):
                if len(rows) == 0:
                    assert arg[r, c] == e and out[r, c] == 0
                else:
                    vals = src[rows, c]
                    best = vals.max() if reduce == "max" else vals.min()
                    assert arg[r, c] == rows[np.flatnonzero(vals == best)[0]]


def test_scatter_first_edge_tie_break():
    src = np.array([[1.0], [3.0], [3.0], [2.0], [3.0]], dtype=np.float32)
    idx = np.array([0, 0, 0, 0, 0])
    out, arg = orc.scatter(src, idx, 2, "max")
    assert out[0, 0] == 3.0 and arg[0, 0] == 1 and arg[1, 0] == 5 and out[1, 0] == 0.0


def test_gcn_norm_semantics():
    """Self-loops are REPLACED (not duplicated), degree is the in-degree incl. the loop, inf -> 0."""
    ei = np.array([[0, 1, 1, 2, 2, 2], [1, 0, 1, 1, 2, 2]])  # self loops at 1 (once) and 2 (twice)
    new_ei, w = orc.gcn_norm(ei, 4, add_self_loops=True)
    assert new_ei.shape[1] == 3 + 4  # 3 non-self edges + one loop per node (incl. isolated node 3)
    assert np.array_equal(new_ei[:, -4:], np.stack([np.arange(4), np.arange(4)]))
    deg = np.array([2, 3, 1, 1], dtype=np.float32)  # in-degree incl. loop
    np.testing.assert_allclose(w, (deg[new_ei[0]] ** -0.5) * (deg[new_ei[1]] ** -0.5), rtol=1e-6)
    # without self loops: isolated destination 3 has degree 0 -> weight 0, not inf
    ei2 = np.array([[3, 0], [0, 1]])
    _, w2 = orc.gcn_norm(ei2, 4, add_self_loops=False)
    assert np.all(np.isfinite(w2)) and w2[0] == 0.0


def test_add_remaining_self_loops_infers_n_from_max_index():
    ei = np.array([[0, 1], [1, 0]])
    out, _ = orc.add_remaining_self_loops(ei)          # optimized_layers.py:164 passes no num_nodes
    assert out.shape[1] == 4 and out.max() == 1
    out, _ = orc.add_remaining_self_loops(ei, num_nodes=5)
    assert out.shape[1] == 7


def test_two_reference_layers_agree_in_the_oracle():
    """SURVEY.md 8a notes 1-2 (layout permutation + pre-self-looped edges)."""
    rng = np.random.default_rng(1)
    n, f, H, B, A = 60, 32, 4, 4, 3
    ei = rng.integers(0, n, size=(2, 400))
    ei = ei[:, ei[0] != ei[1]]
    looped = np.concatenate([ei, np.stack([np.arange(n), np.arange(n)])], axis=1)
    x = rng.standard_normal((n, f)).astype(np.float32)
    bw = [rng.standard_normal((f, f // H)).astype(np.float32) for _ in range(B)]
    cw = rng.standard_normal((H * B * A, f)).astype(np.float32)
    cb = rng.standard_normal(H * B * A).astype(np.float32)
    bias = rng.standard_normal(f).astype(np.float32)
    lay = orc.efficient_graph_conv_forward(x, looped, bw, cw, cb, bias, H, ["symadd", "max", "mean"])
    cw_o = cw.reshape(H, B, A, f).transpose(0, 2, 1, 3).reshape(H * A * B, f)
    cb_o = cb.reshape(H, B, A).transpose(0, 2, 1).reshape(-1)
    opt = orc.egconv_forward(x, ei, np.concatenate(bw, axis=1), cw_o, cb_o, bias, H, B, ["symnorm", "max", "mean"])
    assert rel_err(lay, opt) <= 2e-6


@pytest.mark.parametrize("hidden,H,B,A,layers,extra,total", [
    # EgcZincNet: Embedding(28,h) + 4 x (conv + BatchNorm1d) + mlp([h, h/2, h/4, 1]) -- output/pretrained.txt:41,129
    (168, 8, 4, 1, 4, "zinc", 102861),
    (124, 4, 4, 3, 4, "zinc", 100385),
])
def test_parameter_count_known_answers(hidden, H, B, A, layers, extra, total):
    """Reference known answers (output/pretrained.txt 'Total Params')."""
    conv = orc.layer_param_count(hidden, hidden, H, B, A)
    bn = 2 * hidden
    emb = 28 * hidden
    h2, h4 = hidden // 2, hidden // 4
    mlp = (hidden * h2 + h2) + 2 * h2 + (h2 * h4 + h4) + 2 * h4 + (h4 * 1 + 1)
    assert emb + layers * (conv + bn) + mlp == total


def test_glorot_bound():
    assert abs(orc.glorot_bound(128, 16) - (6 / 144) ** 0.5) < 1e-12
