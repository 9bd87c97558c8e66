"""CPU oracle for the EGC layer forward -- TEST INFRASTRUCTURE, NOT PRODUCT.

This module is a plain-numpy (float32 / int64) restatement of the reference's
algorithm for the one hot path this repository accelerates: the forward of
``EfficientGraphConv`` (reference ``experiments/layers.py:11-228``) and of
``EGConv`` (reference ``experiments/optimized_layers.py:19-286``), including
the third-party operators those two files reach (PyG ``gcn_norm`` /
``add_remaining_self_loops`` / ``MessagePassing.propagate``,
``torch_scatter.scatter``, ``torch_sparse.matmul``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``egc_amd``) never imports it and fails loudly when the HIP library is
missing.

PARITY PIN STATUS: **parity unpinned by reference-owned tests** -- the
reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4), and the third-party libraries it calls
(torch-geometric==2.0, torch-scatter, torch-sparse; Dockerfile:48-54) are
neither vendored under /root/reference nor installable here.  What pins this
oracle instead:
  * ``tests/golden/*.npz`` -- outputs of the reference's OWN layer code
    (layers.py / optimized_layers.py imported from /root/reference in the
    build container by ``tests/golden/make_golden.py``) with only the
    third-party operators substituted by the restatements in this file;
  * the cross-check between the two reference layers (SURVEY.md 8a notes 1-2);
  * parameter-count / repr known answers from output/pretrained.txt;
  * an independent check of ``scatter`` against ``torch.Tensor.index_add_`` /
    ``scatter_reduce_`` (tests/test_oracle.py).

Third-party algorithms restated here (published behaviour of the pinned
versions; call sites in the reference are cited at each function):
  torch-geometric 2.0.x : utils.add_remaining_self_loops, nn.conv.gcn_conv.gcn_norm,
                          nn.conv.MessagePassing.propagate (source_to_target flow)
  torch-scatter 2.0.x   : scatter(reduce=sum|mean|min|max) CPU kernel semantics
  torch-sparse 0.6.x    : matmul(adj_t, x, reduce=...) == the same reductions over CSR rows
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

AGGRS_LAYERS = ("add", "mean", "max", "min", "symadd", "var", "std")  # layers.py:154-159
AGGRS_OPT = ("sum", "mean", "symnorm", "min", "max", "var", "std")  # optimized_layers.py:92-94


# --------------------------------------------------------------------------
# third-party restatements
# --------------------------------------------------------------------------
def maybe_num_nodes(edge_index: np.ndarray, num_nodes=None) -> int:
    """PyG ``utils.num_nodes.maybe_num_nodes``: max index + 1 when not given."""
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.size > 0 else 0


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """PyG 2.0 ``add_remaining_self_loops`` (called at optimized_layers.py:164 WITHOUT
    num_nodes, and from gcn_norm WITH num_nodes).

    Every existing self-loop is dropped, then exactly one self-loop per node
    0..N-1 is appended AFTER the remaining edges (so it is last in edge order).
    """
    edge_index = np.asarray(edge_index, dtype=np.int64)
    n = maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = np.arange(n, dtype=np.int64)
    new_index = np.concatenate([edge_index[:, mask], np.stack([loop, loop])], axis=1)
    new_weight = None
    if edge_weight is not None:
        edge_weight = np.asarray(edge_weight, dtype=F32)
        loop_w = np.full((n,), fill_value, dtype=F32)
        inv = ~mask
        if inv.any():
            loop_w[row[inv]] = edge_weight[inv]
        new_weight = np.concatenate([edge_weight[mask], loop_w])
    return new_index, new_weight


def gcn_norm(edge_index, num_nodes, add_self_loops=True):
    """PyG 2.0 ``gcn_norm(edge_index, None, num_nodes, improved=False, add_self_loops)``
    for a dense COO edge_index (layers.py:173-178, optimized_layers.py:131-137).

    deg is the IN-degree by destination (edge_index[1]) of the (self-looped)
    edge set; w_e = deg[src]^-1/2 * 1 * deg[dst]^-1/2 with inf -> 0.
    """
    edge_index = np.asarray(edge_index, dtype=np.int64)
    n = maybe_num_nodes(edge_index, num_nodes)
    w = np.ones((edge_index.shape[1],), dtype=F32)
    if add_self_loops:
        edge_index, w = add_remaining_self_loops(edge_index, w, 1.0, n)
    row, col = edge_index[0], edge_index[1]
    deg = np.zeros((n,), dtype=F32)
    np.add.at(deg, col, w)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, F32(-0.5), dtype=F32)
    dis[np.isinf(dis)] = 0
    return edge_index, (dis[row] * w * dis[col]).astype(F32)


def scatter(src: np.ndarray, index: np.ndarray, dim_size: int, reduce: str):
    """``torch_scatter.scatter(src, index, dim=0, dim_size=dim_size, reduce=...)``
    CPU semantics (layers.py:203-212 + MessagePassing.aggregate; optimized_layers.py:225-240).

    sum : sequential f32 accumulation in edge order.
    mean: sum / clamp(count, 1).
    min/max: strict compare in edge order (first edge attaining the extremum
             keeps the argument); rows with no edge give 0 and arg == E.
    Returns (out, arg) -- arg is None for sum/mean.
    """
    src = np.ascontiguousarray(src, dtype=F32)
    index = np.asarray(index, dtype=np.int64)
    e = src.shape[0]
    out_shape = (dim_size,) + src.shape[1:]
    if reduce in ("sum", "add"):
        out = np.zeros(out_shape, dtype=F32)
        np.add.at(out, index, src)
        return out, None
    if reduce == "mean":
        out = np.zeros(out_shape, dtype=F32)
        np.add.at(out, index, src)
        cnt = np.zeros((dim_size,), dtype=F32)
        np.add.at(cnt, index, F32(1))
        cnt = np.maximum(cnt, F32(1))
        return (out / cnt.reshape((-1,) + (1,) * (src.ndim - 1))).astype(F32), None
    if reduce in ("max", "min"):
        # stable sort by destination keeps input order inside a row, then a
        # first-occurrence arg-extremum per segment == the sequential strict-compare loop.
        order = np.argsort(index, kind="stable")
        sidx = index[order]
        ssrc = src[order]
        out = np.zeros(out_shape, dtype=F32)
        arg = np.full(out_shape, e, dtype=np.int64)
        if e == 0:
            return out, arg
        bounds = np.flatnonzero(np.diff(sidx)) + 1
        starts = np.concatenate([[0], bounds])
        ends = np.concatenate([bounds, [e]])
        red = np.maximum if reduce == "max" else np.minimum
        vals = red.reduceat(ssrc, starts, axis=0)
        rows = sidx[starts]
        out[rows] = vals
        # first position (in input order) attaining the extremum
        seg_id = np.repeat(np.arange(len(starts)), ends - starts)
        hit = ssrc == vals[seg_id]
        pos = np.where(hit, order.reshape((-1,) + (1,) * (src.ndim - 1)), e)
        arg[rows] = np.minimum.reduceat(pos, starts, axis=0)
        return out, arg
    raise ValueError(f"unknown reduce {reduce}")


def csr_from_coo(edge_index: np.ndarray, num_nodes: int):
    """Stable COO -> CSR keyed by destination (the layout of ``adj_t``:
    experiments/utils.py:107-113).  Returns (rowptr, col(src), edge_id)."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    src, dst = edge_index[0], edge_index[1]
    order = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=num_nodes)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, src[order].astype(np.int64), order.astype(np.int64)


# --------------------------------------------------------------------------
# aggregators shared by both layers
# --------------------------------------------------------------------------
def _var_std(x_j, index, n, std: bool):
    """layers.py:202-216 == optimized_layers.py:237-244."""
    mean, _ = scatter(x_j, index, n, "mean")
    mean_sq, _ = scatter((x_j * x_j).astype(F32), index, n, "mean")
    out = (mean_sq - (mean * mean).astype(F32)).astype(F32)
    if std:
        out = np.sqrt(np.maximum(out, F32(0)) + F32(1e-5), dtype=F32)
    return out


# --------------------------------------------------------------------------
# EfficientGraphConv (experiments/layers.py)
# --------------------------------------------------------------------------
def agg_layer_forward(aggr: str, bases: np.ndarray, edge_index: np.ndarray, add_self_loops: bool):
    """``_AggLayer.forward`` + message + aggregate for one aggregator
    (layers.py:165-219).  Only ``symadd`` touches self-loops (via gcn_norm);
    every other aggregator sees the raw edge list.
    Returns (out [N, B*L], arg or None)."""
    n = bases.shape[0]
    edge_index = np.asarray(edge_index, dtype=np.int64)
    if aggr == "symadd":
        ei, w = gcn_norm(edge_index, n, add_self_loops)
        x_j = (w.reshape(-1, 1) * bases[ei[0]]).astype(F32)  # layers.py:195-197
        return scatter(x_j, ei[1], n, "sum")
    if aggr == "min":
        out, arg = scatter(-bases[edge_index[0]], edge_index[1], n, "max")  # layers.py:190-191
        return (-out).astype(F32), arg
    x_j = bases[edge_index[0]]
    if aggr in ("var", "std"):
        return _var_std(x_j, edge_index[1], n, aggr == "std"), None
    if aggr in ("add", "mean", "max"):
        return scatter(x_j, edge_index[1], n, "sum" if aggr == "add" else aggr)
    raise ValueError(aggr)


def efficient_graph_conv_forward(
    x,
    edge_index,
    bases_weight,  # list of B arrays [F_in, L]
    comb_w,  # [H*B*A, F_in]  (nn.Linear weight)
    comb_b,  # [H*B*A]
    bias,  # [F_out] or None
    num_heads,
    aggrs,
    softmax_weights=False,
    sigmoid_weights=False,
    hardtanh_weights=False,
    add_self_loops=True,
    return_intermediates=False,
):
    """``EfficientGraphConv.forward`` (layers.py:89-140).  Weight column index is
    h*B*A + b*A + a (B major, A minor)."""
    x = np.ascontiguousarray(x, dtype=F32)
    n = x.shape[0]
    nb = len(bases_weight)
    na = len(aggrs)
    bases = np.stack([(x @ np.asarray(w, dtype=F32)).astype(F32) for w in bases_weight], axis=1)
    bases = bases.reshape(n, -1)  # N x BL, column = b*L + l
    aggregated, args = [], {}
    for a in aggrs:
        y, arg = agg_layer_forward(a, bases, edge_index, add_self_loops)
        if arg is not None:
            args[a] = arg
        aggregated.append(y.reshape(n, nb, -1))
    y = np.stack(aggregated, axis=2)  # N x B x A x L
    w = (x @ np.asarray(comb_w, dtype=F32).T + np.asarray(comb_b, dtype=F32)).astype(F32)
    w_pre = w
    if softmax_weights:
        w = w.reshape(n, num_heads, nb * na)
        w = w - w.max(axis=-1, keepdims=True)
        ew = np.exp(w, dtype=F32)
        w = (ew / ew.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    elif sigmoid_weights:
        w = (F32(1) / (F32(1) + np.exp(-w, dtype=F32))).astype(F32)
    elif hardtanh_weights:
        w = np.clip(w, F32(-1), F32(1))
    w = w.reshape(n, num_heads, nb, na, 1)
    z = (w * y[:, None]).astype(F32).sum(axis=(2, 3), dtype=F32)  # N x H x L
    z = z.reshape(n, -1)
    if bias is not None:
        z = z + np.asarray(bias, dtype=F32)
    z = z.astype(F32)
    if return_intermediates:
        return z, {"bases": bases, "weightings": w_pre, "aggregated": y, "args": args}
    return z


# --------------------------------------------------------------------------
# EGConv (experiments/optimized_layers.py)
# --------------------------------------------------------------------------
def egconv_edge_set(edge_index, num_nodes, aggrs, add_self_loops):
    """Graph preparation of ``EGConv.forward`` (optimized_layers.py:125-175) for a dense
    COO edge_index.  Returns (edge_index', symnorm_weight or None).

    NOTE the two branches differ: with ``symnorm`` the self-loops come from
    gcn_norm (num_nodes known => every node gets one); without it they come
    from ``add_remaining_self_loops(edge_index)`` which infers N from the
    largest index present (optimized_layers.py:164)."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    if "symnorm" in aggrs:
        return gcn_norm(edge_index, num_nodes, add_self_loops)
    if add_self_loops:
        ei, _ = add_remaining_self_loops(edge_index)
        return ei, None
    return edge_index, None


def egconv_aggregate(inputs, index, n, aggrs, symnorm_weight):
    """``EGConv.aggregate`` (optimized_layers.py:215-249) -> [N, A, B*L]."""
    outs, args = [], {}
    for a in aggrs:
        if a == "sum":
            out, _ = scatter(inputs, index, n, "sum")
        elif a == "symnorm":
            assert symnorm_weight is not None
            out, _ = scatter((inputs * symnorm_weight.reshape(-1, 1)).astype(F32), index, n, "sum")
        elif a == "mean":
            out, _ = scatter(inputs, index, n, "mean")
        elif a in ("min", "max"):
            out, arg = scatter(inputs, index, n, a)
            args[a] = arg
        elif a in ("var", "std"):
            out = _var_std(inputs, index, n, a == "std")
        else:
            raise ValueError(f'Unknown aggregator "{a}".')
        outs.append(out)
    return np.stack(outs, axis=1), args


def egconv_forward(
    x,
    edge_index,
    bases_weight,  # [F_in, B*L]
    comb_w,  # [H*B*A, F_in]
    comb_b,  # [H*B*A]
    bias,  # [F_out] or None
    num_heads,
    num_bases,
    aggrs,
    add_self_loops=True,
    sigmoid=False,
    return_intermediates=False,
):
    """``EGConv.forward`` (optimized_layers.py:124-210).  Weight column index is
    h*A*B + a*B + b (A major, B minor)."""
    x = np.ascontiguousarray(x, dtype=F32)
    n = x.shape[0]
    na = len(aggrs)
    ei, sw = egconv_edge_set(edge_index, n, aggrs, add_self_loops)
    bases = (x @ np.asarray(bases_weight, dtype=F32)).astype(F32)
    w = (x @ np.asarray(comb_w, dtype=F32).T + np.asarray(comb_b, dtype=F32)).astype(F32)
    w_pre = w
    if sigmoid:
        w = (F32(1) / (F32(1) + np.exp(-w, dtype=F32))).astype(F32)
    aggregated, args = egconv_aggregate(bases[ei[0]], ei[1], n, aggrs, sw)
    w3 = w.reshape(n, num_heads, num_bases * na)
    f_out = bases.shape[1] // num_bases * num_heads
    agg3 = aggregated.reshape(n, na * num_bases, f_out // num_heads)
    out = np.matmul(w3, agg3).astype(F32).reshape(n, f_out)
    if bias is not None:
        out = out + np.asarray(bias, dtype=F32)
    out = out.astype(F32)
    if return_intermediates:
        return out, {"bases": bases, "weightings": w_pre, "aggregated": aggregated, "args": args,
                     "edge_index": ei, "symnorm_weight": sw}
    return out


# --------------------------------------------------------------------------
# initialisers (layers.py:82-87, optimized_layers.py:117-122)
# --------------------------------------------------------------------------
def glorot_bound(fan_in: int, fan_out: int) -> float:
    """PyG ``inits.glorot``: U(-a, a) with a = sqrt(6 / (size(-2) + size(-1)))."""
    return float(np.sqrt(6.0 / (fan_in + fan_out)))


def layer_param_count(f_in, f_out, num_heads, num_bases, num_aggrs, bias=True) -> int:
    """Per-layer parameter count (SURVEY.md section 4 known answers):
    comb Linear (F_in*HBA + HBA) + B bases of F_in x (F_out/H) + bias."""
    hba = num_heads * num_bases * num_aggrs
    return f_in * hba + hba + num_bases * f_in * (f_out // num_heads) + (f_out if bias else 0)


# ---------------------------------------------------------------------------------------------
# relational EGC
# ---------------------------------------------------------------------------------------------
def regconv_forward(x_dict, adj_dict, bases_weight, rel_combs, root_combs, num_heads: int, num_bases: int):
    """REGConv.forward (reference experiments/rmag/models.py:112-148), float32 numpy.

    x_dict[type] = [N_type, F_in]; adj_dict[(src, rel, dst)] = int64 [2, E] with row 0 = SOURCE ids (columns of
    the reference's adj_t) and row 1 = TARGET ids (rows of adj_t); rel_combs["src_rel_dst"] / root_combs[type]
    = (weight [out, in], bias [out]) of the reference's Linears.  ``adj_t.matmul(x, reduce=...)`` is the
    per-target-row reduction of x[source] (torch_sparse spmm), empty rows giving 0 (``scatter`` above)."""
    H, B = num_heads, num_bases
    f32 = np.float32
    bases = {k: (x.astype(f32) @ bases_weight.astype(f32)) for k, x in x_dict.items()}            # :113-115
    L = bases_weight.shape[1] // B
    out = {}
    for k, x in x_dict.items():                                                                    # :117-129
        w, b = root_combs[k]
        weightings = (x.astype(f32) @ w.astype(f32).T + b.astype(f32)).reshape(-1, H, B)
        out[k] = np.matmul(weightings, bases[k].reshape(-1, B, L))                                 # [N, H, L]
    for key, ei in adj_dict.items():                                                               # :131-144
        src, _, dst = key
        n_dst = x_dict[dst].shape[0]
        gathered = bases[src][ei[0]]
        mean, _ = scatter(gathered, ei[1], n_dst, "mean")
        mx, _ = scatter(gathered, ei[1], n_dst, "max")
        aggregated = np.stack([mean, mx], axis=1).reshape(-1, 2 * B, L)
        w, b = rel_combs[f"{key[0]}_{key[1]}_{key[2]}"]
        weightings = (x_dict[dst].astype(f32) @ w.astype(f32).T + b.astype(f32)).reshape(-1, H, 2 * B)
        out[dst] = out[dst] + np.matmul(weightings, aggregated)
    return {k: v.reshape(v.shape[0], H * L).astype(f32) for k, v in out.items()}                    # :146-148 (H * L: empty types too)
