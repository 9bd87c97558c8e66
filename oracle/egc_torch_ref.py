"""Differentiable CPU restatement of the two reference layers -- TEST INFRASTRUCTURE ONLY.

Same algorithm as oracle/egc_oracle.py (reference call sites cited there), written with torch ops so
that autograd yields the gradients the reference's training loops rely on (zinc/configs.py:64-67,
arxiv/configs.py:52-57).  Run in float64 it is the gradient oracle for the HIP backward
(tests/test_backward_gpu.py).  Max/min gradients go to the FIRST entry (edge order; an appended self loop
is last) that attains the extremum -- torch_scatter's CPU arg rule (scatter_max/segment_csr update only on
a strict improvement), which the reference's layers inherit (layers.py:208-219, optimized_layers.py:215-244).
Parity status: unpinned by reference tests (see oracle/egc_oracle.py header).  Never imported by egc_amd.
"""
from __future__ import annotations

import numpy as np
import torch

from . import egc_oracle as orc


def _scatter(src, index, n, reduce):
    f = src.size(1)
    if reduce == "sum":
        return torch.zeros(n, f, dtype=src.dtype).index_add(0, index, src)
    if reduce == "mean":
        out = torch.zeros(n, f, dtype=src.dtype).index_add(0, index, src)
        cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones(index.numel(), dtype=src.dtype))
        return out / cnt.clamp(min=1).view(-1, 1)
    red = "amax" if reduce == "max" else "amin"
    idx = index.view(-1, 1).expand(-1, f)
    with torch.no_grad():
        # Ties are decided on the values ROUNDED TO FLOAT32, the precision the reference computes in: in float64 two
        # sources with identical features can differ in the last bit of x @ W (the BLAS blocks rows differently),
        # which would break an exact tie at random instead of by edge order.
        s32 = src.float()
        ext = torch.zeros(n, f, dtype=s32.dtype).scatter_reduce(0, idx, s32, red, include_self=False)
        e = src.size(0)
        pos = torch.arange(e).view(-1, 1).expand(-1, f)
        pos = torch.where(s32 == ext[index], pos, torch.full_like(pos, e))
        first = torch.full((n, f), e, dtype=torch.int64).scatter_reduce(0, idx, pos, "amin", include_self=True)
        empty = first >= e
    picked = torch.gather(src, 0, first.clamp(max=max(e - 1, 0))) if e > 0 else torch.zeros(n, f, dtype=src.dtype)
    return torch.where(empty, torch.zeros_like(picked), picked)


def _aggregate(a, x_j, index, n, weight):
    if a in ("sum", "add"):
        return _scatter(x_j, index, n, "sum")
    if a in ("symnorm", "symadd"):
        return _scatter(x_j * weight.view(-1, 1), index, n, "sum")
    if a in ("mean", "max", "min"):
        return _scatter(x_j, index, n, a)
    mean = _scatter(x_j, index, n, "mean")
    var = _scatter(x_j * x_j, index, n, "mean") - mean * mean
    return torch.sqrt(torch.relu(var) + 1e-5) if a == "std" else var


def egconv_forward(x, edge_index, bases_weight, comb_w, comb_b, bias, H, B, aggrs, add_self_loops=True, sigmoid=False):
    """EGConv.forward (optimized_layers.py:124-210), differentiable."""
    n = x.size(0)
    ei, sw = orc.egconv_edge_set(np.asarray(edge_index), n, list(aggrs), add_self_loops)
    ei = torch.from_numpy(ei)
    sw = None if sw is None else torch.from_numpy(sw).to(x.dtype)
    bases = x @ bases_weight
    w = x @ comb_w.t() + comb_b
    if sigmoid:
        w = torch.sigmoid(w)
    x_j = bases[ei[0]]
    agg = torch.stack([_aggregate(a, x_j, ei[1], n, sw) for a in aggrs], dim=1)  # [N, A, B*L]
    f_out = bases.size(1) // B * H
    out = torch.matmul(w.view(n, H, B * len(aggrs)), agg.view(n, len(aggrs) * B, f_out // H)).reshape(n, f_out)
    return out if bias is None else out + bias


def efficient_graph_conv_forward(x, edge_index, bases_weight, comb_w, comb_b, bias, H, aggrs, softmax=False,
                                 sigmoid=False, hardtanh=False, add_self_loops=True):
    """EfficientGraphConv.forward (layers.py:89-140), differentiable.  bases_weight: list of B [F_in, L]."""
    n = x.size(0)
    nb, na = len(bases_weight), len(aggrs)
    bases = torch.stack([x @ w for w in bases_weight], dim=1).reshape(n, -1)
    ei_raw = torch.from_numpy(np.asarray(edge_index))
    ys = []
    for a in aggrs:
        if a == "symadd":
            ei, sw = orc.gcn_norm(np.asarray(edge_index), n, add_self_loops)
            ei, sw = torch.from_numpy(ei), torch.from_numpy(sw).to(x.dtype)
            y = _aggregate("symadd", bases[ei[0]], ei[1], n, sw)
        else:
            y = _aggregate(a, bases[ei_raw[0]], ei_raw[1], n, None)
        ys.append(y.view(n, nb, -1))
    y = torch.stack(ys, dim=2)  # N x B x A x L
    w = x @ comb_w.t() + comb_b
    if softmax:
        w = w.view(n, H, nb * na).softmax(dim=-1)
    elif sigmoid:
        w = torch.sigmoid(w)
    elif hardtanh:
        w = torch.nn.functional.hardtanh(w)
    w = w.view(n, H, nb, na, 1)
    z = (w * y.unsqueeze(1)).sum(dim=(2, 3)).reshape(n, -1)
    return z if bias is None else z + bias
