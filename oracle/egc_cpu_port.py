"""Multi-threaded CPU port of the reference's op sequence -- TEST / BASELINE INFRASTRUCTURE ONLY.

``egconv_forward_cpu`` restates ``EGConv.forward`` (reference experiments/optimized_layers.py:124-249)
as the SAME unfused sequence of ATen ops the PyG CPU path executes -- gcn_norm (scatter_add, pow,
two gathers), one index_select gather of ``bases[edge_index[0]]``, one scatter reduction per
aggregator, stack, bmm, bias -- using ``Tensor.index_add_`` / ``scatter_reduce_`` where the reference
calls torch_scatter.  It exists to be TIMED on the GPU box's host cores as ``cpu_baseline``
(kind "port") in bench.py and is checked against the numpy oracle in tests/test_oracle.py.
Parity status: unpinned by reference tests (see oracle/egc_oracle.py header).  Never imported by egc_amd.
"""
from __future__ import annotations

import torch


def _scatter(src, index, n, reduce):
    f = src.size(1)
    if reduce == "sum":
        return torch.zeros(n, f).index_add_(0, index, src)
    if reduce == "mean":
        out = torch.zeros(n, f).index_add_(0, index, src)
        cnt = torch.zeros(n).index_add_(0, index, torch.ones(index.numel())).clamp_(min=1)
        return out / cnt.view(-1, 1)
    red = "amax" if reduce == "max" else "amin"
    return torch.zeros(n, f).scatter_reduce_(0, index.view(-1, 1).expand(-1, f), src, red, include_self=False)


def gcn_norm_cpu(edge_index, n, add_self_loops=True):
    row, col = edge_index[0], edge_index[1]
    w = torch.ones(row.numel())
    if add_self_loops:
        mask = row != col
        loop = torch.arange(n)
        row, col = torch.cat([row[mask], loop]), torch.cat([col[mask], loop])
        w = torch.ones(row.numel())
    deg = torch.zeros(n).index_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    return torch.stack([row, col]), dis[row] * w * dis[col]


def egconv_forward_cpu(x, edge_index, bases_weight, comb_w, comb_b, bias, num_heads, num_bases, aggrs,
                       add_self_loops=True, cached=None):
    """x [N,F_in] f32, edge_index int64 [2,E].  ``cached`` = (edge_index', symnorm_weight) from a
    previous call (the reference's cached=True) or None."""
    n = x.size(0)
    sw = None
    if cached is not None:
        ei, sw = cached
    elif "symnorm" in aggrs:
        ei, sw = gcn_norm_cpu(edge_index, n, add_self_loops)
    elif add_self_loops:
        mask = edge_index[0] != edge_index[1]
        loop = torch.arange(int(edge_index.max()) + 1 if edge_index.numel() else 0)
        ei = torch.cat([edge_index[:, mask], torch.stack([loop, loop])], dim=1)
    else:
        ei = edge_index
    bases = x @ bases_weight
    weightings = torch.addmm(comb_b, x, comb_w.t())
    x_j = bases.index_select(0, ei[0])
    outs = []
    for a in aggrs:
        if a == "symnorm":
            outs.append(_scatter(x_j * sw.view(-1, 1), ei[1], n, "sum"))
        elif a in ("var", "std"):
            mean = _scatter(x_j, ei[1], n, "mean")
            mean_sq = _scatter(x_j * x_j, ei[1], n, "mean")
            o = mean_sq - mean * mean
            outs.append(torch.sqrt(torch.relu(o) + 1e-5) if a == "std" else o)
        else:
            outs.append(_scatter(x_j, ei[1], n, a))
    agg = torch.stack(outs, dim=1)
    f_out = bases.size(1) // num_bases * num_heads
    w3 = weightings.view(n, num_heads, num_bases * len(aggrs))
    a3 = agg.view(n, len(aggrs) * num_bases, f_out // num_heads)
    out = torch.matmul(w3, a3).view(n, f_out)
    if bias is not None:
        out = out + bias
    return out, (ei, sw)
