#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own layer code.

Runs only in the build container (needs /root/reference).  The reference's
``experiments/layers.py`` and ``experiments/optimized_layers.py`` are imported
as-is via importlib; the third-party packages they import (torch_geometric,
torch_scatter, torch_sparse -- absent here, no network) are replaced in
``sys.modules`` by thin shims whose numerical bodies are the restatements in
``oracle/egc_oracle.py``.  Everything reference-OWNED (basis stacking, weight
layout, softmax axis, combine, var/std formula, min-by-negation, self-loop
policy, caching) is therefore executed by the reference's own code.

Only the resulting vectors (inputs, parameters, outputs) are committed; no
reference source travels.  Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import inspect
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import egc_oracle as orc  # noqa: E402

REF = "/root/reference/experiments"


# --------------------------------------------------------------------------
# shims
# --------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def shim_scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert out is None and src.dim() == 2 and dim in (0, -2)
    res, _ = orc.scatter(_np(src), _np(index), int(dim_size), reduce)
    return torch.from_numpy(res)


class SparseTensor:
    """Minimal stand-in for torch_sparse.SparseTensor holding adj_t in COO sorted by
    (row=dst, col=src), optional value."""

    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, is_sorted=False):
        row, col = row.long(), col.long()
        if not is_sorted:
            perm = (row * sparse_sizes[1] + col).argsort(stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.row, self.col, self.value, self.sizes = row, col, value, tuple(sparse_sizes)

    def has_value(self):
        return self.value is not None

    def set_value(self, value, layout=None):
        return SparseTensor(self.row, self.col, value, self.sizes, True)

    def fill_value(self, v):
        return self.set_value(torch.full((self.row.numel(),), float(v)))

    def sparse_size(self, d):
        return self.sizes[d]


def shim_fill_diag(adj, fill_value):
    """torch_sparse.diag.fill_diag: drop existing diagonal entries, insert one per row."""
    n = adj.sizes[0]
    mask = adj.row != adj.col
    loop = torch.arange(n)
    row = torch.cat([adj.row[mask], loop])
    col = torch.cat([adj.col[mask], loop])
    val = None
    if adj.value is not None:
        val = torch.cat([adj.value[mask], torch.full((n,), float(fill_value))])
    elif fill_value != 1.0:
        raise NotImplementedError
    out = SparseTensor(row, col, val, adj.sizes, False)
    if val is None:
        out = out.fill_value(1.0)
    return out


def shim_sparse_matmul(adj, x, reduce="sum"):
    """torch_sparse.matmul(adj_t, x, reduce): per-row reduction of value * x[col]."""
    src = x[adj.col]
    if adj.value is not None:
        src = src * adj.value.view(-1, 1)
    return shim_scatter(src, adj.row, 0, None, adj.sizes[0], "sum" if reduce == "add" else reduce)


def shim_gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False,
                  add_self_loops=True, dtype=None):
    assert edge_weight is None and not improved
    if isinstance(edge_index, SparseTensor):
        adj = edge_index
        if not adj.has_value():
            adj = adj.fill_value(1.0)
        if add_self_loops:
            adj = shim_fill_diag(adj, 1.0)
        deg = torch.zeros(adj.sizes[0]).index_add_(0, adj.row, adj.value)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0.0
        return adj.set_value(dis[adj.row] * adj.value * dis[adj.col])
    ei, w = orc.gcn_norm(_np(edge_index), num_nodes, add_self_loops)
    return torch.from_numpy(ei), torch.from_numpy(w)


def shim_add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    ei, w = orc.add_remaining_self_loops(_np(edge_index), None if edge_attr is None else _np(edge_attr),
                                         1.0 if fill_value is None else fill_value, num_nodes)
    return torch.from_numpy(ei), (None if w is None else torch.from_numpy(w))


class MessagePassing(torch.nn.Module):
    """PyG 2.0 MessagePassing restricted to what the two layer files use:
    source_to_target flow, x_j gather, signature-driven message/aggregate dispatch,
    fused message_and_aggregate for SparseTensor inputs."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self.fuse = type(self).message_and_aggregate is not MessagePassing.message_and_aggregate

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs["x"]
        n = x.size(self.node_dim)
        if isinstance(edge_index, SparseTensor):
            assert self.fuse
            return self.message_and_aggregate(edge_index, x)
        j, i = edge_index[0], edge_index[1]
        avail = dict(kwargs)
        avail.update(x_j=x.index_select(self.node_dim, j), x_i=x.index_select(self.node_dim, i),
                     index=i, ptr=None, dim_size=n, size_i=n, size_j=n)
        msg_args = {k: avail[k] for k in inspect.signature(self.message).parameters}
        out = self.message(**msg_args)
        agg_params = list(inspect.signature(self.aggregate).parameters)[1:]
        out = self.aggregate(out, **{k: avail[k] for k in agg_params})
        return out

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return shim_scatter(inputs, index, self.node_dim, None, dim_size, self.aggr)

    def message_and_aggregate(self, adj_t, x):
        raise NotImplementedError


def shim_glorot(t):
    if t is not None:
        a = orc.glorot_bound(t.size(-2), t.size(-1))
        t.data.uniform_(-a, a)


def shim_zeros(t):
    if t is not None:
        t.data.fill_(0)


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    from typing import Optional, Union
    mod("torch_scatter", scatter=shim_scatter)
    mod("torch_sparse", SparseTensor=SparseTensor, matmul=shim_sparse_matmul)
    mod("torch_sparse.diag", fill_diag=shim_fill_diag)
    mod("torch_geometric")
    mod("torch_geometric.nn", MessagePassing=MessagePassing)
    mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    mod("torch_geometric.nn.conv.gcn_conv", gcn_norm=shim_gcn_norm)
    mod("torch_geometric.nn.inits", glorot=shim_glorot, zeros=shim_zeros)
    mod("torch_geometric.typing", Adj=Union[torch.Tensor, SparseTensor], OptTensor=Optional[torch.Tensor])
    mod("torch_geometric.utils", add_remaining_self_loops=shim_add_remaining_self_loops)


def load_ref(name):
    spec = importlib.util.spec_from_file_location(f"ref_{name}", os.path.join(REF, f"{name}.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# --------------------------------------------------------------------------
# graphs
# --------------------------------------------------------------------------
def rand_graph(rng, n, e, self_loops=0, dups=0, isolated_tail=0):
    """Random directed multigraph; the last `isolated_tail` nodes never appear."""
    hi = n - isolated_tail
    src = rng.integers(0, hi, size=e)
    dst = rng.integers(0, hi, size=e)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if dups:
        k = rng.integers(0, len(src), size=dups)
        src, dst = np.concatenate([src, src[k]]), np.concatenate([dst, dst[k]])
    if self_loops:
        s = rng.integers(0, hi, size=self_loops)
        src, dst = np.concatenate([src, s]), np.concatenate([dst, s])
    perm = rng.permutation(len(src))
    return np.stack([src[perm], dst[perm]]).astype(np.int64)


def cases():
    """(name, kind, cfg) -- kind 'lay' = EfficientGraphConv, 'opt' = EGConv."""
    out = []
    g_plain = dict(n=48, e=200)
    g_messy = dict(n=57, e=260, self_loops=9, dups=25, isolated_tail=3)
    # every aggregator alone, both layers, messy graph (isolated nodes, dups, self loops)
    for a in orc.AGGRS_LAYERS:
        out.append((f"lay_single_{a}", "lay", dict(fin=24, fout=32, H=4, B=2, aggrs=[a], graph=g_messy)))
    for a in orc.AGGRS_OPT:
        out.append((f"opt_single_{a}", "opt", dict(fin=24, fout=32, H=4, B=2, aggrs=[a], graph=g_messy)))
    # shipped combos (run_pretrained.sh / hyperparameters.md) at shipped head/base counts, odd L
    combos = [["symadd"], ["add", "std", "max"], ["symadd", "std", "max"], ["add", "mean", "max"],
              ["symadd", "max", "mean"], ["symadd", "min", "max"], ["mean"]]
    shapes = [(42, 8, 4), (62, 4, 4), (46, 8, 4), (64, 4, 4)]  # (hidden,H,B): L = 5(odd),15,5,16 ... scaled-down
    for i, c in enumerate(combos):
        hid, H, B = shapes[i % len(shapes)]
        hid = (hid // H) * H
        out.append((f"lay_combo_{'-'.join(c)}", "lay",
                    dict(fin=hid, fout=hid, H=H, B=B, aggrs=c, graph=g_messy if i % 2 else g_plain)))
    # true shipped L values: 21 (168/8), 31 (124/4), 23 (184/8)
    out.append(("lay_L21", "lay", dict(fin=168, fout=168, H=8, B=4, aggrs=["symadd"], graph=g_plain)))
    out.append(("lay_L31", "lay", dict(fin=124, fout=124, H=4, B=4, aggrs=["add", "std", "max"], graph=g_messy)))
    out.append(("lay_L23", "lay", dict(fin=184, fout=184, H=8, B=4, aggrs=["symadd"], graph=g_messy)))
    # weight nonlinearities
    for nl in ("softmax", "sigmoid", "hardtanh"):
        out.append((f"lay_{nl}", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["symadd", "max", "mean"],
                                             graph=g_messy, **{nl: True})))
    out.append(("opt_sigmoid", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["symnorm", "max"], graph=g_messy, sigmoid=True)))
    # flags
    out.append(("lay_noselfloops", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["symadd", "mean"], graph=g_messy, add_self_loops=False)))
    out.append(("lay_nobias", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["add", "max"], graph=g_plain, bias=False)))
    out.append(("opt_noselfloops", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["symnorm", "mean", "max"], graph=g_messy, add_self_loops=False)))
    out.append(("opt_noselfloops_nosym", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["sum", "min", "std"], graph=g_messy, add_self_loops=False)))
    out.append(("opt_nobias", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["sum", "var"], graph=g_plain, bias=False)))
    # EGConv WITHOUT symnorm but with self loops: N for the loops is inferred from max index (isolated tail!)
    out.append(("opt_selfloops_inferredN", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["sum", "mean", "max", "std"], graph=g_messy)))
    # north-star combo + head/base extremes + F_in != F_out
    out.append(("opt_northstar_small", "opt", dict(fin=128, fout=128, H=8, B=4, aggrs=["sum", "mean", "max", "symnorm"], graph=g_messy)))
    out.append(("opt_H1B1", "opt", dict(fin=16, fout=16, H=1, B=1, aggrs=["sum"], graph=g_plain)))
    out.append(("lay_H1B1_add", "lay", dict(fin=128, fout=128, H=1, B=1, aggrs=["add"], graph=g_plain)))
    out.append(("opt_H16B16", "opt", dict(fin=32, fout=64, H=16, B=16, aggrs=["symnorm", "min"], graph=g_plain)))
    out.append(("opt_mag_first", "opt", dict(fin=128, fout=352, H=8, B=4, aggrs=["mean"], graph=g_plain)))
    out.append(("opt_all7", "opt", dict(fin=40, fout=40, H=8, B=4, aggrs=list(orc.AGGRS_OPT), graph=g_messy)))
    out.append(("lay_all7", "lay", dict(fin=40, fout=40, H=8, B=4, aggrs=list(orc.AGGRS_LAYERS), graph=g_messy)))
    # exact ties for max/min (integer-valued inputs and weights)
    out.append(("opt_ties", "opt", dict(fin=8, fout=16, H=2, B=2, aggrs=["max", "min", "sum"], graph=dict(n=20, e=120, dups=30), integer=True)))
    out.append(("lay_ties", "lay", dict(fin=8, fout=16, H=2, B=2, aggrs=["max", "min", "add"], graph=dict(n=20, e=120, dups=30), integer=True)))
    # edge cases: no edges at all; single node
    out.append(("opt_noedges", "opt", dict(fin=8, fout=8, H=2, B=2, aggrs=["symnorm", "max", "std"], graph=dict(n=5, e=0))))
    out.append(("lay_noedges", "lay", dict(fin=8, fout=8, H=2, B=2, aggrs=["symadd", "max", "std", "mean"], graph=dict(n=5, e=0))))
    # sparse (adj_t) input path of EGConv / EfficientGraphConv
    out.append(("opt_sparse_symnorm_multi", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["symnorm", "max", "std"], graph=g_plain, sparse=True)))
    out.append(("opt_sparse_mean", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["mean"], graph=g_plain, sparse=True)))
    out.append(("lay_sparse_symadd_max", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["symadd", "max", "mean"], graph=g_plain, sparse=True)))
    # mid-size graph with heavy-tailed degrees for tolerance statistics
    out.append(("opt_mid", "opt", dict(fin=64, fout=64, H=8, B=4, aggrs=["sum", "mean", "max", "symnorm"], graph=dict(n=2000, e=24000, heavy=True))))
    out.append(("lay_mid", "lay", dict(fin=64, fout=64, H=4, B=4, aggrs=["symadd", "std", "max"], graph=dict(n=2000, e=24000, heavy=True))))
    return out


def make_graph(rng, g):
    g = dict(g)
    if g.pop("heavy", False):
        n, e = g["n"], g["e"]
        dst = np.floor(n * rng.random(e) ** 3).astype(np.int64)
        src = rng.integers(0, n, size=e)
        ei = np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])])
        ei = ei[:, ei[0] != ei[1]]
        ei = np.unique(ei, axis=1)
        return ei[:, rng.permutation(ei.shape[1])].astype(np.int64), n
    if g["e"] == 0:
        return np.zeros((2, 0), dtype=np.int64), g["n"]
    return rand_graph(rng, **g), g["n"]


def main():
    install_shims()
    lay = load_ref("layers")
    opt = load_ref("optimized_layers")
    manifest = {}
    for idx, (name, kind, cfg) in enumerate(cases()):
        rng = np.random.default_rng(1000 + idx)
        torch.manual_seed(1000 + idx)
        ei, n = make_graph(rng, cfg["graph"])
        fin, fout, H, B, aggrs = cfg["fin"], cfg["fout"], cfg["H"], cfg["B"], cfg["aggrs"]
        integer = cfg.get("integer", False)
        if integer:
            x = torch.from_numpy(rng.integers(-2, 3, size=(n, fin)).astype(np.float32))
        else:
            x = torch.from_numpy(rng.standard_normal((n, fin)).astype(np.float32))
        if kind == "lay":
            layer = lay.EfficientGraphConv(
                fin, fout, num_heads=H, num_bases=B,
                softmax_weights=cfg.get("softmax", False),
                add_self_loops=cfg.get("add_self_loops", True), bias=cfg.get("bias", True), aggrs=aggrs,
                sigmoid_weights=cfg.get("sigmoid", False), hardtanh_weights=cfg.get("hardtanh", False))
        else:
            layer = opt.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B,
                               add_self_loops=cfg.get("add_self_loops", True), bias=cfg.get("bias", True),
                               sigmoid=cfg.get("sigmoid", False))
        with torch.no_grad():
            for p in layer.parameters():
                if integer:
                    p.copy_(torch.from_numpy(rng.integers(-1, 2, size=tuple(p.shape)).astype(np.float32)))
            if getattr(layer, "bias", None) is not None and not integer:
                layer.bias.copy_(torch.from_numpy(rng.standard_normal(fout).astype(np.float32)))
            ei_t = torch.from_numpy(ei)
            if cfg.get("sparse", False):
                arg = SparseTensor(row=ei_t[1], col=ei_t[0], value=None, sparse_sizes=(n, n), is_sorted=False)
            else:
                arg = ei_t
            out = layer(x, arg) if kind == "opt" else layer(x=x, edge_index=arg)
        sd = {f"param:{k}": _np(v) for k, v in layer.state_dict().items()}
        meta = dict(kind=kind, fin=fin, fout=fout, H=H, B=B, aggrs=aggrs, n=n,
                    softmax=cfg.get("softmax", False), sigmoid=cfg.get("sigmoid", False),
                    hardtanh=cfg.get("hardtanh", False), add_self_loops=cfg.get("add_self_loops", True),
                    bias=cfg.get("bias", True), sparse=cfg.get("sparse", False), repr=repr(layer))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), x=_np(x), edge_index=ei, out=_np(out),
                            meta=json.dumps(meta), **sd)
        manifest[name] = dict(meta, n_edges=int(ei.shape[1]), out_abs_mean=float(out.abs().mean()))
        print(f"{name:36s} N={n:5d} E={ei.shape[1]:6d} out|mean|={out.abs().mean():.4f}")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
