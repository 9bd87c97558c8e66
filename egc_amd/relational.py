"""Relational EGC on the gfx950 kernels: ``REGConv`` (reference experiments/rmag/models.py:75-148).

One shared basis matrix for every node type; per node type a "root" combination of the node's own bases;
per relation (source type, name, target type) the ``mean`` and ``max`` of the source type's bases over the
relation's rectangular adjacency, combined with weightings computed from the TARGET node's features and
accumulated into the target type's output.  The sparse part of every term is the fused aggregate/combine
kernel of this repository (``egc_aggregate_combine_f32`` through the C ABI): the root term runs it on an
identity adjacency, a relation on its ``[N_target, N_source]`` CSR (``egc_graph.n_src_rows``).  The dense
Linears of one node type are ONE GEMM on this repository's matrix-core kernels, in inference and in training
(forward and the gradient GEMMs: ``egc_dense_transform_apply``).

Interface = the reference's: ``REGConv(in_channels, out_channels, num_heads, num_bases)``,
``forward(x_dict, adj_t_dict)`` with ``adj_t_dict[(src, rel, dst)]`` an ``adj_t`` whose rows are targets
(here ``egc_amd.SparseTensor`` / ``CSRGraph`` / ``torch.sparse_csr``); parameter names ``bases_weight``,
``rel_combs.<src>_<rel>_<dst>.{weight,bias}``, ``root_combs.<type>.{weight,bias}`` (state dicts interchange).
The node / edge type lists are the reference's module constants (rmag/models.py:10-26) and may be overridden.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _C
from .functional import (PostOp, egc_aggregate_combine, egc_aggregate_combine_apply, egc_basis_transform,
                         egc_dense_transform_apply, gemm_exact, make_spec, pack_weights)
from .graph import CSRGraph, SparseTensor
from .layers import glorot_

NODE_TYPES = ["author", "field_of_study", "institution", "paper"]          # rmag/models.py:17
EDGE_TYPES = [                                                             # rmag/models.py:18-26
    ("author", "affiliated_with", "institution"),
    ("institution", "to", "author"),
    ("author", "writes", "paper"),
    ("paper", "to", "author"),
    ("paper", "cites", "paper"),
    ("paper", "has_topic", "field_of_study"),
    ("field_of_study", "to", "paper"),
]


def _as_graph(adj_t, n_dst: int, n_src: int) -> CSRGraph:
    if isinstance(adj_t, CSRGraph):
        g = adj_t
    elif isinstance(adj_t, SparseTensor):
        g = adj_t.graph
    elif isinstance(adj_t, torch.Tensor) and adj_t.layout == torch.sparse_csr:
        g = CSRGraph.from_csr(adj_t.crow_indices(), adj_t.col_indices(), n_dst, n_src)
    else:
        raise RuntimeError(f"egc_amd.REGConv: unsupported adjacency type {type(adj_t)}")
    if g.n_nodes != n_dst or g.n_src_rows != n_src:
        raise RuntimeError(f"egc_amd.REGConv: adjacency is [{g.n_nodes}, {g.n_src_rows}], features say [{n_dst}, {n_src}]")
    return g


class REGConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_heads: int, num_bases: int,
                 node_types=None, edge_types=None):
        super().__init__()
        if out_channels % num_heads != 0:
            raise ValueError("out_channels must be divisible by num_heads")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_heads, self.num_bases = num_heads, num_bases
        self.node_types = list(NODE_TYPES if node_types is None else node_types)
        self.edge_types = [tuple(k) for k in (EDGE_TYPES if edge_types is None else edge_types)]
        self.bases_weight = nn.Parameter(torch.empty(in_channels, (out_channels // num_heads) * num_bases))
        self.rel_combs = nn.ModuleDict({f"{k[0]}_{k[1]}_{k[2]}": nn.Linear(in_channels, 2 * num_heads * num_bases)
                                        for k in self.edge_types})
        self.root_combs = nn.ModuleDict({k: nn.Linear(in_channels, num_heads * num_bases) for k in self.node_types})
        H, B = num_heads, num_bases
        # the reference's relation weightings are [h][a][b] (stack of mean, max viewed as 2B rows,
        # rmag/models.py:135-143); the kernels read [h][b][a]: permute the Linear's output rows
        perm = torch.arange(H * 2 * B).view(H, 2, B).transpose(1, 2).reshape(-1)
        self.register_buffer("_hba_rows", perm, persistent=False)
        self._spec_root = make_spec(in_channels, out_channels, H, B, [_C.AGGR_SUM], _C.SET_RAW, _C.SET_RAW,
                                    True, _C.LAYOUT_HBA, _C.ACT_NONE)
        self._spec_rel = make_spec(in_channels, out_channels, H, B, [_C.AGGR_MEAN, _C.AGGR_MAX],
                                   _C.SET_RAW, _C.SET_RAW, True, _C.LAYOUT_HBA, _C.ACT_NONE)
        self._identity = {}
        self._type_cache = {}
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.bases_weight)
        for lin in self.rel_combs.values():
            lin.reset_parameters()
        for lin in self.root_combs.values():
            lin.reset_parameters()
        self._type_cache = {}

    def _identity_graph(self, n: int, device) -> CSRGraph:
        key = (n, device)
        g = self._identity.get(key)
        if g is None:
            g = CSRGraph.from_csr(torch.arange(n + 1, dtype=torch.int32, device=device),
                                  torch.arange(n, dtype=torch.int32, device=device), n)
            self._identity = {key: g} if len(self._identity) > 8 else {**self._identity, key: g}
        return g

    def _type_weights(self, ntype: str):
        """Inference: everything computed from one node type's features -- its bases, its root weightings and the
        weightings of every relation that targets the type -- as ONE GEMM: ([bases_weight | root^T | rel^T ...],
        biases, split-precision planes), cached until a parameter changes in place."""
        rels = [k for k in self.edge_types if k[2] == ntype]
        lins = [self.root_combs[ntype]] + [self.rel_combs[f"{k[0]}_{k[1]}_{k[2]}"] for k in rels]
        params = [self.bases_weight] + [p for lin in lins for p in (lin.weight, lin.bias)]
        key = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._type_cache.get(ntype)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                perm = self._hba_rows
                cols = [self.bases_weight, lins[0].weight.t()] + [lin.weight[perm].t() for lin in lins[1:]]
                wcat = torch.cat(cols, dim=1).contiguous()
                bcat = torch.cat([lins[0].bias] + [lin.bias[perm] for lin in lins[1:]]).contiguous()
                gspec = SimpleNamespace(f_in=self.in_channels, f_g=self._spec_root.f_g,
                                        w_cols=wcat.size(1) - self._spec_root.f_g, ldb=self._spec_root.ldb)
                planes = None if gemm_exact() else pack_weights(gspec, wcat)
            hit = (key, wcat, bcat, planes, gspec, rels)
            self._type_cache[ntype] = hit
        return hit[1:]

    def _forward_inference(self, x_dict, adj_t_dict):
        HB = self.num_heads * self.num_bases
        bases, wt, out, offs = {}, {}, {}, {}
        for key, x in x_dict.items():
            wcat, bcat, planes, gspec, rels = self._type_weights(key)
            ident = self._identity_graph(x.size(0), x.device)
            bases[key], wt[key] = egc_basis_transform(ident, gspec, x, wcat, bcat, planes)   # rmag/models.py:113-143
            offs[key] = {k: HB + 2 * HB * i for i, k in enumerate(rels)}
            out[key] = egc_aggregate_combine(ident, self._spec_root, bases[key], wt[key][:, :HB], None)
        for key, adj_t in adj_t_dict.items():
            src, _, dst = key
            g = _as_graph(adj_t, x_dict[dst].size(0), x_dict[src].size(0))
            o = offs[dst][tuple(key)]
            # the store adds the terms accumulated so far (egc_post.residual, in place)
            egc_aggregate_combine(g, self._spec_rel, bases[src], wt[dst][:, o:o + 2 * HB], None,
                                  post=PostOp(residual=out[dst]), out=out[dst])
        return out

    # -- vertex-partitioned form (SURVEY.md 8e / BASELINE config 5: the ~21 M-edge typed ogbn-mag graph over 2/4/8 GPUs) --
    def partition_graphs(self, part, device=None):
        """The per-relation CSRs of one rank's TypedPartition (egc_amd.partition): rows = the rank's nodes of the target
        type, columns = rows of the rank's basis table [owned rows of all types | halo rows].  Built once per graph."""
        graphs = {}
        for key, e in part.rel_edges.items():
            e = e if device is None else e.to(device)
            graphs[tuple(key)] = CSRGraph.from_edge_index(e.contiguous(), part.n_owned(key[2]), max(part.n_table, 1)).trim_launches()
        return graphs

    def forward_partitioned(self, x_dict, part, graphs, table=None, exchange=None):
        """Inference forward of this rank's share: ``x_dict[type]`` = the rank's OWNED rows of each type (in the order of
        its range), ``graphs`` = partition_graphs(part).  Returns {type: out rows of the owned nodes}.

        One GEMM per node type writes the type's basis rows into its block of the rank's table; ONE all-to-all-v
        (part.plan, all types at once -- the basis matrix is shared, every row has the same width) brings the halo
        rows; the root terms (identity adjacency: no halo row) run while the rows travel; the relation terms follow.
        ``exchange``: replaces part.plan.exchange_start/finish (tests: a simulated exchange); called with the table."""
        HB = self.num_heads * self.num_bases
        ldb = self._spec_root.ldb
        dev = next(iter(x_dict.values())).device
        if table is None:
            table = torch.empty((max(part.n_table, 1), ldb), dtype=torch.float32, device=dev)
        wt, out, offs = {}, {}, {}
        for t in self.node_types:
            x = x_dict[t]
            lo, hi = part.table_rows(t)
            if x.size(0) != hi - lo:
                raise RuntimeError(f"egc_amd.REGConv: {t}: {x.size(0)} rows given, the rank owns {hi - lo}")
            wcat, bcat, planes, gspec, rels = self._type_weights(t)
            offs[t] = {k: HB + 2 * HB * i for i, k in enumerate(rels)}
            if hi > lo:
                ident = self._identity_graph(hi - lo, dev)
                _, wt[t] = egc_basis_transform(ident, gspec, x, wcat, bcat, planes, bases_out=table[lo:hi])
            else:
                wt[t] = torch.empty((0, wcat.size(1) - gspec.f_g), dtype=torch.float32, device=dev)
        handle = None
        if exchange is None:
            if part.plan.live:          # (a plan of build_typed_local_simulation has no process group behind it)
                handle = part.plan.exchange_start(table)
        for t in self.node_types:                      # root terms: the rows' own bases, nothing remote
            lo, hi = part.table_rows(t)
            if hi > lo:
                ident = self._identity_graph(hi - lo, dev)
                out[t] = egc_aggregate_combine(ident, self._spec_root, table[lo:hi], wt[t][:, :HB], None)
            else:
                out[t] = torch.empty((0, self.out_channels), dtype=torch.float32, device=dev)
        if exchange is not None:
            exchange(table)
        elif handle is not None:
            part.plan.exchange_finish(handle)
        for key, g in graphs.items():
            src, _, dst = key
            if g.n_nodes == 0:
                continue
            o = offs[dst][tuple(key)]
            egc_aggregate_combine(g, self._spec_rel, table, wt[dst][:, o:o + 2 * HB], None,
                                  post=PostOp(residual=out[dst]), out=out[dst])
        return out

    def forward(self, x_dict, adj_t_dict):
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                                  any(x.requires_grad for x in x_dict.values()))
        if not needs_grad and (self.num_heads * self.num_bases) % 4 == 0 and \
                all(tuple(k) in self.edge_types for k in adj_t_dict):
            return self._forward_inference(x_dict, adj_t_dict)
        # training: the same ONE GEMM per node type as in inference ([bases | root weightings | weightings of every
        # relation into the type], rmag/models.py:113-143), built from the parameters with differentiable ops and
        # run -- forward and the three gradient GEMMs -- on this repository's kernels (egc_dense_transform_apply)
        HB = self.num_heads * self.num_bases
        f_g, ldb = self._spec_root.f_g, self._spec_root.ldb
        perm = self._hba_rows
        bases, wt, out, offs = {}, {}, {}, {}
        for key, x in x_dict.items():
            rels = [k for k in self.edge_types if k[2] == key and tuple(k) in {tuple(a) for a in adj_t_dict}]
            lins = [self.root_combs[key]] + [self.rel_combs[f"{k[0]}_{k[1]}_{k[2]}"] for k in rels]
            wcat = torch.cat([self.bases_weight, lins[0].weight.t()] + [lin.weight[perm].t() for lin in lins[1:]], dim=1)
            bcat = torch.cat([lins[0].bias] + [lin.bias[perm] for lin in lins[1:]])
            gspec = SimpleNamespace(f_in=self.in_channels, f_g=f_g, w_cols=wcat.size(1) - f_g, ldb=ldb)
            ident = self._identity_graph(x.size(0), x.device)
            bases[key], wt[key] = egc_dense_transform_apply(ident, gspec, x, wcat, bcat)
            offs[key] = {tuple(k): HB + 2 * HB * i for i, k in enumerate(rels)}
            out[key] = egc_aggregate_combine_apply(ident, self._spec_root, bases[key], wt[key][:, :HB].contiguous())  # :117-129
        for key, adj_t in adj_t_dict.items():
            src, _, dst = key
            g = _as_graph(adj_t, x_dict[dst].size(0), x_dict[src].size(0))
            o = offs[dst][tuple(key)]
            out[dst] = out[dst] + egc_aggregate_combine_apply(g, self._spec_rel, bases[src],
                                                              wt[dst][:, o:o + 2 * HB].contiguous())  # :131-144
        return out

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, num_heads={self.num_heads}, "
                f"num_bases={self.num_bases})")
