"""Drop-in ``EfficientGraphConv`` backed by the fused gfx950 kernels.

Mirrors the constructor signature, attributes, parameter names/shapes (state-dict compatible with the
released checkpoints), ``extra_repr`` and error behaviour of the reference class
(experiments/layers.py:11-147) -- but ``forward`` does no tensor math in Python: it hands
(x, graph, parameters) to ``libegc_hip.so``.

Semantics carried over from the reference (SURVEY.md 8a notes):
  * weightings column order is h*B*A + b*A + a (layers.py:127-129);
  * only ``symadd`` sees self-loops (through gcn_norm, layers.py:172-178); add/mean/max/min/var/std
    reduce over the raw edge list even when ``add_self_loops=True``;
  * softmax is over the joint B*A axis per head (layers.py:112-117).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _C
from . import ops
from .functional import (egc_layer_apply, egc_layer_apply_params, gemm_exact, make_spec, pack_layer_weights, pack_weights,
                         pad_bases_columns, padded_basis_stride)
from .graph import CSRGraph, GraphBatch, SparseTensor, graph_from_input

_AGGR_CODE = {"add": _C.AGGR_SUM, "mean": _C.AGGR_MEAN, "max": _C.AGGR_MAX, "min": _C.AGGR_MIN,
              "symadd": _C.AGGR_SYMNORM, "var": _C.AGGR_VAR, "std": _C.AGGR_STD}


def _is_adj_t(edge_index) -> bool:
    """an adjacency object in the reference's sense (layers.py:221: `isinstance(edge_index, SparseTensor)`): this package's
    SparseTensor or a foreign one with torch_sparse's csr() -- NOT this package's own CSRGraph / GraphBatch, which also have a
    csr() and take every aggregator."""
    if isinstance(edge_index, SparseTensor):
        return True
    if isinstance(edge_index, (torch.Tensor, CSRGraph, GraphBatch)):
        return False
    return callable(getattr(edge_index, "csr", None))


def glorot_(t: torch.Tensor):
    """PyG ``inits.glorot``: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class _AggLayer(nn.Module):
    """Parameter-free marker kept so that ``repr(model)`` and the module tree match the reference
    (output/pretrained.txt:57-59).  The aggregation itself runs inside the fused kernel."""

    def __init__(self, aggr, add_self_loops, cache):
        super().__init__()
        assert aggr in _AGGR_CODE, f"unsupported aggregator {aggr!r}"  # PyG's MessagePassing asserts on aggr too
        self.aggr_fun = aggr
        self.add_self_loops = add_self_loops
        self.cache = cache

    def extra_repr(self):
        return self.aggr_fun


class EfficientGraphConv(nn.Module):
    """The EGC layer of the paper (reference: experiments/layers.py:11)."""

    def __init__(self, in_channels, out_channels, num_heads, num_bases, softmax_weights, add_self_loops=True,
                 bias=True, aggrs=None, cache=False, sigmoid_weights=False, hardtanh_weights=False, **kwargs):
        super().__init__()
        assert aggrs is not None
        nonlin = [bool(softmax_weights), bool(sigmoid_weights), bool(hardtanh_weights)]
        assert sum(nonlin) <= 1, "at most one of softmax/sigmoid/hardtanh weights"
        assert out_channels % num_heads == 0

        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_heads, self.num_bases = num_heads, num_bases
        self.softmax_weights, self.sigmoid_weights, self.hardtanh_weights = softmax_weights, sigmoid_weights, hardtanh_weights
        self.add_self_loops = add_self_loops
        self.cache = cache

        self.comb_weights = nn.Linear(in_channels, num_heads * num_bases * len(aggrs))
        self.bases_weight = nn.ParameterList(
            [nn.Parameter(torch.empty(in_channels, out_channels // num_heads)) for _ in range(num_bases)])
        self.aggs = nn.ModuleList([_AggLayer(a, add_self_loops=add_self_loops, cache=cache) for a in aggrs])
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

        act = _C.ACT_SOFTMAX if softmax_weights else _C.ACT_SIGMOID if sigmoid_weights else \
            _C.ACT_HARDTANH if hardtanh_weights else _C.ACT_NONE
        self._spec = make_spec(
            in_channels, out_channels, num_heads, num_bases, [_AGGR_CODE[a] for a in aggrs],
            agg_set=_C.SET_RAW, sym_set=_C.SET_LOOPED if add_self_loops else _C.SET_RAW, loops_all_nodes=True,
            weight_layout=_C.LAYOUT_HBA, weight_act=act,
            basis_stride=padded_basis_stride(out_channels, num_heads, num_bases))
        self._cached_graph = None
        self._wcat_key, self._wcat, self._planes = None, None, None
        self.reset_parameters()

    def reset_parameters(self):
        self.comb_weights.reset_parameters()
        for w in self.bases_weight:
            glorot_(w)
        if self.bias is not None:
            nn.init.zeros_(self.bias)
        self._cached_graph = None
        self._wcat_key, self._wcat, self._planes = None, None, None

    # [bases_weight.0 | ... | bases_weight.B-1 | comb_weights.weight^T], rebuilt when a parameter changes
    def _cat_weights(self):
        sp = self._spec
        w = self.comb_weights.weight
        if torch.is_grad_enabled() and w.is_cuda and w.dtype == torch.float32 and self.num_bases <= 32:
            # training: one launch each way instead of the differentiable cat / pad / transpose chain below
            A = w.size(0) // (self.num_heads * self.num_bases)
            return pack_layer_weights(list(self.bases_weight._parameters.values()), w, None, self.in_channels,
                                      self.num_heads, A, self.num_bases, sp.basis_len, sp.basis_stride, False)[0]
        bases = pad_bases_columns(torch.cat(list(self.bases_weight), dim=1), self.num_bases, sp.basis_len, sp.basis_stride)
        return torch.cat([bases, self.comb_weights.weight.t()], dim=1)

    def _packed_weights(self):
        params = list(self.bases_weight._parameters.values()) + [self.comb_weights.weight]  # (ParameterList.__getitem__ is slow)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return self._cat_weights()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._wcat_key:
            with torch.no_grad():
                self._wcat = self._cat_weights().contiguous()
            self._wcat_key = key
            self._planes = None
        return self._wcat

    def _weight_planes(self, wcat):
        """split-precision planes of wcat for the matrix-core GEMM, rebuilt with the packed weights."""
        if not wcat.is_cuda or gemm_exact() or wcat.requires_grad:
            return None
        if self._planes is None or self._planes.device != wcat.device:
            self._planes = pack_weights(self._spec, wcat)
        return self._planes

    def _train_call(self, x, edge_index):
        """The arguments of functional.egc_layer_apply_params for a training call on a GraphBatch, or None (see EGConv._train_call)."""
        w = self.comb_weights.weight
        if not (w.is_cuda and w.dtype == torch.float32 and self.num_bases <= 32 and x.is_cuda) or ops.use_torch_op():
            return None
        if _is_adj_t(edge_index) and any(a.aggr_fun in ("var", "std") for a in self.aggs):
            return None              # (forward() raises NotImplementedError for it, as layers.py:221-224 does)
        if self.cache and self._cached_graph is not None:
            graph = self._cached_graph
        elif isinstance(edge_index, GraphBatch):
            graph = edge_index
        else:
            if self.cache:
                return None          # (forward() builds and caches the graph first)
            graph = graph_from_input(edge_index, x.size(0))
        sp = self._spec
        A = w.size(0) // (self.num_heads * self.num_bases)
        return (graph, sp, x, self.bias, w, None, self.comb_weights.bias, list(self.bases_weight._parameters.values()),
                self.in_channels, self.num_heads, A, self.num_bases, sp.basis_len, sp.basis_stride, False)

    def forward(self, x, edge_index):
        if _is_adj_t(edge_index) and any(a.aggr_fun in ("var", "std") for a in self.aggs):      # (also a real torch_sparse.SparseTensor)
            raise NotImplementedError  # layers.py:221-224
        if self.cache and self._cached_graph is not None:
            graph = self._cached_graph
        else:
            graph = graph_from_input(edge_index, x.size(0), static=bool(self.cache))
            if self.cache:
                self._cached_graph = graph.trim_launches()
        w = self.comb_weights.weight
        params = list(self.bases_weight._parameters.values())
        if (torch.is_grad_enabled() and w.is_cuda and w.dtype == torch.float32 and self.num_bases <= 32
                and x.is_cuda and (w.requires_grad or any(p.requires_grad for p in params)) and not ops.use_torch_op()):
            # training: parameters in, parameter gradients out, one autograd node (pack + layer + unpack)
            sp = self._spec
            A = w.size(0) // (self.num_heads * self.num_bases)
            return egc_layer_apply_params(graph, sp, x, self.bias, w, None, self.comb_weights.bias, params,
                                          self.in_channels, self.num_heads, A, self.num_bases, sp.basis_len,
                                          sp.basis_stride, False)
        wcat = self._packed_weights()
        return egc_layer_apply(graph, self._spec, x, wcat, self.comb_weights.bias, self.bias,
                               packed=self._weight_planes(wcat))

    def extra_repr(self):
        return (f"(In={self.in_channels}, Out={self.out_channels}, H={self.num_heads}, B={self.num_bases}, "
                f"SL={self.add_self_loops}, SM={self.softmax_weights}, Bias={self.bias is not None})")
