"""The layer call as a ``torch.library`` custom operator ("registered through torch.library with a HIP
dispatch key"), so that ``torch.compile`` / export / autograd see ONE op instead of a Python function that reaches
native code through ctypes:

    torch.ops.egc_amd.layer_forward(x, wcat, bcat, bias, graph_handle, spec_handle) -> out
    torch.ops.egc_amd.layer_forward_train(...) -> (out, bases, weightings, stats, cnt, arg_max, arg_min)
    torch.ops.egc_amd.layer_backward(...)      -> (d_x, d_wcat, d_bcat, d_bias)

Replaces the same reference call sites as egc_layer_forward (layers.py:97-138, optimized_layers.py:177-210; in the
reference autograd derives the backward through those lines).  The graph and the layer description are Python
objects (a CSR with its plan, a C struct), which an operator schema cannot carry: they travel as integer handles into
a registry that keeps them alive as long as the caller does (``handle_of``).  The kernels are registered for the
``cuda`` device type ONLY -- PyTorch-ROCm's name for HIP devices; there is no CPU implementation, so a CPU tensor
raises NotImplementedError from the dispatcher (no fallback, as everywhere in this package).

The modules call the Python functions directly (a custom-op dispatch costs tens of microseconds, which per-batch
graphs of ~50 k nodes would feel); set EGC_USE_TORCH_OP=1, or run under torch.compile, to route them through the op.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import weakref

import torch

from . import _C
from . import functional as F

_REGISTRY: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_NEXT_HANDLE = itertools.count(1)     # handles are never reused: a freed object's handle stays stale for good


def handle_of(obj) -> int:
    """Integer handle of a CSRGraph / LayerSpec for the operator schema; valid while `obj` is alive.  One handle per
    object, drawn from a counter (``id(obj)`` can be handed to a NEW object once the old one is freed, and a stale
    handle would then resolve to it)."""
    h = getattr(obj, "_egc_handle", None)
    if h is None or _REGISTRY.get(h) is not obj:
        h = next(_NEXT_HANDLE)
        try:
            obj._egc_handle = h
        except AttributeError:     # objects without a __dict__: a fresh handle per call, still unique
            pass
        _REGISTRY[h] = obj
    return h


def _get(h: int):
    h = int(h)      # (under torch.compile a handle may arrive as a SymInt: specialise on it, it names one object)
    obj = _REGISTRY.get(h)
    if obj is None:
        raise RuntimeError("egc_amd: stale graph / layer handle (the object behind it has been freed)")
    return obj


@torch.library.custom_op("egc_amd::layer_forward", mutates_args=(), device_types="cuda")
def layer_forward(x: torch.Tensor, wcat: torch.Tensor, bcat: torch.Tensor | None, bias: torch.Tensor | None,
                  graph_handle: int, spec_handle: int) -> torch.Tensor:
    return F.egc_layer_forward(_get(graph_handle), _get(spec_handle), x, wcat, bcat, bias)


@layer_forward.register_fake
def _(x, wcat, bcat, bias, graph_handle, spec_handle):
    return x.new_empty((x.shape[0], _get(spec_handle).f_out))


@torch.library.custom_op("egc_amd::layer_forward_train", mutates_args=(), device_types="cuda")
def layer_forward_train(x: torch.Tensor, wcat: torch.Tensor, bcat: torch.Tensor | None, bias: torch.Tensor | None,
                        graph_handle: int, spec_handle: int
                        ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    graph, spec = _get(graph_handle), _get(spec_handle)
    out, bases, weightings, (stats, cnt, arg_max, arg_min) = F.train_forward_core(graph, spec, x, wcat, bcat, bias)
    def none():     # (a fresh tensor per output: an operator's returns may not alias each other)
        return x.new_empty((0,), dtype=torch.int32)
    return out, bases, weightings, stats, cnt, arg_max if arg_max is not None else none(), arg_min if arg_min is not None else none()


@layer_forward_train.register_fake
def _(x, wcat, bcat, bias, graph_handle, spec_handle):
    # shapes from the same sources as the real operator (egc_aggregate_combine_train): the statistics width is the
    # library's (it includes the 8-bit arg tables), arg_max / arg_min exist only for layers with max / min
    graph, spec = _get(graph_handle), _get(spec_handle)
    n = x.shape[0]
    i32 = dict(dtype=torch.int32)
    codes = [spec.c.aggrs[t] for t in range(spec.c.num_aggrs)]
    stats_w = max(int(_C.load().egc_train_stats_floats(C.byref(spec.c))), 1)

    def arg(code):
        return x.new_empty((n, spec.ldb), **i32) if code in codes else x.new_empty((0,), **i32)
    return (x.new_empty((n, spec.f_out)), x.new_empty((graph.n_src_rows, spec.ldb)), x.new_empty((n, spec.w_cols)),
            x.new_empty((n, stats_w)), x.new_empty((max(n, 1),), **i32), arg(_C.AGGR_MAX), arg(_C.AGGR_MIN))


@torch.library.custom_op("egc_amd::layer_backward", mutates_args=(), device_types="cuda")
def layer_backward(grad_out: torch.Tensor, x: torch.Tensor, wcat: torch.Tensor, bases: torch.Tensor, weightings: torch.Tensor,
                   stats: torch.Tensor, cnt: torch.Tensor, arg_max: torch.Tensor, arg_min: torch.Tensor,
                   graph_handle: int, spec_handle: int, need_bcat: bool, need_bias: bool
                   ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    graph, spec = _get(graph_handle), _get(spec_handle)
    saved = (stats, cnt, arg_max if arg_max.numel() else None, arg_min if arg_min.numel() else None)
    grad_out = grad_out.contiguous()
    d_bases, d_w, d_cat = F.egc_aggregate_combine_backward(graph, spec, bases, weightings, grad_out, saved,
                                                           joint=spec.ldb == spec.f_g)
    halo = graph.halo
    if halo is not None and graph.n_src_rows > graph.n_nodes:
        back = halo.exchange_reverse(d_bases)
        d_bases = d_bases[:graph.n_nodes].index_add(0, halo.send_idx, back)
    if d_cat is None:
        d_cat = torch.cat([d_bases[:, :spec.f_g], d_w], dim=1)
    empty = x.new_empty((0,))
    if need_bcat and need_bias:   # both bias gradients ride along with the weight gradient (egc_weight_grad_ex_f32)
        dwcat, sums, dbias = F._weight_grads(x, d_cat, col_sums=True, extra=grad_out)
    else:
        dwcat, sums = F._weight_grads(x, d_cat, col_sums=need_bcat)
        dbias = F._column_sums(grad_out) if need_bias else empty
    # (outputs own their storage: an operator must not hand out a view into a larger internal array)
    return (F._dx_matmul(d_cat, wcat), dwcat, sums[d_cat.size(1) - spec.w_cols:].clone() if need_bcat else empty, dbias)


@layer_backward.register_fake
def _(grad_out, x, wcat, bases, weightings, stats, cnt, arg_max, arg_min, graph_handle, spec_handle, need_bcat, need_bias):
    spec = _get(spec_handle)
    return (torch.empty_like(x), torch.empty_like(wcat), x.new_empty((spec.w_cols if need_bcat else 0,)),
            x.new_empty((spec.f_out if need_bias else 0,)))


def _setup_context(ctx, inputs, output):
    x, wcat, bcat, bias, graph_handle, spec_handle = inputs
    out, bases, weightings, stats, cnt, arg_max, arg_min = output
    ctx.save_for_backward(x, wcat, bases, weightings, stats, cnt, arg_max, arg_min)
    ctx.handles = (graph_handle, spec_handle)
    ctx.need = (bcat is not None, bias is not None)


def _backward(ctx, g_out, *unused):
    x, wcat, bases, weightings, stats, cnt, arg_max, arg_min = ctx.saved_tensors
    dx, dwcat, dbcat, dbias = layer_backward(g_out, x, wcat, bases, weightings, stats, cnt, arg_max, arg_min,
                                             ctx.handles[0], ctx.handles[1], ctx.need[0], ctx.need[1])
    return dx, dwcat, dbcat if ctx.need[0] else None, dbias if ctx.need[1] else None, None, None


layer_forward_train.register_autograd(_backward, setup_context=_setup_context)


def use_torch_op() -> bool:
    return _C.env_flag("EGC_USE_TORCH_OP") or torch.compiler.is_compiling()


def layer_apply_op(graph, spec, x, wcat, bcat, bias):
    """egc_layer_apply through the registered operators (training form when a gradient is needed)."""
    gh, sh = handle_of(graph), handle_of(spec)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, wcat, bcat, bias)):
        return layer_forward_train(x, wcat, bcat, bias, gh, sh)[0]
    return layer_forward(x, wcat, bcat, bias, gh, sh)
