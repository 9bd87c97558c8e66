"""Operator-level entry point: one EGC layer forward on the current HIP stream.

``egc_layer_forward`` is the Python face of ``egc_layer_forward_f32`` in include/egc_hip.h -- basis
GEMM, fused multi-aggregator reduction and combine -- and mirrors the tail of
``EfficientGraphConv.forward`` (experiments/layers.py:97-138) / ``EGConv.forward``
(optimized_layers.py:177-210).  Torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass

import torch

from . import _C
from .graph import CSRGraph, GraphBatch, _IndexFlag, _require_cuda, _stream_ptr, _device_guard


@dataclass
class LayerSpec:
    """Static shape/semantics of one layer, pre-packed for the C ABI."""
    c: _C.EgcLayer
    f_in: int
    f_out: int
    f_g: int      # bases columns of the GEMM: B * basis_stride (= B * F_out / H when the bases are contiguous)
    w_cols: int   # H * B * A
    basis_len: int = 0      # L = F_out / H
    basis_stride: int = 0   # floats between consecutive bases inside a row (>= L)

    @property
    def ldb(self) -> int:
        return (self.f_g + 3) & ~3

    @property
    def c_addr(self) -> int:
        """Address of the C struct (stable: the spec owns it)."""
        a = self.__dict__.get("_c_addr")
        if a is None:
            a = self.__dict__["_c_addr"] = C.addressof(self.c)
        return a

    @property
    def gemm_flags(self) -> int:
        """Operand precision of the layer's split GEMM (egc_layer_gemm_flags): 0 -- the fast fp16x2 form -- for every layer
        since the variance is accumulated about the row's first entry; EGC_GEMM_STDVAR_24BIT=1 gives std / var layers the
        24-bit-operand form of rounds 2-3 back."""
        return int(_C.load().egc_layer_gemm_flags(C.byref(self.c)))     # (decided in ONE place: the library)


def padded_basis_stride(out_channels: int, num_heads: int, num_bases: int) -> int:
    """Basis stride the layers use: L itself when it is a multiple of 4; otherwise L rounded up to 4 -- each
    basis then owns whole 16-byte slots and the register-resident kernels apply -- as long as the padded row
    still fits their 128 slots (64: one slot per lane; 65-128, the two ogbg-code nets: two slots per lane, round 3;
    beyond that the LDS-based kernels run either way and padding buys nothing)."""
    L = out_channels // num_heads
    Lp = (L + 3) & ~3
    return Lp if (Lp != L and num_bases * Lp <= 512) else L


def pad_bases_columns(w: torch.Tensor, num_bases: int, basis_len: int, basis_stride: int) -> torch.Tensor:
    """[F_in, B * L] -> [F_in, B * stride]: zero weight columns after every basis (differentiable), so the
    GEMM itself writes the padded `bases` layout."""
    if basis_stride == basis_len:
        return w
    f_in = w.size(0)
    return torch.nn.functional.pad(w.reshape(f_in, num_bases, basis_len), (0, basis_stride - basis_len)).reshape(f_in, -1)


def _pack_params(dims, permute, comb_w, comb_b, bases):
    """(wcat [f_in, B Ls + H B A], bcat [H B A] or None) from the parameters: one launch (egc_weights_pack_f32)."""
    lib = _C.load()
    f_in, H, A, B, L, Ls = dims
    dev = comb_w.device
    parts = [b.contiguous() for b in bases]
    cw = comb_w.contiguous()
    cb = comb_b.contiguous() if comb_b is not None else None
    ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    with _device_guard(dev):
        wcat = torch.empty((f_in, B * Ls + H * B * A), dtype=torch.float32, device=dev)
        bcat = torch.empty(H * B * A, dtype=torch.float32, device=dev) if cb is not None else None
        _C.check(lib.egc_weights_pack_f32(ptrs, len(parts), cw.data_ptr(), cb.data_ptr() if cb is not None else None,
                                          f_in, H, A, B, L, Ls, int(permute), wcat.data_ptr(),
                                          bcat.data_ptr() if bcat is not None else None, 0, _stream_ptr(dev)),
                 "egc_weights_pack_f32")
    return wcat, bcat


def _unpack_param_grads(dims, permute, shapes, has_b, dwcat, dbcat):
    """The parameters' gradients (d comb_w, d comb_b or None, [d basis matrices]) from (d wcat, d bcat): the same index
    map read the other way, one launch."""
    lib = _C.load()
    f_in, H, A, B, L, Ls = dims
    dev = dwcat.device if dwcat is not None else dbcat.device
    with _device_guard(dev):
        if dwcat is None:
            dwcat = torch.zeros((f_in, B * Ls + H * B * A), dtype=torch.float32, device=dev)
        dwcat = dwcat.contiguous()
        dcw = torch.empty(shapes[0], dtype=torch.float32, device=dev)
        dcb = dbc = None
        if has_b:
            dbc = (dbcat if dbcat is not None else torch.zeros(H * B * A, dtype=torch.float32, device=dev)).contiguous()
            dcb = torch.empty(shapes[1], dtype=torch.float32, device=dev)
        dparts = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes[2]]
        ptrs = (C.c_void_p * len(dparts))(*[p.data_ptr() for p in dparts])
        _C.check(lib.egc_weights_pack_f32(ptrs, len(dparts), dcw.data_ptr(), dcb.data_ptr() if dcb is not None else None,
                                          f_in, H, A, B, L, Ls, int(permute), dwcat.data_ptr(),
                                          dbc.data_ptr() if dbc is not None else None, 1, _stream_ptr(dev)),
                 "egc_weights_pack_f32")
    return dcw, dcb, dparts


class _PackWeightsFunction(torch.autograd.Function):
    """(wcat, bcat) = the GEMM operand of a layer from its parameters, and the parameters' gradients from (d wcat,
    d bcat): one launch each way (egc_weights_pack_f32) instead of the cat / pad / permute / transpose chain and its
    autograd mirror -- seven or more launches of 5 us per training step.  Inputs: dims, permute flag, comb weight,
    comb bias (or None), then the basis matrices (one [F_in, B L] or B of [F_in, L])."""

    @staticmethod
    def forward(ctx, dims, permute, comb_w, comb_b, *bases):
        wcat, bcat = _pack_params(dims, permute, comb_w, comb_b, bases)
        ctx.dims, ctx.permute, ctx.has_b = dims, permute, comb_b is not None
        ctx.shapes = (comb_w.shape, comb_b.shape if comb_b is not None else None, [b.shape for b in bases])
        if bcat is None:
            bcat = wcat.new_empty(0)
            ctx.mark_non_differentiable(bcat)
        return wcat, bcat

    @staticmethod
    def backward(ctx, dwcat, dbcat):
        dcw, dcb, dparts = _unpack_param_grads(ctx.dims, ctx.permute, ctx.shapes, ctx.has_b, dwcat, dbcat)
        return (None, None, dcw, dcb, *dparts)


def pack_layer_weights(bases, comb_w, comb_b, f_in, H, A, B, L, Ls, permute_hab: bool):
    """Differentiable (wcat [f_in, B Ls + H B A], bcat [H B A] or None) on the device kernel; ``bases`` is a list of one
    [f_in, B L] matrix or of B [f_in, L] matrices (float32 CUDA parameters)."""
    wcat, bcat = _PackWeightsFunction.apply((int(f_in), int(H), int(A), int(B), int(L), int(Ls)), bool(permute_hab),
                                            comb_w, comb_b, *bases)
    return wcat, (bcat if comb_b is not None else None)


def pack_egconv_weights(bases_weight, comb_w, comb_b, f_in, H, A, B, L, Ls):
    """EGConv: one basis matrix, comb rows [h][a][b] permuted to [h][b][a] (optimized_layers.py:195-202)."""
    return pack_layer_weights([bases_weight], comb_w, comb_b, f_in, H, A, B, L, Ls, True)


def make_spec(in_channels, out_channels, num_heads, num_bases, aggr_codes, agg_set, sym_set, loops_all_nodes,
              weight_layout, weight_act, basis_stride: int = 0) -> LayerSpec:
    L = out_channels // num_heads
    stride = basis_stride if basis_stride > L else L
    c = _C.make_layer(in_channels, out_channels, num_heads, num_bases, aggr_codes, agg_set, sym_set,
                      loops_all_nodes, weight_layout, weight_act, stride if stride != L else 0)
    return LayerSpec(c, in_channels, out_channels, num_bases * stride, num_heads * num_bases * len(aggr_codes), L, stride)


def _check_f32(t, name, shape=None):
    _require_cuda(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"egc_amd: {name} must be float32 (got {t.dtype})")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"egc_amd: {name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


def pack_weights(spec: LayerSpec, wcat: torch.Tensor) -> torch.Tensor:
    """Split wcat ([F_in, F_g + W], fp32) into the three bf16 planes the matrix-core GEMM stages
    (egc_basis_pack).  Done once per parameter update; the result is an opaque byte buffer."""
    lib = _C.load()
    _check_f32(wcat, "wcat", (spec.f_in, spec.f_g + spec.w_cols))
    dev = wcat.device
    wcat = wcat.contiguous()
    with _device_guard(dev):
        nbytes = lib.egc_basis_pack_bytes(spec.f_in, spec.f_g, spec.w_cols)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _C.check(lib.egc_basis_pack_ex(wcat.data_ptr(), spec.f_in, spec.f_g, spec.w_cols, getattr(spec, "gemm_flags", 0),
                                       packed.data_ptr(), nbytes, _stream_ptr(dev)), "egc_basis_pack_ex")
    return packed


def gemm_exact() -> bool:
    """EGC_GEMM_EXACT=1 selects the plain fp32-MFMA GEMM instead of the split-precision matrix-core form."""
    return _C.env_flag("EGC_GEMM_EXACT")


def egc_basis_transform(graph: CSRGraph, spec: LayerSpec, x: torch.Tensor, wcat: torch.Tensor,
                        bcat: torch.Tensor | None, packed: torch.Tensor | None = None,
                        bases_out: torch.Tensor | None = None):
    """Step 1 (egc_basis_transform_packed / _f32): returns (bases [n_src_rows, ldb], weightings [N, W]).
    Only the first N rows of bases are written; halo rows are the caller's (partitioned runs).  ``bases_out``: a dense
    float32 [>= N, ldb] block to write the basis rows into instead of a fresh array (a row range of a table shared by
    several node types: relational EGC on a vertex partition); it is returned as ``bases``."""
    lib = _C.load()
    n = graph.n_nodes
    _check_f32(x, "x", (n, spec.f_in))
    _check_f32(wcat, "wcat", (spec.f_in, spec.f_g + spec.w_cols))
    if bcat is not None:
        _check_f32(bcat, "bcat", (spec.w_cols,))
    if x.device != graph.device:
        raise RuntimeError(f"egc_amd: x is on {x.device} but the graph is on {graph.device}")
    dev = x.device
    x = x.contiguous()
    wcat = wcat.contiguous()
    with _device_guard(dev):
        if bases_out is not None:
            if (bases_out.dtype != torch.float32 or bases_out.dim() != 2 or bases_out.size(0) < n or bases_out.size(1) != spec.ldb
                    or not bases_out.is_contiguous() or bases_out.device != dev or bases_out.data_ptr() % 16):
                raise RuntimeError("egc_amd: bases_out must be a dense, 16-byte aligned float32 [>= N, ldb] block on x's device")
            bases = bases_out
        else:
            bases = torch.empty((graph.n_src_rows, spec.ldb), dtype=torch.float32, device=dev)  # owned | halo rows
        weightings = torch.empty((n, spec.w_cols), dtype=torch.float32, device=dev)
        bcat_p = bcat.contiguous().data_ptr() if bcat is not None else None
        if gemm_exact():
            _C.check(lib.egc_basis_transform_f32(x.data_ptr(), wcat.data_ptr(), bcat_p, n, spec.f_in, spec.f_g,
                                                 spec.w_cols, bases.data_ptr(), spec.ldb, weightings.data_ptr(),
                                                 _stream_ptr(dev)), "egc_basis_transform_f32")
        else:
            if packed is None:
                packed = pack_weights(spec, wcat)
            _C.check(lib.egc_basis_transform_packed_ex(x.data_ptr(), packed.data_ptr(), bcat_p, n, spec.f_in, spec.f_g,
                                                       spec.w_cols, getattr(spec, "gemm_flags", 0), bases.data_ptr(),
                                                       spec.ldb, weightings.data_ptr(), _stream_ptr(dev)),
                     "egc_basis_transform_packed_ex")
    return bases, weightings


@dataclass
class PostOp:
    """Caller-side elementwise tail fused into the layer's store (egc_post in include/egc_hip.h):
    out = act((z + bias) * scale + shift) + residual."""
    scale: torch.Tensor | None = None      # [F_out]
    shift: torch.Tensor | None = None      # [F_out]
    residual: torch.Tensor | None = None   # [N, F_out]
    relu: bool = False


def egc_aggregate_combine(graph: CSRGraph, spec: LayerSpec, bases: torch.Tensor, weightings: torch.Tensor,
                          bias: torch.Tensor | None, post: PostOp | None = None, rows: tuple | None = None,
                          out: torch.Tensor | None = None):
    """Steps 2+3 (egc_aggregate_combine_f32 / _post_f32 / _rows_f32): fused multi-aggregator reduction + combine
    (+ fused caller epilogue) -> out [N, F_out].  ``rows = (begin, end)`` finishes only those rows of ``out``."""
    lib = _C.load()
    n = graph.n_nodes
    _check_f32(bases, "bases", (graph.n_src_rows, spec.ldb))
    _check_f32(weightings, "weightings", (n, spec.w_cols))
    if bias is not None:
        _check_f32(bias, "bias", (spec.f_out,))
    dev = bases.device
    # a column block of a wider array (several terms' weightings from one GEMM) is read in place
    ldw = spec.w_cols
    if n > 1 and weightings.stride(1) == 1 and weightings.stride(0) != spec.w_cols and rows is None:
        ldw = int(weightings.stride(0))
    elif not weightings.is_contiguous():
        weightings = weightings.contiguous()
    with _device_guard(dev):
        if out is None:
            out = torch.empty((n, spec.f_out), dtype=torch.float32, device=dev)
        g = graph.c_struct()
        ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)),
                             lib.egc_aggregate_workspace_zero_bytes(C.byref(spec.c), graph.n_nodes, graph.n_edges))
        bias_p = bias.contiguous().data_ptr() if bias is not None else None
        if ldw != spec.w_cols:
            keep = []
            p = None
            if post is not None:
                def sptr(t, shape, name):
                    if t is None:
                        return None
                    _check_f32(t, name, shape)
                    keep.append(t.contiguous())
                    return keep[-1].data_ptr()
                p = _C.EgcPost(sptr(post.scale, (spec.f_out,), "post.scale"), sptr(post.shift, (spec.f_out,), "post.shift"),
                               sptr(post.residual, (n, spec.f_out), "post.residual"), int(bool(post.relu)))
            _C.check(lib.egc_aggregate_combine_strided_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb,
                                                           weightings.data_ptr(), ldw, bias_p,
                                                           C.byref(p) if p is not None else None, out.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                     "egc_aggregate_combine_strided_f32")
        elif rows is not None:
            if post is not None:
                raise RuntimeError("egc_amd: a row range and a fused post-op cannot be combined")
            _C.check(lib.egc_aggregate_combine_rows_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb,
                                                        weightings.data_ptr(), bias_p, out.data_ptr(), int(rows[0]),
                                                        int(rows[1]), ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                     "egc_aggregate_combine_rows_f32")
        elif post is None:
            _C.check(lib.egc_aggregate_combine_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb,
                                                   weightings.data_ptr(), bias_p, out.data_ptr(), None, None,
                                                   ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                     "egc_aggregate_combine_f32")
        else:
            keep = []  # contiguous copies must outlive the launch

            def ptr(t, shape, name):
                if t is None:
                    return None
                _check_f32(t, name, shape)
                keep.append(t.contiguous())
                return keep[-1].data_ptr()
            p = _C.EgcPost(ptr(post.scale, (spec.f_out,), "post.scale"), ptr(post.shift, (spec.f_out,), "post.shift"),
                           ptr(post.residual, (n, spec.f_out), "post.residual"), int(bool(post.relu)))
            _C.check(lib.egc_aggregate_combine_post_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb,
                                                        weightings.data_ptr(), bias_p, C.byref(p), out.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                     "egc_aggregate_combine_post_f32")
    return out


def stdvar_reference(spec: LayerSpec) -> bool:
    """EGC_STDVAR_REFERENCE=1 and a var / std layer: the variance by the reference's own float32 formula, mean(x^2) - mean(x)^2
    (layers.py:203-214; SURVEY.md 8a note 5), on the library's general kernels -- no tile / one-launch path (their kernels
    accumulate the squares about the row's first entry), 24-bit GEMM operands (egc_layer_gemm_flags)."""
    if not _C.env_flag("EGC_STDVAR_REFERENCE"):
        return False
    return any(int(spec.c.aggrs[t]) in (_C.AGGR_VAR, _C.AGGR_STD) for t in range(int(spec.c.num_aggrs)))


def _batch_tile_setup(gb: GraphBatch, spec: LayerSpec, post):
    """What the tile kernels need for this layer on this batch (GraphBatch.tile_setup), or None."""
    if _C.env_flag("EGC_NO_TILE") or stdvar_reference(spec):
        return None
    return gb.tile_setup(spec.c, post is not None and post.scale is not None)


def egc_aggregate_combine_batch(gb: GraphBatch, spec: LayerSpec, bases, weightings, bias, post, tiled):
    """Steps 2 + 3 on tiles of whole graphs (egc_aggregate_combine_batch_f32): the CSR of each tile is built in LDS by the
    workgroup that aggregates it."""
    lib = _C.load()
    _IndexFlag.poll()
    n = gb.n_nodes
    tiles, n_tiles_dev, n_slots, lds_nodes, tmax, emax = tiled
    _check_f32(bases, "bases", (n, spec.ldb))
    _check_f32(weightings, "weightings", (n, spec.w_cols))
    dev = bases.device
    ldw = 0
    if n > 1 and weightings.stride(1) == 1 and weightings.stride(0) != spec.w_cols:
        ldw = int(weightings.stride(0))
    elif not weightings.is_contiguous():
        weightings = weightings.contiguous()
    with _device_guard(dev):
        out = torch.empty((n, spec.f_out), dtype=torch.float32, device=dev)
        keep = []
        p = None
        if post is not None:
            def sptr(t, shape, name):
                if t is None:
                    return None
                _check_f32(t, name, shape)
                keep.append(t.contiguous())
                return keep[-1].data_ptr()
            p = _C.EgcPost(sptr(post.scale, (spec.f_out,), "post.scale"), sptr(post.shift, (spec.f_out,), "post.shift"),
                           sptr(post.residual, (n, spec.f_out), "post.residual"), int(bool(post.relu)))
        ei = gb.edge_index
        needs_max = not bool(spec.c.loops_all_nodes)
        _C.check(lib.egc_aggregate_combine_batch_f32(
            tiles.data_ptr(), n_tiles_dev.data_ptr(), n_slots, lds_nodes, tmax, emax, ei[0].data_ptr(), ei[1].data_ptr(), n,
            gb.max_index().data_ptr() if needs_max else None, C.byref(spec.c), bases.data_ptr(), spec.ldb,
            weightings.data_ptr(), ldw, bias.contiguous().data_ptr() if bias is not None else None,
            C.byref(p) if p is not None else None, out.data_ptr(), gb.status().data_ptr(), _IndexFlag.ptr(),
            _stream_ptr(dev)), "egc_aggregate_combine_batch_f32")
    return out


_BATCH_FUSED_PACKS: "dict[tuple, tuple]" = {}


def _batch_fused_pack(spec: LayerSpec, wcat: torch.Tensor, bcat):
    """The weight planes of egc_layer_forward_batch_fused_f32 (egc_batch_fused_pack), cached on the identity / version of
    the concatenated weights (rebuilt only when a parameter changes)."""
    key = (wcat.data_ptr(), wcat._version, bcat.data_ptr() if bcat is not None else 0,
           bcat._version if bcat is not None else 0, spec.f_in, spec.f_out, spec.w_cols, spec.ldb,
           C.string_at(C.addressof(spec.c), C.sizeof(spec.c)), str(wcat.device), int(_stream_ptr(wcat.device) or 0))
    hit = _BATCH_FUSED_PACKS.get(key)
    if hit is None:
        lib = _C.load()
        _check_f32(wcat, "wcat", (spec.f_in, spec.f_g + spec.w_cols))
        dev = wcat.device
        wc = wcat.contiguous()
        with _device_guard(dev):
            nbytes = int(lib.egc_batch_fused_pack_bytes(C.byref(spec.c)))
            if nbytes <= 0:
                raise RuntimeError("egc_amd: layer outside the envelope of the one-launch batch kernel")
            packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _C.check(lib.egc_batch_fused_pack(C.byref(spec.c), wc.data_ptr(), bcat.contiguous().data_ptr() if bcat is not None else None,
                                              packed.data_ptr(), nbytes, _stream_ptr(dev)), "egc_batch_fused_pack")
        if len(_BATCH_FUSED_PACKS) >= 32:
            _BATCH_FUSED_PACKS.clear()
        hit = (wcat, bcat, packed)       # keep the keyed tensors alive: their addresses are the key
        _BATCH_FUSED_PACKS[key] = hit
    return hit[2]


def _batch_fused_train_pack(spec: LayerSpec, wcat: torch.Tensor, bcat):
    """(packed, packed_t): the forward's and the backward's weight planes of one training step in ONE launch
    (egc_batch_fused_train_pack); not cached -- the parameters change every step."""
    lib = _C.load()
    _check_f32(wcat, "wcat", (spec.f_in, spec.f_g + spec.w_cols))
    dev = wcat.device
    wc = wcat.contiguous()
    with _device_guard(dev):
        nb, nbt = int(lib.egc_batch_fused_pack_bytes(C.byref(spec.c))), int(lib.egc_batch_fused_bwd_pack_bytes(C.byref(spec.c)))
        if nb <= 0 or nbt <= 0:
            raise RuntimeError("egc_amd: layer outside the envelope of the one-launch training path")
        packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        packed_t = torch.empty(nbt, dtype=torch.uint8, device=dev)
        _C.check(lib.egc_batch_fused_train_pack(C.byref(spec.c), wc.data_ptr(), bcat.contiguous().data_ptr() if bcat is not None else None,
                                                packed.data_ptr(), nb, packed_t.data_ptr(), nbt, _stream_ptr(dev)),
                 "egc_batch_fused_train_pack")
    return packed, packed_t


def _batch_fused_train_pack_params(spec: LayerSpec, dims, permute, comb_w, comb_b, bcat_direct, bases):
    """(packed, packed_t) straight from the module's parameters (egc_batch_fused_train_pack_params: the index map of
    egc_weights_pack_f32 inside the pack launch -- no wcat / bcat arrays, one launch instead of two), or None when a parameter is
    not a dense float32 tensor on the device."""
    f_in, H, A, B, L, Ls = dims
    ts = [comb_w, *bases] + [t for t in (comb_b, bcat_direct) if t is not None]
    dev = comb_w.device
    if not all(t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
        return None
    if comb_b is not None and bcat_direct is not None:
        return None
    lib = _C.load()
    with _device_guard(dev):
        nb, nbt = int(lib.egc_batch_fused_pack_bytes(C.byref(spec.c))), int(lib.egc_batch_fused_bwd_pack_bytes(C.byref(spec.c)))
        if nb <= 0 or nbt <= 0:
            return None
        packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        packed_t = torch.empty(nbt, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * len(bases))(*[b.data_ptr() for b in bases])
        st = lib.egc_batch_fused_train_pack_params(C.byref(spec.c), ptrs, len(bases), comb_w.data_ptr(),
                                                   comb_b.data_ptr() if comb_b is not None else None,
                                                   bcat_direct.data_ptr() if bcat_direct is not None else None, H, A, B, L, Ls,
                                                   int(permute), packed.data_ptr(), nb, packed_t.data_ptr(), nbt, _stream_ptr(dev))
        if st != 0:
            return None
    return packed, packed_t


def _batch_fused_setup(gb: GraphBatch, spec: LayerSpec, post, wcat, x=None):
    """(tile_nodes, max_tile_edges) when this layer call can run as ONE launch on the batch, else None.  A layer that asks for
    the 24-bit-operand GEMM (EGC_GEMM_STDVAR_24BIT=1 and std / var) keeps the two-launch path; EGC_NO_FUSED_TILE=1 switches
    the path off."""
    if _C.env_flag("EGC_NO_FUSED_TILE") or _C.env_flag("EGC_NO_TILE") or gemm_exact() or wcat.requires_grad:
        return None
    if spec.gemm_flags != 0:
        return None
    if x is not None and (not x.is_contiguous() or x.data_ptr() % 16 != 0):
        return None          # (a view at a 4-byte offset of a flat buffer: the CSR / tile paths serve it, this launch reads 16-byte pieces)
    return gb.fused_setup(spec.c, post is not None and post.scale is not None)


def egc_layer_forward_batch_fused(gb: GraphBatch, spec: LayerSpec, x, wcat, bcat, bias, post, setup, packed=None):
    """The whole layer on a batch of whole graphs in ONE launch (egc_layer_forward_batch_fused_f32): plan, basis transform +
    weightings on the matrix cores, the tiles' CSR, aggregation and combine -- x and the edge list in, out out."""
    lib = _C.load()
    _IndexFlag.poll()
    n = gb.n_nodes
    tile_nodes, emax = setup
    _check_f32(x, "x", (n, spec.f_in))
    dev = x.device
    if dev != gb.device:
        raise RuntimeError(f"egc_amd: x is on {dev} but the batch is on {gb.device}")
    x = x.contiguous()
    if packed is None:
        packed = _batch_fused_pack(spec, wcat, bcat)
    with _device_guard(dev):
        out = torch.empty((n, spec.f_out), dtype=torch.float32, device=dev)
        keep = []
        p = None
        if post is not None:
            def sptr(t, shape, name):
                if t is None:
                    return None
                _check_f32(t, name, shape)
                keep.append(t.contiguous())
                return keep[-1].data_ptr()
            p = _C.EgcPost(sptr(post.scale, (spec.f_out,), "post.scale"), sptr(post.shift, (spec.f_out,), "post.shift"),
                           sptr(post.residual, (n, spec.f_out), "post.residual"), int(bool(post.relu)))
        ei = gb.edge_index
        needs_max = not bool(spec.c.loops_all_nodes)
        _C.check(lib.egc_layer_forward_batch_fused_f32(
            gb.ptr.data_ptr(), gb.edge_ptr.data_ptr() if gb.edge_ptr is not None else None, gb.n_graphs, ei[0].data_ptr(),
            ei[1].data_ptr(), gb.n_edges, n, gb.max_index().data_ptr() if needs_max else None, C.byref(spec.c), x.data_ptr(),
            packed.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None, C.byref(p) if p is not None else None,
            out.data_ptr(), tile_nodes, emax, gb.status().data_ptr(), _IndexFlag.ptr(), _stream_ptr(dev)),
            "egc_layer_forward_batch_fused_f32")
    return out


def _batch_fused_train_setup(gb: GraphBatch, spec: LayerSpec, x):
    """((tile_nodes, emax) of the forward launch, (tile_nodes, emax) of the backward launch) when a TRAINING call of this layer on
    this batch can run as one launch each way (egc_layer_forward_batch_fused_f32 + egc_layer_backward_batch_fused_f32), else
    None: the CSR path (graph build + GEMM + aggregate + three backward kernels + the dense gradients) then.
    EGC_NO_FUSED_BWD=1 switches the path off.  (Graphs of at most 80 nodes at H = 8: the tile's LDS image also holds d bases
    as 64-bit fixed point, summed by integer LDS atomics.)"""
    if (_C.env_flag("EGC_NO_FUSED_BWD") or _C.env_flag("EGC_NO_FUSED_TILE") or _C.env_flag("EGC_NO_TILE") or gemm_exact()
            or spec.gemm_flags != 0):
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return None
    bs = gb.fused_bwd_setup(spec.c)
    if bs is None:
        return None
    fs = gb.fused_setup(spec.c, False)
    return (fs, bs) if fs is not None else None


class ResidualLink:
    """What lets the gradient of a block's residual branch (x = x + relu(bn(conv(x))), zinc/models.py:70-73) join d x inside the
    conv's backward launch instead of in a pass of autograd's own.  The block (egc_amd.FusedEGCBlock) offers one around its conv
    call (`offer`); the one-launch training path takes it (`taken`); the tail's backward -- which autograd runs first -- then
    leaves its incoming gradient here (`grad`) and reports none for the residual input, and the conv's backward hands it to
    egc_layer_backward_batch_fused_f32 as `d_x_add`.  Only ever used when the residual input IS the conv's input: the offer
    records that tensor, and a layer call whose x is another tensor (a wrapper module that casts / drops out / projects in front
    of the EGC layer) leaves the offer untaken -- autograd then adds the residual gradient itself."""
    __slots__ = ("taken", "grad", "x")
    _local = threading.local()        # the offer lives on the thread that runs the block's forward

    def __init__(self, x=None):
        self.taken, self.grad, self.x = False, None, x

    @classmethod
    def offer(cls, link):
        cls._local.offered = link

    @classmethod
    def take(cls, x=None):
        link = getattr(cls._local, "offered", None)
        cls._local.offered = None
        if link is not None and link.x is not None and link.x is not x:
            return None               # (this call's x is not the block's residual input)
        if link is not None:
            link.taken = True
        return link


def egc_layer_backward_batch_fused(gb: GraphBatch, spec: LayerSpec, x, wcat, packed, grad_out, setup, d_x_add=None, packed_t=None):
    """(d x [N, F_in], d_cat [N, ldb + W]) of one layer on a batch of whole graphs in ONE launch (egc_layer_backward_batch_fused_f32):
    the forward's intermediates are formed again in LDS, nothing was saved but x.  ``d_x_add`` [N, F_in]: added to d x in its store."""
    lib = _C.load()
    _IndexFlag.poll()
    n = gb.n_nodes
    tile_nodes, emax = setup
    dev = x.device
    grad_out = grad_out.contiguous()
    _check_f32(grad_out, "grad_out", (n, spec.f_out))
    if d_x_add is not None:
        d_x_add = d_x_add.contiguous()
        _check_f32(d_x_add, "d_x_add", (n, spec.f_in))
    with _device_guard(dev):
        nb = int(lib.egc_batch_fused_bwd_pack_bytes(C.byref(spec.c)))
        if nb <= 0:
            raise RuntimeError("egc_amd: layer outside the envelope of the one-launch backward")
        stream = _stream_ptr(dev)
        if packed_t is None:
            packed_t = torch.empty(nb, dtype=torch.uint8, device=dev)
            _C.check(lib.egc_batch_fused_bwd_pack(C.byref(spec.c), wcat.data_ptr(), packed_t.data_ptr(), nb, stream), "egc_batch_fused_bwd_pack")
        d_x = torch.empty((n, spec.f_in), dtype=torch.float32, device=dev)
        d_cat = torch.empty((n, spec.ldb + spec.w_cols), dtype=torch.float32, device=dev)
        ei = gb.edge_index
        needs_max = not bool(spec.c.loops_all_nodes)
        _C.check(lib.egc_layer_backward_batch_fused_f32(
            gb.ptr.data_ptr(), gb.edge_ptr.data_ptr() if gb.edge_ptr is not None else None, gb.n_graphs, ei[0].data_ptr(),
            ei[1].data_ptr(), gb.n_edges, n, gb.max_index().data_ptr() if needs_max else None, C.byref(spec.c), x.data_ptr(),
            packed.data_ptr(), packed_t.data_ptr(), grad_out.data_ptr(), d_x.data_ptr(),
            d_x_add.data_ptr() if d_x_add is not None else None, d_cat.data_ptr(), int(d_cat.stride(0)), tile_nodes, emax, gb.status().data_ptr(), _IndexFlag.ptr(), stream), "egc_layer_backward_batch_fused_f32")
    return d_x, d_cat


class _BatchFusedTrainFunction(torch.autograd.Function):
    """The layer on a batch of whole graphs under autograd, one launch each way (round 5): forward = the inference launch
    (egc_layer_forward_batch_fused_f32: nothing but x is kept), backward = egc_layer_backward_batch_fused_f32 (d x, d_cat) + the
    weight gradient x^T d_cat with the bias sums riding along.  Same arguments and gradients as _EGCLayerParamsFunction."""

    @staticmethod
    def forward(ctx, x, bias, comb_w, comb_b, bcat_direct, gb, spec, dims, permute, setups, link, *bases):
        ctx.dims, ctx.permute, ctx.packed_b, ctx.link = dims, permute, comb_b is not None, link
        ctx.shapes = (comb_w.shape, comb_b.shape if comb_b is not None else None, [b.shape for b in bases])
        both = _batch_fused_train_pack_params(spec, dims, permute, comb_w, comb_b, bcat_direct, bases)
        if both is not None:           # the planes of both launches straight from the parameters: no wcat / bcat arrays
            packed, packed_t = both
            wcat = bc = None
        else:
            wcat, bcat = _pack_params(dims, permute, comb_w, comb_b, bases)
            bc = bcat if comb_b is not None else bcat_direct
            packed, packed_t = _batch_fused_train_pack(spec, wcat, bc)
        out = egc_layer_forward_batch_fused(gb, spec, x, wcat, bc, bias, None, setups[0], packed=packed)
        ctx.save_for_backward(x, wcat, packed, packed_t)
        ctx.gb, ctx.spec, ctx.bsetup = gb, spec, setups[1]
        ctx.has_bcat, ctx.has_bias = (comb_b is not None or bcat_direct is not None), bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        need = ctx.needs_input_grad
        x, wcat, packed, packed_t = ctx.saved_tensors
        spec = ctx.spec
        grad_out = grad_out.contiguous()
        add = None
        if ctx.link is not None:                 # (the residual branch's gradient, left by the block's tail: see ResidualLink)
            add, ctx.link.grad = ctx.link.grad, None
            if add is not None and (add.shape != x.shape or add.dtype != torch.float32 or add.device != x.device):
                raise RuntimeError("egc_amd: the residual gradient handed to the conv's backward does not have the shape of x")
        dx, d_cat = egc_layer_backward_batch_fused(ctx.gb, spec, x, wcat, packed, grad_out, ctx.bsetup, add if need[0] else None, packed_t)
        need_w = need[2] or any(need[11:])
        need_b = ctx.has_bcat and (need[3] if ctx.packed_b else need[4])
        need_bias = ctx.has_bias and need[1]
        if need_w and need_b and need_bias and ctx.has_bcat:
            # the usual training call: every parameter takes a gradient -- x^T d_cat and both bias sums land in the parameters'
            # own layouts in the weight-gradient launch's reduction (no d wcat, no unpack launch)
            got = _weight_grads_into_params(x, d_cat, grad_out, ctx.dims, ctx.permute, ctx.shapes, ctx.packed_b)
            if got is not None:
                dcw, dcb, dbc, dparts, dbias = got
                return (dx if need[0] else None, dbias, dcw, dcb, None if ctx.packed_b else dbc, None, None, None, None, None, None,
                        *dparts)
        dwcat = dbcat = dbias = None
        if need_w:
            if need_bias and need_b:
                dwcat, sums, dbias = _weight_grads(x, d_cat, col_sums=True, extra=grad_out)
            else:
                dwcat, sums = _weight_grads(x, d_cat, col_sums=need_b)
            dbcat = sums[d_cat.size(1) - spec.w_cols:] if need_b else None
        elif need_b:
            dbcat = _column_sums(d_cat[:, spec.ldb:].contiguous())
        if need_bias and dbias is None:
            dbias = _column_sums(grad_out)
        dcw = dcb = None
        dparts = [None] * len(ctx.shapes[2])
        if need_w or (need_b and ctx.packed_b):
            dcw, dcb, dparts = _unpack_param_grads(ctx.dims, ctx.permute, ctx.shapes, ctx.packed_b, dwcat,
                                                   dbcat if ctx.packed_b else None)
        return (dx if need[0] else None, dbias, dcw, dcb, None if ctx.packed_b else dbcat, None, None, None, None, None, None, *dparts)


def segment_mean(x: torch.Tensor, seg_ptr: torch.Tensor) -> torch.Tensor:
    """Mean of consecutive row segments of x [N, C]: out[g] = mean(x[seg_ptr[g]:seg_ptr[g+1]]) (egc_segment_mean_f32)."""
    lib = _C.load()
    _check_f32(x, "x")
    x = x.contiguous()
    seg_ptr = seg_ptr.to(device=x.device, dtype=torch.int64).contiguous()
    n_seg = int(seg_ptr.numel()) - 1
    with _device_guard(x.device):
        out = torch.empty((n_seg, x.size(1)), dtype=torch.float32, device=x.device)
        _C.check(lib.egc_segment_mean_f32(x.data_ptr(), seg_ptr.data_ptr(), n_seg, x.size(1), out.data_ptr(),
                                          _stream_ptr(x.device)), "egc_segment_mean_f32")
    return out


def _native_ops(dev):
    """The compiled binding (egc_amd/_native.py) when it is built and `dev` is the current device."""
    from . import _native
    nat = _native.ops()
    if nat is None or dev.index != torch.cuda.current_device():
        return None
    return nat


def _native_train_ops(graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases):
    """The compiled binding for the training call (train_forward / train_backward of egc_amd/csrc_ext), or None: one GPU,
    a square graph without a halo, dense float32 operands, a bias and a combination bias, and the shape envelope in which
    the backward's dense gradients are one-pass kernels (egc_weight_grad_ex_f32 with both column-sum streams riding along,
    the packed d x GEMM, the joint [N, ldb + W] gradient array).  Everything else: the Python path below, same kernels."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.size(0) > 0):
        return None
    nat = _native_ops(x.device)
    if nat is None or not hasattr(nat, "train_forward") or _C.env_flag("EGC_NO_NATIVE_TRAIN"):
        return None
    if graph.halo is not None or graph.n_src_rows != graph.n_nodes or gemm_exact():
        return None
    cb = comb_b if comb_b is not None else bcat_direct
    if bias is None or cb is None or (comb_b is not None and bcat_direct is not None):
        return None
    k = spec.ldb + spec.w_cols
    if not (spec.ldb == spec.f_g and k % 4 == 0 and spec.f_in % 4 == 0 and spec.f_out % 4 == 0):
        return None
    if not (k <= 192 and spec.f_in <= 128 and spec.f_out <= 128):
        # outside the one-pass dense-gradient kernel: the general sequence (round 6: the reference's own 168 / 224 / 296-wide
        # batched nets), where the split GEMMs take the shape and x^T d goes through the exact-fp32 tile grid
        if (k > 384 or spec.f_in > 384 or spec.f_out > 1024 or x.data_ptr() % 16
                or int(_C.load().egc_basis_pack_bytes(k, spec.f_in, 0)) <= 0):
            return None
    for t in (bias, comb_w, cb, *bases):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return None
    if len({tuple(b.shape) for b in bases}) != 1:
        return None
    return nat


def native_block_train(call, bn=None, relu=True, residual=True, with_tail=True):
    """The training call of a layer on a GraphBatch -- with ``with_tail`` the whole block x -> x + relu(bn(conv(x))) of the
    reference's batched nets (zinc/models.py:66-73) -- as ONE autograd node of the compiled binding (egc_torch_ext.cpp:
    batch_block_train), or None when the call is outside its envelope (the Python Functions below then, same kernels).
    ``call``: the arguments of egc_layer_apply_params, as the layer modules' ``_train_call`` returns them."""
    graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases, f_in, H, A, B, L, Ls, permute = call
    if not isinstance(graph, GraphBatch) or _C.env_flag("EGC_NO_NATIVE_TRAIN") or not torch.is_grad_enabled():
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(0) > 1 and x.is_contiguous()):
        return None
    nat = _native_ops(x.device)
    if nat is None or not hasattr(nat, "batch_block_train"):
        return None
    cb = comb_b if comb_b is not None else bcat_direct
    params = [bias, comb_w, cb, *bases]
    if any(t is None or not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.requires_grad) for t in params):
        return None
    if comb_b is not None and bcat_direct is not None:
        return None
    k = spec.ldb + spec.w_cols
    if not (spec.ldb == spec.f_g and k % 4 == 0 and k <= 192 and spec.f_in % 4 == 0 and spec.f_in <= 128
            and spec.f_out % 4 == 0 and spec.f_out <= 128 and k == B * Ls + H * B * A):
        return None
    if graph.n_nodes is None:
        graph.n_nodes = int(x.size(0))
    if x.size(0) != graph.n_nodes or x.device != graph.device:
        return None
    gamma = beta = rm = rv = nt = None
    eps, momentum = 1e-5, 0.1
    if with_tail:
        if not (relu and bn.training and bn.affine and spec.f_out % 4 == 0):
            return None
        gamma, beta, eps = bn.weight, bn.bias, float(bn.eps)
        if any(not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in (gamma, beta)):
            return None
        if bn.track_running_stats:
            rm, rv, nt = bn.running_mean, bn.running_var, bn.num_batches_tracked
            if not (_f32_vec(rm, spec.f_out) and _f32_vec(rv, spec.f_out) and rm.is_cuda and nt is not None and nt.dtype == torch.int64
                    and nt.is_cuda):
                return None
        momentum = -1.0 if bn.momentum is None else float(bn.momentum)
    setups = _batch_fused_train_setup(graph, spec, x)
    if setups is None:
        return None
    _IndexFlag.poll()
    src, dst = graph.rows()
    needs_max = not bool(spec.c.loops_all_nodes)
    return nat.batch_block_train(x, bias, comb_w, comb_b, bcat_direct, gamma, beta, list(bases), rm, rv, nt, graph.ptr, graph.edge_ptr,
                                 src, dst, graph.max_index() if needs_max else None, graph.status(), _IndexFlag.ptr() or 0, spec.c_addr,
                                 _stream_ptr(x.device), (int(f_in), int(H), int(A), int(B), int(L), int(Ls)), bool(permute),
                                 (setups[0][0], setups[0][1], setups[1][0], setups[1][1]), eps, momentum, bool(relu), bool(residual),
                                 bool(with_tail))


def native_csr_block_train(call, bn, relu=True, residual=True):
    """The block x -> x + relu(bn(conv(x))) in training on the CSR path -- layers and batches outside the one-launch training
    envelope: the reference's own 168 - 304-wide batched nets, full graphs -- as ONE autograd node of the compiled binding
    (egc_torch_ext.cpp: csr_block_train = train_forward + the BatchNorm tail / its backward + train_backward + the residual
    gradient), or None outside its envelope (the Python Functions then, same kernels).  ``call``: as for native_block_train; a
    GraphBatch is taken through its CSR."""
    graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases, f_in, H, A, B, L, Ls, permute = call
    if _C.env_flag("EGC_NO_NATIVE_TRAIN") or not torch.is_grad_enabled():
        return None
    if isinstance(graph, GraphBatch):
        if graph.n_nodes is None:
            graph.n_nodes = int(x.size(0))
        if graph.n_nodes != x.size(0):
            return None
        graph = graph.csr()
    if not isinstance(graph, CSRGraph) or x.device != graph.device or x.size(0) != graph.n_nodes or x.size(0) < 2:
        return None
    nat = _native_train_ops(graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases)
    if nat is None or not hasattr(nat, "csr_block_train"):
        return None
    cb = comb_b if comb_b is not None else bcat_direct
    if any(not t.requires_grad for t in (bias, comb_w, cb, *bases)):
        return None
    if not (relu and bn.training and bn.affine and spec.f_out % 4 == 0 and spec.f_out <= 1024):
        return None
    gamma, beta = bn.weight, bn.bias
    if any(not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in (gamma, beta)):
        return None
    rm = rv = nt = None
    if bn.track_running_stats:
        rm, rv, nt = bn.running_mean, bn.running_var, bn.num_batches_tracked
        if not (_f32_vec(rm, spec.f_out) and _f32_vec(rv, spec.f_out) and rm.is_cuda and nt is not None and nt.dtype == torch.int64
                and nt.is_cuda):
            return None
    tg = graph.transposed()
    return nat.csr_block_train(x, bias, comb_w, comb_b, bcat_direct, gamma, beta, list(bases), rm, rv, nt, graph.c_addr(), tg.c_addr(),
                               graph.tensors() + tg.tensors(), graph.workspace_for(spec), spec.c_addr, _stream_ptr(x.device),
                               (int(H), int(A), int(B), int(L), int(Ls)), bool(permute), spec.gemm_flags, float(bn.eps),
                               -1.0 if bn.momentum is None else float(bn.momentum), True, bool(residual), True)


def _layer_forward_one_call(graph: CSRGraph, spec: LayerSpec, x, packed, bcat, bias):
    """The common inference case (one GPU, packed weights at hand, no fused tail) as ONE library call
    (egc_layer_forward_packed: both launches from C) -- small batched graphs are bound by the host side.  Through the
    compiled TORCH_LIBRARY binding when it is built (one dispatcher call: allocations + both launches in C++)."""
    nat = _native_ops(x.device) if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2) else None
    if nat is not None and x.device == graph.device and tuple(x.shape) == (graph.n_nodes, spec.f_in):
        return nat.layer_forward(x if x.is_contiguous() else x.contiguous(), packed,
                                 bcat if (bcat is None or bcat.is_contiguous()) else bcat.contiguous(),
                                 bias if (bias is None or bias.is_contiguous()) else bias.contiguous(),
                                 graph.c_addr(), spec.c_addr, graph.workspace_for(spec), _stream_ptr(x.device), spec.ldb,
                                 spec.w_cols, spec.f_out)
    lib = _C.load()
    n = graph.n_nodes
    _check_f32(x, "x", (n, spec.f_in))
    if x.device != graph.device:
        raise RuntimeError(f"egc_amd: x is on {x.device} but the graph is on {graph.device}")
    dev = x.device
    x = x.contiguous()
    with _device_guard(dev):
        bases = torch.empty((n, spec.ldb), dtype=torch.float32, device=dev)
        weightings = torch.empty((n, spec.w_cols), dtype=torch.float32, device=dev)
        out = torch.empty((n, spec.f_out), dtype=torch.float32, device=dev)
        g = graph.c_struct()
        ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)),
                             lib.egc_aggregate_workspace_zero_bytes(C.byref(spec.c), graph.n_nodes, graph.n_edges))
        _C.check(lib.egc_layer_forward_packed(
            C.byref(g), C.byref(spec.c), x.data_ptr(), packed.data_ptr(),
            bcat.contiguous().data_ptr() if bcat is not None else None,
            bias.contiguous().data_ptr() if bias is not None else None, bases.data_ptr(), spec.ldb,
            weightings.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
            "egc_layer_forward_packed")
    return out


def egc_layer_forward(graph: CSRGraph, spec: LayerSpec, x: torch.Tensor, wcat: torch.Tensor,
                      bcat: torch.Tensor | None, bias: torch.Tensor | None, return_intermediates: bool = False,
                      packed: torch.Tensor | None = None, post: PostOp | None = None):
    """out[N, F_out] for one layer.  wcat = [bases_weight | comb.weight^T]  ([F_in, F_g + W]),
    bcat = comb.bias ([W]) or None, bias = layer bias ([F_out]) or None; ``packed`` = pack_weights(wcat)
    if the caller keeps one (otherwise it is produced here).  On a vertex-partitioned graph the halo rows
    of ``bases`` are exchanged (one all-to-all-v) between the two steps."""
    if isinstance(graph, GraphBatch):
        fsetup = None if return_intermediates else _batch_fused_setup(graph, spec, post, wcat, x)
        if fsetup is not None:
            return egc_layer_forward_batch_fused(graph, spec, x, wcat, bcat, bias, post, fsetup)
        tiled = None if return_intermediates else _batch_tile_setup(graph, spec, post)
        if tiled is None:
            graph = graph.csr()          # outside the tile kernels' envelope: the ordinary path on the same edges
        else:
            bases, weightings = egc_basis_transform(graph, spec, x, wcat, bcat, packed)
            return egc_aggregate_combine_batch(graph, spec, bases, weightings, bias, post, tiled)
    halo = graph.halo if graph.n_src_rows > graph.n_nodes else None
    if (halo is None and post is None and packed is not None and not return_intermediates and not gemm_exact()
            and graph.n_src_rows == graph.n_nodes):
        return _layer_forward_one_call(graph, spec, x, packed, bcat, bias)
    if (halo is None and post is not None and packed is not None and not return_intermediates and not gemm_exact()
            and graph.n_src_rows == graph.n_nodes and x.is_cuda and x.dtype == torch.float32):
        nat = _native_ops(x.device)
        if nat is not None and x.device == graph.device and tuple(x.shape) == (graph.n_nodes, spec.f_in):
            def dense(t):
                return t if (t is None or t.is_contiguous()) else t.contiguous()
            return nat.layer_forward_post(dense(x), packed, dense(bcat), dense(bias), graph.c_addr(), spec.c_addr,
                                          graph.workspace_for(spec), _stream_ptr(x.device), spec.ldb, spec.w_cols, spec.f_out,
                                          spec.gemm_flags, dense(post.scale), dense(post.shift), dense(post.residual),
                                          bool(post.relu))
    bases, weightings = egc_basis_transform(graph, spec, x, wcat, bcat, packed)
    if halo is not None and halo.n_interior is not None and post is None:
        # interior rows (no halo source) are finished while the halo rows of `bases` travel
        handle = halo.exchange_start(bases)
        out = egc_aggregate_combine(graph, spec, bases, weightings, bias, rows=(0, halo.n_interior))
        halo.exchange_finish(handle)
        out = egc_aggregate_combine(graph, spec, bases, weightings, bias, rows=(halo.n_interior, graph.n_nodes), out=out)
    else:
        if halo is not None:
            halo.exchange(bases)
        out = egc_aggregate_combine(graph, spec, bases, weightings, bias, post)
    if return_intermediates:
        return out, bases, weightings
    return out


def egc_aggregate_combine_train(graph: CSRGraph, spec: LayerSpec, bases: torch.Tensor, weightings: torch.Tensor,
                                bias: torch.Tensor | None, between_ranges=None, split_at: int | None = None):
    """Training form of egc_aggregate_combine (egc_aggregate_combine_train_f32): returns
    (out, saved) where ``saved`` = (stats, cnt, arg_max, arg_min) is what the backward consumes.
    ``split_at`` / ``between_ranges``: finish rows [0, split_at) first, call ``between_ranges()`` (e.g. wait for the halo
    rows of ``bases``), then the rest (egc_aggregate_combine_train_rows_f32)."""
    lib = _C.load()
    n = graph.n_nodes
    _check_f32(bases, "bases", (graph.n_src_rows, spec.ldb))
    _check_f32(weightings, "weightings", (n, spec.w_cols))
    # the training kernels take dense rows (strides ldb / W): a column-block view would be read at the wrong rows
    bases, weightings = bases.contiguous(), weightings.contiguous()
    dev = bases.device
    codes = [spec.c.aggrs[t] for t in range(spec.c.num_aggrs)]
    with _device_guard(dev):
        out = torch.empty((n, spec.f_out), dtype=torch.float32, device=dev)
        stats = torch.empty((n, max(int(lib.egc_train_stats_floats(C.byref(spec.c))), 1)), dtype=torch.float32, device=dev)
        cnt = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        arg_max = torch.empty((n, spec.ldb), dtype=torch.int32, device=dev) if _C.AGGR_MAX in codes else None
        arg_min = torch.empty((n, spec.ldb), dtype=torch.int32, device=dev) if _C.AGGR_MIN in codes else None
        g = graph.c_struct()
        ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)),
                             lib.egc_aggregate_workspace_zero_bytes(C.byref(spec.c), graph.n_nodes, graph.n_edges))
        bias_p = bias.contiguous().data_ptr() if bias is not None else None
        amax_p = arg_max.data_ptr() if arg_max is not None else None
        amin_p = arg_min.data_ptr() if arg_min is not None else None
        if split_at is None:
            _C.check(lib.egc_aggregate_combine_train_f32(
                C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb, weightings.data_ptr(), bias_p, out.data_ptr(),
                stats.data_ptr(), cnt.data_ptr(), amax_p, amin_p, ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                "egc_aggregate_combine_train_f32")
        else:
            for lo, hi in ((0, int(split_at)), (int(split_at), n)):
                if lo > 0 and between_ranges is not None:
                    between_ranges()
                _C.check(lib.egc_aggregate_combine_train_rows_f32(
                    C.byref(g), C.byref(spec.c), bases.data_ptr(), spec.ldb, weightings.data_ptr(), bias_p, out.data_ptr(),
                    stats.data_ptr(), cnt.data_ptr(), amax_p, amin_p, lo, hi, ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                    "egc_aggregate_combine_train_rows_f32")
    return out, (stats, cnt, arg_max, arg_min)


def egc_aggregate_combine_backward(graph: CSRGraph, spec: LayerSpec, bases, weightings, grad_out, saved,
                                   joint: bool = False):
    """(d_bases [n_src_rows, ldb], d_weightings [N, W]) through egc_aggregate_combine_backward_f32;
    ``saved`` comes from egc_aggregate_combine_train.  Always returns (d_bases, d_weightings, d_cat): with
    ``joint`` (square graphs) both are column blocks of one [N, ldb + W] array ``d_cat`` -- the left operand of
    the dense gradient GEMMs, no concatenation -- otherwise ``d_cat`` is None."""
    lib = _C.load()
    n = graph.n_nodes
    _check_f32(bases, "bases", (graph.n_src_rows, spec.ldb))
    _check_f32(weightings, "weightings", (n, spec.w_cols))
    _check_f32(grad_out, "grad_out", (n, spec.f_out))
    bases, weightings = bases.contiguous(), weightings.contiguous()  # dense rows, as in the training forward
    dev = bases.device
    stats, cnt, arg_max, arg_min = saved
    tg = graph.transposed()
    with _device_guard(dev):
        d_cat = None
        if joint and graph.n_src_rows == n and (spec.ldb + spec.w_cols) % 4 == 0:
            d_cat = torch.empty((n, spec.ldb + spec.w_cols), dtype=torch.float32, device=dev)
            d_bases, d_w = d_cat[:, :spec.ldb], d_cat[:, spec.ldb:]     # square graph: every d_bases row is written
        else:
            # rectangular graphs: hub rows' partial sums arrive by atomics into a zeroed array
            alloc = torch.empty if graph.n_src_rows == n else torch.zeros
            d_bases = alloc((graph.n_src_rows, spec.ldb), dtype=torch.float32, device=dev)
            d_w = torch.empty((n, spec.w_cols), dtype=torch.float32, device=dev)
        g, t = graph.c_struct(), tg.c_struct()
        nbytes = lib.egc_backward_workspace_bytes_for(C.byref(spec.c), C.byref(g))   # tables + per-entry max / min records
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _C.check(lib.egc_aggregate_combine_backward_f32(
            C.byref(g), C.byref(t), C.byref(spec.c), bases.data_ptr(), spec.ldb, weightings.data_ptr(),
            grad_out.contiguous().data_ptr(), stats.data_ptr(), cnt.data_ptr(),
            arg_max.data_ptr() if arg_max is not None else None, arg_min.data_ptr() if arg_min is not None else None,
            d_bases.data_ptr(), d_bases.stride(0), d_w.data_ptr(), d_w.stride(0), ws.data_ptr(), ws.numel(),
            _stream_ptr(dev)), "egc_aggregate_combine_backward_f32")
    return d_bases, d_w, d_cat   # d_cat is None unless the joint layout was asked for and applies


def _weight_grads(x: torch.Tensor, d: torch.Tensor, col_sums: bool = False, extra: torch.Tensor | None = None):
    """(x^T @ d, d.sum(0) or None[, extra.sum(0)]) for tall x [N, F], d [N, K]: the gradient of [bases_weight |
    comb_weights.weight], of comb_weights.bias and -- with ``extra`` = grad_out -- of the layer's bias (autograd's
    products behind optimized_layers.py:177-178,207-208) in one pass over the operands through egc_weight_grad_ex_f32:
    split-bf16 matrix-core products with fp32-level accuracy over row ranges (outputs of up to 128 x 192) or exact fp32
    products on a grid of output tiles (wider ones), added in a fixed order.  Shapes outside that entry point's envelope
    (a dimension not a multiple of 4) take torch's GEMM on the device.  (Rounds 2 - 5 also sent wide outputs on long
    reductions there; with the tall tiles and the per-XCD row split of round 6 the entry point is level with the library's
    split GEMM at the ogbn-mag widths -- 1.26 against 1.32 ms at 736 k x 352 x 208 -- and ahead below them, so one
    deterministic path serves every width.)  Returns a pair without ``extra``, a triple with it."""
    n, f = x.shape
    k = d.size(1)

    def done(w, s, e):
        return (w, s) if extra is None else (w, s, e)
    if (n == 0 or f % 4 or k % 4 or not x.is_cuda or x.dtype != torch.float32 or d.dtype != torch.float32
            or x.stride(1) != 1 or d.stride(1) != 1 or x.stride(0) % 4 or d.stride(0) % 4
            or x.data_ptr() % 16 or d.data_ptr() % 16):
        return done(_xt_library(x, d), _column_sums(d) if col_sums else None, _column_sums(extra) if extra is not None else None)
    lib = _C.load()
    dev = x.device
    ride = (extra is not None and col_sums and f <= 128 and k <= 192 and extra.dim() == 2 and extra.size(0) == n
            and extra.size(1) % 4 == 0 and extra.size(1) <= 128 and extra.dtype == torch.float32 and extra.stride(1) == 1
            and extra.stride(0) % 4 == 0 and extra.data_ptr() % 16 == 0 and not gemm_exact())   # (the fp32-MFMA form has no third stream)
    with _device_guard(dev):
        out = torch.empty((f, k), dtype=torch.float32, device=dev)
        cs = torch.empty(k, dtype=torch.float32, device=dev) if col_sums else None
        e_cols = extra.size(1) if ride else 0
        es = torch.empty(e_cols, dtype=torch.float32, device=dev) if ride else None
        nbytes = int(lib.egc_weight_grad_ex_workspace_bytes(n, f, k, e_cols))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _C.check(lib.egc_weight_grad_ex_f32(x.data_ptr(), x.stride(0), d.data_ptr(), d.stride(0), n, f, k, out.data_ptr(),
                                            cs.data_ptr() if cs is not None else None,
                                            extra.data_ptr() if ride else None, extra.stride(0) if ride else 0, e_cols,
                                            es.data_ptr() if ride else None, ws.data_ptr(), ws.numel(),
                                            _stream_ptr(dev)), "egc_weight_grad_ex_f32")
    if extra is not None and not ride:
        es = _column_sums(extra)
    return done(out, cs, es)


def _weight_grads_into_params(x, d, extra, dims, permute, shapes, packed_b):
    """x^T @ d, the column sums of d's weightings part and of ``extra`` (= grad_out) written STRAIGHT into gradients of the
    module's own parameters through the pack's index map (egc_weight_grad_params_f32: no d wcat array, no unpack launch), or None
    when the call is outside that entry point's envelope.  Returns (d comb_w, d comb_b or None, d bcat or None, [d basis parts],
    d bias)."""
    f_in, H, A, B, L, Ls = dims
    n, k = x.size(0), d.size(1)
    if (n == 0 or f_in > 128 or k > 192 or f_in % 4 or k % 4 or x.dtype != torch.float32 or d.dtype != torch.float32
            or x.stride(1) != 1 or d.stride(1) != 1 or x.stride(0) % 4 or d.stride(0) % 4 or x.data_ptr() % 16 or d.data_ptr() % 16
            or extra.dim() != 2 or extra.size(0) != n or extra.size(1) % 4 or extra.size(1) > 128 or extra.dtype != torch.float32
            or extra.stride(1) != 1 or extra.stride(0) % 4 or extra.data_ptr() % 16 or gemm_exact()
            or k != B * Ls + H * B * A):
        return None
    lib = _C.load()
    dev = x.device
    with _device_guard(dev):
        dcw = torch.empty(shapes[0], dtype=torch.float32, device=dev)
        dcb = torch.empty(shapes[1], dtype=torch.float32, device=dev) if packed_b else None
        dbc = None if packed_b else torch.empty(H * B * A, dtype=torch.float32, device=dev)
        dparts = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes[2]]
        ptrs = (C.c_void_p * len(dparts))(*[p.data_ptr() for p in dparts])
        e_cols = extra.size(1)
        es = torch.empty(e_cols, dtype=torch.float32, device=dev)
        nbytes = int(lib.egc_weight_grad_ex_workspace_bytes(n, f_in, k, e_cols))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _C.check(lib.egc_weight_grad_params_f32(x.data_ptr(), x.stride(0), d.data_ptr(), d.stride(0), n, f_in, H, A, B, L, Ls,
                                                int(permute), ptrs, len(dparts), dcw.data_ptr(),
                                                dcb.data_ptr() if dcb is not None else None,
                                                dbc.data_ptr() if dbc is not None else None, extra.data_ptr(), extra.stride(0),
                                                e_cols, es.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                 "egc_weight_grad_params_f32")
    return dcw, dcb, dbc, dparts, es


def _xt_library(x: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """x^T @ d on the library GEMM: as one product rocBLAS runs a single tile grid over the tiny output (397 us at
    N = 169k); split into 64 row ranges + a sum it takes 96 us."""
    n = x.size(0)
    splits = 64
    if n < 64 * splits:
        return x.t() @ d
    m = (n // splits) * splits
    out = torch.bmm(x[:m].view(splits, m // splits, -1).transpose(1, 2), d[:m].view(splits, m // splits, -1)).sum(0)
    if m < n:
        out = out + x[m:].t() @ d[m:]
    return out


def _xt_matmul(x: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    return _weight_grads(x, d)[0]


def _column_sums(t: torch.Tensor) -> torch.Tensor:
    """t.sum(0) for a float32 matrix (or a column block of one) through egc_column_sums_f32: one pass at the
    memory rate instead of torch's generic reduction (26 us per 87 MB operand at config 2)."""
    n, c = t.shape
    if (not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1 or c % 4 or c > 1024 or n == 0
            or (n > 1 and t.stride(0) % 4) or t.data_ptr() % 16):
        return t.sum(0)
    lib = _C.load()
    dev = t.device
    with _device_guard(dev):
        parts = max(1, min(1024, (n + 127) // 128))     # partial rows: one workgroup each, then a small torch sum
        out = torch.empty((parts, c), dtype=torch.float32, device=dev)
        _C.check(lib.egc_column_sums_f32(t.data_ptr(), n, int(t.stride(0)) if n > 1 else c, c, out.data_ptr(), parts,
                                         _stream_ptr(dev)), "egc_column_sums_f32")
        if parts == 1:
            return out[0]
        total = torch.empty(c, dtype=torch.float32, device=dev)
        _C.check(lib.egc_sum_partials_f32(out.data_ptr(), parts, c, total.data_ptr(), _stream_ptr(dev)), "egc_sum_partials_f32")
    return total


def _dx_matmul(d_cat: torch.Tensor, wcat: torch.Tensor) -> torch.Tensor:
    """d_cat [N, F_g + W] @ wcat^T [F_g + W, F_in] (the gradient w.r.t. x) on the split-precision matrix-core GEMM
    of the forward (egc_basis_pack / egc_basis_transform_packed with no weightings block): 95 us instead of the
    128 us of the fp32 library GEMM at config 2, 44 instead of 79 for an EGC-S layer; same fp32-level accuracy."""
    f_in, k = wcat.size(0), wcat.size(1)
    n = d_cat.size(0)
    if (gemm_exact() or f_in % 4 != 0 or k % 4 != 0 or n == 0 or not d_cat.is_cuda or not d_cat.is_contiguous()
            or d_cat.data_ptr() % 16):      # (rows of d_cat must be 16-byte aligned for the split-precision kernels)
        return d_cat @ wcat.t()
    lib = _C.load()
    dev = d_cat.device
    with _device_guard(dev):
        w = wcat.detach().contiguous()          # [f_in, k]: the transpose of this GEMM's operand, packed where it lies
        nbytes = lib.egc_basis_pack_bytes(k, f_in, 0)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dx = torch.empty((n, f_in), dtype=torch.float32, device=dev)
        stream = _stream_ptr(dev)
        _C.check(lib.egc_basis_pack_transposed(w.data_ptr(), k, k, f_in, 0, packed.data_ptr(), nbytes, stream),
                 "egc_basis_pack_transposed")
        _C.check(lib.egc_basis_transform_packed(d_cat.data_ptr(), packed.data_ptr(), None, n, k, f_in, 0, dx.data_ptr(),
                                                f_in, None, stream), "egc_basis_transform_packed")
    return dx


class _EGCLayerFunction(torch.autograd.Function):
    """Autograd around the fused forward.  The sparse part of the backward (gradients w.r.t. bases and the
    pre-activation weightings) runs in the HIP kernels of egc_backward.hip; the dense rest is three plain
    GEMMs and two column sums (torch.matmul -> rocBLAS), exactly what autograd produces for the reference's
    ``x @ bases_weight`` / ``comb_weights(x)``.  ``wcat`` / ``bcat`` are built from the module parameters with
    differentiable torch ops, so their gradients flow on to the parameters by themselves."""

    @staticmethod
    def forward(ctx, x, wcat, bcat, bias, graph, spec):
        return _layer_train_forward(ctx, x, wcat, bcat, bias, graph, spec)

    @staticmethod
    def backward(ctx, grad_out):
        need = ctx.needs_input_grad
        dx, dwcat, dbcat, dbias = _layer_train_backward(ctx, grad_out, need[0], need[1], ctx.has_bcat and need[2],
                                                        ctx.has_bias and need[3])
        return dx, dwcat, dbcat, dbias, None, None


def train_forward_core(graph, spec, x, wcat, bcat, bias):
    """(out, bases, weightings, saved) of the training forward, halo exchange included (shared with egc_amd/ops.py)."""
    bases, weightings = egc_basis_transform(graph, spec, x, wcat, bcat, None)
    halo = graph.halo if (graph.halo is not None and graph.n_src_rows > graph.n_nodes) else None
    if halo is not None and halo.n_interior is not None:
        # vertex partition, interior rows first: they are finished while the halo rows of `bases` travel (the overlapped
        # form of the inference path, egc_layer_forward)
        handle = halo.exchange_start(bases)
        out, saved = egc_aggregate_combine_train(graph, spec, bases, weightings, bias, split_at=halo.n_interior,
                                                 between_ranges=lambda: halo.exchange_finish(handle))
    else:
        if halo is not None:
            halo.exchange(bases)   # vertex partition: halo rows of `bases` from their owners
        out, saved = egc_aggregate_combine_train(graph, spec, bases, weightings, bias)
    return out, bases, weightings, saved


def _layer_train_forward(ctx, x, wcat, bcat, bias, graph, spec):
    out, bases, weightings, saved = train_forward_core(graph, spec, x, wcat, bcat, bias)
    ctx.save_for_backward(x, wcat, bases, weightings)
    ctx.graph, ctx.spec, ctx.saved = graph, spec, saved
    ctx.has_bcat, ctx.has_bias = bcat is not None, bias is not None
    return out


def _layer_train_backward(ctx, grad_out, need_x, need_wcat, need_bcat, need_bias):
    """(dx, d wcat, d bcat, d bias) of the fused layer; what is not needed is None."""
    x, wcat, bases, weightings = ctx.saved_tensors
    spec = ctx.spec
    grad_out = grad_out.contiguous()
    d_bases, d_w, d_cat = egc_aggregate_combine_backward(ctx.graph, spec, bases, weightings, grad_out, ctx.saved,
                                                         joint=spec.ldb == spec.f_g)
    halo = ctx.graph.halo
    if halo is not None and ctx.graph.n_src_rows > ctx.graph.n_nodes:
        # gradients collected for other ranks' vertices go home (reverse all-to-all-v) and are added there
        back = halo.exchange_reverse(d_bases)
        d_bases = d_bases[:ctx.graph.n_nodes].index_add(0, halo.send_idx, back)
    if d_cat is None:
        d_cat = torch.cat([d_bases[:, :spec.f_g], d_w], dim=1)         # [N, F_g + W]
    dx = _dx_matmul(d_cat, wcat) if need_x else None
    dwcat = dbcat = dbias = None
    if need_wcat:
        # the column sums of d_w (comb bias) and of grad_out (the layer's bias) ride along with x^T d_cat
        if need_bias and need_bcat:
            dwcat, sums, dbias = _weight_grads(x, d_cat, col_sums=True, extra=grad_out)
        else:
            dwcat, sums = _weight_grads(x, d_cat, col_sums=need_bcat)
        dbcat = sums[d_cat.size(1) - spec.w_cols:] if need_bcat else None
    elif need_bcat:
        dbcat = _column_sums(d_w)
    if need_bias and dbias is None:
        dbias = _column_sums(grad_out)
    return dx, dwcat, dbcat, dbias


class _EGCLayerParamsFunction(torch.autograd.Function):
    """_PackWeightsFunction and _EGCLayerFunction as ONE autograd node: the layer straight from the module's parameters
    (comb weight, comb bias, basis matrices), their gradients straight back -- one ``apply`` and one backward node per
    layer instead of two (a small batch's training step is bound by exactly that).
    ``comb_b`` goes through the pack (EGConv: rows permuted with the weight's); ``bcat_direct`` is a combination bias
    already in the operand's order (EfficientGraphConv), used and differentiated as it is."""

    @staticmethod
    def forward(ctx, x, bias, comb_w, comb_b, bcat_direct, graph, spec, dims, permute, *bases):
        ctx.dims, ctx.permute, ctx.packed_b = dims, permute, comb_b is not None
        ctx.shapes = (comb_w.shape, comb_b.shape if comb_b is not None else None, [b.shape for b in bases])
        nat = _native_train_ops(graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases)
        ctx.native = nat is not None
        if nat is not None:
            # pack + planes + GEMM + training aggregate as ONE compiled call (egc_amd/csrc_ext: train_forward)
            f_in, H, A, B, L, Ls = dims
            out, wcat, _, bases_t, weightings, stats, cnt, arg_max, arg_min = nat.train_forward(
                x, comb_w, comb_b, bcat_direct, list(bases), bias, graph.c_addr(), spec.c_addr, graph.workspace_for(spec),
                _stream_ptr(x.device), H, A, B, L, Ls, bool(permute), spec.gemm_flags)
            ctx.save_for_backward(x, wcat, bases_t, weightings, stats, cnt, arg_max, arg_min)
            ctx.graph, ctx.spec = graph, spec
            ctx.has_bcat, ctx.has_bias = True, True
            return out
        wcat, bcat = _pack_params(dims, permute, comb_w, comb_b, bases)
        return _layer_train_forward(ctx, x, wcat, bcat if comb_b is not None else bcat_direct, bias, graph, spec)

    @staticmethod
    def backward(ctx, grad_out):
        need = ctx.needs_input_grad
        if ctx.native:
            x, wcat, bases_t, weightings, stats, cnt, arg_max, arg_min = ctx.saved_tensors
            f_in, H, A, B, L, Ls = ctx.dims
            nat = _native_ops(x.device)
            shapes = ctx.shapes
            res = nat.train_backward(grad_out, x, wcat, bases_t, weightings, stats, cnt, arg_max, arg_min, ctx.graph.c_addr(),
                                     ctx.graph.transposed().c_addr(), ctx.spec.c_addr, _stream_ptr(x.device), H, A, B, L, Ls,
                                     bool(ctx.permute), bool(ctx.packed_b), bool(need[0]), list(shapes[0]),
                                     list(shapes[1]) if shapes[1] is not None else [0], len(shapes[2]), list(shapes[2][0]))
            dx, dcw, dcb_or_bcat, dbias = res[0] if need[0] else None, res[1], res[2], res[3]
            return (dx, dbias, dcw, dcb_or_bcat if ctx.packed_b else None, None if ctx.packed_b else dcb_or_bcat,
                    None, None, None, None, *res[4:])
        need_w = need[2] or any(need[9:])
        need_b = ctx.has_bcat and (need[3] if ctx.packed_b else need[4])
        dx, dwcat, dbcat, dbias = _layer_train_backward(ctx, grad_out, need[0], need_w, need_b, ctx.has_bias and need[1])
        dcw = dcb = None
        dparts = [None] * len(ctx.shapes[2])
        if need_w or (need_b and ctx.packed_b):
            dcw, dcb, dparts = _unpack_param_grads(ctx.dims, ctx.permute, ctx.shapes, ctx.packed_b, dwcat,
                                                   dbcat if ctx.packed_b else None)
        return (dx, dbias, dcw, dcb, None if ctx.packed_b else dbcat, None, None, None, None, *dparts)


class _DenseTransformFunction(torch.autograd.Function):
    """Autograd around the dense half of a layer alone: (bases [N, ldb], weightings [N, W]) = x @ [B | C] (+ bcat on
    the weightings block) through egc_basis_transform, with the gradients w.r.t. x / wcat / bcat through the
    repository's own GEMMs (_dx_matmul, _weight_grads) -- what autograd produces for the reference's
    ``torch.matmul(x, bases_weight)`` + one ``Linear`` per weightings block (rmag/models.py:113-143), as ONE forward
    GEMM per node type instead of one per Linear."""

    @staticmethod
    def forward(ctx, x, wcat, bcat, graph, gspec):
        bases, weightings = egc_basis_transform(graph, gspec, x, wcat, bcat, None)
        ctx.save_for_backward(x, wcat)
        ctx.gspec, ctx.has_bcat = gspec, bcat is not None
        return bases, weightings

    @staticmethod
    def backward(ctx, d_bases, d_w):
        x, wcat = ctx.saved_tensors
        f_g = ctx.gspec.f_g
        d_cat = torch.cat([d_bases[:, :f_g], d_w], dim=1)   # pad columns of `bases` carry no gradient
        dx = _dx_matmul(d_cat, wcat) if ctx.needs_input_grad[0] else None
        need_bcat = ctx.has_bcat and ctx.needs_input_grad[2]
        dwcat = dbcat = None
        if ctx.needs_input_grad[1]:
            dwcat, sums = _weight_grads(x, d_cat, col_sums=need_bcat)
            dbcat = sums[f_g:] if need_bcat else None
        elif need_bcat:
            dbcat = _column_sums(d_w.contiguous())
        return dx, dwcat, dbcat, None, None


def egc_dense_transform_apply(graph, gspec, x, wcat, bcat):
    """egc_basis_transform with autograd when any input requires a gradient (``gspec``: f_in, f_g, w_cols, ldb)."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, wcat, bcat)):
        return _DenseTransformFunction.apply(x, wcat, bcat, graph, gspec)
    return egc_basis_transform(graph, gspec, x, wcat, bcat, None)


class _AggregateCombineFunction(torch.autograd.Function):
    """Autograd around the fused aggregate/combine alone: ``bases`` [n_src_rows, ldb] and the pre-activation
    ``weightings`` [N, W] (layout [h][b][a]) come from differentiable torch ops of the caller (relational EGC:
    the basis table of the SOURCE node type, the combination Linear of the TARGET type)."""

    @staticmethod
    def forward(ctx, bases, weightings, bias, graph, spec):
        out, saved = egc_aggregate_combine_train(graph, spec, bases, weightings, bias)
        ctx.save_for_backward(bases, weightings)
        ctx.graph, ctx.spec, ctx.saved, ctx.has_bias = graph, spec, saved, bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        bases, weightings = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        d_bases, d_w, _ = egc_aggregate_combine_backward(ctx.graph, ctx.spec, bases, weightings, grad_out, ctx.saved)
        dbias = _column_sums(grad_out) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return d_bases, d_w, dbias, None, None


def _moment_partials(a: torch.Tensor, b=None, scale=None, shift=None, relu=True, keep=None, keep_scale=1.0,
                     count_inc=None, n_valid=None) -> torch.Tensor:
    """[parts][2][C] float64 partial sums (sum_r g, sum_r g * b) over row blocks through egc_column_moments_f64; b is None:
    g = a and the second sum is the second moment of a; else g = a * keep * keep_scale * [b * scale + shift > 0] (dropout
    mask if given, ReLU mask if ``relu``).  The blocks are added by the finalize kernels (egc_bn_forward_finalize /
    egc_bn_backward_finalize)."""
    lib = _C.load()
    n, c = a.shape
    dev = a.device
    with _device_guard(dev):
        parts = max(1, min(1024, (n + 127) // 128))
        out = torch.empty((parts, 2, c), dtype=torch.float64, device=dev)
        _C.check(lib.egc_column_moments_f64(a.data_ptr(), b.data_ptr() if b is not None else None,
                                            scale.data_ptr() if scale is not None else None,
                                            shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                            keep.data_ptr() if keep is not None else None, float(keep_scale), n, c,
                                            out.data_ptr(), parts, count_inc.data_ptr() if count_inc is not None else None,
                                            n_valid.data_ptr() if n_valid is not None else None,
                                            _stream_ptr(dev)), "egc_column_moments_f64")
    return out


def _f32_vec(t, c):
    """A [C] parameter / buffer the finalize kernels may read in place (float32, dense), else None."""
    return t is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == c


class _BatchNormActResidualFunction(torch.autograd.Function):
    """out = act(batch_norm(h; batch statistics) * gamma + beta) + residual -- the training-mode tail of the
    reference's blocks (zinc/models.py:66-72) in two streaming passes each way plus ONE per-channel launch between them
    (egc_tail.hip), which also updates the module's running statistics when they are passed.  ``keep`` ([N, C] uint8,
    0 = dropped) with ``keep_scale`` = 1 / (1 - p) puts a dropout between the activation and the residual add, as the
    ogbn-arxiv net has it (arxiv/norm_models.py:34-40).  Returns (out, batch mean, biased batch variance), both float64."""

    @staticmethod
    def forward(ctx, h, residual, gamma, beta, eps, relu, running_mean, running_var, momentum, n_tracked, keep, keep_scale,
                n_valid, sync=None, res_link=None):
        lib = _C.load()
        n, c = h.shape
        dev = h.device
        ctx.res_link = res_link
        h = h.contiguous()
        gamma_c = gamma.detach().contiguous().float() if gamma is not None else None
        beta_c = beta.detach().contiguous().float() if beta is not None else None
        res = residual.contiguous() if residual is not None else None
        ctx.sync = sync
        with _device_guard(dev):
            n_parts = max(1, min(1024, (n + 127) // 128))
            parts = torch.empty((n_parts, 2, c), dtype=torch.float64, device=dev)
            stats = torch.empty((3, c), dtype=torch.float64, device=dev)     # mean | biased variance | 1 / std
            affine = torch.empty((2, c), dtype=torch.float32, device=dev)    # scale | shift
            out = torch.empty_like(h)
            stream = _stream_ptr(dev)
            # statistics pass (which also bumps num_batches_tracked) + the per-channel step: one call, and with a sync word
            # and few partial blocks one launch (egc_bn_forward_stats_f32)
            _C.check(lib.egc_bn_forward_stats_f32(
                h.data_ptr(), n, c, parts.data_ptr(), n_parts,
                n_tracked.data_ptr() if (n_tracked is not None and running_mean is not None) else None,
                n_valid.data_ptr() if n_valid is not None else None,
                gamma_c.data_ptr() if gamma_c is not None else None,
                beta_c.data_ptr() if beta_c is not None else None, float(eps), stats.data_ptr(), affine.data_ptr(),
                running_mean.data_ptr() if running_mean is not None else None,
                running_var.data_ptr() if running_var is not None else None,
                -1.0 if momentum is None else float(momentum),
                n_tracked.data_ptr() if n_tracked is not None else None,
                sync.data_ptr() if sync is not None else None, stream), "egc_bn_forward_stats_f32")
            _C.check(lib.egc_affine_act_residual_f32(h.data_ptr(), affine[0].data_ptr(), affine[1].data_ptr(),
                                                     res.data_ptr() if res is not None else None, int(relu),
                                                     keep.data_ptr() if keep is not None else None, float(keep_scale),
                                                     n, c, out.data_ptr(),
                                                     n_valid.data_ptr() if n_valid is not None else None, stream),
                     "egc_affine_act_residual_f32")
        ctx.save_for_backward(h, affine, stats, gamma_c, keep, n_valid)
        ctx.keep_scale = float(keep_scale)
        ctx.set_materialize_grads(False)     # (mean / var carry no gradient: no zero-filled stand-ins per backward)
        ctx.relu, ctx.has_res, ctx.has_gamma, ctx.has_beta = bool(relu), residual is not None, gamma is not None, beta is not None
        mean, var = stats[0], stats[1]
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, dout, _dmean, _dvar):
        lib = _C.load()
        h, affine, stats, gamma_c, keep, n_valid = ctx.saved_tensors
        n, c = h.shape
        dev = h.device
        if dout is None:
            return (None,) * 15
        dout = dout.contiguous()
        dh = dgamma = dbeta = None
        if ctx.needs_input_grad[0] or (ctx.has_gamma and ctx.needs_input_grad[2]) or (ctx.has_beta and ctx.needs_input_grad[3]):
            masked = ctx.relu or keep is not None or n_valid is not None
            if not masked:
                s1 = _column_sums(dout).double()
                sgh = (dout.double() * h.double()).sum(0) if n else torch.zeros(c, dtype=torch.float64, device=dev)
                parts = torch.stack([s1, sgh]).unsqueeze(0).contiguous()
            with _device_guard(dev):
                out5 = torch.empty((5, c), dtype=torch.float32, device=dev)   # d gamma | d beta | coef_g | coef_h | coef_1
                stream = _stream_ptr(dev)
                if masked:      # sum g, sum g h (g = dout * dropout mask * relu mask) + the per-channel step: one call
                    n_parts = max(1, min(1024, (n + 127) // 128))
                    parts = torch.empty((n_parts, 2, c), dtype=torch.float64, device=dev)
                    sync = ctx.sync
                    _C.check(lib.egc_bn_backward_stats_f32(
                        dout.data_ptr(), h.data_ptr(), affine[0].data_ptr(), affine[1].data_ptr(), int(ctx.relu),
                        keep.data_ptr() if keep is not None else None, ctx.keep_scale, n, c, parts.data_ptr(), n_parts,
                        n_valid.data_ptr() if n_valid is not None else None, stats.data_ptr(),
                        gamma_c.data_ptr() if gamma_c is not None else None, out5.data_ptr(),
                        sync.data_ptr() if sync is not None else None, stream), "egc_bn_backward_stats_f32")
                else:
                    _C.check(lib.egc_bn_backward_finalize(parts.data_ptr(), parts.size(0), c, n, stats.data_ptr(),
                                                          gamma_c.data_ptr() if gamma_c is not None else None, out5.data_ptr(),
                                                          n_valid.data_ptr() if n_valid is not None else None, stream),
                             "egc_bn_backward_finalize")
                dgamma = out5[0] if ctx.has_gamma and ctx.needs_input_grad[2] else None
                dbeta = out5[1] if ctx.has_beta and ctx.needs_input_grad[3] else None
                if ctx.needs_input_grad[0]:
                    dh = torch.empty_like(h)
                    _C.check(lib.egc_affine_act_backward_f32(dout.data_ptr(), h.data_ptr(), affine[0].data_ptr(),
                                                             affine[1].data_ptr(), int(ctx.relu),
                                                             keep.data_ptr() if keep is not None else None, ctx.keep_scale,
                                                             out5[2].data_ptr(), out5[3].data_ptr(), out5[4].data_ptr(), n, c,
                                                             dh.data_ptr(), n_valid.data_ptr() if n_valid is not None else None,
                                                             stream), "egc_affine_act_backward_f32")
        dres = dout if ctx.has_res and ctx.needs_input_grad[1] else None
        if dres is not None and ctx.res_link is not None:
            ctx.res_link.grad, dres = dres, None      # (joins d x inside the conv's backward launch: ResidualLink)
        return dh, dres, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


def batch_norm_act_residual_supported(h: torch.Tensor) -> bool:
    return (h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.size(0) > 1 and h.size(1) % 4 == 0
            and h.size(1) <= 1024)


def batch_norm_act_residual(h, residual, gamma, beta, eps: float, relu: bool, running_mean=None, running_var=None,
                            momentum=None, num_batches_tracked=None, keep=None, keep_scale: float = 1.0, n_valid=None,
                            sync=None, res_link=None):
    """Training-mode BatchNorm1d (batch statistics) -> optional ReLU -> optional residual add, fused
    (_BatchNormActResidualFunction): returns (out, batch mean [C] float64, biased batch variance [C] float64).
    With ``running_mean`` / ``running_var`` (float32 [C], dense) the running statistics are updated in the same launch
    that finishes the batch statistics, as nn.BatchNorm1d does: unbiased variance, ``momentum``, or -- momentum None --
    the cumulative average over ``num_batches_tracked`` (a device int64 scalar, INCREMENTED here when given).
    ``keep`` / ``keep_scale``: dropout between the activation and the residual add (see the Function).
    ``n_valid`` (device int64 scalar): only the first n_valid rows are real -- the rest is the padding of a batch brought
    to a recording's static shape; statistics and gradients are those of nn.BatchNorm1d on the real rows.
    ``sync`` (device int32 scalar, zero; the caller's for the lifetime of its module): lets the statistics pass and the
    per-channel step of small inputs be ONE launch each way (egc_bn_forward_stats_f32)."""
    c = h.size(1)
    if running_mean is not None and not (_f32_vec(running_mean, c) and _f32_vec(running_var, c)
                                         and (momentum is not None or num_batches_tracked is not None)):
        raise RuntimeError("egc_amd: running statistics must be dense float32 [C] tensors")
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or num_batches_tracked.numel() != 1
                                            or num_batches_tracked.device != h.device):
        raise RuntimeError("egc_amd: num_batches_tracked must be an int64 scalar on the device of h")
    if keep is not None and (keep.dtype != torch.uint8 or keep.shape != h.shape or not keep.is_contiguous()
                             or keep.device != h.device):
        raise RuntimeError("egc_amd: the dropout mask must be a dense uint8 tensor of the shape of h")
    if n_valid is not None and (n_valid.dtype != torch.int64 or n_valid.numel() != 1 or n_valid.device != h.device):
        raise RuntimeError("egc_amd: n_valid must be an int64 scalar on the device of h")
    return _BatchNormActResidualFunction.apply(h, residual, gamma, beta, float(eps), bool(relu), running_mean, running_var,
                                               momentum, num_batches_tracked, keep, float(keep_scale), n_valid, sync, res_link)


def egc_aggregate_combine_apply(graph, spec, bases, weightings, bias=None):
    """egc_aggregate_combine with autograd when any input requires a gradient."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (bases, weightings, bias)):
        return _AggregateCombineFunction.apply(bases, weightings, bias, graph, spec)
    return egc_aggregate_combine(graph, spec, bases, weightings, bias)


def _as_csr(graph):
    return graph.csr() if isinstance(graph, GraphBatch) else graph


def egc_layer_apply_params(graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases, f_in, H, A, B, L, Ls, permute_hab):
    """The training-path layer call from the module parameters (one autograd node: _EGCLayerParamsFunction).  ``bases``:
    one [f_in, B L] matrix or B [f_in, L] matrices; ``comb_b`` a combination bias to permute with the weight's rows,
    ``bcat_direct`` one already in the operand's order (pass exactly one of the two, or neither)."""
    if isinstance(graph, GraphBatch) and spec.ldb == spec.f_g:
        setups = _batch_fused_train_setup(graph, spec, x)
        if setups is not None:
            if getattr(ResidualLink._local, "offered", None) is None:        # (a block's Python tail would hand its gradient over)
                out = native_block_train((graph, spec, x, bias, comb_w, comb_b, bcat_direct, bases, f_in, H, A, B, L, Ls, permute_hab),
                                         with_tail=False)
                if out is not None:
                    return out
            return _BatchFusedTrainFunction.apply(x, bias, comb_w, comb_b, bcat_direct, graph, spec,
                                                  (int(f_in), int(H), int(A), int(B), int(L), int(Ls)), bool(permute_hab), setups,
                                                  ResidualLink.take(x) if x.requires_grad else None, *bases)
    return _EGCLayerParamsFunction.apply(x, bias, comb_w, comb_b, bcat_direct, _as_csr(graph), spec,
                                         (int(f_in), int(H), int(A), int(B), int(L), int(Ls)), bool(permute_hab), *bases)


def egc_layer_apply(graph, spec, x, wcat, bcat, bias, packed=None):
    from . import ops   # torch.library registration of the same call (egc_amd/ops.py)
    if ops.use_torch_op():
        return ops.layer_apply_op(_as_csr(graph), spec, x, wcat, bcat, bias)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, wcat, bcat, bias)):
        return _EGCLayerFunction.apply(x, wcat, bcat, bias, _as_csr(graph), spec)
    return egc_layer_forward(graph, spec, x, wcat, bcat, bias, packed=packed)
