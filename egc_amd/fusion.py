"""Caller-side pieces around the layer, fused.

The reference's graph nets wrap every EGC layer the same way (zinc/models.py:66-72, mol/pna_style_models.py:71-78,
cifar/models.py:67-74):   x = conv(x, edge_index);  x = bn(x);  x = relu(x);  x = x + identity   and finish with
``global_mean_pool(x, batch)``; the ogbn-arxiv net puts ``F.dropout`` between the ReLU and the residual add
(arxiv/norm_models.py:34-40).  In eval mode BatchNorm1d is a per-channel affine map, so the whole tail folds into the
store of the fused aggregate/combine kernel (``egc_aggregate_combine_post_f32``): three elementwise passes over
[N, F_out] and three launches less per layer.  With batch statistics (training) the statistics need every row of the
layer's output first, so the tail is its own two streaming passes each way (``batch_norm_act_residual``:
statistics, then normalise + ReLU + residual in one pass; backward the two BatchNorm sums with the ReLU mask
recomputed, then one pass for the gradient) instead of PyTorch's pass per operator.
"""
from __future__ import annotations

import torch

from . import _C
import torch.nn as nn

from .functional import (PostOp, ResidualLink, native_block_train, native_csr_block_train, batch_norm_act_residual, batch_norm_act_residual_supported, egc_layer_forward,
                         segment_mean)
from .graph import GraphBatch, graph_from_input


class FusedEGCBlock(nn.Module):
    """conv -> BatchNorm1d -> ReLU (-> dropout) (-> + input) as one module; ``conv`` is an ``egc_amd.EfficientGraphConv``
    or ``egc_amd.EGConv``, ``bn`` the ``nn.BatchNorm1d`` that follows it in the reference nets (shared, not copied:
    state dicts keep their keys), ``dropout`` the probability of the arxiv net's ``F.dropout(x, p, self.training)``
    (0 for the other nets).  The dropout mask is drawn from torch's generator on the input's device (one byte per
    element), so ``torch.manual_seed`` reproduces a run; the mask of the last training forward stays in
    ``last_keep_mask`` for inspection."""

    def __init__(self, conv: nn.Module, bn: nn.BatchNorm1d | None = None, relu: bool = True, residual: bool = True,
                 dropout: float = 0.0):
        super().__init__()
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout}")
        self.conv, self.bn, self.relu, self.residual, self.dropout = conv, bn, relu, residual, float(dropout)
        self.last_keep_mask = None
        self._affine, self._affine_key = None, None
        # arrival counter of the one-launch form of the statistics step (egc_bn_forward_stats_f32; opt-in, EGC_BN_ONE_LAUNCH=1:
        # measured slower than the two launches when the step is replayed as a hipGraph); not state
        self.register_buffer("_bn_sync", torch.zeros(1, dtype=torch.int32), persistent=False)

    def _dropping(self):
        return self.dropout > 0.0 and self.training

    def _plain(self, x, edge_index, identity=None):
        identity = x if identity is None else identity
        h = self.conv(x=x, edge_index=edge_index) if hasattr(self.conv, "aggs") else self.conv(x, edge_index)
        if self.bn is not None:
            h = self.bn(h)
        if self.relu:
            h = torch.relu(h)
        if self._dropping():
            h = torch.nn.functional.dropout(h, self.dropout, True)
        return identity + h if self.residual else h

    def _batch_stats(self, x, edge_index, identity=None, n_valid=None):
        """conv, then BatchNorm1d on batch statistics -> ReLU -> + input in two passes; the running statistics are
        updated as nn.BatchNorm1d does (momentum or cumulative average, unbiased variance)."""
        bn = self.bn
        # A GraphBatch in training: the whole block as ONE autograd node of the compiled binding (csrc_ext: batch_block_train --
        # the launches of the path below issued from C++, the residual branch's gradient handed to the conv's backward launch
        # inside the node), when the call is inside its envelope; else the Python Functions below, same kernels.
        if (identity is None and n_valid is None and not self._dropping() and self.relu
                and bn.training and hasattr(self.conv, "_train_call") and torch.is_grad_enabled()
                and not (self.residual and _C.env_flag("EGC_NO_RESIDUAL_LINK"))):
            call = self.conv._train_call(x, edge_index)
            if call is not None:
                out = native_block_train(call, bn, relu=True, residual=bool(self.residual), with_tail=True)
                if out is None:      # outside the one-launch envelope (the reference's wide nets, full graphs): the CSR path's node
                    out = native_csr_block_train(call, bn, relu=True, residual=bool(self.residual))
                if out is not None:
                    return out
        # the residual branch's gradient may join d x inside the conv's backward launch (functional.ResidualLink): offered when
        # the residual adds the conv's own input, taken by the one-launch training path of a GraphBatch
        link = None
        if self.residual and identity is None and torch.is_grad_enabled() and x.requires_grad and not _C.env_flag("EGC_NO_RESIDUAL_LINK"):
            link = ResidualLink(x)
        identity = x if identity is None else identity
        ResidualLink.offer(link)
        try:
            h = self.conv(x=x, edge_index=edge_index) if hasattr(self.conv, "aggs") else self.conv(x, edge_index)
        finally:
            ResidualLink.offer(None)
        link = link if (link is not None and link.taken) else None
        if not batch_norm_act_residual_supported(h) or (self.residual and identity.shape != h.shape):
            if n_valid is not None:
                raise RuntimeError("egc_amd: n_valid needs the fused training tail (float32 CUDA activations, channels % 4 == 0)")
            h = bn(h)
            h = torch.relu(h) if self.relu else h
            if self._dropping():
                h = torch.nn.functional.dropout(h, self.dropout, True)
            return identity + h if self.residual else h
        keep = None
        if self._dropping():
            keep = torch.empty(h.shape, dtype=torch.uint8, device=h.device).bernoulli_(1.0 - self.dropout)
            self.last_keep_mask = keep
        track = bn.training and bn.track_running_stats
        in_place = (track and bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous()
                    and bn.running_var.dtype == torch.float32 and bn.running_var.is_contiguous())
        counted = in_place and bn.num_batches_tracked is not None and bn.num_batches_tracked.dtype == torch.int64
        if track and not counted:
            with torch.no_grad():
                bn.num_batches_tracked += 1      # (else: bumped by the statistics pass, on the device)
        out, mean, var = batch_norm_act_residual(
            h, identity if self.residual else None, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.eps,
            self.relu, bn.running_mean if in_place else None, bn.running_var if in_place else None, bn.momentum,
            bn.num_batches_tracked if counted else None, keep, 1.0 / (1.0 - self.dropout), n_valid,
            sync=self._bn_sync if (self._bn_sync.device == h.device and _C.env_flag("EGC_BN_ONE_LAUNCH")) else None,
            res_link=link)
        if track and not in_place:               # running statistics kept in another dtype: torch's arithmetic
            with torch.no_grad():
                n = h.size(0)
                m = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
                bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
                bn.running_var.mul_(1 - m).add_((var * (n / (n - 1))).to(bn.running_var.dtype), alpha=m)
        return out

    def forward(self, x, edge_index, identity=None, n_valid=None):
        """``identity``: what the residual adds when it is not the layer's input itself -- the CIFAR net drops out the
        layer's input but adds the undropped activations back (cifar/models.py:64-71): ``block(drop(x), ei, identity=x)``.
        ``n_valid`` (device int64 scalar): the batch occupies the first n_valid rows, the rest is padding up to the
        static shape of a hipGraph recording (egc_amd.GraphedStep); BatchNorm's batch statistics then count the real
        rows only, so one recording serves batches of every size up to the padded one with unchanged numerics."""
        bn = self.bn
        if bn is not None and (bn.training or not bn.track_running_stats):
            return self._batch_stats(x, edge_index, identity, n_valid)   # batch statistics: the tail is its own two passes
        if self._dropping():                              # (dropout without batch statistics: torch's operators)
            return self._plain(x, edge_index, identity)
        fusable = not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                                    or (identity is not None and identity.requires_grad)))
        if not fusable:
            return self._plain(x, edge_index, identity)
        identity = x if identity is None else identity
        conv = self.conv
        scale = shift = None
        if bn is not None:
            # eval-mode BatchNorm is a per-channel affine map: computed once per state of the module (five tiny torch
            # kernels per call otherwise -- most of a small batch's forward); keyed on the version counters, as the layers'
            # packed weights are: in-place updates (optimizer steps, load_state_dict) are seen, writes through .data are not
            srcs = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
            key = tuple((t.data_ptr(), t._version) for t in srcs) + (bn.eps,)
            if self._affine_key != key:
                with torch.no_grad():
                    inv = torch.rsqrt(bn.running_var + bn.eps)
                    sc = inv * bn.weight if bn.affine else inv
                    sh = (bn.bias if bn.affine else 0) - bn.running_mean * sc
                self._affine, self._affine_key = (sc.contiguous(), sh.contiguous()), key
            scale, shift = self._affine
        post = PostOp(scale, shift, identity if self.residual else None, self.relu)
        graph = graph_from_input(edge_index, x.size(0))
        if hasattr(conv, "aggs"):      # EfficientGraphConv
            wcat = conv._packed_weights()
            return egc_layer_forward(graph, conv._spec, x, wcat, conv.comb_weights.bias, conv.bias,
                                     packed=conv._weight_planes(wcat), post=post)
        wcat, bcat = conv._packed_weights()   # EGConv
        spec = conv._spec_coo if ((isinstance(edge_index, torch.Tensor) and edge_index.layout == torch.strided)
                                  or isinstance(edge_index, GraphBatch)) else conv._spec_adj
        return egc_layer_forward(graph, spec, x, wcat, bcat, conv.bias, packed=conv._weight_planes(spec, wcat), post=post)


class _SegmentMeanFunction(torch.autograd.Function):
    """out[g] = mean of the rows of graph g (a sorted batch vector's segments); d x[r] = d out[batch[r]] / count."""

    @staticmethod
    def forward(ctx, x, seg, batch):
        ctx.save_for_backward(seg, batch)
        return segment_mean(x, seg)

    @staticmethod
    def backward(ctx, dout):
        seg, batch = ctx.saved_tensors
        counts = (seg[1:] - seg[:-1]).clamp_(min=1).to(dout.dtype)
        return (dout / counts[:, None]).index_select(0, batch), None, None


def global_mean_pool(x: torch.Tensor, batch: torch.Tensor, size: int | None = None) -> torch.Tensor:
    """torch_geometric.nn.global_mean_pool for a SORTED batch vector (graphs are contiguous in a PyG batch; the
    readout of the reference's batched nets, zinc/models.py:73): a segmented mean on the device
    (``egc_segment_mean_f32``), differentiable.  Pass ``size`` (the number of graphs) where nothing may be read back
    to the host -- inside a hipGraph recording; without it the number of graphs comes from ``batch.max()``."""
    n_graphs = int(batch.max()) + 1 if size is None else int(size)
    seg = torch.searchsorted(batch, torch.arange(n_graphs + 1, device=batch.device, dtype=batch.dtype))
    if torch.is_grad_enabled() and x.requires_grad:
        return _SegmentMeanFunction.apply(x, seg, batch)
    return segment_mean(x, seg)
