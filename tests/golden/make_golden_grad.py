#!/usr/bin/env python3
"""Generate tests/golden/grad_*.npz: forward AND backward of the REFERENCE's own layer code.

The reference's backward is PyTorch autograd through ``experiments/layers.py:89-140`` /
``experiments/optimized_layers.py:177-208``.  This script imports those files as-is (as make_golden.py does)
but installs DIFFERENTIABLE shims for the absent third-party packages:

  * ``torch_scatter.scatter``          index_add_ (sum), sum / clamp(count, 1) (mean), and for max / min a gather of
                                       the FIRST entry attaining the extremum -- torch_scatter's CPU arg rule (its
                                       scatter_max / scatter_min update the argument only on a strict improvement),
                                       ties decided at float32 precision (the precision the reference computes in);
  * ``gcn_norm`` / ``add_remaining_self_loops``   the oracle's restatements: index manipulation and constant
                                       weights, nothing to differentiate;
  * ``MessagePassing.propagate``       index_select + the layer's own message / aggregate (as in make_golden.py).

Each fixture holds the inputs, the parameters, an upstream gradient, the forward output and the gradients w.r.t. x
and every parameter -- computed twice, in float32 (the reference's own precision) and in float64 (the reference's
code run in double: what the HIP backward is held to), plus the max / min ``arg`` positions where the layer has them.
Runs only in the build container (needs /root/reference); only vectors are committed.

Usage:  python tests/golden/make_golden_grad.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the non-differentiable shim set; pieces are replaced below)
from oracle import egc_oracle as orc  # noqa: E402

ARGS = {}  # name -> list of arg tensors recorded by the scatter shim (max / min), in call order


def diff_scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """Differentiable torch_scatter.scatter(src [E, F], index [E], dim=0 / -2, dim_size, reduce)."""
    assert out is None and src.dim() == 2 and dim in (0, -2)
    n, f = int(dim_size), src.size(1)
    if reduce in ("sum", "add"):
        return torch.zeros(n, f, dtype=src.dtype).index_add(0, index, src)
    if reduce == "mean":
        s = torch.zeros(n, f, dtype=src.dtype).index_add(0, index, src)
        cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones(index.numel(), dtype=src.dtype))
        return s / cnt.clamp(min=1).view(-1, 1)
    assert reduce in ("max", "min")
    red = "amax" if reduce == "max" else "amin"
    e = src.size(0)
    idx = index.view(-1, 1).expand(-1, f)
    with torch.no_grad():
        s32 = src.float()  # ties at the reference's precision
        ext = torch.zeros(n, f, dtype=s32.dtype).scatter_reduce(0, idx, s32, red, include_self=False)
        pos = torch.arange(e).view(-1, 1).expand(-1, f)
        pos = torch.where(s32 == ext[index], pos, torch.full_like(pos, e))
        first = torch.full((n, f), e, dtype=torch.int64).scatter_reduce(0, idx, pos, "amin", include_self=True)
        empty = first >= e
    ARGS.setdefault("args", []).append((reduce, first.clone()))
    picked = torch.gather(src, 0, first.clamp(max=max(e - 1, 0))) if e > 0 else torch.zeros(n, f, dtype=src.dtype)
    return torch.where(empty, torch.zeros_like(picked), picked)


def diff_gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False, add_self_loops=True, dtype=None):
    assert edge_weight is None and not improved and isinstance(edge_index, torch.Tensor)
    ei, w = orc.gcn_norm(edge_index.numpy(), num_nodes, add_self_loops)
    return torch.from_numpy(ei), torch.from_numpy(w).to(dtype if dtype is not None else torch.float32)


class DiffMessagePassing(mg.MessagePassing):
    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return diff_scatter(inputs, index, self.node_dim, None, dim_size, self.aggr)


def install():
    mg.install_shims()
    sys.modules["torch_scatter"].scatter = diff_scatter
    sys.modules["torch_geometric.nn.conv.gcn_conv"].gcn_norm = diff_gcn_norm
    sys.modules["torch_geometric.nn"].MessagePassing = DiffMessagePassing
    sys.modules["torch_geometric.nn.conv"].MessagePassing = DiffMessagePassing


def cases():
    g_messy = dict(n=57, e=260, self_loops=9, dups=25, isolated_tail=3)
    g_plain = dict(n=48, e=200)
    g_hub = dict(n=400, e=5200, heavy=True)          # heavy-tailed: rows with hundreds of entries
    out = [
        ("grad_opt_northstar", "opt", dict(fin=128, fout=128, H=8, B=4, aggrs=["sum", "mean", "max", "symnorm"], graph=g_messy)),
        ("grad_opt_northstar_hub", "opt", dict(fin=64, fout=128, H=8, B=4, aggrs=["sum", "mean", "max", "symnorm"], graph=g_hub)),
        ("grad_opt_symnorm", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["symnorm"], graph=g_messy)),
        ("grad_opt_minstdvar", "opt", dict(fin=24, fout=32, H=4, B=2, aggrs=["min", "std", "var"], graph=g_messy)),
        ("grad_opt_noselfloops", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["sum", "max", "std"], graph=g_messy, add_self_loops=False)),
        ("grad_opt_sigmoid", "opt", dict(fin=32, fout=32, H=4, B=4, aggrs=["symnorm", "mean"], graph=g_messy, sigmoid=True)),
        ("grad_opt_mag_first", "opt", dict(fin=128, fout=352, H=8, B=4, aggrs=["mean"], graph=g_plain)),
        ("grad_opt_ties", "opt", dict(fin=8, fout=16, H=2, B=2, aggrs=["max", "min", "sum"], graph=dict(n=20, e=120, dups=30), integer=True)),
        ("grad_opt_L129", "opt", dict(fin=16, fout=129, H=1, B=2, aggrs=["sum", "max"], graph=g_plain)),   # ldb != F_g
        ("grad_lay_egcm", "lay", dict(fin=64, fout=64, H=4, B=4, aggrs=["symadd", "std", "max"], graph=g_messy)),
        ("grad_lay_egcs", "lay", dict(fin=168, fout=168, H=8, B=4, aggrs=["symadd"], graph=g_plain)),
        ("grad_lay_softmax", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["symadd", "max", "mean"], graph=g_messy, softmax=True)),
        ("grad_lay_hardtanh", "lay", dict(fin=32, fout=32, H=4, B=4, aggrs=["add", "min", "var"], graph=g_messy, hardtanh=True)),
        ("grad_lay_ties", "lay", dict(fin=8, fout=16, H=2, B=2, aggrs=["max", "min", "add"], graph=dict(n=20, e=120, dups=30), integer=True)),
        ("grad_lay_hub", "lay", dict(fin=48, fout=64, H=8, B=4, aggrs=["symadd", "max", "mean"], graph=g_hub)),
    ]
    return out


def build(lay, opt, kind, cfg):
    fin, fout, H, B, aggrs = cfg["fin"], cfg["fout"], cfg["H"], cfg["B"], cfg["aggrs"]
    if kind == "lay":
        return lay.EfficientGraphConv(fin, fout, num_heads=H, num_bases=B, softmax_weights=cfg.get("softmax", False),
                                      add_self_loops=cfg.get("add_self_loops", True), bias=True, aggrs=aggrs,
                                      sigmoid_weights=cfg.get("sigmoid", False), hardtanh_weights=cfg.get("hardtanh", False))
    return opt.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B, add_self_loops=cfg.get("add_self_loops", True),
                      bias=True, sigmoid=cfg.get("sigmoid", False))


def run(layer, kind, x, ei_t, gout, dtype):
    layer = layer.to(dtype)
    for p in layer.parameters():
        p.grad = None
    xx = x.to(dtype).clone().requires_grad_(True)
    ARGS.clear()
    out = layer(xx, ei_t) if kind == "opt" else layer(x=xx, edge_index=ei_t)
    out.backward(gout.to(dtype))
    grads = {k: v.grad.detach().clone() for k, v in layer.named_parameters()}
    return out.detach(), xx.grad.detach(), grads, list(ARGS.get("args", []))


def main():
    install()
    lay = mg.load_ref("layers")
    opt = mg.load_ref("optimized_layers")
    manifest = {}
    for idx, (name, kind, cfg) in enumerate(cases()):
        rng = np.random.default_rng(5000 + idx)
        torch.manual_seed(5000 + idx)
        ei, n = mg.make_graph(rng, cfg["graph"])
        integer = cfg.get("integer", False)
        if integer:
            x = torch.from_numpy(rng.integers(-2, 3, size=(n, cfg["fin"])).astype(np.float32))
        else:
            x = torch.from_numpy(rng.standard_normal((n, cfg["fin"])).astype(np.float32))
        gout = torch.from_numpy(rng.standard_normal((n, cfg["fout"])).astype(np.float32))
        layer = build(lay, opt, kind, cfg)
        with torch.no_grad():
            for p in layer.parameters():
                if integer:
                    p.copy_(torch.from_numpy(rng.integers(-1, 2, size=tuple(p.shape)).astype(np.float32)))
            if not integer:
                layer.bias.copy_(torch.from_numpy(rng.standard_normal(cfg["fout"]).astype(np.float32)))
        sd = {f"param:{k}": v.detach().numpy().copy() for k, v in layer.state_dict().items()}
        ei_t = torch.from_numpy(ei)
        out32, gx32, gp32, args32 = run(layer, kind, x, ei_t, gout, torch.float32)
        out64, gx64, gp64, _ = run(layer, kind, x, ei_t, gout, torch.float64)
        layer.float()
        meta = dict(kind=kind, fin=cfg["fin"], fout=cfg["fout"], H=cfg["H"], B=cfg["B"], aggrs=cfg["aggrs"], n=n,
                    softmax=cfg.get("softmax", False), sigmoid=cfg.get("sigmoid", False), hardtanh=cfg.get("hardtanh", False),
                    add_self_loops=cfg.get("add_self_loops", True), bias=True, sparse=False,
                    arg_reduces=[r for r, _ in args32])
        extra = {f"arg:{i}": a.numpy().astype(np.int32) for i, (_, a) in enumerate(args32)}
        np.savez_compressed(
            os.path.join(HERE, f"{name}.npz"), x=x.numpy(), edge_index=ei, gout=gout.numpy(), out=out32.numpy(),
            out64=out64.numpy(), grad_x=gx32.numpy(), grad_x64=gx64.numpy(), meta=json.dumps(meta), **sd, **extra,
            **{f"grad:{k}": v.numpy() for k, v in gp32.items()}, **{f"grad64:{k}": v.numpy() for k, v in gp64.items()})
        d = float((gx32.double() - gx64).abs().max() / max(1.0, float(gx64.abs().max())))
        manifest[name] = dict(meta, n_edges=int(ei.shape[1]), f32_vs_f64_grad_x=d)
        print(f"{name:28s} N={n:5d} E={ei.shape[1]:6d} |grad_x|max={float(gx64.abs().max()):9.3f}  f32 vs f64 grad_x: {d:.2e}")
    with open(os.path.join(HERE, "MANIFEST_GRAD.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
