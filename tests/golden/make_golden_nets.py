#!/usr/bin/env python3
"""Generate tests/golden/net_*.npz and relgrad_*.npz: forward AND backward of the REFERENCE's own CALLERS of the layer.

SURVEY.md 8(c), last row: "a golden forward of one synthetic batch captured through the shimmed import" of the
reference's nets.  This script imports, as they are, from /root/reference/experiments:

    zinc/models.py          EgcZincNet     (Embedding -> 4 x [EfficientGraphConv -> BatchNorm1d -> ReLU -> + x] ->
                                            global_mean_pool -> mlp; zinc/models.py:17-74,92-135)
    arxiv/norm_models.py    EgcArxivNet    (Linear -> 3 x [conv -> bn -> relu -> dropout -> + x] -> Linear ->
                                            log_softmax; arxiv/norm_models.py:13-43,98-131)
    mag/models.py           EGC            (EGConv x 3 on an adj_t, cached=True, [:, :349], log_softmax; mag/models.py:16-69)
    rmag/models.py          REGConv        (gradients this time; the forward fixtures are make_golden_rel.py's)

together with the reference's own layers.py / optimized_layers.py / utils.py (mlp) underneath them.  The absent
third-party packages are the DIFFERENTIABLE shims of make_golden_grad.py (scatter with torch_scatter's first-edge
arg rule, gcn_norm / add_remaining_self_loops as constants) plus ``global_mean_pool`` (sum / clamp(count, 1) over the
batch vector) and a differentiable ``torch_sparse.matmul`` / ``SparseTensor.matmul``.  Every net runs forward (eval
mode with given running statistics, then training mode with batch statistics) and backward, in float32 (the
reference's own precision) and in float64 (what the HIP path is held to); dropout probabilities are 0 (a random mask
cannot be a fixture).  Only vectors are committed: inputs, state dict, upstream gradient, outputs, gradients.

Runs only in the build container (needs /root/reference).  Usage:  python tests/golden/make_golden_nets.py
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_grad as mgg  # noqa: E402
from egc_amd.workloads import heavy_tailed_graph, zinc_like_batch  # noqa: E402  (seeded synthetic inputs)

REF = "/root/reference/experiments"


def global_mean_pool(x, batch, size=None):
    n = int(batch.max()) + 1 if size is None else int(size)
    s = torch.zeros(n, x.size(1), dtype=x.dtype).index_add(0, batch, x)
    cnt = torch.zeros(n, dtype=x.dtype).index_add(0, batch, torch.ones(batch.numel(), dtype=x.dtype))
    return s / cnt.clamp(min=1).view(-1, 1)


def diff_sparse_matmul(adj, x, reduce="sum"):
    """torch_sparse.matmul(adj_t, x, reduce), differentiable w.r.t. x: per-row reduction of value * x[col]."""
    src = x[adj.col]
    if adj.value is not None:
        src = src * adj.value.to(x.dtype).view(-1, 1)
    return mgg.diff_scatter(src, adj.row, 0, None, adj.sizes[0], "sum" if reduce == "add" else reduce)


class _Absent(torch.nn.Module):
    """Baseline convolutions the model files import by name and these fixtures never construct."""

    def __init__(self, *a, **k):
        raise RuntimeError("a baseline layer of torch_geometric: not part of the EGC path, not shimmed")


def install():
    mgg.install()
    tg = sys.modules["torch_geometric.nn"]
    tg.global_mean_pool = global_mean_pool
    tg.global_add_pool = tg.global_max_pool = None
    for name in ("GATConv", "GATv2Conv", "GCNConv", "GINConv", "PNAConv", "SAGEConv", "RGCNConv"):
        setattr(tg, name, _Absent)
    def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False, add_self_loops=True, dtype=None):
        if isinstance(edge_index, mg.SparseTensor):      # adj_t: make_golden.py's restatement (constants, float32 as PyG's)
            return mg.shim_gcn_norm(edge_index, edge_weight, num_nodes, improved, add_self_loops, dtype)
        return mgg.diff_gcn_norm(edge_index, edge_weight, num_nodes, improved, add_self_loops, dtype)
    sys.modules["torch_geometric.nn.conv.gcn_conv"].gcn_norm = gcn_norm
    sys.modules["torch_sparse"].matmul = diff_sparse_matmul
    mg.SparseTensor.matmul = lambda self, x, reduce="sum": diff_sparse_matmul(self, x, reduce)
    # the reference's packages under their own names (its model files say `from experiments.layers import ...`);
    # this repository's `experiments/` re-export package must not be the one that answers
    pkg = types.ModuleType("experiments")
    pkg.__path__ = [REF]
    sys.modules["experiments"] = pkg
    for sub in ("utils", "layers", "optimized_layers"):
        _load(f"experiments.{sub}", os.path.join(REF, f"{sub}.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _randomise(net, rng):
    """Make every term visible: BatchNorm affine + running statistics away from their defaults, layer biases non-zero."""
    with torch.no_grad():
        for name, m in net.named_modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)))
                m.bias.copy_(torch.from_numpy((0.3 * rng.standard_normal(m.bias.shape)).astype(np.float32)))
                m.running_mean.copy_(torch.from_numpy((0.3 * rng.standard_normal(m.bias.shape)).astype(np.float32)))
                m.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2.0, m.bias.shape).astype(np.float32)))
        for name, p in net.named_parameters():
            if name.endswith("bias") and p.dim() == 1 and "comb" not in name and float(p.abs().sum()) == 0.0:
                p.copy_(torch.from_numpy((0.2 * rng.standard_normal(p.shape)).astype(np.float32)))


def _run(net, call, gout, dtype, state):
    """eval forward, then a training forward + backward, from the SAME saved state; returns outputs and gradients."""
    net.load_state_dict(state)
    net = net.to(dtype)
    net.eval()
    for m in net.modules():         # cached=True layers pin the first graph: same graph every time, but reset anyway
        if hasattr(m, "_cached_edge_index"):
            m._cached_edge_index = m._cached_adj_t = None
    with torch.no_grad():
        out_eval = call(net, dtype, False)[0].detach().clone()
    net.train()
    for p in net.parameters():
        p.grad = None
    out, leaf = call(net, dtype, True)
    out.backward(gout.to(dtype))
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
    gx = leaf.grad.detach().clone() if leaf is not None else None
    return out_eval, out.detach().clone(), grads, gx


def _save(name, meta, inputs, state, gout, r32, r64, manifest):
    arrays = {f"in:{k}": v for k, v in inputs.items()}
    arrays.update({f"param:{k}": v.numpy() for k, v in state.items()})
    arrays["gout"] = gout.numpy()
    for tag, (oe, ot, grads, gx) in (("32", r32), ("64", r64)):
        arrays[f"out_eval{tag}"] = oe.numpy()
        arrays[f"out_train{tag}"] = ot.numpy()
        arrays.update({f"grad{tag}:{k}": v.numpy() for k, v in grads.items()})
        if gx is not None:
            arrays[f"grad_x{tag}"] = gx.numpy()
    # the reference's own float32 against its float64: what a float32 implementation of this net can be held to
    # (per parameter, relative to the parameter's own largest gradient -- floored at 1 % of the net's largest one: a
    # Linear / layer bias in front of a BatchNorm on batch statistics has an analytically ZERO gradient, of which
    # float32 keeps 1e-8 of noise)
    dist = {}
    gscale = max(float(v.abs().max()) for v in r64[2].values())
    for k in r64[2]:
        a, b = r32[2][k].double(), r64[2][k]
        dist[k] = float((a - b).abs().max() / max(1e-2 * gscale, float(b.abs().max())))
    d_out = float((r32[1].double() - r64[1]).abs().max() / max(1.0, float(r64[1].abs().max())))
    meta = dict(meta, f32_vs_f64_out_train=d_out, f32_vs_f64_grad_max=max(dist.values()), f32_vs_f64_grad=dist,
                grad_scale=gscale)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
    manifest[name] = {k: v for k, v in meta.items() if k != "f32_vs_f64_grad"}
    print(f"{name:26s} out f32-vs-f64 {d_out:.2e}   worst gradient f32-vs-f64 {max(dist.values()):.2e}")


def zinc_cases(manifest):
    zm = _load("experiments.zinc.models", os.path.join(REF, "zinc", "models.py"))
    for name, hidden, H, B, aggrs, seed in (("net_zinc_egcs", 56, 8, 4, ["symadd"], 11),
                                            ("net_zinc_egcm", 48, 4, 4, ["add", "std", "max"], 12),
                                            ("net_zinc_plumbing", 32, 1, 1, ["add"], 13),      # BASELINE config 1's shape
                                            # four bases of 16 channels: the shape of the one-launch training path of a batch
                                            ("net_zinc_b64", 64, 4, 4, ["symadd", "max", "mean"], 14)):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        atom, ei, n, batch = zinc_like_batch(24, seed=seed)
        net = zm.EgcZincNet(hidden, 4, 0.0, True, readout="mean", heads=H, bases=B, aggrs=aggrs)
        _randomise(net, rng)
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        n_graphs = int(batch.max()) + 1
        gout = torch.from_numpy(rng.standard_normal((n_graphs, 1)).astype(np.float32))
        data = types.SimpleNamespace(x=atom.view(-1, 1), edge_index=ei, batch=batch)

        def call(net, dtype, train):
            return net(data), None
        r32 = _run(net, call, gout, torch.float32, state)
        r64 = _run(net, call, gout, torch.float64, state)
        meta = dict(net="EgcZincNet", ref="zinc/models.py:17-74,92-135", hidden=hidden, layers=4, H=H, B=B, aggrs=aggrs,
                    residual=True, n=n, n_graphs=n_graphs)
        _save(name, meta, dict(atom=atom.numpy(), edge_index=ei.numpy(), batch=batch.numpy()), state, gout, r32, r64, manifest)


def arxiv_cases(manifest):
    am = _load("experiments.arxiv.norm_models", os.path.join(REF, "arxiv", "norm_models.py"))
    for name, hidden, H, B, aggrs, seed in (("net_arxiv_egcs", 48, 8, 4, ["symadd"], 21),
                                            ("net_arxiv_egcm", 64, 4, 4, ["symadd", "max", "mean"], 22)):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        n = 260
        ei = heavy_tailed_graph(n, 900, seed=seed)
        x = torch.from_numpy(rng.standard_normal((n, am.NUM_FEATURES)).astype(np.float32))
        net = am.EgcArxivNet(hidden, 3, 0.0, True, heads=H, bases=B, aggrs=aggrs)
        _randomise(net, rng)
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        gout = torch.from_numpy(rng.standard_normal((n, am.NUM_CLASSES)).astype(np.float32))

        def call(net, dtype, train):
            leaf = x.to(dtype).clone().requires_grad_(train)
            return net(leaf, ei), (leaf if train else None)
        r32 = _run(net, call, gout, torch.float32, state)
        r64 = _run(net, call, gout, torch.float64, state)
        meta = dict(net="EgcArxivNet", ref="arxiv/norm_models.py:13-43,98-131", hidden=hidden, layers=3, H=H, B=B,
                    aggrs=aggrs, residual=True, dropout=0.0, n=n)
        _save(name, meta, dict(x=x.numpy(), edge_index=ei.numpy()), state, gout, r32, r64, manifest)


def mag_cases(manifest):
    mm = _load("experiments.mag.models", os.path.join(REF, "mag", "models.py"))
    for name, hidden, H, B, aggrs, seed in (("net_mag_symnorm", 64, 8, 4, ["symnorm"], 31),
                                            ("net_mag_mean", 32, 4, 4, ["mean"], 32)):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        n = 220
        ei = heavy_tailed_graph(n, 800, seed=seed)            # symmetric, as mag/configs.py:84-85 makes adj_t
        x = torch.from_numpy(rng.standard_normal((n, mm.IN_FEATURES)).astype(np.float32))
        net = mm.EGC(hidden, 3, 0.0, H, B, aggrs)
        _randomise(net, rng)
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        gout = torch.from_numpy(rng.standard_normal((n, mm.OUT_TRUE)).astype(np.float32))
        adj_t = mg.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(n, n))

        def call(net, dtype, train):
            leaf = x.to(dtype).clone().requires_grad_(train)
            return net(leaf, adj_t), (leaf if train else None)
        r32 = _run(net, call, gout, torch.float32, state)
        r64 = _run(net, call, gout, torch.float64, state)
        meta = dict(net="mag EGC", ref="mag/models.py:16-69", hidden=hidden, layers=3, H=H, B=B, aggrs=aggrs, dropout=0.0, n=n,
                    out_true=mm.OUT_TRUE)
        _save(name, meta, dict(x=x.numpy(), edge_index=ei.numpy()), state, gout, r32, r64, manifest)


def regconv_grad_cases(manifest):
    """Gradients of the reference's REGConv (rmag/models.py:75-148) w.r.t. every node type's features and every
    parameter, on the inputs of the forward fixtures rel_*.npz (make_golden_rel.py)."""
    rm = _load("experiments.rmag.models", os.path.join(REF, "rmag", "models.py"))
    for src_name in ("rel_small", "rel_mag_shape"):
        z = np.load(os.path.join(HERE, f"{src_name}.npz"))
        meta = json.loads(bytes(z["meta"]).decode())
        rng = np.random.default_rng(41 + len(src_name))
        conv = rm.REGConv(meta["fin"], meta["fout"], meta["H"], meta["B"])
        conv.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p_")})
        sizes = {k: z[f"x_{k}"].shape[0] for k in meta["node_types"]}
        adj = {}
        for i, key in enumerate(meta["edge_types"]):
            ei = torch.from_numpy(z[f"ei_{i}"])
            adj[tuple(key)] = mg.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(sizes[key[2]], sizes[key[0]]))
        gout = {k: torch.from_numpy(rng.standard_normal((sizes[k], meta["fout"])).astype(np.float32)) for k in sizes}
        res = {}
        for tag, dtype in (("32", torch.float32), ("64", torch.float64)):
            conv = conv.to(dtype)
            for p in conv.parameters():
                p.grad = None
            xs = {k: torch.from_numpy(z[f"x_{k}"]).to(dtype).requires_grad_(True) for k in sizes}
            out = conv(xs, adj)
            sum((out[k] * gout[k].to(dtype)).sum() for k in sizes).backward()
            res[tag] = (out, {k: v.grad.detach().clone() for k, v in xs.items()},
                        {k: v.grad.detach().clone() for k, v in conv.named_parameters()})
        arrays = {}
        worst = 0.0
        for tag, (out, gx, gp) in res.items():
            for k in sizes:
                arrays[f"out{tag}_{k}"] = out[k].detach().numpy()
                arrays[f"grad_x{tag}_{k}"] = gx[k].numpy()
            arrays.update({f"grad{tag}:{k}": v.numpy() for k, v in gp.items()})
        for k, v in res["64"][2].items():
            worst = max(worst, float((res["32"][2][k].double() - v).abs().max() / max(1e-30, float(v.abs().max()))))
        for k in sizes:
            arrays[f"gout_{k}"] = gout[k].numpy()
            worst = max(worst, float((res["32"][1][k].double() - res["64"][1][k]).abs().max()
                                     / max(1e-30, float(res["64"][1][k].abs().max()))))
        m = dict(forward_fixture=src_name, ref="rmag/models.py:75-148", f32_vs_f64_grad_max=worst)
        arrays["meta"] = np.frombuffer(json.dumps(m).encode(), dtype=np.uint8)
        name = src_name.replace("rel_", "relgrad_")
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
        manifest[name] = m
        print(f"{name:26s} worst gradient f32-vs-f64 {worst:.2e}")


def main():
    install()
    manifest = {}
    zinc_cases(manifest)
    arxiv_cases(manifest)
    mag_cases(manifest)
    regconv_grad_cases(manifest)
    with open(os.path.join(HERE, "MANIFEST_NETS.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
