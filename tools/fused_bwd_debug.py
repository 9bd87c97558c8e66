"""The one-launch backward of the batch path (egc_layer_backward_batch_fused_f32) against the CSR path's backward on the same
batch: gradients w.r.t. x and every parameter, layer by layer shape (development aid; the parity tests are tests/test_fused_bwd_gpu.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_batch_tile_gpu import _messy_batch
dev = torch.device("cuda:0")


def ring_graphs(sizes):
    src, dst, off = [], [], 0
    for s in sizes:
        for i in range(s):
            for d in (1, 2, 5):
                j = (i + d) % s
                src += [off + i, off + j]; dst += [off + j, off + i]
        off += s
    return torch.tensor([src, dst], dtype=torch.long), off, torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.long)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(name, make, ei, n, ptr, mx, fin):
    torch.manual_seed(0)
    conv = make().to(dev).train()
    x0 = torch.randn(n, fin, device=dev)
    go = torch.randn(n, conv.out_channels, device=dev)
    res = {}
    for path in ("csr", "fused"):
        conv.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        g = ei.to(dev) if path == "csr" else egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=mx, num_nodes=n)
        out = conv(x, g) if isinstance(conv, egc_amd.EGConv) else conv(x=x, edge_index=g)
        out.backward(go)
        if path == "fused":
            g.check()
            ran = [k[-1] for k, v in g._setups.items() if isinstance(k, tuple) and isinstance(k[-1], str) and v]
        res[path] = (out.detach(), x.grad.detach(), {k: v.grad.detach().clone() for k, v in conv.named_parameters()})
    o = rel(res["fused"][0], res["csr"][0]); dx = rel(res["fused"][1], res["csr"][1])
    bad = ((res["fused"][1] - res["csr"][1]).abs().max(dim=1).values / res["csr"][1].abs().max() > 1e-5).nonzero().flatten().cpu().numpy()
    gp = {k: rel(res["fused"][2][k], res["csr"][2][k]) for k in res["csr"][2]}
    print(f"{name}: N={n} ran={ran} out {o:.1e} dx {dx:.1e} bad dx rows {len(bad)} {bad[:8]} params " + " ".join(f"{k}:{v:.1e}" for k, v in gp.items()), flush=True)


ns = lambda: egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4)
lay = lambda: egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=["symadd", "max", "mean"])
egs = lambda: egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=["symadd"])
d64 = lambda: egc_amd.EGConv(64, 64, aggrs=["sum", "max"], num_heads=4, num_bases=4)
for sizes in ([20], [40], [70], [20, 20, 30], [30] * 40):
    ei, n, ptr = ring_graphs(sizes)
    os.environ["EGC_FT_GRID"] = "1" if len(sizes) <= 3 else "4"
    run(f"north star rings {sizes[:4]}", ns, ei, n, ptr, max(sizes), 128)
os.environ.pop("EGC_FT_GRID", None)
ei, n, ptr = _messy_batch(7, max_size=80)
for name, mk, fin in (("north star messy", ns, 128), ("EGC-M layers.py messy", lay, 128), ("EGC-S layers.py messy", egs, 128), ("d64 sum+max messy", d64, 64)):
    run(name, mk, ei, n, ptr, 80, fin)

# ---- a stack of blocks (conv -> BatchNorm(train) -> ReLU -> + x), as the nets have it
import torch.nn as nn
from egc_amd import workloads as wl
for nb in (1, 2, 4):
    ei, n, batch = wl.molecule_batch(256, seed=0)
    sizes = torch.bincount(batch, minlength=256)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
    torch.manual_seed(0)
    blocks = nn.ModuleList([egc_amd.FusedEGCBlock(ns(), nn.BatchNorm1d(128)) for _ in range(nb)]).to(dev).train()
    params = list(blocks.parameters())
    x0 = torch.randn(n, 128, device=dev)
    go = torch.randn(n, 128, device=dev)
    res = {}
    for path in ("csr", "fused"):
        for p in params: p.grad = None
        g = ei.to(dev) if path == "csr" else egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=int(sizes.max()), num_nodes=n)
        h = x0
        hs = []
        for b in blocks:
            h = b(h, g); hs.append(h.detach())
        h.backward(go)
        if path == "fused": g.check()
        res[path] = (hs, [p.grad.clone() for p in params])
    print(f"{nb} blocks: outputs per block " + " ".join(f"{rel(a, b):.1e}" for a, b in zip(res['fused'][0], res['csr'][0])) +
          " | param grads " + " ".join(f"{rel(a, b):.1e}" for a, b in zip(res['fused'][1], res['csr'][1])), flush=True)

# ---- full molhiv batch, one layer: where do the gradients differ?
ei, n, batch = wl.molecule_batch(2048, seed=0)
sizes = torch.bincount(batch, minlength=2048)
ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
run("north star molhiv b2048", ns, ei, n, ptr, int(sizes.max()), 128)
_, ei, n, batch = wl.zinc_like_batch(128, seed=0)
sizes = torch.bincount(batch, minlength=128)
ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
run("north star zinc b128", ns, ei, n, ptr, int(sizes.max()), 128)
