"""Eager 4-block training step on the ZINC / molhiv batches handed over as GraphBatch: what the measured host time depends on
(x with / without a gradient, the `ei.add_(0)` that stands for a new batch, warm-up length)."""
import os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
for name, (ei, n, batch) in (("zinc b128", wl.zinc_like_batch(128, seed=0)[1:]), ("molhiv b2048", wl.molecule_batch(2048, seed=0))):
    ei, batch = ei.to(dev), batch.to(dev)
    sizes = torch.bincount(batch)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
    mx = int(sizes.max())
    torch.manual_seed(0)
    blocks = nn.ModuleList([egc_amd.FusedEGCBlock(egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4),
                                                  nn.BatchNorm1d(128)) for _ in range(4)]).to(dev).train()
    params = list(blocks.parameters())
    gout = torch.randn(n, 128, device=dev)
    for xgrad in (True, False):
        for touch in (False, True):
            x = torch.randn(n, 128, device=dev).requires_grad_(xgrad)
            def step():
                for p in params: p.grad = None
                if touch: ei.add_(0)
                g = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n)
                h = x
                for b in blocks: h = b(h, g)
                h.backward(gout)
            for warm, it in ((5, 50), (20, 200)):
                for _ in range(warm): step()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(it): step()
                torch.cuda.synchronize()
                print(f"{name}: x.requires_grad={xgrad} ei.add_(0)={touch} warm {warm} iters {it}: {(time.perf_counter() - t0) / it * 1e6:.0f} us", flush=True)
