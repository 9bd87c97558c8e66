"""Experiment (round 6): in the replayed 4-block training step, the weight-gradient launches (x^T d_cat + its reduction: 39 us per molhiv
layer, 19 us per ZINC layer) feed nothing but the parameters' gradients -- run them on a FORKED branch of the hipGraph, beside the next
block's backward chain, joined at the end of the step.  Python Functions path (EGC_NO_NATIVE_TRAIN=1), the weight-gradient helper
wrapped; replay time with and without the fork, gradients compared."""
import os, sys, time
os.environ["EGC_NO_NATIVE_TRAIN"] = "1"
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import functional as F, workloads as wl
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
    x, gout = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
    side = torch.cuda.Stream()
    pending = []
    real = F._weight_grads_into_params
    fork = [False]

    def wrapped(*a, **k):
        if not fork[0]:
            return real(*a, **k)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = real(*a, **k)
        pending.append((a, out))
        return out
    F._weight_grads_into_params = wrapped

    def step():
        g = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n)
        h = x
        for b in blocks:
            h = b(h, g)
        h.backward(gout)
        if fork[0]:
            torch.cuda.current_stream().wait_stream(side)
            pending.clear()

    def wall(fn, it=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
    res = {}
    for mode in (False, True):
        fork[0] = mode
        graphed = egc_amd.GraphedStep(step, params=params)
        t = wall(graphed)
        res[mode] = (t, [p.grad.clone() for p in params])
        del graphed
    same = all(torch.equal(a, b) for a, b in zip(res[False][1], res[True][1]))
    print(f"{name}: one hipGraph, single branch {res[False][0]:.0f} us; weight gradients on a forked branch {res[True][0]:.0f} us; gradients identical: {same}", flush=True)
    F._weight_grads_into_params = real
