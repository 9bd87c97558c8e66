#!/usr/bin/env python3
"""Host time per eval-mode layer call (a graph so small that the GPU is never the bound): the compiled TORCH_LIBRARY
binding against the ctypes path, for the plain layer and for the fused conv -> BatchNorm(eval) -> ReLU -> + x block."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd  # noqa: E402
from egc_amd import _native  # noqa: E402


def wall(fn, iters=150, reps=15):
    """Host time per call: bursts of `iters` calls into an EMPTY queue (a burst short enough that the launches never
    wait for the GPU -- the kernels of one call take ~15 us of GPU time, a call less than that of host time otherwise
    shows up as the GPU's), median over the bursts."""
    for _ in range(200):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        out.append((time.perf_counter() - t0) / iters * 1e6)
    torch.cuda.synchronize()
    return sorted(out)[reps // 2]


def main():
    dev = torch.device("cuda:0")
    n, e = 64, 300
    ei = torch.randint(0, n, (2, e), device=dev)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    bn = torch.nn.BatchNorm1d(128).to(dev).eval()
    block = egc_amd.FusedEGCBlock(conv, bn).eval()
    x = torch.randn(n, 128, device=dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    res = {}
    for mode in ("native", "ctypes"):
        if mode == "ctypes":
            os.environ["EGC_NO_NATIVE_EXT"] = "1"
            _native._TRIED, _native._OPS = False, None
        with torch.no_grad():
            res[mode] = (wall(lambda: conv(x, g)), wall(lambda: block(x, g)), wall(lambda: conv(x, ei)))
    for mode, (a, b, c) in res.items():
        print(f"{mode:7s}: layer call on a CSRGraph {a:6.1f} us   fused block {b:6.1f} us   layer call on a cached edge_index {c:6.1f} us")


if __name__ == "__main__":
    main()
