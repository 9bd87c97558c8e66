#!/usr/bin/env python3
"""What the `weightings` round trip costs at config 2 (SURVEY.md 8f rank 3 on full graphs).

Run once per library (EGC_HIP_LIB selects a diagnostic build, tools/build_variant.sh):
  default                          GEMM (bases + weightings) and the aggregate launch as shipped
  -DEGC_DIAG_GEMM_NO_W_STORE       the GEMM with the weightings' stores dropped         -> what the 86.7 MB write costs
  -DEGC_DIAG_GEMM_NO_W             the GEMM computing and writing `bases` only          -> the most a fused form's GEMM can save
  -DEGC_DIAG_W_ROW0                the aggregate with every row reading weightings row 0 -> the most its read can save
Cache states of the aggregate launch: as in the layer (right behind the GEMM), alone back to back, and after a 1 GiB write to
another buffer with `bases` / `weightings` / neither touched again."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd  # noqa: E402
from egc_amd import functional as Fn, workloads as wl  # noqa: E402


def ev_time(fn, iters=50, reps=5, before=None):
    ts = []
    for _ in range(reps):
        tot = 0.0
        if before is None:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            e.synchronize()
            tot = s.elapsed_time(e) / iters
        else:
            n = max(iters // 5, 5)
            for _ in range(n):
                before()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                e.synchronize()
                tot += s.elapsed_time(e)
            tot /= n
        ts.append(tot)
    return sorted(ts)[reps // 2] * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ei, n = wl.arxiv_like(seed=0)
    g = egc_amd.CSRGraph.from_edge_index(ei.to(dev), n).trim_launches()
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    spec = conv._spec_coo
    x = torch.randn(n, 128, device=dev)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)     # 1 GiB
    with torch.no_grad():
        wcat, bcat = conv._packed_weights()
        planes = conv._weight_planes(spec, wcat)
        bases, wts = Fn.egc_basis_transform(g, spec, x, wcat, bcat, planes)

        def gemm():
            Fn.egc_basis_transform(g, spec, x, wcat, bcat, planes)

        def agg():
            Fn.egc_aggregate_combine(g, spec, bases, wts, conv.bias)

        def layer():
            conv(x, g)
        for _ in range(20):
            layer()
        lib = os.environ.get("EGC_HIP_LIB", "default library")
        print(f"[{lib}] config 2: N={n}, E_in={ei.size(1)}")
        print(f"  layer (GEMM + aggregate, module call)        {ev_time(layer):8.1f} us")
        print(f"  GEMM alone, back to back                     {ev_time(gemm):8.1f} us")
        print(f"  aggregate alone, back to back                {ev_time(agg):8.1f} us")
        print(f"  aggregate, single launch, nothing flushed    {ev_time(agg, before=lambda: None):8.1f} us   (event overhead of a single launch included)")
        print(f"  aggregate after a 1 GiB write (all cold)     {ev_time(agg, before=lambda: flush.fill_(1.0)):8.1f} us")
        print(f"  .. after the write, `bases` read again       {ev_time(agg, before=lambda: (flush.fill_(1.0), bases.sum())):8.1f} us   (weightings cold)")
        print(f"  .. after the write, `weightings` read again  {ev_time(agg, before=lambda: (flush.fill_(1.0), wts.sum())):8.1f} us   (bases cold)")
        print(f"  .. after the write, both read again          {ev_time(agg, before=lambda: (flush.fill_(1.0), bases.sum(), wts.sum())):8.1f} us")


if __name__ == "__main__":
    main()
