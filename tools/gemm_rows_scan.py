"""The basis-transform GEMM of the reference's wide batched nets (168 / 224 / 296 wide) over a range of row counts, as a hipGraph
of 20 calls (so that host time does not count): time = fixed part + rows x slope says whether a launch is bound by its prologue
or by its per-tile work.  EGC_HIP_LIB selects an experiment build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egc_amd import functional as F
import egc_amd

dev = torch.device("cuda:0")
nets = [(224, 4, 4, ["add", "mean", "max"]), (296, 8, 4, ["symadd"]), (168, 8, 4, ["symadd"]), (304, 8, 8, ["symadd"])]
rows = [int(v) for v in os.environ.get("EGC_SCAN_ROWS", "6596,13192,26385,52771,105542,211084").split(",")]
for hid, H, B, aggrs in nets:
    conv = egc_amd.EfficientGraphConv(hid, hid, H, B, False, aggrs=aggrs).to(dev).eval()
    spec = conv._spec
    wcat = torch.randn(hid, spec.f_g + spec.w_cols, device=dev)
    bcat = torch.randn(spec.w_cols, device=dev)
    planes = F.pack_weights(spec, wcat)
    pts = []
    for n in rows:
        x = torch.randn(n, hid, device=dev)
        g = egc_amd.CSRGraph.from_edge_index(torch.zeros((2, 1), dtype=torch.long, device=dev), n)
        for _ in range(3):
            F.egc_basis_transform(g, spec, x, wcat, bcat, planes)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    F.egc_basis_transform(g, spec, x, wcat, bcat, planes)
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(5): gr.replay()
            e1.record(s); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        pts.append((n, us))
        nbytes = n * (hid + spec.ldb + spec.w_cols) * 4
        print(f"{hid}/H{H}/B{B} -> {spec.ldb}+{spec.w_cols}  N={n}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s", flush=True)
    (n0, t0), (n1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / (n1 - n0)
    print(f"    fixed ~ {t0 - slope * n0:.1f} us, slope {slope * 1e3:.3f} us per 1000 rows ({(hid + spec.ldb + spec.w_cols) * 4 / slope / 1e6:.2f} TB/s marginal)")
