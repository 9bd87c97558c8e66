"""Times the basis-transform GEMM alone (egc_basis_transform_packed) with HIP events: config-2 and config-3 row counts
at the north-star width (F_in 128 -> 64 bases + 128 weightings), the ogbn-mag shape with --mag.  EGC_HIP_LIB selects
an experiment build (tools/build_variant.sh); a build with -DEGC_GEMM_STAMPS prints its stamps to stderr."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egc_amd import functional as F, _C
import egc_amd

dev = torch.device("cuda:0")
shapes = [(169343, 128, 128, 8, 4, 4), (52771, 128, 128, 8, 4, 4), (240730, 128, 128, 8, 4, 4)]
if "--mag" in sys.argv:
    shapes = [(736389, 352, 352, 8, 4, 1)]
if "--wide" in sys.argv:     # F_in > 128 layer shapes of the reference's nets on the arxiv-sized graph (hidden, H, B, A)
    shapes = [(169343, h, h, H, B, A) for h, H, B, A in ((136, 4, 4, 3), (184, 8, 4, 1), (224, 4, 4, 3), (296, 8, 4, 1), (300, 4, 4, 3), (304, 8, 8, 1))]
for n, fin, fout, H, B, A in shapes:
    aggrs = ["sum", "mean", "max", "symnorm"][:A] if A > 1 else ["symnorm"]
    if "--wide" in sys.argv:
        conv = egc_amd.EfficientGraphConv(fin, fout, H, B, False, aggrs=["symadd", "max", "mean"][:A]).to(dev).eval()
        spec = conv._spec
    else:
        conv = egc_amd.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=B).to(dev).eval()
        spec = conv._spec_coo
    x = torch.randn(n, fin, device=dev)
    g = egc_amd.CSRGraph.from_edge_index(torch.zeros((2, 1), dtype=torch.long, device=dev), n)
    with torch.no_grad():
        wcat = torch.randn(fin, spec.f_g + spec.w_cols, device=dev)
        bcat = torch.randn(spec.w_cols, device=dev)
    planes = F.pack_weights(spec, wcat)
    for _ in range(5):
        F.egc_basis_transform(g, spec, x, wcat, bcat, planes)
    torch.cuda.synchronize()
    reps = int(os.environ.get("EGC_REPS", "40"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        F.egc_basis_transform(g, spec, x, wcat, bcat, planes)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    nbytes = n * (fin + spec.ldb + spec.w_cols) * 4
    print(f"N={n} F_in={fin} -> {spec.ldb}+{spec.w_cols}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s")
