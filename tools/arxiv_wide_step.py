"""Forward + backward of ONE block x + relu(bn(conv(x))) at the reference's ogbn-arxiv net widths (184 / H8 / B4 symadd, 136 / H4 / B4
symadd, max, mean: hyperparameters.md) on the arxiv-shaped graph: which kernels the full-graph training step of those nets runs on
(run under rocprofv3 --kernel-trace --stats; development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
ei, n = wl.arxiv_like(); ei = ei.to(dev)
g = egc_amd.CSRGraph.from_edge_index(ei, n)
for hid, H, B, aggrs in ((184, 8, 4, ["symadd"]), (136, 4, 4, ["symadd", "max", "mean"])):
    torch.manual_seed(0)
    blk = egc_amd.FusedEGCBlock(egc_amd.EfficientGraphConv(hid, hid, H, B, False, aggrs=aggrs), torch.nn.BatchNorm1d(hid)).to(dev).train()
    x = torch.randn(n, hid, device=dev, requires_grad=True); go = torch.randn(n, hid, device=dev)
    def step():
        blk.zero_grad(set_to_none=True); x.grad = None
        blk(x, g).backward(go)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"{hid}/H{H}/B{B} {'+'.join(aggrs)}: block forward + backward {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
