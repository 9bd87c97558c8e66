"""Times forward + backward of ONE north-star EGConv layer on the ogbn-arxiv-shaped graph through autograd (the
workload behind profiles/r01_training_step_kernel_stats.csv and the training numbers of DESIGN.md)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
ei, n = wl.arxiv_like(); ei = ei.to(dev)
g = egc_amd.CSRGraph.from_edge_index(ei, n)
layer = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev)
x = torch.randn(n, 128, device=dev, requires_grad=True)
go = torch.randn(n, 128, device=dev)
def fwd_bwd():
    layer.zero_grad(set_to_none=True)   # as the reference's loops do every step (zinc/configs.py:64-67)
    x.grad = None
    out = layer(x, g)
    out.backward(go)
for _ in range(3): fwd_bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): fwd_bwd()
torch.cuda.synchronize()
print("fwd+bwd ms", (time.perf_counter() - t0) / 10 * 1e3)
with torch.no_grad():
    for _ in range(3): layer(x, g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): layer(x, g)
    torch.cuda.synchronize(); print("fwd only ms (module call)", (time.perf_counter() - t0) / 10 * 1e3)
