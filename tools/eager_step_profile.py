"""Host-side profile (cProfile) of the eager small-batch training step: 4 x FusedEGCBlock forward + backward on a ZINC-shaped
batch of 128 molecules -- where the host time of the eager step goes (the GPU work is ~0.45 ms, the eager step 1.2-1.6 ms)."""
import cProfile, io, os, pstats, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
from egc_amd.fusion import FusedEGCBlock
dev = torch.device("cuda:0")
_, ei, n, batch = wl.zinc_like_batch()
ei = ei.to(dev)
torch.manual_seed(0)
blocks = nn.ModuleList([FusedEGCBlock(egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4),
                                      nn.BatchNorm1d(128)) for _ in range(4)]).to(dev).train()
g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
x = torch.randn(n, 128, device=dev)
gout = torch.randn(n, 128, device=dev)
params = list(blocks.parameters())
def step():
    for p in params: p.grad = None
    h = x
    for b in blocks: h = b(h, g)
    h.backward(gout)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print(f"eager step {(time.perf_counter() - t0) / 200 * 1e6:.0f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "38")))
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
