"""Host-side profile (cProfile) of the eager training step on a ZINC-shaped batch of 128 molecules with the batch handed over as an
egc_amd.GraphBatch (a NEW GraphBatch every step, as a loader gives it): where the host time goes once the GPU work is 0.37 ms."""
import cProfile, io, os, pstats, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
from egc_amd.fusion import FusedEGCBlock
dev = torch.device("cuda:0")
_, ei, n, batch = wl.zinc_like_batch(128, seed=0)
ei, batch = ei.to(dev), batch.to(dev)
sizes = torch.bincount(batch, minlength=128)
ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
mx = int(sizes.max())
torch.manual_seed(0)
# EGC_STEP_SHAPE="hidden,H,B,aggr+aggr,self_loops[,lay]": another layer shape, e.g. the reference's ZINC EGC-S layer "168,8,4,symadd,1,lay"
shape = os.environ.get("EGC_STEP_SHAPE", "128,8,4,sum+mean+max+symnorm,1").split(",")
HID, HEADS, BASES, AGGRS, LOOPS = int(shape[0]), int(shape[1]), int(shape[2]), shape[3].split("+"), shape[4] != "0"
def make_conv():
    if len(shape) > 5 and shape[5] == "lay":
        return egc_amd.EfficientGraphConv(HID, HID, HEADS, BASES, False, aggrs=AGGRS, add_self_loops=LOOPS)
    return egc_amd.EGConv(HID, HID, aggrs=AGGRS, num_heads=HEADS, num_bases=BASES, add_self_loops=LOOPS)
blocks = nn.ModuleList([FusedEGCBlock(make_conv(), nn.BatchNorm1d(HID)) for _ in range(4)]).to(dev).train()
x = torch.randn(n, HID, device=dev).requires_grad_(True)
gout = torch.randn(n, HID, device=dev)
params = list(blocks.parameters())
def step():
    for p in params: p.grad = None
    g = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n)
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
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "40")))
print("\n".join(l[:160] for l in s.getvalue().splitlines()))
