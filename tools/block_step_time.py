"""Times one training step (forward + backward) of the reference nets' block  conv -> BatchNorm1d -> ReLU -> + input
(zinc/models.py:66-72) on the ogbn-arxiv-shaped graph: FusedEGCBlock's two-pass tail (egc_tail.hip) against the same
block with PyTorch's separate operators (the numbers of DESIGN.md section 5)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
ei, n = wl.arxiv_like(); ei = ei.to(dev)
g = egc_amd.CSRGraph.from_edge_index(ei, n)
conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev)
bn = torch.nn.BatchNorm1d(128).to(dev)
block = egc_amd.FusedEGCBlock(conv, bn).train()
x = torch.randn(n, 128, device=dev, requires_grad=True)
go = torch.randn(n, 128, device=dev)
def step(fn):
    block.zero_grad(set_to_none=True); x.grad = None
    fn(x, g).backward(go)
def ev(fn, it=20):
    for _ in range(5): step(fn)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): step(fn)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
print("block training step, fused tail   ms %.3f" % ev(block))
print("block training step, torch tail   ms %.3f" % ev(block._plain))
def conv_only(x, g): return conv(x, g)
print("conv alone (no tail)              ms %.3f" % ev(conv_only))
# the ogbn-arxiv net's own block (arxiv/norm_models.py:34-40): ... -> ReLU -> F.dropout(p = 0.2) -> + input
block = egc_amd.FusedEGCBlock(conv, bn, dropout=0.2).train()
print("arxiv block (dropout 0.2), fused   ms %.3f" % ev(block))
print("arxiv block (dropout 0.2), torch   ms %.3f" % ev(block._plain))
