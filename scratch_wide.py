import sys, torch
sys.path.insert(0, "/root/repo")
import egc_amd
from egc_amd import workloads as wl, functional as F
dev = torch.device("cuda:0")
ei, n = wl.arxiv_like(); ei = ei.to(dev)
g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
print("long rows, chunks:", g.long_row_stats(), "entries in long rows:", int((g.rowptr[1:] - g.rowptr[:-1])[(g.rowptr[1:] - g.rowptr[:-1]) > 64].sum()), "of", ei.size(1))
for d, H, B, aggrs in [(300, 4, 4, ["symadd", "min", "max"]), (304, 8, 8, ["symadd"])]:
    layer = egc_amd.EfficientGraphConv(d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs).to(dev).eval()
    x = torch.randn(n, d, device=dev)
    sp = layer._spec
    with torch.no_grad():
        wcat = layer._packed_weights()
        bases, wt = F.egc_basis_transform(g, sp, x, wcat, layer.comb_weights.bias, layer._weight_planes(wcat))
        for _ in range(20): F.egc_aggregate_combine(g, sp, bases, wt, layer.bias)
torch.cuda.synchronize()
