"""Every layer shape of the reference's trained nets (hyperparameters.md, output/pretrained.txt) on the ogbn-arxiv-shaped
graph: layer time, aggregate-launch time and the rate of its algorithmic bytes (the table of DESIGN.md section 5)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it * 1e3
ei, n = wl.arxiv_like(); ei = ei.to(dev)
g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
e_eff = ei.size(1) + n
# hyperparameters.md of the reference: zinc / CIFAR / molhiv / arxiv / code, EGC-S and EGC-M (+ the 128/H8/B4 EGC-M flavour)
shapes = [(128, 8, 4, ["symadd", "max", "mean"]), (124, 4, 4, ["add", "std", "max"]), (128, 4, 4, ["symadd", "std", "max"]),
          (136, 4, 4, ["symadd", "max", "mean"]), (168, 8, 4, ["symadd"]), (184, 8, 4, ["symadd"]), (224, 4, 4, ["add", "mean", "max"]),
          (296, 8, 4, ["symadd"]), (300, 4, 4, ["symadd", "min", "max"]), (304, 8, 8, ["symadd"])]
for d, H, B, aggrs in shapes:
    layer = egc_amd.EfficientGraphConv(d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs).to(dev).eval()
    x = torch.randn(n, d, device=dev)
    sp = layer._spec
    with torch.no_grad():
        t = ev(lambda: layer(x=x, edge_index=g))
        from egc_amd import functional as F
        wcat = layer._packed_weights()
        bases, wt = F.egc_basis_transform(g, sp, x, wcat, layer.comb_weights.bias, layer._weight_planes(wcat))
        ta = ev(lambda: F.egc_aggregate_combine(g, sp, bases, wt, layer.bias))
    gather = e_eff * sp.ldb * 4
    alg = gather + n * (sp.w_cols + d) * 4
    print(f"{d}/H{H}/B{B} {','.join(aggrs):22s} slots {sp.ldb // 4:3d}  layer {t:7.1f} us  aggregate {ta:7.1f} us  "
          f"aggregate bytes {alg / 1e6:7.1f} MB -> {alg / ta / 1e6:5.2f} TB/s")
