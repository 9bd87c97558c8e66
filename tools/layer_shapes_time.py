"""Event-timed layer forward (and per-batch COO->CSR build) for the shapes of BASELINE.json's configs 2-5 and the
reference's trained nets: the table of DESIGN.md section 5."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
def ev_time(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3
def run(name, ei, n, layer, f_in):
    ei = ei.to(dev); x = torch.randn(n, f_in, device=dev); layer = layer.to(dev).eval()
    t_csr = ev_time(lambda: egc_amd.CSRGraph.from_edge_index(ei, n), 10, 2)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    kw = (lambda: layer(x, g)) if isinstance(layer, egc_amd.EGConv) else (lambda: layer(x=x, edge_index=g))
    with torch.no_grad():
        t_layer = ev_time(kw)
        os.environ["EGC_FORCE_GENERIC"] = "1"
        t_generic = ev_time(kw)
        del os.environ["EGC_FORCE_GENERIC"]
    e = ei.size(1)
    print(f"{name:34s} N={n:7d} E={e:8d}  csr {t_csr:7.1f} us  layer {t_layer:7.1f} us  (generic agg path: {t_generic:7.1f})  "
          f"{(e + n) / t_layer * 1e-3:6.2f} G edges/s")
torch.manual_seed(0)
ei, n, _ = wl.molecule_batch(2048)
run("C3 molhiv b2048 EGC-M 128/8/4/4", ei, n, egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]), 128)
run("C3 molhiv b2048 224/H4/B4 add,mean,max", ei, n, egc_amd.EfficientGraphConv(224, 224, num_heads=4, num_bases=4, softmax_weights=False, aggrs=["add", "mean", "max"]), 224)
ei, n, _ = wl.knn_superpixel_batch(2048)
run("C4 cifar b2048 EGC-M 128/8/4/4", ei, n, egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]), 128)
ei64, n64, _ = wl.knn_superpixel_batch(64)
run("C4 cifar b64 EGC-M 128/8/4/4", ei64, n64, egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]), 128)
ei, n = wl.mag_like()
run("C5 mag 352/H8/B4 symnorm", ei, n, egc_amd.EGConv(352, 352, aggrs=["symnorm"], num_heads=8, num_bases=4), 352)
run("C5 mag 352/H8/B4 mean", ei, n, egc_amd.EGConv(352, 352, aggrs=["mean"], num_heads=8, num_bases=4), 352)
run("C5 mag first layer 128->352 symnorm", ei, n, egc_amd.EGConv(128, 352, aggrs=["symnorm"], num_heads=8, num_bases=4), 128)
ei, n = wl.arxiv_like()
run("C2 arxiv EGC-S 184/H8/B4 symadd", ei, n, egc_amd.EfficientGraphConv(184, 184, num_heads=8, num_bases=4, softmax_weights=False, aggrs=["symadd"]), 184)
run("C2 arxiv EGC-M 136/H4/B4 symadd,max,mean", ei, n, egc_amd.EfficientGraphConv(136, 136, num_heads=4, num_bases=4, softmax_weights=False, aggrs=["symadd", "max", "mean"]), 136)
run("C2 arxiv north star", ei, n, egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]), 128)
