"""Training step of the reference's batched nets' core -- 4 x [EGConv -> BatchNorm1d(train) -> ReLU -> + x], forward + backward -- on a
ZINC-shaped batch of 128 and a molhiv-shaped batch of 2048 graphs, with the layer taking (a) the COO edge list (per-batch graph
build + GEMM + aggregate, three backward kernels + dense gradients) and (b) an egc_amd.GraphBatch (one launch each way:
egc_layer_forward_batch_fused_f32 / egc_layer_backward_batch_fused_f32); eager and as ONE hipGraph.  EGC_SMALL_ONLY=zinc|molhiv|cifar (the
CIFAR10-superpixel-shaped batch of 2048 graphs, 241 k nodes, runs only on request).
EGC_STEP_SHAPE="hidden,H,B,aggr+aggr+...,self_loops" runs the blocks at another layer shape, e.g. the reference's own molhiv net
"224,4,4,sum+mean+max,0" (run_pretrained.sh:24) or "296,8,4,symnorm,1" (:23); a sixth field "lay" builds experiments/layers.py's EfficientGraphConv instead of EGConv:
"224,4,4,add+mean+max,1,lay" is the reference's molhiv EGC-M layer as its net constructs it."""
import os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
only = os.environ.get("EGC_SMALL_ONLY", "")
shape = os.environ.get("EGC_STEP_SHAPE", "128,8,4,sum+mean+max+symnorm,1").split(",")
HID, HEADS, BASES, AGGRS, LOOPS = int(shape[0]), int(shape[1]), int(shape[2]), shape[3].split("+"), shape[4] != "0"
KIND = shape[5] if len(shape) > 5 else "opt"      # "lay": experiments/layers.py's EfficientGraphConv (aggregator names add / symadd / ...), as the reference's batched nets use it


def make_conv():
    if KIND == "lay":
        return egc_amd.EfficientGraphConv(HID, HID, HEADS, BASES, False, aggrs=AGGRS, add_self_loops=LOOPS)
    return egc_amd.EGConv(HID, HID, aggrs=AGGRS, num_heads=HEADS, num_bases=BASES, add_self_loops=LOOPS)
for name, gen, G in (("zinc b128", lambda: wl.zinc_like_batch(128, seed=0)[1:], 128), ("molhiv b2048", lambda: wl.molecule_batch(2048, seed=0), 2048),
                     ("cifar b2048", lambda: wl.knn_superpixel_batch(2048, seed=0), 2048)):     # (cifar: only on request -- EGC_SMALL_ONLY=cifar)
    if (only and only not in name) or (not only and name.startswith("cifar")):
        continue
    ei, n, batch = gen()
    ei, batch = ei.to(dev), batch.to(dev)
    sizes = torch.bincount(batch, minlength=G)
    mx = int(sizes.max())
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
    torch.manual_seed(0)
    blocks = nn.ModuleList([egc_amd.FusedEGCBlock(make_conv(),
                                                  nn.BatchNorm1d(HID)) for _ in range(4)]).to(dev).train()
    params = list(blocks.parameters())
    x = torch.randn(n, HID, device=dev)
    gout = torch.randn(n, HID, device=dev)
    grads = {}
    for label, graph_of in (("COO (CSR path)", lambda: ei), ("GraphBatch (one launch each way)", lambda: egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n))):
        def step():
            g = graph_of()
            h = x
            for b in blocks:
                h = b(h, g)
            h.backward(gout)
        def eager():
            for p in params: p.grad = None
            ei.add_(0)
            step()
        def wall(fn, it=50):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(it): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
        t_eager = wall(eager)
        grads[label] = [p.grad.clone() for p in params]
        graphed = egc_amd.GraphedStep(step, params=params)
        t_graph = wall(graphed)
        del graphed
        print(f"{name} N={n}: {label}: eager step {t_eager:.0f} us, hipGraph replay {t_graph:.0f} us", flush=True)
    a, b = list(grads.values())
    scale = max(float(q.abs().max()) for q in a)     # (one scale for all: the conv bias' true gradient is 0 in front of a BatchNorm)
    print(f"   parameter gradients, GraphBatch vs COO path: max abs diff / largest gradient {max(float((p - q).abs().max()) for p, q in zip(b, a)) / scale:.2e}", flush=True)
