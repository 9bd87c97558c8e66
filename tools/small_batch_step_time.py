"""The reference's small-batch regime (zinc/configs.py: 128 graphs per step): 4 x [EGConv -> BatchNorm1d(train) -> ReLU ->
+ x], forward + backward, eager against the whole step replayed as ONE hipGraph (egc_amd.GraphedStep); DESIGN.md section 5."""
import sys, os, time, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
from egc_amd.fusion import FusedEGCBlock
dev = torch.device("cuda:0")
only = os.environ.get("EGC_SMALL_ONLY", "")     # "zinc" / "molhiv": one workload (kernel statistics per size)
for name, (ei, n, batch) in (("zinc-like batch 128", wl.zinc_like_batch()[1:]), ("molhiv batch 2048", wl.molecule_batch())):
    if only and only not in name:
        continue
    ei = ei.to(dev)
    torch.manual_seed(0)
    blocks = nn.ModuleList([FusedEGCBlock(egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4),
                                          nn.BatchNorm1d(128)) for _ in range(4)]).to(dev).train()
    g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
    x = torch.randn(n, 128, device=dev)
    gout = torch.randn(n, 128, device=dev)
    params = list(blocks.parameters())
    def step():
        h = x
        for b in blocks: h = b(h, g)
        h.backward(gout)
    def wall(fn, it=100):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
    def eager():
        for p in params: p.grad = None
        step()
    t_eager = wall(eager)
    ref = [p.grad.clone() for p in params]
    # capture: warm up on a side stream, then record forward + backward into static .grad buffers
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            for p in params: p.grad = None
            step()
    torch.cuda.current_stream().wait_stream(s)
    for p in params: p.grad = None
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        step()
    cg.replay(); torch.cuda.synchronize()
    err = max(float((p.grad - r).abs().max() / r.abs().max().clamp_min(1e-30)) for p, r in zip(params, ref))
    t_graph = wall(cg.replay)
    print(f"{name}: N={n}  eager step {t_eager:.0f} us, hipGraph replay {t_graph:.0f} us, grads max rel diff vs eager {err:.2e}", flush=True)
