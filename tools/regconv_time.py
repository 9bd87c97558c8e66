"""Relational EGC (REGConv) forward on the ogbn-mag-shaped heterogeneous workload (workloads.rmag_like):
the numbers of DESIGN.md section 7."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
from egc_amd import workloads as wl
dev = torch.device("cuda:0")
nodes, rel = wl.rmag_like()
print({k: v.shape[1] for k, v in rel.items()}, sum(v.shape[1] for v in rel.values()))
adj = {}
t0 = time.perf_counter()
for (s, r, d), ei in rel.items():
    ei = ei.to(dev)
    adj[(s, r, d)] = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(nodes[d], nodes[s]))
torch.cuda.synchronize(); print("CSR builds ms", (time.perf_counter() - t0) * 1e3)
for fin, fout, H, B in ((128, 64, 4, 4), (128, 128, 8, 4)):
    conv = egc_amd.REGConv(fin, fout, H, B).to(dev).eval()
    x = {k: torch.randn(n, fin, device=dev) for k, n in nodes.items()}
    with torch.no_grad():
        for _ in range(3): out = conv(x, adj)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = conv(x, adj)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    e = sum(v.shape[1] for v in rel.values())
    print(f"REGConv {fin}->{fout} H{H} B{B}: {ms:.3f} ms/forward, {e / ms / 1e6:.2f} G edges/s")
