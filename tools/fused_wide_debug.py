import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd
dev = torch.device("cuda:0")

def ring_graphs(sizes):
    src, dst, off = [], [], 0
    for s in sizes:
        for i in range(s):
            for d in (1, 2, 5):
                j = (i + d) % s
                src += [off + i, off + j]; dst += [off + j, off + i]
        off += s
    ei = torch.tensor([src, dst], dtype=torch.long)
    # group by graph (already), keep order
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.long)
    return ei, off, ptr

def run(sizes, hidden, H, B, aggrs, grid=None, maxn=None):
    if grid: os.environ["EGC_FT_GRID"] = str(grid)
    else: os.environ.pop("EGC_FT_GRID", None)
    ei, n, ptr = ring_graphs(sizes)
    torch.manual_seed(0)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs).to(dev).eval()
    x = torch.randn(n, hidden, device=dev)
    with torch.no_grad():
        ref = conv(x=x, edge_index=ei.to(dev))
        gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=maxn or max(sizes), num_nodes=n)
        out = conv(x=x, edge_index=gb)
    try:
        gb.check()
    except Exception as ex:
        print("   check:", ex)
    fused = any(isinstance(k, tuple) and k[-1] == "fused" and v for k, v in gb._setups.items())
    err = (out - ref).abs().max(dim=1).values / ref.abs().max()
    bad = (err > 1e-5).nonzero().flatten().cpu().numpy()
    print(f"sizes={sizes[:6]}{'...' if len(sizes)>6 else ''} hidden={hidden} grid={grid} fused={fused} max err {float(err.max()):.2e} bad rows {len(bad)} of {n}: {bad[:12]} .. {bad[-6:] if len(bad) else ''}")

for sizes, grid in (([20], None), ([40], None), ([70], None), ([100], None), ([150], None), ([20, 20], 1), ([20] * 8, 1), ([40, 40, 40], 1),
                    ([70, 70], 1), ([20] * 40, 1), ([30] * 64, 4)):
    run(sizes, 168, 8, 4, ["symadd"], grid)
run([100], 224, 4, 4, ["add", "mean", "max"], None, 96)
run([90], 224, 4, 4, ["add", "mean", "max"], None)
run([60, 60], 296, 8, 4, ["symadd"], 1)
run([20], 304, 8, 8, ["symadd"], None)
run([60, 50, 40, 64], 304, 8, 8, ["symadd"], 1)
run([20], 300, 4, 4, ["symadd", "min", "max"], None)
run([60, 50, 40, 64] * 4, 300, 4, 4, ["symadd", "min", "max"], 2)
