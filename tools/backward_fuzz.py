"""A hunt, not a test: random layers (both classes, widths / heads / bases / aggregator lists incl. max and min) on random
graphs with hubs on both sides, rows of 0-5 entries (record overflow), ties and self loops -- the gradients of the record
path (default) against the arg-byte path (EGC_BWD_NO_REC=1) and the all-separate record builder (EGC_BWD_REC_SEPARATE=1).
Prints every case that differs by more than 3e-6 (scale-relative) or raises.  usage: python tools/backward_fuzz.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import egc_amd

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LAY = ["add", "mean", "max", "min", "symadd"]
OPT = ["sum", "mean", "max", "min", "symnorm"]
bad = 0
worst = 0.0
for case in range(n_cases):
    kind = "lay" if rng.random() < 0.5 else "opt"
    H = int(rng.choice([1, 2, 4, 8])); B = int(rng.choice([1, 2, 4, 8]))
    L = int(rng.choice([4, 8, 12, 16, 21, 31, 32, 56, 64]))
    if B * ((L + 3) // 4 * 4) > 256: L = 16
    hidden = H * L
    A = int(rng.integers(1, 5))
    pool = LAY if kind == "lay" else OPT
    names = list(rng.choice(pool, size=A, replace=False))
    if not {"max", "min"} & set(names): names[0] = "max"
    n = int(rng.choice([40, 700, 5000, 30000]))
    e = int(rng.choice([n, 5 * n, 14 * n]))
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    for _ in range(int(rng.integers(0, 4))):                       # destination hubs
        k = min(e, int(rng.choice([65, 70, 257, 300, 3000]))); ei[1, rng.integers(0, e, size=k)] = int(rng.integers(0, n))
    for _ in range(int(rng.integers(0, 3))):                       # source hubs
        k = min(e, int(rng.choice([65, 300, 2000]))); ei[0, rng.integers(0, e, size=k)] = int(rng.integers(0, n))
    if rng.random() < 0.5: ei[0, -(e // 20 + 1):] = ei[1, -(e // 20 + 1):]     # self loops
    keep = (ei[1] < n - n // 5) | (rng.random(e) < 0.1)          # the last rows: few entries (records overflow)
    ei = torch.from_numpy(ei[:, keep]).to(dev)
    asl = bool(rng.random() < 0.8)
    torch.manual_seed(case)
    fin = int(rng.choice([hidden, 16, 48, 128]))
    if kind == "opt":
        conv = egc_amd.EGConv(fin, hidden, aggrs=names, num_heads=H, num_bases=B, add_self_loops=asl).to(dev)
    else:
        fin = hidden
        conv = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=names, add_self_loops=asl).to(dev)
    x0 = torch.randn(n, fin, device=dev)
    if rng.random() < 0.5: x0[torch.from_numpy(rng.integers(1, n, size=n // 3)).to(dev)] = x0[0].clone()
    gout = torch.randn(n, hidden, device=dev)
    desc = (case, kind, fin, hidden, H, B, L, names, asl, n, int(ei.size(1)))

    def grads():
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        out = conv(x, ei) if kind == "opt" else conv(x=x, edge_index=ei)
        out.backward(gout)
        return [x.grad.clone()] + [p.grad.clone() for p in conv.parameters()]
    try:
        for v in ("EGC_BWD_NO_REC", "EGC_BWD_REC_SEPARATE"): os.environ.pop(v, None)
        rec = grads()
        os.environ["EGC_BWD_NO_REC"] = "1"
        ref = grads()
        del os.environ["EGC_BWD_NO_REC"]
        os.environ["EGC_BWD_REC_SEPARATE"] = "1"
        sep = grads()
        del os.environ["EGC_BWD_REC_SEPARATE"]
        torch.cuda.synchronize()
        err = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(rec + sep, ref + ref))
        worst = max(worst, err)
        if not err <= 3e-6:
            bad += 1
            print("MISMATCH", err, desc, flush=True)
    except Exception as ex:
        bad += 1
        print("EXC", repr(ex)[:200], desc, flush=True)
print(f"{n_cases} cases, {bad} bad, largest difference between the paths {worst:.2e}")
