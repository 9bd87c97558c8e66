"""A hunt, not a test: the block x -> x + relu(bn(conv(x))) in training through the compiled nodes (batch_block_train /
csr_block_train) against the Python Functions (EGC_NO_NATIVE_TRAIN=1) at random widths, heads, bases, aggregator lists and
batches -- outputs, d x and every parameter gradient bit for bit, except the bias of a conv in front of the BatchNorm (noise
around zero on both paths: tests/test_native_ext.py::_same_gradient).  usage: python tools/native_node_fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import egc_amd

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LAY = ["add", "mean", "max", "min", "symadd", "std", "var"]
bad = 0
for case in range(n_cases):
    H = int(rng.choice([1, 2, 4, 8])); B = int(rng.choice([1, 2, 4, 8]))
    L = int(rng.choice([4, 8, 16, 21, 28, 31, 32, 37, 38, 56, 64, 75]))
    hidden = H * L
    if hidden > 384 or hidden % 4: continue
    A = int(rng.integers(1, 4))
    names = list(rng.choice(LAY, size=A, replace=False))
    n_graphs = int(rng.choice([3, 60, 400, 1500]))
    sizes = rng.integers(2, int(rng.choice([10, 40, 70])) + 1, size=n_graphs)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    src, dst = [], []
    for gi in range(n_graphs):
        m = int(sizes[gi]); e = int(rng.integers(0, 4 * m + 1))
        src.append(ptr[gi] + rng.integers(0, m, size=e)); dst.append(ptr[gi] + rng.integers(0, m, size=e))
    ei = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64))
    n = int(ptr[-1])
    as_batch = bool(rng.random() < 0.5)
    residual = bool(rng.random() < 0.7)
    x0, go = torch.randn(n, hidden), torch.randn(n, hidden)
    res = {}
    try:
        for mode in ("native", "python"):
            if mode == "python": os.environ["EGC_NO_NATIVE_TRAIN"] = "1"
            else: os.environ.pop("EGC_NO_NATIVE_TRAIN", None)
            torch.manual_seed(case)
            blocks = torch.nn.ModuleList([egc_amd.FusedEGCBlock(egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=names),
                                                                torch.nn.BatchNorm1d(hidden), residual=residual) for _ in range(2)]).to(dev).train()
            x = x0.to(dev).requires_grad_(True)
            g = egc_amd.GraphBatch(ei.to(dev), ptr=torch.from_numpy(ptr).to(dev), max_nodes=int(sizes.max())) if as_batch else ei.to(dev)
            h = x
            for b in blocks: h = b(h, g)
            h.backward(go.to(dev))
            res[mode] = (h.detach().clone(), x.grad.clone(), [(k, p.grad.clone()) for k, p in blocks.named_parameters()], h.grad_fn.name())
    finally:
        os.environ.pop("EGC_NO_NATIVE_TRAIN", None)
    ok = torch.equal(res["native"][0], res["python"][0]) and torch.equal(res["native"][1], res["python"][1])
    for (k, a), (_, b) in zip(res["native"][2], res["python"][2]):
        if k.endswith("conv.bias"):
            noise = 4e-6 * n ** 0.5 * float(go.abs().max()) * 4
            ok = ok and float(a.abs().max()) <= noise and float(b.abs().max()) <= noise
        else:
            ok = ok and torch.equal(a, b)
    if not ok or "BlockTrainFn" not in res["native"][3]:
        bad += 1
        print(f"case {case}: hidden {hidden} H{H} B{B} {names} graphs {n_graphs} N {n} batch {as_batch} residual {residual}: node {res['native'][3]} DIFFERS" if not ok else
              f"case {case}: hidden {hidden} H{H} B{B} {names}: python path taken ({res['native'][3]})", flush=True)
print(f"{n_cases} cases, {bad} reported")
