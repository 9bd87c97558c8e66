#!/usr/bin/env python3
"""The reference's own batched layer shapes (run_pretrained.sh:7-48, hyperparameters.md) on the batches they were trained on
(ZINC b128, CIFAR b2048 / b64-like, molhiv b2048): which path serves each (one launch / plan + GEMM + tile kernel / CSR build +
GEMM + aggregate) and what each path costs.  HIP-event medians of module calls on one GraphBatch (same batch) and with a new
GraphBatch per call.  EGC_SHAPES_ONLY=<substring> restricts the shapes (for rocprofv3 passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd  # noqa: E402
from egc_amd import workloads as wl  # noqa: E402


def med(fn, iters=30, reps=5):
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[reps // 2] * 1e3


SHAPES = [  # (name, dataset, hidden, H, B, aggrs)   -- run_pretrained.sh line in the comment
    ("zinc EGC-S", "zinc", 168, 8, 4, ["symadd"]),                    # :7
    ("zinc EGC-M", "zinc", 124, 4, 4, ["add", "std", "max"]),         # :8
    ("cifar EGC-S", "cifar", 168, 8, 4, ["symadd"]),                  # :12
    ("cifar EGC-M", "cifar", 128, 4, 4, ["symadd", "std", "max"]),    # :13
    ("molhiv EGC-S", "molhiv", 296, 8, 4, ["symadd"]),                # :23
    ("molhiv EGC-M", "molhiv", 224, 4, 4, ["add", "mean", "max"]),    # :24
    ("code EGC-S (on the ZINC batch)", "zinc", 304, 8, 8, ["symadd"]),               # :47 (64-row tiles: graphs of at most 64 nodes)
    ("code EGC-M (on the ZINC batch)", "zinc", 300, 4, 4, ["symadd", "min", "max"]),   # :48
    ("north star EGC-M d128", "molhiv", 128, 8, 4, ["symadd", "max", "mean"]),
    # round 6: the code nets on a code-SHAPED batch (egc_amd.workloads.code_like_batch: 128 ASTs of ~125 nodes, up to 250)
    ("code EGC-S (code-shaped batch)", "code", 304, 8, 8, ["symadd"]),               # :47
    ("code EGC-M (code-shaped batch)", "code", 300, 4, 4, ["symadd", "min", "max"]),   # :48
    ("north star EGC-M d128 (code-shaped batch)", "code", 128, 8, 4, ["symadd", "max", "mean"]),
]


def main():
    dev = torch.device("cuda:0")
    only = os.environ.get("EGC_SHAPES_ONLY", "")
    data = {}
    for name, ds, d, H, B, aggrs in SHAPES:
        if only and only not in name:
            continue
        if ds not in data:
            if ds == "zinc":
                _, ei, n, batch = wl.zinc_like_batch(128, seed=0)
                G = 128
            elif ds == "code":
                ei, n, batch = wl.code_like_batch(128, seed=0)
                G = 128
            elif ds == "cifar":
                ei, n, batch = wl.knn_superpixel_batch(2048, seed=0)
                G = 2048
            else:
                ei, n, batch = wl.molecule_batch(2048, seed=0)
                G = 2048
            ei, batch = ei.to(dev), batch.to(dev)
            sizes = torch.bincount(batch, minlength=G)
            ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
            data[ds] = (ei, n, ptr, int(sizes.max()))
        ei, n, ptr, mx = data[ds]
        torch.manual_seed(0)
        layer = egc_amd.EfficientGraphConv(d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs).to(dev).eval()
        x = torch.randn(n, d, device=dev)
        comp = n * d * 8 + int(ei.size(1)) * 16 + ptr.numel() * 8
        line = f"{name:34s} {d}/H{H}/B{B} {','.join(aggrs):14s} N={n} E={ei.size(1)} max graph {mx} compulsory {comp / 1e6:6.1f} MB |"
        with torch.no_grad():
            g = egc_amd.CSRGraph.from_edge_index(ei, n)
            ref = layer(x=x, edge_index=g)
            t_csr = med(lambda: layer(x=x, edge_index=egc_amd.CSRGraph.from_edge_index(ei, n)))
            line += f" csr build+layer {t_csr:7.1f} us |"
            for label, env in (("one-launch", {}), ("tile", {"EGC_NO_FUSED_TILE": "1"})):
                os.environ.update(env)
                try:
                    gb = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n)
                    out = layer(x=x, edge_index=gb)
                    gb.check()
                    ran = [k[-1] if isinstance(k[-1], str) else "tile" for k, v in gb._setups.items() if v]
                    err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1))
                    t_new = med(lambda: layer(x=x, edge_index=egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n)))
                    t_same = med(lambda: layer(x=x, edge_index=gb))
                    line += f" {label}: ran {'/'.join(ran) or 'csr'} new-batch {t_new:7.1f} same {t_same:7.1f} us err {err:.1e} |"
                finally:
                    for k in env:
                        os.environ.pop(k, None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
