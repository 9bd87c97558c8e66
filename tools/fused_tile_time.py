#!/usr/bin/env python3
"""Configs 1 / 3 / 4 of BASELINE.json (ZINC batch of 128, 2048 molecules, 2048 superpixel graphs; north-star layer): the
whole layer as ONE launch (egc_layer_forward_batch_fused_f32) against the two-launch tile path (plan + GEMM + agg_tile_kernel)
and the ordinary path (graph build + GEMM + aggregate).  HIP-event medians; every figure includes the construction of the
GraphBatch (a new batch every step), with and without the graphs' edge offsets."""
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
    return sorted(ts)[reps // 2]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    only = os.environ.get("EGC_TILE_ONLY", "")
    for name, gen, G in (("molhiv b2048", lambda: wl.molecule_batch(2048, seed=0), 2048),
                         ("cifar b2048", lambda: wl.knn_superpixel_batch(2048, seed=0), 2048),
                         ("zinc b128", lambda: wl.zinc_like_batch(128, seed=0)[1:], 128)):
        if only and not name.startswith(only):
            continue
        ei, n, batch = gen()
        ei, batch = ei.to(dev), batch.to(dev)
        sizes = torch.bincount(batch, minlength=G)
        mx = int(sizes.max())
        ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
        eptr = torch.searchsorted(batch[ei[1]].contiguous(), torch.arange(G + 1, device=dev))
        x = torch.randn(n, 128, device=dev)
        res = {}
        with torch.no_grad():
            g = egc_amd.CSRGraph.from_edge_index(ei, n)
            ref = conv(x, g)
            for label, env, ep in (("fused, edge offsets given", "0", eptr), ("fused, offsets searched", "0", None),
                                   ("two-launch tile path, offsets given", "1", eptr)):
                os.environ["EGC_NO_FUSED_TILE"] = env
                gb = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n, edge_ptr=ep)
                out = conv(x, gb)
                gb.check()
                err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1))

                def one_go():
                    conv(x, egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n, edge_ptr=ep))
                res[label] = (med(one_go), med(lambda: conv(x, gb)), err)
            os.environ["EGC_NO_FUSED_TILE"] = "0"
            t_build = med(lambda: egc_amd.CSRGraph.from_edge_index(ei, n))
            t_layer = med(lambda: conv(x, g))
        e_eff = int(ei.size(1)) + n
        comp = n * 128 * 4 * 2 + int(ei.size(1)) * 16 + (G + 1) * 16    # x + out + edge list + offsets
        print(f"{name}: N={n} E={ei.size(1)} max graph {mx}; compulsory bytes {comp / 1e6:.1f} MB")
        print(f"   ordinary: build {t_build * 1e3:.1f} + layer {t_layer * 1e3:.1f} = {(t_build + t_layer) * 1e3:.1f} us")
        for k, (a, b, err) in res.items():
            print(f"   {k}: new batch every call {a * 1e3:.1f} us, same batch {b * 1e3:.1f} us  "
                  f"({e_eff / a / 1e6:.2f} G edges/s, {comp / a / 1e9 * 1e3:.0f} GB/s of compulsory bytes)  rel err {err:.1e}")


if __name__ == "__main__":
    main()
