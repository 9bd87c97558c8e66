#!/usr/bin/env python3
"""Configs 3 / 4 of BASELINE.json (2048 molecules / 2048 superpixel graphs, north-star layer): the per-batch graph
build + layer on the ordinary path (egc_graph_build: five launches; GEMM; aggregate) against the tile path
(egc_batch_plan: one launch; GEMM; agg_tile_kernel with the CSR built in LDS).  HIP-event medians."""
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
    for name, gen, G, mx in (("molhiv b2048", lambda: wl.molecule_batch(2048, seed=0), 2048, 222),
                             ("cifar b2048", lambda: wl.knn_superpixel_batch(2048, seed=0), 2048, 150),
                             ("zinc b128", lambda: wl.zinc_like_batch(128, seed=0)[1:], 128, 37)):
        if only and not name.startswith(only):
            continue
        ei, n, batch = gen()
        ei, batch = ei.to(dev), batch.to(dev)
        ptr = torch.searchsorted(batch, torch.arange(G + 1, device=dev))
        eptr = torch.searchsorted(batch[ei[1]], torch.arange(G + 1, device=dev))   # a PyG batch carries these (collation's slices)
        x = torch.randn(n, 128, device=dev)
        with torch.no_grad():
            g = egc_amd.CSRGraph.from_edge_index(ei, n)
            gb = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n, edge_ptr=eptr)
            ref = conv(x, g)
            out = conv(x, gb)
            gb.check()
            err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1))
            t_build = med(lambda: egc_amd.CSRGraph.from_edge_index(ei, n))
            t_layer = med(lambda: conv(x, g))
            slot = next(iter(gb._plans))
            t_plan = med(lambda: egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n, edge_ptr=eptr).plan(slot))
            t_tile = med(lambda: conv(x, gb))
            def both():
                conv(x, egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=mx, num_nodes=n, edge_ptr=eptr))
            t_both = med(both)
        tiles = next(iter(gb._plans.values()))
        setup = next(iter(gb._setups.values()))
        print(f"{name}: N={n} E={ei.size(1)}  ordinary: build {t_build * 1e3:.1f} us + layer {t_layer * 1e3:.1f} us = "
              f"{(t_build + t_layer) * 1e3:.1f} us | tile path: plan {t_plan * 1e3:.1f} us + layer {t_tile * 1e3:.1f} us, "
              f"plan+layer in one go {t_both * 1e3:.1f} us  ({int(tiles[1][0])} tiles of {tiles[2]} slots, slot {setup[0]}, "
              f"lds nodes {setup[1]}, tmax {setup[2]}, emax {setup[3]})  rel err vs ordinary {err:.1e}")


if __name__ == "__main__":
    main()
