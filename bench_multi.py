"""The partitioned workloads of bench.py (imported by it; same command line, same one-line JSON contract).

``mag`` (default at --gpus N > 1) -- BASELINE config 5 as STRONG scaling: ONE ogbn-mag-shaped graph
(workloads.mag_like: N = 736,389, ~10.8 M symmetrised heavy-tailed edges; reference graph mag/configs.py:73-88),
one EGConv 352 -> 352 / H8 / B4 / symnorm layer (mag/models.py:23-53), vertex-partitioned over the ranks after the
locality-improving, work-balanced renumbering of egc_amd.partition.locality_partition (computed on every rank's
GPU from the same edge list, no communication).  Per step and rank: basis GEMM on the owned rows -> ONE
all-to-all-v of the halo rows of `bases` (RCCL over xGMI), overlapped with the aggregation of the interior rows ->
boundary rows.  `value` = E_eff of the whole graph / max-over-ranks step time.  Rank 0 also times the same layer
unpartitioned on its own GPU in the same run (`strong_scaling.t1_ms`), so every line carries its own baseline.

``arxiv-weak`` -- round 1's weak-scaling synthetic (one arxiv-sized vertex range per rank, 5 % cross edges).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import torch

from bench import HBM_PEAK_GBS, METRIC, log, roofline_terms, time_region

MAG_F, MAG_HEADS, MAG_BASES = 352, 8, 4


def _mag_args():
    spec = os.environ.get("EGC_BENCH_MAG_COMMUNITIES", "")   # "K:p_in" plants community structure (workloads.mag_like)
    if not spec:
        return 0, 0.0
    k, p = spec.split(":")
    return int(k), float(p)


def run(args, world, rank, local, workload):
    import egc_amd
    from egc_amd import _C, partition
    from egc_amd.functional import pack_weights
    from egc_amd import workloads as wl
    dist = None
    if world > 1:
        import torch.distributed as dist
    dev = torch.device("cuda", local)
    lib = _C.load()
    torch.manual_seed(args.seed)

    if workload == "mag":
        f_in = f_out = MAG_F
        conv = egc_amd.EGConv(MAG_F, MAG_F, aggrs=["symnorm"], num_heads=MAG_HEADS, num_bases=MAG_BASES, cached=True)
        comm_k, p_in = _mag_args()
        ei_cpu, n_global = wl.mag_like(seed=args.seed, communities=comm_k, p_in=p_in)
        desc = (f"ogbn-mag-shaped homogeneous graph (N={n_global}, E_in={ei_cpu.size(1)} symmetrised heavy-tailed"
                + (f", planted communities K={comm_k} p_in={p_in}" if comm_k else "") +
                "), EGConv 352->352 H=8 B=4 symnorm, CSR cached")
        scaling = "strong"
    else:
        f_in = f_out = 128
        conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4, cached=True)
        ei_cpu, n_global = wl.partitioned_arxiv_like(rank, world, seed=args.seed)
        desc = ("one arxiv-sized vertex range per GPU of an N-times larger graph, 5% cross-partition edges, "
                "EGC-M d=128 H=8 B=4 sum+mean+max+symnorm")
        scaling = "weak"
    with torch.no_grad():
        conv.bias.normal_()
    conv = conv.to(dev).eval()
    spec = conv._spec_coo
    wcat, bcat = conv._packed_weights()
    planes = pack_weights(spec, wcat)
    bias = conv.bias.detach()
    ldb = spec.ldb
    has_sym = "symnorm" in conv.aggregators

    # ---- partition -------------------------------------------------------------------------------
    overlap = os.environ.get("EGC_BENCH_NO_OVERLAP", "0") in ("", "0")
    reorder = os.environ.get("EGC_BENCH_NO_REORDER", "0") in ("", "0")
    part_info = None
    ei_dev = ei_cpu.to(dev)
    if workload == "mag":
        e_global = int(ei_dev.size(1))
        if world > 1:
            t0 = time.perf_counter()
            naive_bounds = partition.vertex_ranges(n_global, world)
            q_naive = partition.partition_quality(ei_dev, naive_bounds)
            if reorder:
                order, new_of_old, bounds = partition.agree_on_partition(
                    *partition.locality_partition(ei_dev, n_global, world))
                ei_dev = new_of_old[ei_dev]
                q = partition.partition_quality(ei_dev, bounds)
            else:
                bounds, q = naive_bounds, q_naive
            torch.cuda.synchronize(dev)
            part_info = {"reorder": "balanced label propagation (partition.locality_partition)" if reorder else "none",
                         "setup_s": time.perf_counter() - t0,
                         "contiguous_split": {k: q_naive[k] for k in ("cross_edge_frac", "halo_rows_per_rank", "max_peer_rows",
                                                                      "entries_per_rank")},
                         "used": {k: q[k] for k in ("cross_edge_frac", "halo_rows_per_rank", "max_peer_rows",
                                                    "entries_per_rank", "rows_per_rank")}}
            owned = partition.local_edges(ei_dev, bounds[rank], bounds[rank + 1])
            ei_local, plan = partition.build_distributed(owned, n_global, interior_first=overlap, bounds=bounds)
            graph = egc_amd.CSRGraph.from_partition(ei_local, plan, global_max_index=n_global - 1).trim_launches()
            n = plan.n_local
            e_in = int(ei_local.size(1))
            halo_stats = dict(plan.stats, halo_over_local=plan.n_halo / max(plan.n_local, 1))
            del owned, ei_local
        else:
            graph = egc_amd.CSRGraph.from_edge_index(ei_dev, n_global).trim_launches()
            n, e_in, halo_stats = n_global, e_global, None
        total_e_eff = float(e_global + n_global)
    else:
        ei_local, plan = partition.build_distributed(ei_dev, n_global, interior_first=overlap)
        graph = egc_amd.CSRGraph.from_partition(ei_local, plan, global_max_index=n_global - 1).trim_launches()
        n, e_in = plan.n_local, int(ei_local.size(1))
        halo_stats = dict(plan.stats, halo_over_local=plan.n_halo / max(plan.n_local, 1))
        total_e_eff = None
    if workload != "mag" or world == 1:
        del ei_dev
    e_eff = e_in + n

    torch.manual_seed(args.seed + 1 + rank)
    x = torch.randn(n, f_in, device=dev)
    bases = torch.empty((graph.n_src_rows, ldb), device=dev)
    weightings = torch.empty((n, spec.w_cols), device=dev)
    out = torch.empty((n, f_out), device=dev)
    g = graph.c_struct()
    ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def gemm_only():
        _C.check(lib.egc_basis_transform_packed(x.data_ptr(), planes.data_ptr(), bcat.data_ptr(), n, f_in, spec.f_g,
                                                spec.w_cols, bases.data_ptr(), ldb, weightings.data_ptr(), stream),
                 "egc_basis_transform_packed")

    def agg_rows(lo, hi):
        _C.check(lib.egc_aggregate_combine_rows_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), ldb,
                                                    weightings.data_ptr(), bias.data_ptr(), out.data_ptr(), lo, hi,
                                                    ws.data_ptr(), ws.numel(), stream), "egc_aggregate_combine_rows_f32")

    def agg_only():
        agg_rows(0, n)

    n_int = graph.halo.n_interior if (world > 1 and graph.halo is not None) else None

    def step():  # GEMM on owned rows -> halo all-to-all-v (RCCL) || interior rows -> boundary rows
        gemm_only()
        if world == 1:
            agg_only()
        elif n_int is None:
            graph.halo.exchange(bases)
            agg_only()
        else:
            handle = graph.halo.exchange_start(bases)
            agg_rows(0, n_int)
            graph.halo.exchange_finish(handle)
            agg_rows(n_int, n)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(20):
        step()
    sync_all()
    reps = max(10, min(args.steps, 50))
    agg_ms = time_region(agg_only, reps, lambda: torch.cuda.synchronize(dev))
    gemm_ms = time_region(gemm_only, reps, lambda: torch.cuda.synchronize(dev))
    exch_ms = None
    if world > 1:
        sync_all()
        exch_ms = time_region(lambda: graph.halo.exchange(bases), reps, sync_all)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if total_e_eff is None:
            tot = torch.tensor([float(e_eff)], device=dev, dtype=torch.float64)
            dist.all_reduce(tot)
            total_e_eff = float(tot.item())
    elif total_e_eff is None:
        total_e_eff = float(e_eff)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_e_eff / (elapsed / args.steps)

    # ---- the same layer unpartitioned on ONE GPU, timed by rank 0 in the same run (strong-scaling baseline) ----
    strong = None
    if workload == "mag" and world > 1:
        if rank == 0:
            del bases, weightings, out, x
            ei_full = ei_cpu.to(dev)
            g1 = egc_amd.CSRGraph.from_edge_index(ei_full, n_global).trim_launches()
            x1 = torch.randn(n_global, f_in, device=dev)
            with torch.no_grad():
                for _ in range(5):
                    conv(x1, g1)
                t1 = time_region(lambda: conv(x1, g1), 20, lambda: torch.cuda.synchronize(dev))
            strong = {"t1_ms": t1, "tN_ms": ms_per_step, "speedup": t1 / ms_per_step,
                      "note": "t1 = the same layer on the whole graph on rank 0's GPU alone, same run"}
        dist.barrier()

    terms = roofline_terms(n, e_eff, f_in, spec.f_g, f_out, spec.w_cols, symnorm=has_sym)
    agg_bytes = terms["aggregate_launch"]
    agg_gbs = agg_bytes / (agg_ms * 1e-3) / 1e9
    result = {
        "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "gemm": ("fp16x2-split, long-k form (egc_gemm_f16x2k.hip; the bf16x3 kernel when an operand leaves fp16's range)"
                 if f_in > 128 else "fp16x2-split"),
        "data": "synthetic",
        "config": {"workload": desc, "n_nodes_global": n_global, "n_nodes_rank0": n, "e_in_rank0": e_in,
                   "e_eff_total": total_e_eff, "layer": "EGConv",
                   "parallelism": "single GPU" if world == 1 else
                   f"1-D vertex partition x{world}, one halo all-to-all-v per layer" + (", interior rows overlapped" if overlap else ""),
                   "halo_rank0": halo_stats, "partition": part_info},
        "roofline": {"bound": "hbm", "kernel": "egc::agg_fast_kernel on rank 0's rows (all rows, halo already present)",
                     "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg_gbs / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": agg_bytes, "launch_ms": agg_ms},
        "kernels_ms_rank0": {"basis_gemm": gemm_ms, "aggregate_combine_all_rows": agg_ms, "halo_exchange_alone": exch_ms},
        "strong_scaling": strong,
    }
    if rank == 0:
        log(f"rank 0: gemm {gemm_ms:.4f} ms, aggregate {agg_ms:.4f} ms, exchange {exch_ms} ms, step {ms_per_step:.4f} ms")
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
