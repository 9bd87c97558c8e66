"""The vertex-partitioned workloads of bench.py (imported by it; same command line, same one-line JSON contract).

At --gpus N > 1 the line is STRONG scaling of exactly what north_star names (BASELINE.json):

* headline (`value`, `ms_per_step`, `metric`): BASELINE config 2 -- the SAME ogbn-arxiv-shaped graph, layer and E_eff as
  the N = 1 line (workloads.arxiv_like, EGC-M d=128 H=8 B=4 sum+mean+max+symnorm), vertex-partitioned over the ranks;
* `strong_scaling.config5_mag_homogeneous`: ONE ogbn-mag-shaped homogeneous graph (workloads.mag_like: N = 736,389,
  ~10.8 M symmetrised edges; mag/configs.py:73-88), EGConv 352 -> 352 / H8 / B4 / symnorm (mag/models.py:23-53);
* `strong_scaling.config5_rmag_typed`: the ~21 M-edge typed ogbn-mag graph (workloads.rmag_like: 1.94 M nodes of 4 types,
  42.1 M CSR entries in the 7 relations of rmag/models.py:18-26) through REGConv 128 -> 128 / H8 / B4
  (rmag/models.py:75-148), every node type cut into per-rank ranges, ONE all-to-all-v per layer for all types.

Per step and rank: basis GEMM on the owned rows -> ONE all-to-all-v of the halo rows of `bases` (RCCL over xGMI),
overlapped with the rows that need no halo row -> the remaining rows.  `value` = E_eff of the WHOLE graph / max-over-ranks
step time.  Every workload object carries `t1_ms` (the same layer on the whole graph on rank 0's GPU alone, same run),
the halo statistics, the largest single peer message and a PREDICTED exchange time = that message / 153 GB/s (xGMI is
point-to-point: SURVEY.md 8e), next to the exchange measured alone, so the measured step can be held against the model.

``--workload arxiv | mag | rmag`` runs one of them as the headline; ``arxiv-weak`` keeps round 1's weak-scaling synthetic.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import torch

from bench import AGGRS, BASES, F_IN, F_OUT, HBM_PEAK_GBS, HEADS, METRIC, log, roofline_terms, time_region

MAG_F, MAG_HEADS, MAG_BASES = 352, 8, 4
XGMI_LINK_GBS = 153.0      # one direct link per peer pair, 7 links per GPU


def _mag_args():
    spec = os.environ.get("EGC_BENCH_MAG_COMMUNITIES", "")   # "K:p_in" plants community structure (workloads.mag_like)
    if not spec:
        return 0, 0.0
    k, p = spec.split(":")
    return int(k), float(p)


class _Ctx:
    def __init__(self, args, world, rank, local):
        self.args, self.world, self.rank = args, world, rank
        self.dev = torch.device("cuda", local)
        self.dist = None
        import torch.distributed as dist
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            self.dist = dist
        # the partitioned code path: always at world > 1; at world 1 when bench.py was told to initialise the process group
        # anyway (EGC_BENCH_FORCE_PARTITION=1: the RCCL calls with no peer to talk to -- all a one-GPU box can offer)
        self.part = self.dist is not None
        # tests only: shrink every workload (EGC_BENCH_SCALE=0.05) so that a functional run takes seconds
        self.scale = float(os.environ.get("EGC_BENCH_SCALE", "1") or 1)
        self.overlap = os.environ.get("EGC_BENCH_NO_OVERLAP", "0") in ("", "0")
        self.reorder = os.environ.get("EGC_BENCH_NO_REORDER", "0") in ("", "0")

    def sync_all(self):
        torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return v
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """Every rank's `obj` (small python values) as a list on every rank."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def timed_steps(self, step, steps, warmup):
        """The contract's timed region: `warmup` untimed steps, then EXACTLY `steps`, bracketed by barrier +
        synchronize on both sides, MAX over ranks.  Returns seconds for the `steps` steps."""
        for _ in range(warmup):
            step()
        self.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync_all()
        return self.max_over_ranks(time.perf_counter() - t0)


def _exchange_model(ctx, plan, row_bytes):
    """Halo statistics of all ranks + the xGMI model of the exchange (largest single peer message / link rate)."""
    mine = dict(n_local=plan.n_local, n_halo=plan.n_halo, n_send=int(plan.send_idx.numel()),
                max_peer_rows_recv=max(plan.recv_splits, default=0), max_peer_rows_sent=max(plan.send_splits, default=0),
                n_interior=plan.n_interior)
    allr = ctx.gather(mine)
    peer = max(max(r["max_peer_rows_recv"], r["max_peer_rows_sent"]) for r in allr)
    return dict(rows_owned_per_rank=[r["n_local"] for r in allr], halo_rows_per_rank=[r["n_halo"] for r in allr],
                rows_sent_per_rank=[r["n_send"] for r in allr], interior_rows_per_rank=[r["n_interior"] for r in allr],
                row_bytes=row_bytes, max_peer_rows=peer, max_peer_bytes=peer * row_bytes,
                halo_bytes_per_rank_max=max(r["n_halo"] for r in allr) * row_bytes,
                predicted_exchange_ms=peer * row_bytes / (XGMI_LINK_GBS * 1e9) * 1e3,
                model="largest single peer message / 153 GB/s (one xGMI link per peer pair; SURVEY.md 8e)")


# ---------------------------------------------------------------------------------------------------------------
# homogeneous graphs: config 2 (arxiv) and config 5 (mag), strong scaling
# ---------------------------------------------------------------------------------------------------------------
def strong_homogeneous(ctx, name, conv, f_in, ei_cpu, n_global, desc, steps, warmup):
    import egc_amd
    from egc_amd import _C, partition
    from egc_amd.functional import pack_weights
    lib = _C.load()
    dev, world, rank, dist = ctx.dev, ctx.world, ctx.rank, ctx.dist
    f_out = conv.out_channels
    with torch.no_grad():
        conv.bias.normal_()
    conv = conv.to(dev).eval()
    spec = conv._spec_coo
    wcat, bcat = conv._packed_weights()
    planes = pack_weights(spec, wcat)
    bias = conv.bias.detach()
    ldb = spec.ldb
    has_sym = "symnorm" in conv.aggregators
    e_global = int(ei_cpu.size(1))
    total_e_eff = float(e_global + n_global)      # EGConv convention: every aggregator also traverses one self loop per node

    ei_dev = ei_cpu.to(dev)
    part_info, plan = None, None
    if ctx.part:
        t0 = time.perf_counter()
        naive_bounds = partition.vertex_ranges(n_global, world)
        q_naive = partition.partition_quality(ei_dev, naive_bounds)
        if ctx.reorder:
            order, new_of_old, bounds = partition.agree_on_partition(*partition.locality_partition(ei_dev, n_global, world))
            ei_dev = new_of_old[ei_dev]
            q = partition.partition_quality(ei_dev, bounds)
        else:
            bounds, q = naive_bounds, q_naive
        torch.cuda.synchronize(dev)
        t_part = time.perf_counter() - t0
        owned = partition.local_edges(ei_dev, bounds[rank], bounds[rank + 1])
        ei_local, plan = partition.build_distributed(owned, n_global, interior_first=ctx.overlap, bounds=bounds)
        graph = egc_amd.CSRGraph.from_partition(ei_local, plan, global_max_index=n_global - 1).trim_launches()
        torch.cuda.synchronize(dev)
        part_info = {"reorder": "balanced label propagation (partition.locality_partition)" if ctx.reorder else "none",
                     "renumber_s": t_part, "setup_s": time.perf_counter() - t0,
                     "contiguous_split": {k: q_naive[k] for k in ("cross_edge_frac", "halo_rows_per_rank", "max_peer_rows",
                                                                  "entries_per_rank")},
                     "used": {k: q[k] for k in ("cross_edge_frac", "halo_rows_per_rank", "max_peer_rows",
                                                "entries_per_rank", "rows_per_rank")}}
        n, e_in = plan.n_local, int(ei_local.size(1))
        del owned, ei_local
    else:
        graph = egc_amd.CSRGraph.from_edge_index(ei_dev, n_global, build="sort").trim_launches()
        n, e_in = n_global, e_global
    del ei_dev
    e_eff = e_in + n

    torch.manual_seed(ctx.args.seed + 1 + rank)
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

    n_int = plan.n_interior if (plan is not None) else None

    def step():  # GEMM on owned rows -> halo all-to-all-v (RCCL) || interior rows -> boundary rows
        gemm_only()
        if plan is None:
            agg_only()
        elif n_int is None:
            plan.exchange(bases)
            agg_only()
        else:
            handle = plan.exchange_start(bases)
            agg_rows(0, n_int)
            plan.exchange_finish(handle)
            agg_rows(n_int, n)

    for _ in range(10):
        step()
    ctx.sync_all()
    reps = max(10, min(steps, 50))
    lsync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
    agg_ms = time_region(agg_only, reps, lsync)
    gemm_ms = time_region(gemm_only, reps, lsync)
    kernels = {"basis_gemm": gemm_ms, "aggregate_combine_all_rows": agg_ms}
    exch = None
    if plan is not None:
        if n_int is not None:
            kernels["aggregate_interior_rows"] = time_region(lambda: agg_rows(0, n_int), reps, lsync) if n_int > 0 else 0.0
            kernels["aggregate_boundary_rows"] = time_region(lambda: agg_rows(n_int, n), reps, lsync) if n_int < n else 0.0
        kernels["send_pack"] = time_region(lambda: plan.pack_send(bases), reps, lsync)
        ctx.sync_all()
        kernels["halo_exchange_alone"] = time_region(lambda: plan.exchange(bases), reps, ctx.sync_all)
        exch = _exchange_model(ctx, plan, ldb * 4)
        exch["measured_exchange_alone_ms_rank0"] = kernels["halo_exchange_alone"]
        hidden = kernels.get("aggregate_interior_rows", 0.0)
        exch["predicted_step_ms"] = (gemm_ms + kernels["send_pack"] + max(exch["predicted_exchange_ms"], hidden)
                                     + kernels.get("aggregate_boundary_rows", agg_ms))
        exch["predicted_step_model"] = ("rank 0: GEMM + send pack + max(predicted exchange, interior rows) + boundary rows "
                                        "(RCCL launch latency not modelled)")

    elapsed = ctx.timed_steps(step, steps, warmup)
    ms_per_step = elapsed / steps * 1e3
    value = total_e_eff / (elapsed / steps)

    terms = roofline_terms(n, e_eff, f_in, spec.f_g, f_out, spec.w_cols, symnorm=has_sym)
    agg_bytes = terms["aggregate_launch"]
    agg_gbs = agg_bytes / (agg_ms * 1e-3) / 1e9
    terms_global = roofline_terms(n_global, int(total_e_eff), f_in, spec.f_g, f_out, spec.w_cols, symnorm=has_sym)

    # ---- the same layer unpartitioned on ONE GPU, timed by rank 0 in the same run (strong-scaling baseline) ----
    del bases, weightings, out, x, graph
    torch.cuda.empty_cache()
    t1 = None
    if rank == 0:
        if plan is None:
            t1 = ms_per_step
        else:
            g1 = egc_amd.CSRGraph.from_edge_index(ei_cpu.to(dev), n_global, build="sort").trim_launches()
            x1 = torch.randn(n_global, f_in, device=dev)
            with torch.no_grad():
                for _ in range(5):
                    conv(x1, g1)
                t1 = time_region(lambda: conv(x1, g1), max(10, min(steps, 50)), lsync)
            del g1, x1
            torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
    rec = {
        "workload": desc, "value": value, "unit": "edges/s", "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
        "t1_ms": t1, "speedup_vs_1gpu": (t1 / ms_per_step) if t1 else None,
        "n_nodes_global": n_global, "e_in_global": e_global, "e_eff_total": total_e_eff, "layer": "EGConv",
        "parallelism": "single GPU" if plan is None else
        f"1-D vertex partition x{world}, one halo all-to-all-v per layer" + (", interior rows overlapped" if ctx.overlap else ""),
        "layer_frac_whole_job": terms_global["layer"] / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
        "exchange": exch, "partition": part_info,
        "kernels_ms_rank0": kernels,
        "roofline_rank0": {"bound": "hbm", "kernel": "egc::agg_fast_kernel on rank 0's rows (all rows, halo already present)",
                           "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg_gbs / HBM_PEAK_GBS,
                           "traffic": None, "algorithmic_bytes_per_launch": agg_bytes, "launch_ms": agg_ms},
    }
    if rank == 0:
        log(f"[{name}] x{world}: step {ms_per_step:.4f} ms (t1 {t1 if t1 is None else round(t1, 4)} ms), gemm {gemm_ms:.4f}, "
            f"aggregate {agg_ms:.4f}" + (f", exchange alone {kernels['halo_exchange_alone']:.4f} ms "
                                         f"(predicted {exch['predicted_exchange_ms']:.4f}), halo rows rank0 {plan.n_halo}"
                                         if plan is not None else ""))
    return rec


# ---------------------------------------------------------------------------------------------------------------
# the typed ogbn-mag graph through REGConv, every node type partitioned
# ---------------------------------------------------------------------------------------------------------------
def strong_typed(ctx, steps, warmup, scale=1.0):
    import egc_amd
    from egc_amd import partition
    from egc_amd import workloads as wl
    dev, world, rank, dist = ctx.dev, ctx.world, ctx.rank, ctx.dist
    torch.manual_seed(ctx.args.seed)
    nodes, rel = wl.rmag_like(seed=ctx.args.seed, scale=scale * ctx.scale)
    entries = sum(int(v.shape[1]) for v in rel.values())
    conv = egc_amd.REGConv(F_IN, F_OUT, HEADS, BASES).to(dev).eval()
    node_types = list(conv.node_types)
    lsync = lambda: torch.cuda.synchronize(dev)   # noqa: E731
    t0 = time.perf_counter()
    layout = partition.typed_layout(nodes, rel, world, node_types=node_types)    # same on every rank, no communication
    if ctx.part:
        part = partition.build_typed_distributed({k: v.to(dev) for k, v in rel.items()}, layout)
    else:
        part = partition.build_typed_local_simulation({k: v.to(dev) for k, v in rel.items()}, layout)[0]
    graphs = conv.partition_graphs(part, dev)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t0
    torch.manual_seed(ctx.args.seed + 1 + rank)
    x = {t: torch.randn(part.n_owned(t), F_IN, device=dev) for t in node_types}
    ldb = conv._spec_root.ldb
    table = torch.empty((max(part.n_table, 1), ldb), device=dev)

    def step():
        with torch.no_grad():
            conv.forward_partitioned(x, part, graphs, table=table)

    for _ in range(5):
        step()
    ctx.sync_all()
    exch, kernels = None, {}
    if ctx.part:
        reps = max(5, min(steps, 20))
        kernels["send_pack"] = time_region(lambda: part.plan.pack_send(table), reps, lsync)
        ctx.sync_all()
        kernels["halo_exchange_alone"] = time_region(lambda: part.plan.exchange(table), reps, ctx.sync_all)
        exch = _exchange_model(ctx, part.plan, ldb * 4)
        exch["measured_exchange_alone_ms_rank0"] = kernels["halo_exchange_alone"]
    elapsed = ctx.timed_steps(step, steps, warmup)
    ms_per_step = elapsed / steps * 1e3
    local_entries = ctx.gather(sum(int(e.size(1)) for e in part.rel_edges.values()))
    del graphs, table, x, part
    torch.cuda.empty_cache()
    t1 = None
    if rank == 0:
        if not ctx.part:
            t1 = ms_per_step
        else:
            adj = {}
            for (s, r, d), ei in rel.items():
                ei = ei.to(dev)
                adj[(s, r, d)] = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(nodes[d], nodes[s]))
            x1 = {k: torch.randn(nn_, F_IN, device=dev) for k, nn_ in nodes.items()}
            with torch.no_grad():
                for _ in range(3):
                    conv(x1, adj)
                t1 = time_region(lambda: conv(x1, adj), max(5, min(steps, 20)), lsync)
            del adj, x1
            torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
    rec = {
        "workload": f"ogbn-mag-shaped typed graph ({sum(nodes.values())} nodes of {len(nodes)} types, {entries} CSR entries "
                    f"in {len(rel)} relations; rmag/models.py:18-26), REGConv {F_IN}->{F_OUT} H={HEADS} B={BASES} "
                    "(rmag/models.py:75-148)",
        "value": entries / (ms_per_step * 1e-3), "unit": "edges/s", "ms_per_step": ms_per_step, "steps": steps,
        "warmup": warmup, "t1_ms": t1, "speedup_vs_1gpu": (t1 / ms_per_step) if t1 else None, "entries_total": entries,
        "entries_per_rank": local_entries, "layer": "REGConv",
        "parallelism": "single GPU" if not ctx.part else
        f"every node type cut into {world} cost-balanced ranges; ONE all-to-all-v per layer for all types (shared basis "
        "matrix); root terms overlapped with the exchange",
        "exchange": exch, "partition": {"setup_s": setup_s, "rows_per_type_rank0": {t: layout.owned(t, 0)[1] - layout.owned(t, 0)[0]
                                                                                      for t in node_types}},
        "kernels_ms_rank0": kernels,
    }
    if rank == 0:
        log(f"[rmag typed] x{world}: step {ms_per_step:.4f} ms (t1 {t1 if t1 is None else round(t1, 4)} ms)" +
            (f", exchange alone {kernels['halo_exchange_alone']:.4f} ms (predicted {exch['predicted_exchange_ms']:.4f})"
             if ctx.part else ""))
    return rec


# ---------------------------------------------------------------------------------------------------------------
def _arxiv_weak(ctx):
    """Round 1's weak-scaling synthetic: one arxiv-sized vertex range per rank, 5 % cross edges."""
    import egc_amd
    from egc_amd import _C, partition
    from egc_amd.functional import pack_weights
    from egc_amd import workloads as wl
    lib = _C.load()
    args, dev, world, rank, dist = ctx.args, ctx.dev, ctx.world, ctx.rank, ctx.dist
    conv = egc_amd.EGConv(128, 128, aggrs=AGGRS, num_heads=8, num_bases=4, cached=True)
    ei_cpu, n_global = wl.partitioned_arxiv_like(rank, world, seed=args.seed)
    with torch.no_grad():
        conv.bias.normal_()
    conv = conv.to(dev).eval()
    spec = conv._spec_coo
    wcat, bcat = conv._packed_weights()
    planes = pack_weights(spec, wcat)
    bias = conv.bias.detach()
    ldb = spec.ldb
    ei_local, plan = partition.build_distributed(ei_cpu.to(dev), n_global, interior_first=ctx.overlap)
    graph = egc_amd.CSRGraph.from_partition(ei_local, plan, global_max_index=n_global - 1).trim_launches()
    n, e_in = plan.n_local, int(ei_local.size(1))
    e_eff = e_in + n
    x = torch.randn(n, 128, device=dev)
    bases = torch.empty((graph.n_src_rows, ldb), device=dev)
    weightings = torch.empty((n, spec.w_cols), device=dev)
    out = torch.empty((n, 128), device=dev)
    g = graph.c_struct()
    ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        _C.check(lib.egc_basis_transform_packed(x.data_ptr(), planes.data_ptr(), bcat.data_ptr(), n, 128, spec.f_g,
                                                spec.w_cols, bases.data_ptr(), ldb, weightings.data_ptr(), stream),
                 "egc_basis_transform_packed")
        handle = plan.exchange_start(bases) if world > 1 else None
        n_int = plan.n_interior if (world > 1 and plan.n_interior is not None) else n
        for lo, hi in ((0, n_int), (n_int, n)):
            if lo == n_int and handle is not None:
                plan.exchange_finish(handle)
            if hi > lo:
                _C.check(lib.egc_aggregate_combine_rows_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), ldb,
                                                            weightings.data_ptr(), bias.data_ptr(), out.data_ptr(), lo, hi,
                                                            ws.data_ptr(), ws.numel(), stream), "egc_aggregate_combine_rows_f32")

    elapsed = ctx.timed_steps(step, args.steps, args.warmup)
    tot = torch.tensor([float(e_eff)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
    ms = elapsed / args.steps * 1e3
    return {"metric": METRIC, "value": float(tot.item()) / (elapsed / args.steps), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "gemm": "fp16x2-split", "data": "synthetic",
            "config": {"workload": "one arxiv-sized vertex range per GPU of an N-times larger graph, 5% cross-partition "
                                   "edges, EGC-M d=128 H=8 B=4 sum+mean+max+symnorm", "layer": "EGConv",
                       "parallelism": f"1-D vertex partition x{world}"},
            "roofline": None, "cpu_baseline": None}


def _arxiv(ctx, steps, warmup):
    import egc_amd
    from egc_amd import workloads as wl
    torch.manual_seed(ctx.args.seed)
    conv = egc_amd.EGConv(F_IN, F_OUT, aggrs=AGGRS, num_heads=HEADS, num_bases=BASES, cached=True)
    ei_cpu, n_global = wl.arxiv_like(seed=ctx.args.seed)
    if ctx.scale != 1.0:
        n_global = max(64, int(wl.ARXIV_NODES * ctx.scale))
        ei_cpu = wl.heavy_tailed_graph(n_global, max(64, int(wl.ARXIV_DIRECTED_EDGES * ctx.scale)), ctx.args.seed)
    desc = (f"ogbn-arxiv-shaped full graph: N={n_global}, heavy-tailed symmetrised E_in={ei_cpu.size(1)} (+N self loops), "
            "EGC-M d=128 H=8 B=4 aggrs=sum+mean+max+symnorm, CSR cached -- the graph of the N = 1 line")
    return strong_homogeneous(ctx, "arxiv", conv, F_IN, ei_cpu, n_global, desc, steps, warmup)


def _mag(ctx, steps, warmup):
    import egc_amd
    from egc_amd import workloads as wl
    torch.manual_seed(ctx.args.seed)
    conv = egc_amd.EGConv(MAG_F, MAG_F, aggrs=["symnorm"], num_heads=MAG_HEADS, num_bases=MAG_BASES, cached=True)
    comm_k, p_in = _mag_args()
    ei_cpu, n_global = wl.mag_like(seed=ctx.args.seed, communities=comm_k, p_in=p_in)
    if ctx.scale != 1.0:
        n_global = max(64, int(wl.MAG_NODES * ctx.scale))
        ei_cpu = wl.heavy_tailed_graph(n_global, max(64, int(wl.MAG_DIRECTED_EDGES * ctx.scale)), ctx.args.seed, comm_k, p_in)
    desc = (f"ogbn-mag-shaped homogeneous graph (N={n_global}, E_in={ei_cpu.size(1)} symmetrised heavy-tailed"
            + (f", planted communities K={comm_k} p_in={p_in}" if comm_k else "") +
            "; mag/configs.py:73-88), EGConv 352->352 H=8 B=4 symnorm (mag/models.py:23-53), CSR cached")
    return strong_homogeneous(ctx, "mag", conv, MAG_F, ei_cpu, n_global, desc, steps, warmup)


def run(args, world, rank, local, workload):
    ctx = _Ctx(args, world, rank, local)
    dist = ctx.dist
    if workload == "arxiv-weak":
        line = _arxiv_weak(ctx)
    else:
        side_steps, side_warm = max(5, min(args.steps, 50)), max(2, min(args.warmup, 10))
        nested = {}
        if workload in ("all", "arxiv"):
            head = _arxiv(ctx, args.steps, args.warmup)
            if workload == "all":
                for key, fn in (("config5_mag_homogeneous", lambda: _mag(ctx, side_steps, side_warm)),
                                ("config5_rmag_typed", lambda: strong_typed(ctx, side_steps, side_warm))):
                    try:
                        nested[key] = fn()
                    except Exception as ex:   # noqa: BLE001 -- recorded in the line; the headline stands on its own
                        nested[key] = {"error": repr(ex)[:500]}
                        log(f"[{key}] failed on rank {rank}: {ex!r}")
        elif workload == "mag":
            head = _mag(ctx, args.steps, args.warmup)
        elif workload == "rmag":
            head = strong_typed(ctx, args.steps, args.warmup)
        else:
            raise SystemExit(f"unknown workload {workload}")
        is_homog = head.get("layer") == "EGConv"
        line = {
            "metric": METRIC, "value": head["value"], "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if world > 1 else None, "vs_baseline": None, "dtype": "f32",
            "bench_scale": ctx.scale if ctx.scale != 1.0 else None,
            "gemm": "fp16x2-split (22-bit operands, fp32 accumulate); long-k form for F_in > 128", "data": "synthetic",
            "config": {"workload": head["workload"], "layer": head["layer"], "parallelism": head["parallelism"],
                       "e_eff_total": head.get("e_eff_total", head.get("entries_total"))},
            "roofline": head.get("roofline_rank0") if is_homog else None, "cpu_baseline": None,
            "strong_scaling": dict({("config2_arxiv" if workload in ("all", "arxiv") else workload): head}, **nested),
        }
    if rank == 0:
        import bench
        bench.emit(line)      # compact line on stdout (bounded size), the full record in bench_detail.json + stderr
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
