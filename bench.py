#!/usr/bin/env python3
"""Benchmark of the one hot path: EGC-M layer forward on an ogbn-arxiv-shaped graph (BASELINE
config 2: N = 169,343, ~2.33 M symmetrised edges + N self loops, d = 128, H = 8, B = 4,
aggregators sum+mean+max+symnorm, fp32).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full layer forward through the C ABI (basis GEMM + fused aggregate/combine) with every input
already resident in HBM and the CSR pre-built (static graph, the reference's ``cached=True``).  Rank 0 prints
ONE JSON line.  For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL)
and the line is STRONG scaling of the same config-2 graph, vertex-partitioned over the ranks (so the N = 1 point of
the curve is this file's N = 1 line), with BASELINE config 5 -- the homogeneous ogbn-mag shape and the ~21 M-edge
typed graph through a partitioned REGConv -- nested under ``strong_scaling`` (bench_multi.py).

Extra objects on the JSON line:
  roofline      -- the dominant kernel (fused aggregate+combine launch): SURVEY.md 8(d) algorithmic bytes of
                   that launch (gather + col + rowptr + deg + out; `weightings` are NOT algorithmic) / its
                   HIP-event time, against the 8 TB/s HBM peak.
  cpu_baseline  -- the oracle's multi-threaded CPU port of the reference op sequence
                   (oracle/egc_cpu_port.py), timed on this host's cores on the same workload.
  other_configs -- measured in this run: configs 3 / 4 (per-batch CSR build reported separately), config 5
                   (homogeneous ogbn-mag shape) and the training step of config 2.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling of the same guide (SURVEY.md 8d asks for both)
AGGRS = ["sum", "mean", "max", "symnorm"]
F_IN = F_OUT = 128
HEADS, BASES = 8, 4
METRIC = "EGC-M layer fwd edges/sec on ogbn-arxiv; achieved HBM GB/s vs roofline"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("EGC_BENCH_FORCE_PARTITION", "0") not in ("", "0") and "RANK" in os.environ
    if world > 1 or force:   # (force: the partitioned code path with RCCL at world size 1, tests/test_bench_multi_gpu.py)
        import torch.distributed as dist
        backend = os.environ.get("EGC_BENCH_BACKEND", "nccl")  # "gloo": functional smoke test on a shared GPU
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def spawn_ranks(n_gpus, argv):
    """``python bench.py --gpus N`` with N > 1 and no RANK in the environment: start the N ranks ourselves, as children
    (``python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>``), relay rank 0's ONE JSON line and
    exit with the launcher's code.  Called before anything in this process has touched the GPU (no torch.cuda call has
    run yet); this process never initialises it -- it only waits for the child."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", str(port)),
           os.path.abspath(__file__)] + list(argv)
    log("bench.py: starting", n_gpus, "ranks:", " ".join(cmd))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
    lines = []
    for ln in proc.stdout:            # ranks other than 0 print nothing on stdout; stderr goes straight through
        if ln.startswith("{"):
            lines.append(ln.rstrip("\n"))
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    for ln in lines:
        print(ln, flush=True)
    if rc == 0 and len(lines) != 1:
        log(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}")
        rc = 1
    raise SystemExit(rc)


def time_region(fn, iters, sync=None):
    """HIP-event timing of `iters` back-to-back calls on the current stream (the stream the library launches
    on: torch's current stream is what is handed to the C ABI); returns ms per call."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    (sync or torch.cuda.synchronize)()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def time_region_median(fn, iters, repeats=5):
    """Median of `repeats` time_region measurements: the side figures (other_configs) are short, partly host-bound
    regions, and one stall of the shared host inside a single region (seen: 80 ms in a 10-call region) would
    otherwise be reported as the figure.  The headline (K timed steps) is not touched by this."""
    return sorted(time_region(fn, iters) for _ in range(repeats))[repeats // 2]


def cpu_baseline(ei, n, x, conv_state, runs=5):
    """Time the CPU port (oracle) on the same workload: full config-2 layer forward, graph prep cached.
    torch's scatter kernels do not scale to hundreds of threads, so a few thread counts are tried and the
    best one is reported (with the number of threads it used)."""
    from oracle.egc_cpu_port import egconv_forward_cpu
    ncpu = os.cpu_count() or 1
    args = (x, ei, conv_state["bases_weight"], conv_state["comb_weight.weight"], conv_state["comb_weight.bias"],
            conv_state["bias"], HEADS, BASES, AGGRS)
    torch.set_num_threads(min(16, ncpu))
    out, cached = egconv_forward_cpu(*args)  # warm-up; also builds the cached gcn_norm edge set
    best = None
    for threads in sorted({min(t, ncpu) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(threads)
        egconv_forward_cpu(*args, cached=cached)
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            out, _ = egconv_forward_cpu(*args, cached=cached)
            times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        log(f"  cpu port, {threads:3d} threads: {med * 1e3:8.1f} ms/forward (median of {runs})")
        if best is None or med < best[0]:
            best = (med, threads)
    return out, best[0], best[1], int(cached[0].size(1))


def cpu_leg(conv_cpu_state, x_cpu, ei_cpu, H, B, aggrs, e_eff, budget_s=12.0, max_runs=5, add_self_loops=True):
    """The CPU port on one of the side configs (same inputs as the GPU run): median of up to `max_runs` forwards within about
    `budget_s` seconds, at the thread count the headline's baseline settled on (8: torch's scatter kernels do not scale
    further).  Returns the cpu_baseline object and the port's output."""
    from oracle.egc_cpu_port import egconv_forward_cpu
    st = conv_cpu_state
    threads = min(8, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    args = (x_cpu, ei_cpu, st["bases_weight"], st["comb_weight.weight"], st["comb_weight.bias"], st["bias"], H, B, aggrs, add_self_loops)
    t0 = time.perf_counter()
    out, cached = egconv_forward_cpu(*args)                  # warm-up; builds the cached gcn_norm edge set
    first = time.perf_counter() - t0
    times = []
    while len(times) < max_runs and (not times or sum(times) + first < budget_s):
        t0 = time.perf_counter()
        out, _ = egconv_forward_cpu(*args, cached=cached)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": e_eff / med, "unit": "edges/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"the full workload of this record, gcn_norm cached; median of {len(times)} forwards of {med * 1e3:.1f} ms on "
                      f"{threads} torch threads"}, out


def side_traffic(key):
    """HBM bytes per launch of a side config's dominant kernel from the committed counter passes (profiles/pmc_traffic.json:
    side_configs; FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes), or (None, None)."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pj.get("side_configs", {}).get(key)
        if ent:
            return ent["hbm_bytes_per_launch"], "profiles/pmc_traffic.json side_configs." + key + " (" + ent.get("collected", "") + ")"
    except Exception:     # noqa: BLE001
        pass
    return None, None


def roofline_terms(n, e_eff, f_in, f_g, f_out, w_cols, symnorm):
    """SURVEY.md 8(d) bytes.  `aggregate_launch` = what the fused aggregate+combine launch must move by that
    model (weightings are "counted as fused (not materialised)"); `layer` = the survey's whole-layer figure."""
    from egc_amd.workloads import algorithmic_bytes
    t = algorithmic_bytes(n, e_eff, f_in, f_g, f_out, w_cols, symnorm=symnorm)
    t["aggregate_launch"] = t["gather"] + t["col"] + t["rowptr"] + t["deg"] + t["out"]
    return t


# -------------------------------------------------------------------------------------------------
# the ONE JSON line: compact (the driver keeps a bounded tail of stdout: round 5's 23 KB line did not parse).
# The full record goes to bench_detail.json (and to stderr); the line carries the contract keys, `roofline`,
# `cpu_baseline`, the kernel times and -- per side config -- {workload, layer_ms | step_ms, frac, traffic} only.
# -------------------------------------------------------------------------------------------------
LINE_LIMIT_BYTES = 6000
DETAIL_FILE = "bench_detail.json"


def _r(v, sig=6):
    """Floats to `sig` significant digits (the line's budget goes to fields, not to digits); containers recursively."""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _shorten_strings(v, n):
    if isinstance(v, str):
        return _short(v, n)
    if isinstance(v, dict):
        return {k: _shorten_strings(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_shorten_strings(x, n) for x in v][:16]
    return v


def _side_record(rec):
    """One side config on the line: what it is, how long it took, its fraction and counter traffic.  Everything else of the
    record (every path's timings, byte models, its own cpu_baseline sample text) is in the detail file."""
    if not isinstance(rec, dict):
        return _short(rec, 120)
    out = {"workload": _short(rec.get("workload", ""), 72)}
    for k in ("layer_ms", "step_ms", "eager_step_ms", "hipgraph_replay_ms", "eager_ms", "coo_hipgraph_replay_ms", "csr_path_step_ms",
              "speedup_vs_csr_path"):
        if k in rec:
            out[k] = rec[k]
    rf = rec.get("roofline") if isinstance(rec.get("roofline"), dict) else {}
    frac = rf.get("frac", rec.get("layer_frac", rec.get("step_frac")))
    if frac is not None:
        out["frac"] = frac
    if "layer_frac" in rec and rf:
        out["layer_frac"] = rec["layer_frac"]
    if rf:
        out["traffic"] = rf.get("traffic")
    cb = rec.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        out["cpu_edges_per_s"] = cb["value"]
        out["cpu_cores"] = cb.get("cores")
    if "roofline_error" in rec:
        out["error"] = _short(rec["roofline_error"], 120)
    return out


def compact_line(result):
    """The line the driver parses, built from the full record.  Keeps every key of the contract; shrinks prose."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "gemm", "data")
    line = {k: result[k] for k in keep if k in result}
    if "gemm" in line:
        line["gemm"] = _short(line["gemm"], 100)
    cfg = dict(result.get("config", {}))
    if "workload" in cfg:
        cfg["workload"] = _short(cfg["workload"], 200)
    line["config"] = cfg
    rf = result.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                               "algorithmic_bytes_per_launch", "launch_ms", "frac_vs_measured_copy_ceiling") if k in rf}
        if "kernel" in line["roofline"]:
            line["roofline"]["kernel"] = _short(line["roofline"]["kernel"], 80)
    else:
        line["roofline"] = rf
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "host_cores", "hip_vs_port_rel_err", "error")
                                if k in cb}
        if "sample" in line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = _short(line["cpu_baseline"]["sample"], 200)
    else:
        line["cpu_baseline"] = cb
    for k in ("kernels_ms", "layer_frac", "layer_frac_by_gemm", "layer_algorithmic_bytes", "ms_per_step_median_of_5_regions"):
        if k in result:
            line[k] = result[k]
    sl = result.get("std_layer")
    if isinstance(sl, dict):
        line["std_layer"] = {k: sl[k] for k in ("layer_ms", "layer_frac", "error") if k in sl}
    oc = result.get("other_configs")
    if isinstance(oc, dict):
        line["other_configs"] = {k: _side_record(v) for k, v in oc.items()}
    if "bench_scale" in result:
        line["bench_scale"] = result["bench_scale"]
    ss = result.get("strong_scaling")      # bench_multi.py: per workload a flat record of numbers + a few descriptions
    if isinstance(ss, dict):
        line["strong_scaling"] = {k: _shorten_strings(v, 80) for k, v in ss.items()}
    line["detail"] = DETAIL_FILE
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT_BYTES:      # never print a line the driver cannot hold: drop side records, largest first
        for field in ("other_configs", "strong_scaling"):
            oc = line.get(field) or {}
            while len(text) > LINE_LIMIT_BYTES and oc:
                oc.pop(max(oc, key=lambda k: len(json.dumps(oc[k]))))
                line[field + "_truncated"] = True
                text = json.dumps(line, separators=(",", ":"))
    return text


def emit(result):
    """Full record -> bench_detail.json (next to this file; EGC_BENCH_DETAIL overrides the path) and stderr; compact line -> stdout."""
    path = os.environ.get("EGC_BENCH_DETAIL", os.path.join(ROOT, DETAIL_FILE))
    full = json.dumps(result, indent=1)
    try:
        with open(path, "w") as f:
            f.write(full + "\n")
    except OSError as ex:
        log(f"bench.py: could not write {path}: {ex!r}")
    log("full record (" + path + "):")
    log(json.dumps(result))
    print(compact_line(result), flush=True)


# -------------------------------------------------------------------------------------------------
# other configs (N = 1): measured in the same run, module-level calls (what a caller of the layer pays)
# -------------------------------------------------------------------------------------------------
def measure_layer_config(name, ei_cpu, n, conv, f_in, dev, per_batch_csr, iters=30, batch=None, max_nodes=None, traffic_key=None,
                         self_loops=True):
    """`batch` (+ `max_nodes`): the workload is a PyG-style batch of small graphs -- measured on the ordinary path (per-batch
    egc_graph_build + GEMM + aggregate) AND on the tile path (egc_amd.GraphBatch: one plan launch per batch; GEMM; one
    launch that builds each tile's CSR in LDS and aggregates from LDS); the record's headline fields are the faster one's."""
    import egc_amd
    ei = ei_cpu.to(dev)
    x_cpu = torch.randn(n, f_in)
    x = x_cpu.to(dev)
    state_cpu = {k: v.detach().clone() for k, v in conv.state_dict().items()}
    conv = conv.to(dev).eval()
    spec = conv._spec_coo
    e_in = int(ei.size(1))
    # EGConv convention (optimized_layers.py:127-175): every aggregator traverses one self loop per node -- unless the layer is built
    # with add_self_loops=False and without symnorm (the edge sets of EfficientGraphConv's add / mean / max, layers.py:166-193)
    e_eff = e_in + (n if self_loops else 0)
    rec = {"workload": name, "n_nodes": n, "e_in": e_in, "e_eff": e_eff}
    with torch.no_grad():
        if per_batch_csr:
            for _ in range(3):
                egc_amd.CSRGraph.from_edge_index(ei, n)
            rec["csr_build_ms"] = time_region_median(lambda: egc_amd.CSRGraph.from_edge_index(ei, n), 10)
            g = egc_amd.CSRGraph.from_edge_index(ei, n)     # a per-batch graph: nothing is read back to the host
        else:
            g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()   # static graph (cached=True)
        for _ in range(5):
            conv(x, g)
        rec["layer_ms"] = time_region_median(lambda: conv(x, g), iters, 3)
    has_sym = "symnorm" in conv.aggregators
    t = roofline_terms(n, e_eff, f_in, spec.f_g, conv.out_channels, spec.w_cols, has_sym)
    rec["algorithmic_bytes"] = t["layer"]
    rec["edges_per_s"] = e_eff / (rec["layer_ms"] * 1e-3)
    rec["layer_frac"] = t["layer"] / (rec["layer_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if per_batch_csr:
        tot = rec["layer_ms"] + rec["csr_build_ms"]
        rec["edges_per_s_incl_csr"] = e_eff / (tot * 1e-3)
        rec["frac_incl_csr"] = t["layer"] / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS
        rec["path"] = "csr: egc_graph_build (5 launches) + GEMM + agg_fast_kernel"
    if batch is not None:
        g_count = int(batch.max()) + 1
        sizes = torch.bincount(batch.to(dev), minlength=g_count)
        ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])   # = batch.ptr of a PyG Batch
        mx = int(sizes.max())          # the loader's largest graph (a dataset property; `max_nodes` is its declared bound)
        # the graphs' edge offsets: PyG's collation has them on the host (batch._slice_dict); NOT part of the timed region below,
        # which is why the headline of this record is the form that does not need them
        eptr = torch.searchsorted(batch.to(dev)[ei[1]].contiguous(), torch.arange(g_count + 1, device=dev))
        compulsory = n * (f_in + conv.out_channels) * 4 + e_in * 16 + (g_count + 1) * 8     # x + out + edge_index + ptr
        paths = {}
        with torch.no_grad():
            ref = conv(x, egc_amd.CSRGraph.from_edge_index(ei, n))
            for key, env, ep, label in (
                    ("fused", "0", None, "ONE launch (egc_layer_forward_batch_fused_f32): plan + GEMM + per-tile CSR + aggregate + combine; "
                                         "edge ranges found in the launch from edge_index + batch.ptr"),
                    ("fused_edge_ptr", "0", eptr, "the same with the graphs' edge offsets supplied (computed outside the timed region)"),
                    ("tile", "1", eptr, "two launches + plan: egc_batch_plan + GEMM + agg_tile_kernel (edge offsets supplied)")):
                os.environ["EGC_NO_FUSED_TILE"] = env
                try:
                    def make():
                        return egc_amd.GraphBatch(ei, ptr=ptr, num_nodes=n, max_nodes=mx if key != "tile" else (max_nodes or 256),
                                                  edge_ptr=ep)
                    gb = make()
                    out = conv(x, gb)
                    gb.check()
                    err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1))
                    one_shot = any(isinstance(k, tuple) and k[-1] == "fused" and v for k, v in gb._setups.items())
                    if (key != "tile") != one_shot:
                        raise RuntimeError(f"path {key}: the expected kernel did not run")

                    def one_go():
                        conv(x, make())
                    for _ in range(20):      # untimed: the first path measured here was otherwise timed on clocks still ramping
                        one_go()
                    both_ms = time_region_median(one_go, iters, 3)
                    same_ms = time_region_median(lambda: conv(x, gb), iters, 3)
                    paths[key] = {"path": label, "new_batch_every_call_ms": both_ms, "same_batch_ms": same_ms,
                                  "edges_per_s": e_eff / (both_ms * 1e-3),
                                  "frac_survey_bytes": t["layer"] / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "frac_compulsory_bytes": compulsory / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "rel_err_vs_csr_path": err}
                finally:
                    os.environ.pop("EGC_NO_FUSED_TILE", None)
        rec["csr_path"] = {k: rec[k] for k in ("csr_build_ms", "layer_ms", "edges_per_s_incl_csr", "frac_incl_csr", "layer_frac", "path")}
        rec["compulsory_bytes"] = compulsory     # what the one-launch path has to move (SURVEY 8d's figure credits a gather per edge)
        rec["survey_model_bytes"] = t["layer"]
        rec["max_graph_nodes"] = mx
        rec.update(paths)
        # headline fields: the one-launch path in the form that needs only what a PyG `Batch` carries on the device (edge_index +
        # batch.ptr; edge ranges found inside the launch) -- the forms that are handed the graphs' edge offsets (built outside the
        # timed region) stay side fields.  Its fractions are of the bytes the launch HAS to move (x + out + edge_index
        # + ptr): SURVEY 8(d)'s per-edge gather is served from LDS here, so that model's bytes over the time is a speed-up
        # against the model, not a fraction of the HBM peak -- printed as `survey_model_ratio`.
        if "fused" in paths:
            pb = paths["fused"]
            rec.update(csr_build_ms=0.0, layer_ms=pb["new_batch_every_call_ms"], edges_per_s=pb["edges_per_s"],
                       layer_frac=pb["frac_compulsory_bytes"], edges_per_s_incl_csr=pb["edges_per_s"],
                       frac_incl_csr=pb["frac_compulsory_bytes"], frac_compulsory_bytes=pb["frac_compulsory_bytes"],
                       survey_model_ratio=pb["frac_survey_bytes"], path="fused: " + pb["path"])
    # ---- the dominant kernel of this config against its own bytes, and the CPU port timed beside it ----
    try:
        with torch.no_grad():
            if batch is not None and "fused" in rec and rec.get("path", "").startswith("fused"):
                # the whole layer IS one kernel: its frac on SURVEY's bytes (a gather per edge credited) and on what it moves
                key = rec["path"].split(":")[0]
                ms_k = rec[key]["same_batch_ms"]
                traffic, traffic_src = side_traffic(traffic_key)
                rec["roofline"] = {"bound": "hbm", "kernel": "egc::fused_tile_kernel (plan + GEMM + CSR + aggregate + combine)",
                                   "achieved": rec["compulsory_bytes"] / (ms_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": rec["compulsory_bytes"] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "bytes": "compulsory: x + out + edge_index + ptr (the launch gathers from LDS, not from HBM)",
                                   "survey_model_ratio": t["layer"] / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "launch_ms": ms_k, "traffic": traffic, "traffic_source": traffic_src,
                                   "note": "launch_ms includes the host's call (HIP events around module calls on one batch); "
                                           "survey_model_ratio = SURVEY 8(d) bytes / time / peak: not a fraction of the peak (may exceed 1)"}
                ref_out = conv(x, egc_amd.GraphBatch(ei, ptr=ptr, num_nodes=n, max_nodes=mx))
            else:
                from egc_amd import functional as Fn
                wcat, bcat = conv._packed_weights()
                planes = conv._weight_planes(spec, wcat)
                bases, wts = Fn.egc_basis_transform(g, spec, x, wcat, bcat, planes)
                agg_ms = time_region_median(lambda: Fn.egc_aggregate_combine(g, spec, bases, wts, conv.bias), iters, 3)
                gemm_ms = time_region_median(lambda: Fn.egc_basis_transform(g, spec, x, wcat, bcat, planes), iters, 3)
                agg_bytes = t["gather"] + t["col"] + t["rowptr"] + t["deg"] + t["out"]
                rec["kernels_ms"] = {"basis_gemm": gemm_ms, "aggregate_combine": agg_ms}
                rec["roofline"] = {"bound": "hbm", "kernel": "egc::agg_fast_kernel (fused aggregate+combine)",
                                   "achieved": agg_bytes / (agg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": agg_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": agg_ms,
                                   "traffic": side_traffic(traffic_key)[0], "traffic_source": side_traffic(traffic_key)[1],
                                   "gemm_frac": (n * 4 * (f_in + spec.ldb + spec.w_cols)) / (gemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                ref_out = conv(x, g)
        if not getattr(measure_layer_config, "no_cpu", False):
            cb, cpu_out = cpu_leg(state_cpu, x_cpu, ei_cpu, conv.num_heads, conv.num_bases, list(conv.aggregators), e_eff,
                                  add_self_loops=self_loops)
            cb["hip_vs_port_rel_err"] = float((ref_out.cpu() - cpu_out).abs().max() / max(1.0, float(cpu_out.abs().max())))
            rec["cpu_baseline"] = cb
    except Exception as ex:   # noqa: BLE001 -- a side figure never takes the record down
        rec["roofline_error"] = repr(ex)[:300]
    log(f"  {name}: " + ", ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()
                                  if k != "workload"))
    del g, x, ei
    torch.cuda.empty_cache()
    return rec


def training_step_bytes(n, e_eff, f_in=F_IN, f_out=F_OUT, ldb=64, w_cols=128, stat_k=3):
    """What one training step of the north-star layer (forward + backward, config 2) has to move, pass by pass -- the
    backward's counterpart of SURVEY.md 8(d)'s forward figure.  Per CSR entry the forward gathers one
    basis row (ldb floats + a 4-byte index); the backward's source pass gathers, per transposed entry, the destination's two
    gradient rows (sum-like and max-routed: 2 x ldb floats) and one 64-byte record (arg positions as bytes), plus its index."""
    row = ldb * 4
    t = {
        "fwd_gemm": n * 4 * (f_in + ldb + w_cols),
        "fwd_aggregate_train": e_eff * (row + 4) + n * 8 + n * 4 * (f_out + w_cols) + n * stat_k * row + n * (ldb + 4),
        "bwd_destination": n * 4 * (f_out + w_cols + w_cols) + n * stat_k * row + n * 2 * row + e_eff * 64,
        "bwd_source": e_eff * (2 * row + 64 + 4) + n * 8 + n * row,
        "weight_gradient": n * 4 * (f_in + ldb + w_cols),
        "dx_gemm": n * 4 * (ldb + w_cols + f_in),
    }
    t["total"] = sum(t.values())
    return t


def _north_star_layer():
    import egc_amd
    return egc_amd.EGConv(F_IN, F_OUT, aggrs=AGGRS, num_heads=HEADS, num_bases=BASES)


def other_configs(dev, seed):
    """The side figures.  Every section stands alone: a failure in one is recorded (``<section>_error``) and leaves the
    others and the headline line untouched."""
    out = {}
    for name, section in (("layer_configs", _oc_layer_configs), ("config2_training", _oc_config2_training),
                          ("small_batches", _oc_small_batches)):
        try:
            section(out, dev, seed)
        except Exception as ex:   # noqa: BLE001 -- reported in the line, never fatal for it
            out[name + "_error"] = repr(ex)[:500]
            log(f"  other_configs section {name} failed: {ex!r}")
            try:
                torch.cuda.synchronize(dev)
                torch.cuda.empty_cache()
            except Exception:     # noqa: BLE001
                pass
    return out


def _oc_layer_configs(out, dev, seed):
    import egc_amd
    from egc_amd import workloads as wl
    torch.manual_seed(seed)
    ns = _north_star_layer
    _, ei, n, batch = wl.zinc_like_batch(128, seed=seed)
    out["config1_zinc_b128"] = measure_layer_config(
        "ZINC-shaped batch of 128 molecules (BASELINE config 1: EGC-S plumbing shape), EGConv d=128 H=1 B=1 sum", ei, n,
        egc_amd.EGConv(F_IN, F_OUT, aggrs=["sum"], num_heads=1, num_bases=1), F_IN, dev, True, batch=batch, max_nodes=37)
    out["config1_zinc_b128_egcm"] = measure_layer_config(
        "the same batch through the north-star EGC-M layer (d=128 H=8 B=4 sum+mean+max+symnorm)", ei, n, ns(), F_IN, dev, True,
        batch=batch, max_nodes=37)
    ei, n, batch = wl.molecule_batch(2048, seed=seed)
    out["config3_molhiv_b2048"] = measure_layer_config(
        "ogbg-molhiv-shaped batch of 2048 graphs, EGC-M d=128 H=8 B=4 sum+mean+max+symnorm", ei, n, ns(), F_IN, dev, True,
        batch=batch, max_nodes=222, traffic_key="config3_molhiv_b2048")
    # config 3 at the reference's OWN molhiv net (run_pretrained.sh:24, hyperparameters.md: hidden 224, H 4, B 4, add + mean + max on
    # the raw edges -- EfficientGraphConv's edge sets, expressed through EGConv(add_self_loops=False)): the WIDE one-launch form
    # (egc_fused_tile_wide*.hip) against plan + GEMM + agg_tile_kernel and the CSR path, all timed in this record
    out["config3_molhiv_b2048_ref224"] = measure_layer_config(
        "ogbg-molhiv-shaped batch of 2048 graphs, the reference's molhiv EGC-M layer d=224 H=4 B=4 add+mean+max (raw edges)", ei, n,
        egc_amd.EGConv(224, 224, aggrs=["sum", "mean", "max"], num_heads=4, num_bases=4, add_self_loops=False), 224, dev, True,
        batch=batch, max_nodes=222, traffic_key="config3_molhiv_b2048_ref224", self_loops=False)
    ei, n, batch = wl.knn_superpixel_batch(2048, seed=seed)
    out["config4_cifar_b2048"] = measure_layer_config(
        "CIFAR10-superpixel-shaped batch of 2048 8-NN graphs, EGC-M d=128 H=8 B=4 sum+mean+max+symnorm", ei, n, ns(),
        F_IN, dev, True, batch=batch, max_nodes=150, traffic_key="config4_cifar_b2048")
    # the reference's ogbg-code EGC-S layer on a code-SHAPED batch (run_pretrained.sh:47; egc_amd.workloads.code_like_batch: ASTs of ~125
    # nodes, up to 250): no LDS tile of a CU holds a 250-node graph's 1,280-byte `bases` rows, so this is the CSR path with its
    # per-batch graph build
    ei, n, batch = wl.code_like_batch(128, seed=seed)
    out["code_b128_ref304"] = measure_layer_config(
        "ogbg-code2-shaped batch of 128 ASTs, the reference's code EGC-S layer d=304 H=8 B=8 symadd (CSR path: per-batch graph build)", ei, n,
        egc_amd.EGConv(304, 304, aggrs=["symnorm"], num_heads=8, num_bases=8), 304, dev, True)
    ei, n = wl.mag_like(seed=seed)
    out["config5_mag_homogeneous_1gpu"] = measure_layer_config(
        "ogbn-mag-shaped homogeneous graph (mag/configs.py:73-88), EGConv 352->352 H=8 B=4 symnorm (mag/models.py:23-53)",
        ei, n, egc_amd.EGConv(352, 352, aggrs=["symnorm"], num_heads=8, num_bases=4), 352, dev, False, iters=10,
        traffic_key="config5_mag_aggregate")


def _oc_config2_training(out, dev, seed):
    import egc_amd
    from egc_amd import workloads as wl
    torch.manual_seed(seed)
    ns = _north_star_layer
    # training step of config 2: forward + backward of one north-star layer through autograd
    ei, n = wl.arxiv_like(seed=seed)
    ei = ei.to(dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
    layer = ns().to(dev)
    x = torch.randn(n, F_IN, device=dev, requires_grad=True)
    go = torch.randn(n, F_OUT, device=dev)

    def fwd_bwd():   # as the reference's loops: gradients are cleared every step (zinc/configs.py:64-67: optimizer.zero_grad())
        layer.zero_grad(set_to_none=True)
        x.grad = None
        layer(x, g).backward(go)
    for _ in range(3):
        fwd_bwd()
    ms = time_region_median(fwd_bwd, 10, 3)
    tb = training_step_bytes(n, int(ei.size(1)) + n)
    out["config2_training_step"] = {"workload": "config 2, forward + backward of one EGConv layer through autograd (gradients cleared every step)",
                                    "step_ms": ms, "edges_per_s": (int(ei.size(1)) + n) / (ms * 1e-3),
                                    "algorithmic_bytes": tb, "step_frac": tb["total"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "byte model of DESIGN.md section 5 (what each pass has to move; per-kernel times: "
                                            "profiles/r04_training_step_kernel_stats.csv)"}
    log(f"  training step (config 2): {ms:.4f} ms")
    # the reference nets' block (zinc/models.py:66-72): conv -> BatchNorm1d (batch statistics) -> ReLU -> + input
    bn = torch.nn.BatchNorm1d(F_OUT).to(dev)
    block = egc_amd.FusedEGCBlock(layer, bn).train()

    def block_step(fn):
        block.zero_grad(set_to_none=True)
        x.grad = None
        fn(x, g).backward(go)
    for fn in (block, block._plain):
        for _ in range(3):
            block_step(fn)
    ms_fused = time_region_median(lambda: block_step(block), 10, 3)
    ms_torch = time_region_median(lambda: block_step(block._plain), 10, 3)
    out["config2_training_block"] = {"workload": "config 2, forward + backward of conv -> BatchNorm1d(train) -> ReLU -> + input",
                                     "step_ms": ms_fused, "step_ms_with_torch_tail": ms_torch}
    log(f"  training block (config 2): {ms_fused:.4f} ms (torch tail: {ms_torch:.4f} ms)")
    # the ogbn-arxiv net's own block has a dropout in front of the residual add (arxiv/norm_models.py:34-40, p = 0.2)
    block = egc_amd.FusedEGCBlock(layer, bn, dropout=0.2).train()
    for fn in (block, block._plain):
        for _ in range(3):
            block_step(fn)
    ms_fused = time_region_median(lambda: block_step(block), 10, 3)
    ms_torch = time_region_median(lambda: block_step(block._plain), 10, 3)
    out["config2_training_block_arxiv_net"] = {
        "workload": "config 2, forward + backward of conv -> BatchNorm1d(train) -> ReLU -> dropout(0.2) -> + input",
        "step_ms": ms_fused, "step_ms_with_torch_tail": ms_torch}
    log(f"  arxiv-net training block (config 2, dropout 0.2): {ms_fused:.4f} ms (torch tail: {ms_torch:.4f} ms)")


def _oc_small_batches(out, dev, seed):
    import egc_amd
    from egc_amd import workloads as wl
    ns = _north_star_layer
    # the batch sizes the reference actually trains at (zinc/configs.py: 128 graphs per batch): a few thousand nodes,
    # where the step is bound by what launches the kernels -- eager against the whole step replayed as one hipGraph
    def ref168():     # the reference's ZINC EGC-S layer as its net constructs it (zinc/models.py:125-135, run_pretrained.sh:7)
        return egc_amd.EfficientGraphConv(168, 168, 8, 4, False, aggrs=["symadd"])
    def ref224():     # the reference's molhiv EGC-M layer (mol/pna_style_models.py, run_pretrained.sh:23)
        return egc_amd.EfficientGraphConv(224, 224, 4, 4, False, aggrs=["add", "mean", "max"])
    def ref296():     # the reference's molhiv EGC-S layer (run_pretrained.sh:23)
        return egc_amd.EfficientGraphConv(296, 296, 8, 4, False, aggrs=["symadd"])
    for key, (ei, n, bvec), label, make, width in (
            ("zinc_b128_training_step", wl.zinc_like_batch(128, seed=seed)[1:], "ZINC-shaped batch of 128 molecules", ns, F_IN),
            ("molhiv_b2048_training_step", wl.molecule_batch(2048, seed=seed), "molhiv-shaped batch of 2048 molecules", ns, F_IN),
            ("zinc_b128_ref168_training_step", wl.zinc_like_batch(128, seed=seed)[1:],
             "ZINC-shaped batch of 128 molecules, the reference's own EGC-S net width (EfficientGraphConv 168 / H8 / B4 symadd: CSR path)", ref168, 168),
            ("molhiv_b2048_ref224_training_step", wl.molecule_batch(2048, seed=seed),
             "molhiv-shaped batch of 2048 molecules, the reference's own EGC-M net width (EfficientGraphConv 224 / H4 / B4 add, mean, max: CSR path)", ref224, 224),
            ("molhiv_b2048_ref296_training_step", wl.molecule_batch(2048, seed=seed),
             "molhiv-shaped batch of 2048 molecules, the reference's own EGC-S net width (EfficientGraphConv 296 / H8 / B4 symadd: CSR path)", ref296, 296)):
        torch.manual_seed(seed)
        blocks = torch.nn.ModuleList([egc_amd.FusedEGCBlock(make(), torch.nn.BatchNorm1d(width)) for _ in range(4)]).to(dev).train()
        params = list(blocks.parameters())
        ei = ei.to(dev)
        xs, gos = torch.randn(n, width, device=dev), torch.randn(n, width, device=dev)
        # what a PyG batch carries besides edge_index: the graphs' node offsets (Batch.ptr); with them the layer runs as one
        # launch each way (egc_layer_forward_batch_fused_f32 / egc_layer_backward_batch_fused_f32), no graph build at all
        sizes = torch.bincount(bvec.to(dev))
        ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sizes, 0)])
        max_nodes = int(sizes.max())
        as_batch = [True]

        def step():
            g = egc_amd.GraphBatch(ei, ptr=ptr, max_nodes=max_nodes, num_nodes=n) if as_batch[0] else ei
            h = xs
            for b in blocks:
                h = b(h, g)        # (COO in: the per-batch graph build is part of the step)
            h.backward(gos)

        def eager():
            for p in params:
                p.grad = None
            ei.add_(0)             # a new batch: the graph cache must not serve the previous build
            step()

        def wall(fn, iters=50):
            for _ in range(5):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / iters * 1e3
        # (eager steps are host-bound and the host is shared: the median of three regions, not one)
        ms_eager = sorted(wall(eager) for _ in range(3))[1]
        graphed = egc_amd.GraphedStep(step, params=params)
        ms_graph = wall(graphed)
        del graphed
        as_batch[0] = False
        ms_eager_coo = sorted(wall(eager) for _ in range(3))[1]
        graphed = egc_amd.GraphedStep(step, params=params)
        ms_graph_coo = wall(graphed)
        out[key] = {"workload": f"{label} (N={n}, E={int(ei.size(1))}): 4 x [EGConv -> BatchNorm1d(train) -> ReLU -> + x], forward + backward",
                    "path": "edge_index + the batch's node offsets (egc_amd.GraphBatch): one launch each way per layer, no graph build",
                    "eager_step_ms": ms_eager, "hipgraph_replay_ms": ms_graph,
                    "coo_path": "edge_index only: per-batch graph build + GEMM + aggregate, three backward kernels (rounds 2-4)",
                    "coo_eager_step_ms": ms_eager_coo, "coo_hipgraph_replay_ms": ms_graph_coo}
        log(f"  {key}: eager {ms_eager:.4f} ms, one hipGraph {ms_graph:.4f} ms (edge_index only: {ms_eager_coo:.4f} / {ms_graph_coo:.4f})")
        del graphed
        if "ref224" in key or "ref296" in key:       # (training only: the line the driver parses has room for one more record each, not two)
            del blocks, params
            continue
        # the same net serving: eval mode (BatchNorm / ReLU / residual in the aggregate kernel's store), graph build included
        blocks.eval()
        outs = []

        def infer():
            with torch.no_grad():
                h = xs
                for b in blocks:
                    h = b(h, ei)
            outs[:] = [h]

        def eager_infer():
            ei.add_(0)
            infer()
        ms_eager = wall(eager_infer)
        graphed = egc_amd.GraphedStep(infer)
        ms_graph = wall(graphed)
        out[key.replace("training_step", "inference")] = {
            "workload": f"{label}: graph build + 4 fused blocks, eval-mode forward", "eager_ms": ms_eager,
            "hipgraph_replay_ms": ms_graph}
        log(f"  {key.replace('training_step', 'inference')}: eager {ms_eager:.4f} ms, one hipGraph {ms_graph:.4f} ms")
        del graphed, blocks, params


def bench_rmag(args, world, dev):
    """The heterogeneous ogbn-mag shape (BASELINE.json's "~21M edges": rmag/models.py:18-26, 1.94 M nodes of 4 types,
    21.1 M typed edges + reverses / symmetrisation = 42.1 M CSR entries in 7 relations) through one REGConv layer
    (rmag/models.py:75-148) on ONE GPU; the relational layer has no partitioned form."""
    import egc_amd
    from egc_amd import workloads as wl
    if world > 1:
        raise SystemExit("bench_rmag is the one-GPU form; N > 1 goes through bench_multi.strong_typed")
    torch.manual_seed(args.seed)
    nodes, rel = wl.rmag_like(seed=args.seed)
    adj = {}
    for (s, r, d), ei in rel.items():
        ei = ei.to(dev)
        adj[(s, r, d)] = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(nodes[d], nodes[s]))
    conv = egc_amd.REGConv(F_IN, F_OUT, HEADS, BASES).to(dev).eval()
    x = {k: torch.randn(n, F_IN, device=dev) for k, n in nodes.items()}
    entries = sum(int(v.shape[1]) for v in rel.values())
    with torch.no_grad():
        for _ in range(max(args.warmup, 3)):
            conv(x, adj)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            conv(x, adj)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
    emit({
        "metric": METRIC, "value": entries / (ms * 1e-3), "unit": "edges/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": None, "vs_baseline": None,
        "dtype": "f32", "gemm": "fp16x2 / bf16x3 split, one GEMM per node type", "data": "synthetic",
        "config": {"workload": f"ogbn-mag-shaped typed graph ({sum(nodes.values())} nodes of {len(nodes)} types, "
                               f"{entries} CSR entries in {len(rel)} relations), REGConv {F_IN}->{F_OUT} H={HEADS} B={BASES}",
                   "layer": "REGConv", "parallelism": "single GPU"},
        "roofline": None, "cpu_baseline": None})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--workload", default=None, choices=[None, "all", "arxiv", "mag", "arxiv-weak", "rmag"],
                    help="default: arxiv (config 2) on one GPU; on several GPUs `all` = config 2 strong scaling as the "
                         "headline with config 5 (homogeneous mag, typed rmag through a partitioned REGConv) nested; "
                         "arxiv / mag / rmag run one of them alone")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])     # does not return
    world, rank, local = dist_setup(args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    workload = args.workload or ("arxiv" if world == 1 else "all")
    if workload == "rmag" and world == 1:
        return bench_rmag(args, world, dev)
    forced = os.environ.get("EGC_BENCH_FORCE_PARTITION", "0") not in ("", "0") and "RANK" in os.environ
    if forced and args.workload is None:
        workload = "all"
    if workload != "arxiv" or world > 1 or forced:
        import bench_multi  # the partitioned workloads live in their own file
        return bench_multi.run(args, world, rank, local, workload)

    import egc_amd
    from egc_amd import _C
    from egc_amd.functional import LayerSpec, gemm_exact, pack_weights
    from egc_amd.workloads import arxiv_like

    lib = _C.load()
    torch.manual_seed(args.seed)
    conv = egc_amd.EGConv(F_IN, F_OUT, aggrs=AGGRS, num_heads=HEADS, num_bases=BASES, cached=True)
    with torch.no_grad():
        conv.bias.normal_()
    state_cpu = {k: v.detach().clone() for k, v in conv.state_dict().items()}
    conv = conv.to(dev).eval()
    spec: LayerSpec = conv._spec_coo
    wcat, bcat = conv._packed_weights()
    planes = pack_weights(spec, wcat)  # split-precision weight planes, rebuilt only when parameters change
    bias = conv.bias.detach()
    ldb = spec.ldb
    ei_cpu, n = arxiv_like(seed=args.seed)
    ei = ei_cpu.to(dev)
    graph = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()  # static graph (the reference's cached=True)
    torch.manual_seed(args.seed + 1)
    x_cpu = torch.randn(n, F_IN)
    x = x_cpu.to(dev)
    e_in = int(ei.size(1))
    e_eff = e_in + n  # EGConv convention: every aggregator also traverses one self loop per node

    bases = torch.empty((graph.n_src_rows, ldb), device=dev)
    weightings = torch.empty((n, spec.w_cols), device=dev)
    out = torch.empty((n, F_OUT), device=dev)
    ws = torch.zeros(max(lib.egc_aggregate_workspace_bytes(C.byref(spec.c), n, e_in), 1), dtype=torch.uint8, device=dev)
    g = graph.c_struct()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def gemm_only():
        _C.check(lib.egc_basis_transform_packed(x.data_ptr(), planes.data_ptr(), bcat.data_ptr(), n, F_IN, spec.f_g,
                                                spec.w_cols, bases.data_ptr(), ldb, weightings.data_ptr(), stream),
                 "egc_basis_transform_packed")

    def gemm_exact_only():   # the plain fp32-MFMA GEMM (EGC_GEMM_EXACT=1), for the side field
        _C.check(lib.egc_basis_transform_f32(x.data_ptr(), wcat.data_ptr(), bcat.data_ptr(), n, F_IN, spec.f_g,
                                             spec.w_cols, bases.data_ptr(), ldb, weightings.data_ptr(), stream),
                 "egc_basis_transform_f32")

    nb24 = lib.egc_basis_pack_bytes(F_IN, spec.f_g, spec.w_cols)
    planes24 = torch.empty(nb24, dtype=torch.uint8, device=dev)
    _C.check(lib.egc_basis_pack_ex(wcat.data_ptr(), F_IN, spec.f_g, spec.w_cols, 1, planes24.data_ptr(), nb24, stream), "egc_basis_pack_ex")

    def gemm_24bit_only():   # the 24-bit-operand form (EGC_GEMM_24BIT: three bf16 planes per operand), for the side field
        _C.check(lib.egc_basis_transform_packed_ex(x.data_ptr(), planes24.data_ptr(), bcat.data_ptr(), n, F_IN, spec.f_g, spec.w_cols, 1,
                                                   bases.data_ptr(), ldb, weightings.data_ptr(), stream), "egc_basis_transform_packed_ex")

    def agg_only():
        _C.check(lib.egc_aggregate_combine_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), ldb,
                                               weightings.data_ptr(), bias.data_ptr(), out.data_ptr(), None, None,
                                               ws.data_ptr(), ws.numel(), stream), "egc_aggregate_combine_f32")

    def step():  # one full layer forward through the C ABI
        _C.check(lib.egc_layer_forward_packed(C.byref(g), C.byref(spec.c), x.data_ptr(), planes.data_ptr(),
                                              bcat.data_ptr(), bias.data_ptr(), bases.data_ptr(), ldb,
                                              weightings.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                              stream), "egc_layer_forward_packed")

    def sync_all():
        torch.cuda.synchronize(dev)

    # ---- per-kernel HIP-event timing on the launch stream.  Done BEFORE the contract's warm-up + timed region,
    # so that region runs at the clocks the chip holds in steady state rather than during the first
    # milliseconds after idle (the whole region is ~30 ms at the default step count).
    for _ in range(100):  # untimed: bring the chip out of idle clocks before anything is measured
        step()
    reps = max(50, min(args.steps, 200))
    agg_ms = time_region(agg_only, reps)
    gemm_ms = time_region(gemm_only, reps)
    gemm_exact_ms = time_region(gemm_exact_only, 20)
    gemm_24bit_ms = time_region(gemm_24bit_only, 20)
    gemm_only()  # leave the split-precision intermediates in place
    step_ms_events = time_region(step, reps)

    # ---- warm-up, then EXACTLY --steps timed steps bracketed by synchronize ----
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / args.steps * 1e3
    value = float(e_eff) / (elapsed / args.steps)
    # robustness of the headline (not part of its definition): four more regions of the same length, the median of the five
    regions = [ms_per_step]
    for _ in range(4):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        regions.append((time.perf_counter() - t0) / args.steps * 1e3)

    terms = roofline_terms(n, e_eff, F_IN, spec.f_g, F_OUT, spec.w_cols, symnorm=True)
    agg_bytes = terms["aggregate_launch"]
    agg_gbs = agg_bytes / (agg_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            pj = json.load(open(pmc_file))
            traffic = pj.get("aggregate_kernel_hbm_bytes_per_launch")
            traffic_src = "profiles/pmc_traffic.json (" + pj.get("collected", "rocprofv3 --pmc passes, not this run") + ")"
        except Exception:
            traffic = None

    result = {
        "metric": METRIC,
        "value": value, "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": None,   # one GPU: neither weak nor strong
        "vs_baseline": None,
        "dtype": "f32", "gemm": "exact fp32 MFMA" if gemm_exact() else
                 "fp16x2-split (3 x v_mfma_f32_32x32x16_f16 per k-step, fp32 accumulate, 22-bit operands)",
        "data": "synthetic",
        "config": {"workload": "ogbn-arxiv-shaped full graph: N=169343, heavy-tailed symmetrised "
                               f"E_in={e_in} (+N self loops => E_eff={e_eff}), EGC-M d=128 H=8 B=4 "
                               "aggrs=sum+mean+max+symnorm, CSR cached",
                   "n_nodes": n, "e_in": e_in, "e_eff": e_eff, "layer": "EGConv", "parallelism": "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "egc::agg_fast_kernel (fused aggregate+combine, one launch)",
                     "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg_gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": agg_bytes, "launch_ms": agg_ms,
                     "frac_vs_measured_copy_ceiling": agg_gbs / HBM_COPY_CEILING_GBS,
                     "frac_incl_materialised_weightings": (agg_bytes + terms["weightings"]) / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "SURVEY.md 8(d): gather + col + rowptr + deg + out of the launch; the materialised "
                             "weightings it also reads are waste, not algorithmic bytes"},
        "kernels_ms": {"basis_gemm": gemm_ms, "aggregate_combine": agg_ms, "layer_forward": step_ms_events,
                       "basis_gemm_exact_fp32": gemm_exact_ms, "layer_forward_gemm_exact_fp32": gemm_exact_ms + agg_ms,
                       "basis_gemm_24bit": gemm_24bit_ms, "layer_forward_gemm_24bit": gemm_24bit_ms + agg_ms},
        "layer_algorithmic_bytes": terms["layer"],
        "layer_achieved_gbs": terms["layer"] / (step_ms_events * 1e-3) / 1e9,
        "layer_frac": terms["layer"] / (step_ms_events * 1e-3) / 1e9 / HBM_PEAK_GBS,
        # the precision / speed trade of the GEMM, on the line: the default (fp16x2: 22-bit operands, componentwise error 1.3e-7
        # against float64 -- below the fp32-MFMA kernel's 3.2e-7, tests/test_gemm_gpu.py), the 24-bit-operand form (bf16x3: what
        # std / var layers ran in rounds 2-3, on request since), the plain fp32-MFMA kernel
        "layer_frac_by_gemm": {"fp16x2_default": terms["layer"] / (step_ms_events * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "bf16x3_24bit": terms["layer"] / ((gemm_24bit_ms + agg_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "exact_fp32_mfma": terms["layer"] / ((gemm_exact_ms + agg_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "ms_per_step_regions": regions, "ms_per_step_median_of_5_regions": sorted(regions)[2],
    }

    # ---- a std layer at this size (VERDICT r3 #4c): the same layer with `mean` replaced by `std` -- the same algorithmic bytes.
    # Since round 4 such layers run the default GEMM (the variance is accumulated about the row's first entry: no cancellation
    # left to amplify the 22-bit operand split, profiles/r04_stdvar_shift.md); timed through the module on the cached graph.
    try:
        torch.manual_seed(args.seed + 7)
        conv_std = egc_amd.EGConv(F_IN, F_OUT, aggrs=["sum", "std", "max", "symnorm"], num_heads=HEADS, num_bases=BASES,
                                  cached=True).to(dev).eval()

        spec_s = conv_std._spec_coo
        wcat_s, bcat_s = conv_std._packed_weights()
        planes_s = pack_weights(spec_s, wcat_s)
        bias_s = conv_std.bias.detach()
        ws_s = torch.zeros(max(lib.egc_aggregate_workspace_bytes(C.byref(spec_s.c), n, e_in), 1), dtype=torch.uint8, device=dev)

        def std_step():   # the same C-ABI call as the headline's step (same shapes: the intermediates are shared)
            _C.check(lib.egc_layer_forward_packed(C.byref(g), C.byref(spec_s.c), x.data_ptr(), planes_s.data_ptr(),
                                                  bcat_s.data_ptr(), bias_s.data_ptr(), bases.data_ptr(), ldb,
                                                  weightings.data_ptr(), out.data_ptr(), ws_s.data_ptr(), ws_s.numel(),
                                                  stream), "egc_layer_forward_packed")
        for _ in range(20):
            std_step()
        std_ms = time_region_median(std_step, reps)
        result["std_layer"] = {"aggregators": "sum+std+max+symnorm", "gemm_flags": int(spec_s.gemm_flags),
                               "layer_ms": std_ms, "layer_frac": terms["layer"] / (std_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "formula": "variance evaluated as E[(x-s)^2] - (E[x]-s)^2 about the row's first entry s -- a DELIBERATE "
                                          "departure from the reference's float32 E[x^2] - E[x]^2 (layers.py:203-216): same number in exact "
                                          "arithmetic, no cancellation; such layers are held to 1e-4 against the reference-generated float32 "
                                          "fixtures and to 1e-5 against float64 (DESIGN.md section 4, profiles/r04_stdvar_shift.md)",
                               "note": "egc_layer_forward_packed on the cached CSR graph; same byte model as the headline layer"}
        gemm_only()
        agg_only()   # `out` and the intermediates back to the headline layer's
    except Exception as exc:   # (a side field must not take the line down)
        result["std_layer"] = {"error": repr(exc)}

    log("algorithmic bytes (SURVEY.md 8d), per forward:")
    for k, v in terms.items():
        log(f"  {k:18s} {v / 1e6:10.2f} MB")
    log(f"kernel ms: gemm {gemm_ms:.4f} (exact fp32: {gemm_exact_ms:.4f})  aggregate+combine {agg_ms:.4f}  layer {step_ms_events:.4f}")
    out_main = out.clone()
    measure_layer_config.no_cpu = bool(args.no_cpu_baseline)
    if not args.no_other_configs:
        log("other configs:")
        result["other_configs"] = other_configs(dev, args.seed)
    if not args.no_cpu_baseline:
        try:
            ref_out, cpu_s, threads, e_cached = cpu_baseline(ei_cpu, n, x_cpu, state_cpu)
            err = float((out_main.cpu() - ref_out).abs().max() / max(1.0, float(ref_out.abs().max())))
            log(f"cpu port: {cpu_s * 1e3:.1f} ms/forward on {threads} threads; HIP vs CPU port rel err {err:.2e}")
            result["cpu_baseline"] = {
                "value": e_eff / cpu_s, "unit": "edges/s", "cores": threads, "threads": threads,
                "host_cores": os.cpu_count(), "kind": "port",   # `cores` = the threads actually used (the contract's field)
                "sample": f"full config-2 layer forward (E_eff={e_cached}), gcn_norm cached; best of 8/16/32/64 "
                          f"torch threads, median of 5 runs each after a warm-up; {cpu_s * 1e3:.1f} ms per forward "
                          f"on a {os.cpu_count()}-core host (torch's scatter / index kernels stop scaling at 8-32 threads: in "
                          f"plain words, the host is used as an {threads}-core machine)",
                "hip_vs_port_rel_err": err}
        except Exception as ex:   # noqa: BLE001 -- the measured line is printed either way, with the failure in it
            log(f"cpu baseline failed: {ex!r}")
            result["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": "failed",
                                      "error": repr(ex)[:500]}
    emit(result)


if __name__ == "__main__":
    main()
