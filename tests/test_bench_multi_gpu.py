"""bench.py's multi-GPU line, functionally, on the one GPU a test box has: (i) two gloo ranks sharing the GPU run all
three strong-scaling workloads (config 2 arxiv, config 5 homogeneous mag, the typed mag graph through the partitioned
REGConv) at a reduced scale and print ONE line that carries all of them, each with its own single-GPU time, halo
statistics and predicted exchange time; (ii) the same code path with RCCL (backend nccl) at world size 1 -- process-group
set-up with a device id, the set-up all-to-alls, the per-layer all-to-all-v -- which is all of the RCCL path a one-GPU
box can execute (the 2 / 4 / 8-GPU curve is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, port, env_extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3",
           "--warmup", "1"]
    env = dict(os.environ, EGC_BENCH_SCALE="0.05", **env_extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 6000, len(lines[0])      # the driver keeps a bounded tail of stdout
    return json.loads(lines[0])


def _check_line(line, world, partitioned):
    assert line["n_gpus"] == world and line["unit"] == "edges/s" and line["value"] > 0
    assert "ogbn-arxiv" in line["metric"] and "ogbn-arxiv" in line["config"]["workload"]
    assert line["scaling"] == ("strong" if world > 1 else None)
    ss = line["strong_scaling"]
    assert set(ss) == {"config2_arxiv", "config5_mag_homogeneous", "config5_rmag_typed"}
    assert line["value"] == ss["config2_arxiv"]["value"] and line["ms_per_step"] == ss["config2_arxiv"]["ms_per_step"]
    for key, rec in ss.items():
        assert "error" not in rec, (key, rec)
        assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["t1_ms"] > 0
        if partitioned:
            ex = rec["exchange"]
            assert len(ex["halo_rows_per_rank"]) == world and ex["row_bytes"] % 16 == 0
            assert ex["max_peer_bytes"] == ex["max_peer_rows"] * ex["row_bytes"]
            # (the line carries six significant digits: bench.compact_line)
            assert abs(ex["predicted_exchange_ms"] - ex["max_peer_bytes"] / 153e9 * 1e3) <= 1e-5 * ex["predicted_exchange_ms"] + 1e-12
            assert ex["measured_exchange_alone_ms_rank0"] >= 0
            if world > 1:
                assert ex["max_peer_rows"] > 0 and sum(ex["halo_rows_per_rank"]) > 0
    assert ss["config5_rmag_typed"]["layer"] == "REGConv" and len(ss["config5_rmag_typed"]["entries_per_rank"]) == world
    assert sum(ss["config5_rmag_typed"]["entries_per_rank"]) == ss["config5_rmag_typed"]["entries_total"]


def test_all_three_workloads_two_gloo_ranks_on_one_gpu():
    line = _run(2, 29551, {"EGC_BENCH_BACKEND": "gloo"})
    _check_line(line, 2, True)


def test_all_three_workloads_rccl_world_size_one():
    line = _run(1, 29553, {"EGC_BENCH_FORCE_PARTITION": "1"})
    _check_line(line, 1, True)


def test_bench_starts_its_own_ranks_when_given_gpus_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK in the environment (the form the N = 1 driver command has): bench.py spawns
    the two ranks itself as children of a process that never touched the GPU, and relays rank 0's one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EGC_BENCH_SCALE="0.05", EGC_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    _check_line(line, 2, True)
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
