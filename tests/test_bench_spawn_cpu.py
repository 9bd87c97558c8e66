"""bench.py --gpus N without a launcher: the self-spawn path, as far as a box without a GPU can take it -- the children
are started under torch.distributed.run, fail loudly (no MI355X here: there is no CPU fallback), and the non-zero exit
code comes back through the parent, which itself never initialises the GPU."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU form of this test is tests/test_bench_multi_gpu.py")
def test_spawned_ranks_failure_is_propagated():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert "No HIP GPUs are available" in r.stderr or "needs an MI355X" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_line_is_compact_and_keeps_the_contract(tmp_path, capsys, monkeypatch):
    """The driver keeps a bounded tail of stdout (round 5's 23 KB line came back unparsed): the printed line stays under
    6 KB whatever the side records hold, carries every contract key with `roofline` and `cpu_baseline`, and the full record
    goes to the detail file."""
    import json
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))      # a real full record (23 KB)
    assert len(json.dumps(full)) > 20000
    # ... made worse: more side records, longer prose
    for i in range(12):
        full["other_configs"][f"extra_{i}"] = dict(full["other_configs"]["config4_cifar_b2048"], workload="w" * 400)
    full["roofline"]["note"] = "n" * 3000
    monkeypatch.setenv("EGC_BENCH_DETAIL", str(tmp_path / "detail.json"))
    bench.emit(full)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == full["metric"] and abs(line["value"] / full["value"] - 1) < 1e-5
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
    for rec in line["other_configs"].values():
        assert set(rec) <= {"workload", "layer_ms", "step_ms", "eager_step_ms", "hipgraph_replay_ms", "eager_ms",
                            "coo_hipgraph_replay_ms", "csr_path_step_ms", "speedup_vs_csr_path", "frac", "layer_frac", "traffic",
                            "cpu_edges_per_s", "cpu_cores", "error"}
    detail = json.load(open(tmp_path / "detail.json"))
    assert detail["roofline"]["note"] == "n" * 3000 and len(detail["other_configs"]) == len(full["other_configs"])
