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
