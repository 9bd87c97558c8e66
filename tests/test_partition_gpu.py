"""Training on a vertex partition (forward halo exchange + reverse exchange of the halo rows of d_bases):
two gloo ranks sharing the GPU must reproduce the single-device gradients (tests/partition_train_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_training_step_matches_single_device():
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(here, "partition_train_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_the_rccl_code_path_at_world_size_one():
    """One rank, backend nccl (= RCCL): process-group set-up with a device id, the all-to-alls of build_distributed, the
    halo exchange calls of the forward and the backward and the gradient all-reduce all run through RCCL -- with no
    peer to talk to, which is all a one-GPU box can offer (the 2 / 4 / 8-GPU runs are the driver's)."""
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(here, "partition_train_worker.py")]
    env = dict(os.environ, EGC_TEST_BACKEND="nccl")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
