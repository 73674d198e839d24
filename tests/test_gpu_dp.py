"""Data parallelism on the device: 2 ranks (gloo rendezvous, both on cuda:0) vs one process on the full batch.

Covers what the N > 1 path adds to every engine: row sharding, the single gradient all-reduce with
grad_scale = 1/N, the psum of the VQ EMA statistics, global-norm clipping on the REDUCED gradient, and
launch-plan replay with a collective inside the plan."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_equal_one_process_on_the_full_batch():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "dp_gpu_worker.py")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-12000:]
    assert "DP-GPU-OK" in out.stdout


def test_rccl_one_rank_rehearsal():
    """the N > 1 code path with the real backend: a 1-rank RCCL communicator on cuda:0 and PM_FORCE_DP=1 (asynchronous
    bucketed all-reduce on the communication stream, inside the replayed launch plan); trajectories equal the plain runs"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")], capture_output=True, text=True,
                         env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "RCCL-1RANK-OK" in out.stdout
