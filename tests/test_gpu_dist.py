"""The N > 1 bundle-adjustment path with a REAL process group on the GPU: two fresh child processes (gloo, both
on GPU 0 of a one-GPU box) each own one point shard; the sharded LM loop runs inside the C-ABI with the all-reduce
supplied as a callback (vo_ba_set_allreduce).  Every rank asserts equality with the unsharded device solve:
identical LM decisions and erase masks, poses within 1e-9 / 1e-8."""
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_process_sharded_local_ba_through_the_c_abi():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "dist_worker.py"), "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=str(ROOT))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("-> OK") == 4 and "MISMATCH" not in out, out[-4000:]


@pytest.mark.timeout(900)
def test_two_process_sharded_global_ba_500_keyframes():
    """BASELINE config 4's reduced system (500 key-frames, 3000 x 3000) sharded over two ranks: both derive the same
    nested-dissection key-frame order, sum their parts of the reduced system in the Cholesky storage, and reproduce the
    unsharded solve (same LM decisions, cost within 1e-9, poses within 1e-8)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "dist_worker.py"), "--backend", "gloo", "--case", "gba500"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=840, env=env, cwd=str(ROOT))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("-> OK") == 2 and "MISMATCH" not in out, out[-4000:]
