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
    nested-dissection key-frame order and reproduce the unsharded solve (same LM decisions, cost within 1e-9, poses within
    1e-8) both ways: with the per-rank segment factorisation (each rank eliminates its own segments, the separator block
    is all-reduced) and with the replicated factorisation of the all-reduced packed system (VO_BA_SEGMENTS=0)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "dist_worker.py"), "--backend", "gloo", "--case", "gba500"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=840, env=env, cwd=str(ROOT))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("-> OK") == 4 and "MISMATCH" not in out, out[-4000:]  # 2 ranks x (segments, replicated)


@pytest.mark.timeout(1500)
def test_bench_gpus_2_starts_two_ranks_and_reports_the_sharded_legs():
    """`python bench.py --gpus 2` without a launcher: bench.py itself starts two fresh ranks (before touching the GPU) and relays
    rank 0's JSON line -- n_gpus = 2, frames sharded without a collective, and under world > 1 the sharded config-3 leg, the
    replicas leg and the sharded config-4 global BA (payload of its all-reduce reported).  gloo: both ranks share GPU 0."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "64",
           "--no-bruteforce", "--no-single-stream"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1400, env=env, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["collective_backend"] == "gloo" and "RCCL has not been exercised" in d["rccl_note"]
    assert "replicas" in d["local_ba"] and "sharding" in d["local_ba"] and d["local_ba"]["lm_iters_per_s"] > 0
    g = d["global_ba"]
    assert g["lm_iters_per_s"] > 0 and 1.0 < g["allreduce_payload_MB"] < 15.0 and g["key_frame_order"]["parts"] > 1
    sg = g["segment_factorisation"]  # the same solve through the per-rank segment factorisation (opt-in form)
    assert sg["lm_iters_per_s"] > 0 and sg["first_separator_tile_column"] == 24 and sg["allreduce_payload_MB"] < g["allreduce_payload_MB"]
    assert sg["allreduce_calls_per_solve"] == 2 * g["allreduce_calls_per_solve"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    sc = d["ba_scaling"]  # the SCALE question answered inside the run: node aggregate over rank 0 alone, plus the sharded forms
    assert sc["n_gpus"] == 2 and sc["replicas"]["x_one_gpu"] > 0 and sc["replicas"]["one_gpu_lm_iters_per_s"] > 0
    for k in ("sharded_local_ba", "sharded_global_ba_replicated", "sharded_global_ba_segments"):
        assert sc[k]["x_one_gpu"] > 0, k


@pytest.mark.timeout(900)
def test_bench_rccl_process_group_of_one():
    """torch.distributed with the **nccl** backend (= RCCL) on the hardware, as a process group of ONE rank:
    `bench.py --dist-at-one-rank` runs the SHARDED forms of the config-3 and config-4 legs on handles with
    VO_BA_OPT_COLLECTIVES_AT_ONE_RANK, so that every collective of the LM loop is a real dist.all_reduce on RCCL through the
    `_allreduce` callback the multi-GPU run uses -- the torch-RCCL path has then run on the hardware before the 8-GPU run."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--backend", "nccl", "--dist-at-one-rank", "--steps", "2", "--warmup", "1",
           "--batch", "64", "--no-bruteforce", "--no-single-stream", "--no-cpu-baseline", "--collective-timeout", "120"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1
    s3 = d["local_ba"]["sharded_at_one_rank"]
    assert s3["backend"] == "nccl" and s3["allreduce_calls_per_solve"] >= 2 * d["local_ba"]["iterations_per_solve"]
    assert 0 < s3["ms_per_solve"] < 20 * d["local_ba"]["ms_per_solve"]
    g = d["global_ba_sharded_at_one_rank"]
    assert g["lm_iters_per_s"] > 0 and g["allreduce_calls_per_solve"] >= 2 and 1.0 < g["allreduce_payload_MB"] < 15.0
