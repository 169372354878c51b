"""RCCL carrying the collectives of the sharded LM loop on a real GPU -- as far as a ONE-GPU box allows: RCCL refuses two
ranks on one device, so examples/rccl_sharded_ba.cpp (the C++ host a deployment would use: ncclCommInitRank,
vo_ba_set_allreduce with ncclAllReduce(ncclDouble, ncclSum) on the handle's stream) runs as a single rank with
`--collectives-at-one-rank` (vo_ba_set_option(h, VO_BA_OPT_COLLECTIVES_AT_ONE_RANK, 1)).  The library then runs the sharded form of its loop (partial-sum payloads, k_ba_reduce,
two callbacks per LM iteration, the closing merge) on the one shard, every collective is a real ncclAllReduce enqueued
on the BA stream between the loop's kernels, and the result must equal the plain one-GPU solve.  What this does NOT show:
bytes crossing xGMI (world > 1) -- that is the driver's multi-GPU run."""
import os
import pathlib
import re
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.mark.timeout(900)
def test_rccl_host_runs_the_sharded_loop_on_one_rank(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not pathlib.Path(hipcc).exists() or not pathlib.Path("/opt/rocm/include/rccl/rccl.h").exists():
        pytest.skip("hipcc / rccl.h not present")
    from vo_slam_test_amd import build
    so = build.build()
    exe = tmp_path / "rccl_sharded_ba"
    cmd = [hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", str(ROOT / "examples" / "rccl_sharded_ba.cpp"),
           f"-I{ROOT / 'include'}", f"-L{so.parent}", "-lvo_hip", "-L/opt/rocm/lib", "-lrccl", f"-Wl,-rpath,{so.parent}",
           "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    prob = tmp_path / "problem.bin"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "dump_ba_problem.py"), str(prob)], capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0 and prob.exists(), r.stderr[-2000:]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", VO_NCCL_ID_FILE=str(tmp_path / "id"),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([str(exe), str(prob), "--collectives-at-one-rank"], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "RCCL all-reduce self-check on the BA stream ok" in out
    m = re.search(r"(\d+) LM iterations in .* (\d+) RCCL all-reduces in 3 solves", out)
    assert m, out[-2000:]
    iters, calls = int(m.group(1)), int(m.group(2))
    assert iters >= 2 and calls >= 3 * 2 * iters  # two per LM iteration (+ the closing merges are skipped at one rank)
    m = re.search(r"max \|pose difference\| ([0-9.e+-]+), erase masks (\w+)", out)
    assert m and float(m.group(1)) < 1e-9 and m.group(2) == "identical", out[-2000:]
