"""CPU: the RCCL driver of the sharded BA (examples/rccl_sharded_ba.cpp: vo_ba_set_allreduce with
ncclAllReduce(ncclDouble, ncclSum) on the handle's stream) compiles against /opt/rocm/include/rccl/rccl.h and links
against librccl and libvo_hip.so.  Compile + link only: running it needs GPUs (the driver's multi-GPU node);
bench.py --gpus N --backend nccl is the RCCL path the driver measures."""
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_rccl_driver_compiles_and_links(tmp_path):
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
    # the binary imports the collective and the C-ABI it registers it with
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True).stdout
    for sym in ("ncclAllReduce", "ncclCommInitRank", "vo_ba_set_allreduce", "vo_ba_set_shard", "vo_ba_local_ba"):
        assert sym in nm, sym
    # without an argument it prints its usage and exits before touching a device
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
