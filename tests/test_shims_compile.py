"""Boundary compile check: the C++ shims of include/myslam_shim/ (the reference's class names and signatures over
the C-ABI) are parsed by g++ -fsyntax-only against declaration-only stubs of the cv:: / Eigen:: / Sophus:: / DBoW3::
and reference types they touch (tests/shim_stubs/).  Catches typos, missing headers and signature drift in this
image, which has none of those libraries; it pins no parity and links nothing."""
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
STUBS = ROOT / "tests" / "shim_stubs"


@pytest.mark.parametrize("tu", ["orbextractor", "matcher", "optimizer", "mappoint", "frame", "sim3solver", "localmapping", "map"])
def test_shim_parses(tu):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    # gnu++14 like the reference's own build (its KeyFrameAndPose typedef, loopClosing.h, pairs std::map with an
    # allocator of a different value_type, which strict ISO mode rejects)
    r = subprocess.run([gxx, "-std=gnu++14", "-fsyntax-only", "-Wall", "-Werror", f"-I{STUBS}", f"-I{STUBS / 'thirdparty'}", f"-I{ROOT / 'include'}",
                        str(STUBS / f"tu_{tu}.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_header_is_plain_c():
    """include/vo_hip.h is a C header: parses as C99 with -pedantic"""
    gcc = shutil.which("gcc")
    src = "#include \"vo_hip.h\"\nint main(void) { return VO_OK; }\n"
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT / 'include'}", "-x", "c", "-"],
                       input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shim_run_driver_builds_here():
    """tests/shim_run/shim_run.cpp (the driver that EXECUTES the shims on the GPU box, tests/test_gpu_shims_run.py) compiles
    against its functional stand-ins and links with libvo_hip.so in this container too -- no GPU needed for that"""
    import tempfile
    from vo_slam_test_amd import build
    gxx = shutil.which("g++")
    so = build.build()
    run = ROOT / "tests" / "shim_run"
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([gxx, "-std=gnu++14", "-O0", "-Wall", "-Werror", f"-I{run}", f"-I{run / 'thirdparty'}", f"-I{ROOT / 'include'}",
                            str(run / "shim_run.cpp"), f"-L{so.parent}", "-lvo_hip", f"-Wl,-rpath,{so.parent}", "-L/opt/rocm/lib",
                            "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", f"{d}/shim_run"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
