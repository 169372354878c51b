"""Build libvo_hip.so (the C-ABI product library) in-tree with hipcc for gfx950.

`python -m vo_slam_test_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "libvo_hip.so"
ARCH = "gfx950"

# (source, extra flags).  orb/match decide integer results from float expressions and must not be
# FMA-contracted (x86-64 reference build has no FMA); the FP64 BA kernels may contract.
SOURCES = [
    ("vo_common.hip", []),
    ("orb.hip", ["-ffp-contract=off"]),
    ("match.hip", ["-ffp-contract=off"]),
    ("guided.hip", ["-ffp-contract=off"]),
    ("track.hip", ["-ffp-contract=off"]),
    ("tracker.hip", ["-ffp-contract=off"]),
    ("ba.hip", ["-ffp-contract=fast"]),
    ("pose_graph.hip", ["-ffp-contract=fast"]),
    ("chol.hip", ["-ffp-contract=fast"]),
    ("loop.hip", ["-ffp-contract=off"]),
    ("dataset_io.cpp", []),
]
COMMON = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and pathlib.Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "vo_hip.h"]
    return any(d.stat().st_mtime > t for d in deps if d.exists())


def build(force: bool = False, verbose: bool = False) -> pathlib.Path:
    if not force and not needs_build():
        return OUT
    cc = hipcc()
    objdir = PKG / "_obj"
    objdir.mkdir(exist_ok=True)
    objs = []
    for src, extra in SOURCES:
        s = CSRC / src
        if not s.exists():
            continue
        o = objdir / (s.stem + ".o")
        objs.append(str(o))
        # per-file rebuild: an object is current if it is newer than its source and every header / .inc of csrc/ and include/
        deps = [s, PKG.parent / "include" / "vo_hip.h", pathlib.Path(__file__)] + [d for d in CSRC.glob("*") if d.suffix in (".h", ".inc")]
        if not force and o.exists() and all(o.stat().st_mtime > d.stat().st_mtime for d in deps if d.exists()):
            continue
        cmd = [cc, *COMMON, *extra, "-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(OUT), *objs, "-lz"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
