"""Developer tool: phase breakdown of k_ba_solve from s_memrealtime stamps.

Builds a VO_BA_STAMPS variant of the library into tools/_stamp/libvo_stamp.so (git-ignored, ships
with gpurun), then on the GPU box:  python tools/ba_stamps.py run
"""
import ctypes as C
import pathlib
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "_stamp" / "libvo_stamp.so"
sys.path.insert(0, str(ROOT))


def build():
    from vo_slam_test_amd import build as B
    OUT.parent.mkdir(exist_ok=True)
    objs = []
    for src, extra in B.SOURCES:
        o = OUT.parent / (src + ".ba.o")
        subprocess.run([B.hipcc(), *B.COMMON, *extra, "-DVO_BA_STAMPS", "-c", str(B.CSRC / src), "-o", str(o)], check=True)
        objs.append(str(o))
    subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(OUT), *objs, "-lz"], check=True)
    print("built", OUT)


def run():
    import torch  # noqa: F401  (HIP runtime first)
    from vo_slam_test_amd import _lib, synth
    _lib.SO = OUT
    L = _lib.lib()
    pr = synth.make_lba_problem(0)
    ba = _lib.BundleAdjuster(pr)
    ba.local_ba()
    st = np.zeros(64, np.uint64)
    L.vo_ba_debug_stamps(ba._h, st.ctypes.data_as(C.c_void_p))
    d = np.diff(st[:8].astype(np.int64)) / 100.0  # s_memrealtime ticks at 100 MHz -> us
    rel = lambda i: (int(st[i]) - int(st[16])) / 100.0
    print("gemm block 0 (us from its entry): hinv table %.2f, MFMA loop done %.2f, end %.2f; solve kernel starts at %.2f"
          % (rel(17), rel(18), rel(20), rel(0)))
    print("  gemm, latest over all tile blocks (us from block 0 entry): MFMA loop done %.2f, last-arriver ticket %.2f, slab sums stored %.2f" % (rel(23), rel(22), rel(21)))
    print("  camera role, latest block end %.2f; solve kernel end %.2f; backsub block 0 entry %.2f (all from gemm block 0 entry)" % (rel(35), rel(7), rel(36)))
    print("  camera role, latest block start %.2f, latest wave edges done %.2f, latest block sum done %.2f" % (rel(42), rel(43), rel(44)))
    print("  camera role, first block (us from gemm block 0 entry): start %.2f, edges done %.2f, block sum done %.2f" % (rel(32), rel(33), rel(34)))
    rb = lambda i: (int(st[i]) - int(st[36])) / 100.0
    print("  back-substitution kernel, block 0 (us from its entry): point steps %.2f, candidate linearised %.2f, block sums %.2f; last block: ticket %.2f, update done %.2f" % (rb(37), rb(38), rb(39), rb(40), rb(41)))
    print("  shader clock during the solve: %.0f MHz" % ((int(st[31]) - int(st[30])) / ((int(st[7]) - int(st[0])) / 100.0)))
    print("  LDLt phases summed over the block columns (us): diag loads %.2f, ldl6 %.2f, panel %.2f, barriers %.2f, trailing %.2f" % tuple(int(st[48 + i]) / 100.0 for i in range(5)))
    print("  G prefetch +%.2f, pose prefetch +%.2f" % ((int(st[10]) - int(st[0])) / 100.0, (int(st[11]) - int(st[0])) / 100.0))
    print(f"  prefetch drained at +{(int(st[8]) - int(st[0])) / 100.0:.2f} us, slab sums done at +{(int(st[9]) - int(st[0])) / 100.0:.2f} us")
    names = ["slab sums+scale", "assemble+gmax", "LDLt", "back-subst", "dots", "cand poses", "block_sum"]
    for nm, v in zip(names, d):
        print(f"{nm:18s} {v:8.2f} us")
    print(f"{'total':18s} {(int(st[7]) - int(st[0])) / 100.0:8.2f} us")
    ba.close()


if __name__ == "__main__":
    build() if len(sys.argv) < 2 or sys.argv[1] == "build" else run()
