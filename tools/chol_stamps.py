"""Developer tool (GPU box): phase timeline of the critical Cholesky tasks from a -DVO_CHOL_STAMPS build.
usage: tools/chol_stamps.py [n]   (n x n SPD system, default 2994)"""
import ctypes as C, pathlib, subprocess, sys
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2994
out = ROOT / "tools" / "_stamp" / "libvo_chol_stamps.so"
out.parent.mkdir(exist_ok=True)
srcs = ["vo_common.hip", "chol.hip", "pose_graph.hip"]
objs = []
for s in srcs:
    o = out.parent / (s + ".o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-DVO_CHOL_STAMPS",
                    "-c", str(ROOT / "vo_slam_test_amd" / "csrc" / s), "-o", str(o)], check=True)
    objs.append(str(o))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *objs], check=True)
import torch  # noqa
L = C.CDLL(str(out))
rng = np.random.default_rng(0)
M = rng.normal(size=(n, n)) / np.sqrt(n)
A = M @ M.T + np.eye(n)
b = rng.normal(size=n)
Al, x = np.ascontiguousarray(np.tril(A)), b.copy()
st = np.zeros(2 * 64 * 16, np.uint64)
L.vo_chol_debug_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
for rep in range(3):
    Al, x = np.ascontiguousarray(np.tril(A)), b.copy()
    rc = L.vo_chol_debug_solve(n, Al.ctypes.data, x.ctypes.data, st.ctypes.data, len(st))
print("rc", rc, "residual", np.abs(A @ x - b).max())
m = (n + 63) // 64
S = st.reshape(-1, 16)[:2 * m].astype(np.int64)
t0 = S[0, 6] if S[0, 6] else S[S[:, 6] > 0, 6].min()
print("stamps in us relative to the first diagonal tile's publication (100 MHz counter)")
print("col | diag: wait_last  seen  mfma_done  chol_done  published | sub: wait_last seen mfma_done diag_seen loaded trsm_done published")
for j in range(0, m, max(1, m // 12)):
    d, u = S[2 * j], S[2 * j + 1]
    f = lambda v: "%8.1f" % ((v - t0) / 100.0) if v else "       -"
    print("%3d | " % j + " ".join(f(d[k]) for k in (0, 1, 2, 3, 6)) + " | " + " ".join(f(u[k]) for k in (0, 1, 2, 3, 4, 5, 6)))
pub = S[0:2 * m:2, 6]
print("mean column period %.1f us; factorisation %.1f us; backward substitution done %.1f us after the last diagonal tile" % (
    np.diff(pub[pub > 0]).mean() / 100.0, (pub.max() - t0) / 100.0, (S[0, 10] - pub.max()) / 100.0))
