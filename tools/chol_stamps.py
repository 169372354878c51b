"""Developer tool (GPU box): phase timeline of the critical Cholesky tasks from a -DVO_CHOL_STAMPS build.
usage: tools/chol_stamps.py [n] [parts]   (n x n SPD system, default 2994; parts > 0: a cyclic band of half-width 288
(the reduced camera system of BASELINE config 4) in a nested-dissection order of `parts` segments, as ba.hip picks it)"""
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
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-DVO_CHOL_STAMPS", *(["-D" + a for a in sys.argv[3:]]),
                    "-c", str(ROOT / "vo_slam_test_amd" / "csrc" / s), "-o", str(o)], check=True)
    objs.append(str(o))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *objs], check=True)
import torch  # noqa
L = C.CDLL(str(out))
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0)
if parts == 0:
    M = rng.normal(size=(n, n)) / np.sqrt(n)
    A = M @ M.T + np.eye(n)
else:
    nf, w = n // 6, 48
    idx = np.arange(nf)
    d = np.abs(idx[:, None] - idx[None, :])
    keep = np.minimum(d, nf - d) <= w                         # covisible key-frames
    M = rng.normal(size=(n, n))
    A = (M + M.T) * np.kron(keep, np.ones((6, 6)))[:n, :n]
    A += (np.abs(A).sum(1).max() + 1.0) * np.eye(n)
    in_segs = nf - parts * w
    base = in_segs // parts // 32 * 32
    order, seps = [], []
    pos = 0
    for g in range(parts):
        ln = base if g < parts - 1 else in_segs - base * (parts - 1)
        order += list(range(pos, pos + ln)); pos += ln
        seps.append(list(range(pos, pos + w))); pos += w
    rank = [((g + 1) & -(g + 1)).bit_length() for g in range(parts)]
    for g in sorted(range(parts), key=lambda g: rank[g]):
        order += seps[g]
    rows = (6 * np.array(order)[:, None] + np.arange(6)[None, :]).ravel()
    rows = np.concatenate([rows, np.arange(6 * nf, n)])
    A = A[np.ix_(rows, rows)]
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
for j in range(0, m, max(1, m // 12) if parts == 0 else 1):
    d, u = S[2 * j], S[2 * j + 1]
    f = lambda v: "%8.1f" % ((v - t0) / 100.0) if v else "       -"
    print("%3d | " % j + " ".join(f(d[k]) for k in (0, 1, 2, 3, 6)) + " | " + " ".join(f(u[k]) for k in (0, 1, 2, 3, 4, 5, 6)))
dd = S[0:2 * m:2]
ok = (dd[:, 14] > 0) & (dd[:, 13] > 0)
if ok.any():
    q = dd[ok]
    print("diagonal tile, panel 1 (mean over columns): pivots %.2f us, panel stores + trailing MFMA %.2f us, store drain + barrier + flag %.2f us; "
          "tile: accumulators->LDS to first panel flagged %.2f us" % (
              (q[:, 11] - q[:, 14]).mean() / 100, (q[:, 12] - q[:, 11]).mean() / 100, (q[:, 13] - q[:, 12]).mean() / 100,
              (q[:, 14] - q[:, 2]).mean() / 100))
pub = S[0:2 * m:2, 6]
print("backward chain: started %.1f us, reached column 0 at %.1f us, done %.1f us" % tuple((S[0, k] - t0) / 100.0 for k in (8, 9, 10)))
print("backward chain: %.1f us waiting for requested tiles at the top of the steps, %d steps took the slow path (far link late)" % (S[1, 8] / 100.0, S[1, 9]))
print("mean column period %.1f us; factorisation %.1f us; backward substitution done %.1f us after the last diagonal tile" % (
    np.diff(pub[pub > 0]).mean() / 100.0, (pub.max() - t0) / 100.0, (S[0, 10] - pub.max()) / 100.0))
