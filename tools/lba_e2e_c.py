"""Developer tool (GPU box): the three C entry points of the per-key-frame local BA (vo_ba_reset, vo_ba_local_ba,
vo_ba_get_state) on arguments marshalled once -- what the C++ shim pays -- timed one by one; median of 200."""
import sys, pathlib, time, ctypes
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402
lb = synth.make_lba_problem(0)
keep = vo.BundleAdjuster(lb); keep.local_ba()
L = vo.lib()
a = {k: np.ascontiguousarray(v) for k, v in lb.items() if isinstance(v, np.ndarray)}
rargs = (keep._h, len(lb["poses"]), vo._p(a["poses"]), vo._p(a["fixed"]), len(lb["points"]), vo._p(a["points"]), len(lb["e_cam"]),
         vo._p(a["e_cam"]), vo._p(a["e_pt"]), vo._p(a["e_obs"]), vo._p(a["e_inv_sigma"]), vo._p(a["cam"]))
erase = np.zeros(len(lb["e_cam"]), np.uint8); sums = (vo.LmSummary * 2)()
po, px = np.zeros((len(lb["poses"]), 6)), np.zeros((len(lb["points"]), 3))
largs = (keep._h, None, vo._p(erase), ctypes.byref(sums)); gargs = (keep._h, vo._p(po), vo._p(px))
T = []
for _ in range(220):
    t0 = time.perf_counter(); L.vo_ba_reset(*rargs); t1 = time.perf_counter(); L.vo_ba_local_ba(*largs); t2 = time.perf_counter()
    L.vo_ba_get_state(*gargs); t3 = time.perf_counter()
    T.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
T = np.array(T[20:]) * 1e3
m = np.median(T, axis=0)
print(f"vo_ba_reset {m[0]:.4f} ms, vo_ba_local_ba {m[1]:.4f} ms, vo_ba_get_state {m[2]:.4f} ms, per call {m[3]:.4f} ms (min {T[:,3].min():.4f})")
