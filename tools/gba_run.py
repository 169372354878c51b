"""Developer tool: config-4-size global BA (500 key-frames x 50k points) through the large-system path."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402

from vo_slam_test_amd import _lib, synth  # noqa: E402

import os  # noqa: E402
if os.environ.get('VO_PAIRS'):
    _lib.set_option(4, int(os.environ['VO_PAIRS']))  # VO_OPT_BA_PAIRS_KERNEL
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
t0 = time.perf_counter()
pr = synth.make_global_ba_problem(0, n_kf=n_kf, n_pts=n_pts)
deg = np.bincount(pr["e_pt"], minlength=n_pts)
print(f"problem: {n_kf} KF, {n_pts} points, {len(pr['e_cam'])} edges, obs/point mean {deg[deg > 0].mean():.1f} max {deg.max()} "
      f"(generated in {time.perf_counter() - t0:.1f} s)")
t0 = time.perf_counter()
ba = _lib.BundleAdjuster(pr)
s = ba.solve(0.0, 0.0, 1)          # builds the device structures, one iteration
print(f"create + first iteration: {time.perf_counter() - t0:.2f} s; key-frame order {ba.debug_order()}")
ba.set_state(pr["poses"], pr["points"])
for its in (5, 10):
    ba.set_state(pr["poses"], pr["points"])
    t0 = time.perf_counter()
    s = ba.solve(float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815))), its)
    dt = time.perf_counter() - t0
    print(f"{s.iterations} LM iterations ({s.accepted} accepted, termination {s.termination}) in {dt * 1e3:.1f} ms = "
          f"{s.iterations / dt:.1f} iters/s; cost {s.initial_cost:.6g} -> {s.final_cost:.6g}")
poses, pts = ba.state()
print("pose error before/after", np.abs(pr["poses"] - pr["poses_true"]).max(), np.abs(poses - pr["poses_true"]).max())
ba.close()
