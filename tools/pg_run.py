"""Developer tool: config-4-size pose graph on the device, with timing."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402

from vo_slam_test_amd import _lib, synth  # noqa: E402

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = synth.make_pose_graph(7, n_kf=n_kf, drift=0.004, extra_edges=4)
_lib.Optimizer.solvePoseGraphLoop(synth.make_pose_graph(0, n_kf=12))
t0 = time.perf_counter()
q, t, s = _lib.Optimizer.solvePoseGraphLoop(g)
dt = time.perf_counter() - t0
const = 0.5 * len(g["e_i"])
print(f"{n_kf} KF, {len(g['e_i'])} edges: iterations {s.iterations} accepted {s.accepted} termination {s.termination} "
      f"cost {s.initial_cost - const:.4g} -> {s.final_cost - const:.4g} (+ {const} constant) in {dt * 1e3:.1f} ms")
print("max translation error before/after", np.abs(g["trans"] - g["true_trans"]).max(), np.abs(t - g["true_trans"]).max())
