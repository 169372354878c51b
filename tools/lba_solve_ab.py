"""Developer tool (GPU box): the local-BA solve alone (config 3: set_state + vo_ba_local_ba through a pre-created handle) for the
library named by VO_HIP_LIB; median / min of 60.  Used to A/B libraries of different rounds."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402
lb = synth.make_lba_problem(0)
ba = vo.BundleAdjuster(lb)
_, sums, _ = ba.local_ba()
its = sum(int(s.iterations) for s in sums)
t = []
for _ in range(64):
    ba.set_state(lb["poses"], lb["points"])
    t0 = time.perf_counter(); ba.local_ba(); t.append((time.perf_counter() - t0) * 1e3)
t = t[4:]
print(f"solve: median {np.median(t):.4f} ms, min {np.min(t):.4f} ms, {its} LM iterations -> {its / np.median(t) * 1e3:.0f} it/s")
