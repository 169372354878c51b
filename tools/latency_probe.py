"""Developer tool: single-frame latency of the host-buffer entry points (what a tracking thread sees)."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402

from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
f0 = synth.make_frame(0)
f1, dx, dy = synth.make_shifted(f0, 0)
for _ in range(5):
    k0, d0 = ext(f0)
ts = []
for i in range(50):
    t0 = time.perf_counter()
    k1, d1 = ext(f1 if i & 1 else f0)
    ts.append(time.perf_counter() - t0)
print(f"vo_orb_extract (640x480 host image in, {len(k1)} key-points + descriptors out): median {np.median(ts) * 1e3:.3f} ms, "
      f"min {np.min(ts) * 1e3:.3f} ms")
ts = []
for i in range(50):
    t0 = time.perf_counter()
    D = vo.hamming_matrix(d0, d1)
    ts.append(time.perf_counter() - t0)
print(f"vo_hamming_matrix ({len(d0)} x {len(d1)}, host in/out): median {np.median(ts) * 1e3:.3f} ms")
pr = synth.make_pose_problem(0)
vo.Optimizer.solvePoseOnlySE3([pr])
ts = []
for i in range(50):
    t0 = time.perf_counter()
    vo.Optimizer.solvePoseOnlySE3([pr])
    ts.append(time.perf_counter() - t0)
print(f"vo_pose_only_solve (1 frame x {len(pr['pts'])} obs, host in/out): median {np.median(ts) * 1e3:.3f} ms")
# guided matchers (what tracking calls once or twice per frame)
k0, d0 = ext(f0)
k1, d1 = ext(f1)
sf = np.array([1.2 ** i for i in range(8)], np.float32)
rng = np.random.default_rng(0)
z = rng.uniform(0.8, 4.5, len(k1)).astype(np.float32)
ur1 = (k1["x"] - np.float32(40.0) / z).astype(np.float32)
cur = vo.FrameArrays(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
q = dict(flags=np.full(len(k0), 3, np.uint8), u=(k0["x"] + dx).astype(np.float32), v=(k0["y"] + dy).astype(np.float32),
         invz=np.full(len(k0), 0.5, np.float32), octave=k0["octave"].astype(np.int32), angle=k0["angle"].astype(np.float32),
         desc=np.ascontiguousarray(d0))
m = vo.Matcher(0.8)
ts = []
for i in range(30):
    t0 = time.perf_counter()
    n, assigned = m.searchByProjection_frame(cur, q, 15.0, 40.0, 0, True, sf)
    ts.append(time.perf_counter() - t0)
print(f"vo_match_frame_projection ({len(k0)} map points -> {len(k1)} features, {n} matches): median {np.median(ts) * 1e3:.3f} ms")
