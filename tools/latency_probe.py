"""Developer tool: one camera stream through vo_tracker (batch 1, host image + raw depth in, pose out): wall time per
frame and the per-stage device times (HIP events)."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
from vo_slam_test_amd import _lib as vo, synth
from vo_slam_test_amd.tracking import load_maps

W, H = 640, 480
cam5 = synth.CAM.astype(np.float32)
inv = float(np.float32(1.0) / np.float32(synth.DEPTH_SCALE))
img, dep = synth.make_frames(1, start=5), np.stack([synth.make_depth(5)])
ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
k, d = ext(img[0])
ext.close()
z = dep[0][np.clip(k["y"].astype(int), 0, H - 1), np.clip(k["x"].astype(int), 0, W - 1)].astype(np.float32) * np.float32(inv)
mp = synth.make_tracking_map(k["x"], k["y"], k["octave"], k["angle"], d, np.where(z > 0, z, -1).astype(np.float32), seed=0)
t = vo.Tracker(1, cam5, synth.DIST, W, H, max_last=1100, max_local=2200, inv_depth_scale=inv, single_stream=True)
load_maps(t, [mp], 1100, 2200)
h_img, h_dep = np.ascontiguousarray(img), np.ascontiguousarray(dep).view(np.uint16)
for timing in (False, True):
    t.set_timing(timing)
    lat = []
    for i in range(60):
        t0 = time.perf_counter()
        t.track(h_img, h_dep)
        r = t.results()
        lat.append(time.perf_counter() - t0)
    print("timing events", timing, ": median %.4f ms per frame, inliers %d, status %d" % (np.median(lat[10:]) * 1e3, r["n_inliers"][0], r["status"][0]))
    if timing:
        ms, n = t.get_timing()
        print({k: round(v / max(n, 1), 4) for k, v in ms.items()}, "sum %.4f" % (sum(ms.values()) / max(n, 1)))
t.close()
