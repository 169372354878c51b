#!/usr/bin/env python3
"""Fixture g10_tracking.npz (round 2, regenerated in round 3 with the culling step and Frame::isInFrame in the path):
one synthetic RGB-D frame through the tracked-frame path of the oracle (tests/track_ref.py) --
key-points, undistorted coordinates, uRight / depth (frame.cpp:36-133), the searchByProjection assignments against the
last frame's points (matcher.cpp:18-148), the first pose-only solve, cullingOutliersBeforeLocalMap
(visualOdometry.cpp:864-886), Frame::isInFrame of the local map points with the refined pose (frame.cpp:145-190), the
search against them (:274-353), the second solve and trackLocalMap's inlier count -- plus a batch of
Sim3 RANSAC hypotheses (sim3Solver.cpp:98-269) and a median-descriptor query (mappoint.cpp:118-179).  Inputs AND
expected outputs; the oracle is the source (the reference ships no vectors: "parity unpinned").

    python tests/golden/make_g10_tracking.py
"""
import ctypes as C
import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import oracle_lib as O  # noqa: E402
from vo_slam_test_amd import synth  # noqa: E402

IDX, W, H = 77, 640, 480


def main():
    out = {}
    img, raw = synth.make_frame(IDX), synth.make_depth(IDX)
    inv = np.float32(1.0) / np.float32(synth.DEPTH_SCALE)
    cam5 = synth.CAM.astype(np.float32)
    p = O.orb_params()
    sf = np.array(list(p.scale)[:8], np.float32)
    k, d, _ = O.extract(p, img)
    n = len(k)
    x, y = np.ascontiguousarray(k["x"]), np.ascontiguousarray(k["y"])
    ux, uy = np.zeros(n, np.float32), np.zeros(n, np.float32)
    O.lib().orc_undistort_points(n, x, y, cam5[:4].copy(), synth.DIST.ctypes.data, ux, uy)
    dimg = np.zeros((H, W), np.float32)
    O.lib().orc_depth_to_float(np.ascontiguousarray(raw).reshape(-1), H * W, float(inv), dimg.reshape(-1))
    ur, dep = np.zeros(n, np.float32), np.zeros(n, np.float32)
    O.lib().orc_find_depth(n, x, y, ux, dimg, W, H, W, float(cam5[4]), ur, dep)
    T, pose6, la, lo = synth.make_tracking_map(ux, uy, k["octave"], k["angle"], d, dep, seed=IDX)
    from track_ref import track_frame
    w = track_frame(O, k, d, ux, uy, ur, T, pose6, la, lo, cam5, sf, W, H)
    out.update(image=img, depth_raw=raw, inv_depth_scale=np.float32(inv), cam5=cam5, dist=synth.DIST,
               kp_x=x, kp_y=y, kp_octave=k["octave"], kp_angle=k["angle"], desc=d, ux=ux, uy=uy, uright=ur, depth=dep,
               Tcw=T, pose0=pose6, last_points=la["points"], last_flags=la["flags"], last_octave=la["octave"],
               last_angle=la["angle"], last_desc=la["desc"],
               local_points=lo["points"], local_normals=lo["normals"], local_min_dist=lo["min_dist"],
               local_max_dist=lo["max_dist"], local_valid=lo["valid"], local_link=lo["link"], local_desc=lo["desc"],
               assigned_last=w["assigned_last"], n_last=np.int32(w["n_last"]), pose_1=w["pose_1"], inliers_1=np.int32(w["inliers_1"]),
               observed_inliers_1=np.int32(w["observed_inliers_1"]),
               local_flags=w["local_flags"], local_u=w["local_u"], local_v=w["local_v"], local_ur=w["local_ur"],
               local_level=w["local_level"], local_viewcos=w["local_viewcos"],
               assigned_local=w["assigned_local"], n_local=np.int32(w["n_local"]), pose_2=w["pose_2"],
               inliers_2=np.int32(w["inliers_2"]), n_tracked=np.int32(w["n_tracked"]))
    n0, n1, i1, i2 = w["n_last"], w["n_local"], w["inliers_1"], w["inliers_2"]
    # Sim3 RANSAC hypotheses
    from test_gpu_loop import _sim3_data
    pc1, pc2, px1, px2, me1, me2, cam, tri, _ = _sim3_data(11, n=200)
    K, m = len(tri), len(pc1)
    oc, oflag, osim = np.zeros(K, np.int32), np.zeros((K, m), np.uint8), np.zeros((K, 13))
    O.lib().orc_sim3_ransac_eval(m, np.ascontiguousarray(pc1), np.ascontiguousarray(pc2), np.ascontiguousarray(px1),
                                 np.ascontiguousarray(px2), me1, me2, cam, K, tri, 1, oc, oflag, osim)
    out.update(s3_pc1=pc1, s3_pc2=pc2, s3_px1=px1, s3_px2=px2, s3_me1=me1, s3_me2=me2, s3_cam=cam, s3_tri=tri,
               s3_counts=oc, s3_flags=np.packbits(oflag, axis=1), s3_sims=osim)
    # median descriptor of 40 observations of one map point
    rng = np.random.default_rng(4)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    obs = np.repeat(base[None], 40, 0)
    for r in range(40):
        bits = rng.choice(256, rng.integers(0, 30), replace=False)
        for b in bits:
            obs[r, b // 8] ^= np.uint8(1 << (b % 8))
    best = O.lib().orc_median_descriptor(np.ascontiguousarray(obs), 40)
    out.update(md_obs=obs, md_best=np.int32(best))
    np.savez_compressed(HERE / "g10_tracking.npz", **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith(("assigned", "n_", "inl", "s3_counts"))},
          int(n0), int(n1), int(i1), int(i2))


if __name__ == "__main__":
    main()
