"""The tracked-frame path of the reference, stage by stage on the CPU oracle (test infrastructure): what
VisualOdometry::trackWithMotion + trackLocalMap do to one frame (src/visualOdometry.cpp:228-251, 286-300, 745-775,
864-886).  Used by tests/test_gpu_tracking.py, tests/golden/make_g10_tracking.py and bench.py's cpu_baseline leg."""
import ctypes as C

import numpy as np


def project_last(T, P, pf, cam, W, H):
    """the projection prologue of Matcher::searchByProjection(Frame*, Frame*), matcher.cpp:41-64"""
    x = T[0] * P[:, 0] + T[1] * P[:, 1] + T[2] * P[:, 2] + T[9]
    y = T[3] * P[:, 0] + T[4] * P[:, 1] + T[5] * P[:, 2] + T[10]
    zc = T[6] * P[:, 0] + T[7] * P[:, 1] + T[8] * P[:, 2] + T[11]
    z = zc.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (np.float32(1.0) / z).astype(np.float32)
        u = (np.float64(cam[0]) * x / zc + np.float64(cam[2])).astype(np.float32)
        v = (np.float64(cam[1]) * y / zc + np.float64(cam[3])).astype(np.float32)
    ok = ((pf & 1) == 1) & ~(z < 0) & ~((u < 0) | (u > W)) & ~((v < 0) | (v > H))
    flags = np.where(ok, 1 | (pf & 2), 0).astype(np.uint8)
    return flags, np.where(ok, u, 0).astype(np.float32), np.where(ok, v, 0).astype(np.float32), np.where(ok, invz, 0).astype(np.float32)


def is_in_frame(orc, pose6, local, valid, cam5, W, H, sf1, n_levels=8):
    n = len(valid)
    fl, lv = np.zeros(n, np.uint8), np.zeros(n, np.int32)
    u, v, ur, vc = (np.zeros(n, np.float32) for _ in range(4))
    orc.lib().orc_is_in_frame(n, np.ascontiguousarray(pose6, np.float64), np.ascontiguousarray(local["points"], np.float64),
                              np.ascontiguousarray(local["normals"], np.float64), np.ascontiguousarray(local["min_dist"], np.float32),
                              np.ascontiguousarray(local["max_dist"], np.float32), np.ascontiguousarray(valid, np.uint8),
                              np.ascontiguousarray(cam5, np.float32), 0.0, float(W), 0.0, float(H), float(sf1), n_levels,
                              fl, u, v, ur, lv, vc)
    return fl, u, v, ur, lv, vc


def local_map_stage(orc, of, k, ux, uy, ur, fpt, has, fobs, p1, a_first, n_first_list, local, cam5, sf, W, H, th_radius, ratio, solve):
    """trackLocalMap from searchLocalMapPoints on (visualOdometry.cpp:745-775, :289-300): isInFrame with the refined pose
    (points the first stage matched are skipped through `link`), the search, the second solve, the inlier count"""
    n = len(k)
    last_matched = np.zeros(n_first_list, bool)
    last_matched[a_first[a_first >= 0]] = True
    valid = np.asarray(local["valid"], np.uint8).copy()
    lk = np.asarray(local.get("link", np.full(len(valid), -1)), np.int64)
    skip = (lk >= 0) & last_matched[np.clip(lk, 0, max(n_first_list - 1, 0))] if n_first_list > 0 else np.zeros(len(valid), bool)
    valid[skip] = 0
    fl, lu, lv_, lur, llev, lvc = is_in_frame(orc, p1, local, valid, cam5, W, H, sf[1])
    a1 = np.full(n, -1, np.int32)
    n1 = 0
    if len(fl) > 0:
        n1 = orc.lib().orc_match_local_map(C.byref(of.c), len(fl), fl, lu, lv_, lur, llev, lvc, np.ascontiguousarray(local["desc"]),
                                           float(th_radius), float(ratio), sf, fobs, a1)
    new = a1 >= 0
    fpt[new] = np.asarray(local["points"])[a1[new]]
    fobs[new] = (fl[a1[new]] >> 1) & 1
    has = has | new
    p2, out2, i2, idx2 = solve(p1, has)
    n_tracked = int(fobs[idx2[~out2]].sum())
    foutl = np.zeros(n, np.uint8)
    foutl[idx2[out2]] = 1
    return dict(feature_outlier=foutl, local_flags=fl, local_u=lu, local_v=lv_, local_ur=lur, local_level=llev, local_viewcos=lvc, assigned_local=a1,
                n_local=n1, pose_2=p2, inliers_2=i2, n_tracked=n_tracked, has=has)


def track_frame_ref_keyframe(orc, k, d, ux, uy, ur, pose6_last, kf, node_of_frame_feature, local, cam5, sf, W=640, H=480,
                             th_radius=3.0, ratio=0.8, ref_ratio=0.7):
    """VisualOdometry::trackRefKeyFrame (visualOdometry.cpp:256-277) + trackLocalMap on the oracle: searchByBoW(key-frame,
    frame) with Matcher(0.7), the key-frame's map points into the frame's slots, pose = frame_last_->Tcw_, solve, culling.
    kf: dict(points, flags, angle, desc, nodes = per-feature node ids); node_of_frame_feature: the frame's FeatureVector."""
    import oracle_lib as olib
    n = len(k)
    of = orc.FrameData(ux, uy, k["octave"], k["angle"], ur, d)
    nk = len(kf["flags"])
    okf = orc.FrameData(np.zeros(nk, np.float32), np.zeros(nk, np.float32), np.zeros(nk, np.int32), np.ascontiguousarray(kf["angle"], np.float32),
                        np.full(nk, -1, np.float32), np.ascontiguousarray(kf["desc"]))
    ba, bb = olib.BowData(kf["nodes"]), olib.BowData(node_of_frame_feature)
    a0 = np.full(n, -1, np.int32)
    va = (np.asarray(kf["flags"]) & 1).astype(np.uint8)
    n0 = orc.lib().orc_match_bow(C.byref(okf.c), va, C.byref(ba.c), C.byref(of.c), np.ones(n, np.uint8), C.byref(bb.c), 0, float(ref_ratio), 1, a0)
    fpt, has, fobs = np.zeros((n, 3)), a0 >= 0, np.zeros(n, np.uint8)
    fpt[has] = np.asarray(kf["points"])[a0[has]]
    fobs[has] = (np.asarray(kf["flags"])[a0[has]] >> 1) & 1
    cam_d = np.asarray(cam5, np.float64)

    def solve(pose_in, has_now):
        idx = np.nonzero(has_now)[0]
        pr = dict(pts=np.ascontiguousarray(fpt[idx]),
                  obs=np.ascontiguousarray(np.stack([ux[idx], uy[idx], ur[idx]], 1).astype(np.float64)),
                  inv_sigma=np.ascontiguousarray(1.0 / sf[k["octave"][idx]].astype(np.float64)), cam=cam_d, pose0=pose_in)
        pose, outl, ninl, _, _ = orc.pose_only(pr)
        return pose, np.asarray(outl, bool), ninl, idx

    p1, out1, i1, idx1 = solve(np.asarray(pose6_last, np.float64), has)
    n_obs1 = int(fobs[idx1[~out1]].sum())
    has[idx1[out1]] = False
    fobs[idx1[out1]] = 0
    # matched points -- kept or culled -- carry visualIdxOfFrame_ == frame id (:752, :881): searchLocalMapPoints skips them
    out = dict(assigned_first=a0, n_first=n0, pose_1=p1, inliers_1=i1, observed_inliers_1=n_obs1)
    out.update(local_map_stage(orc, of, k, ux, uy, ur, fpt, has.copy(), fobs, p1, a0, nk, local, cam5, sf, W, H, th_radius, ratio, solve))
    return out


def track_first(orc, k, d, ux, uy, ur, T, pose6, last, cam5, sf, W=640, H=480, radius=15.0, retry=True):
    """trackWithMotion (visualOdometry.cpp:224-255): projection, searchByProjection against the last frame's points, the
    2 x radius retry, solvePoseOnlySE3, cullingOutliersBeforeLocalMap -> state dict for local_map_stage"""
    n = len(k)
    of = orc.FrameData(ux, uy, k["octave"], k["angle"], ur, d)
    cam_d = np.asarray(cam5, np.float64)
    nl = len(last["flags"])
    a0 = np.full(n, -1, np.int32)
    n0, retried = 0, False
    qf = np.zeros(0, np.uint8)
    if nl > 0:
        qf, qu, qv, qz = project_last(T, np.asarray(last["points"]), np.asarray(last["flags"]), cam5, W, H)
        args = (np.ascontiguousarray(last["octave"], np.int32), np.ascontiguousarray(last["angle"], np.float32), np.ascontiguousarray(last["desc"]))
        n0 = orc.lib().orc_match_frame_projection(C.byref(of.c), len(qf), qf, qu, qv, qz, *args, float(radius), float(cam5[4]), 0, 1, 8, sf,
                                                  np.zeros(n, np.uint8), a0)
        if retry and n0 < 20:  # visualOdometry.cpp:241-245: clear, search again at twice the radius
            retried = True
            a0 = np.full(n, -1, np.int32)
            n0 = orc.lib().orc_match_frame_projection(C.byref(of.c), len(qf), qf, qu, qv, qz, *args, float(2 * radius), float(cam5[4]), 0, 1, 8,
                                                      sf, np.zeros(n, np.uint8), a0)
    fpt, has, fobs = np.zeros((n, 3)), a0 >= 0, np.zeros(n, np.uint8)
    fpt[has] = np.asarray(last["points"])[a0[has]]
    fobs[has] = (qf[a0[has]] >> 1) & 1

    def solve(pose_in, has_now):
        idx = np.nonzero(has_now)[0]
        pr = dict(pts=np.ascontiguousarray(fpt[idx]),
                  obs=np.ascontiguousarray(np.stack([ux[idx], uy[idx], ur[idx]], 1).astype(np.float64)),
                  inv_sigma=np.ascontiguousarray(1.0 / sf[k["octave"][idx]].astype(np.float64)), cam=cam_d, pose0=pose_in)
        pose, outl, ninl, _, _ = orc.pose_only(pr)
        return pose, np.asarray(outl, bool), ninl, idx

    p1, out1, i1, idx1 = solve(np.asarray(pose6, np.float64), has)
    # cullingOutliersBeforeLocalMap (:864-886)
    n_obs1 = int(fobs[idx1[~out1]].sum())
    has[idx1[out1]] = False
    fobs[idx1[out1]] = 0
    status = (1 if n0 < 20 else 0) | (2 if n_obs1 < 10 else 0)
    return dict(of=of, retried=retried, assigned_last=a0, n_last=n0, pose_1=p1, inliers_1=i1, observed_inliers_1=n_obs1, fpt=fpt, has=has,
                fobs=fobs, solve=solve, n_last_list=nl, status=status)


def track_frame(orc, k, d, ux, uy, ur, T, pose6, last, local, cam5, sf, W=640, H=480, radius=15.0, th_radius=3.0, ratio=0.8,
                retry=True):
    """-> dict of every intermediate result of the path for one frame (k, d: the oracle's key-points and descriptors;
    ux, uy, ur: undistorted coordinates and uRight; T [12], pose6: the pose estimate; last / local: the map as
    synth.make_tracking_map builds it)"""
    s1 = track_first(orc, k, d, ux, uy, ur, T, pose6, last, cam5, sf, W, H, radius, retry)
    out = {kk: s1[kk] for kk in ("retried", "assigned_last", "n_last", "pose_1", "inliers_1", "observed_inliers_1")}
    # searchLocalMapPoints (:745-775): matched points -- kept or culled -- carry visualIdxOfFrame_ == frame id and are skipped
    out.update(local_map_stage(orc, s1["of"], k, ux, uy, ur, s1["fpt"], s1["has"].copy(), s1["fobs"], s1["pose_1"], s1["assigned_last"],
                               s1["n_last_list"], local, cam5, sf, W, H, th_radius, ratio, s1["solve"]))
    return out
