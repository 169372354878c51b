#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (inputs AND expected outputs).

The reference (guisongchen/vo_slam_test) has no tests, golden vectors or fixtures and cannot be
built in this image, so these vectors come from the CPU restatement in oracle/ on seeded synthetic
inputs ("parity unpinned").  They pin (a) the oracle against regressions and
(b) the HIP path on the GPU box, where neither /root/reference nor this script's inputs are needed.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
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


def g1_extract():
    out = {}
    for tag, (w, h, nf, nl, idx) in {"vga": (640, 480, 1000, 8, 0), "qvga": (320, 240, 500, 6, 5)}.items():
        img = synth.make_frame(idx, w=w, h=h, n_rect=600 if w == 640 else 200, n_blob=150 if w == 640 else 60)
        p = O.orb_params(nf, 1.2, nl, 20, 7)
        kps, desc, npl = O.extract(p, img, cap=nf + 64)
        out[f"{tag}_image"] = img
        out[f"{tag}_params"] = np.array([nf, nl, 20, 7], np.int32)
        out[f"{tag}_kps"] = kps
        out[f"{tag}_desc"] = desc
        out[f"{tag}_per_level"] = npl
    np.savez_compressed(HERE / "g1_extract.npz", **out)


def g2_fast():
    img = synth.make_frame(2)
    out = {}
    for k, (x0, y0) in enumerate([(100, 80), (400, 300)]):
        crop = np.ascontiguousarray(img[y0:y0 + 64, x0:x0 + 64])
        out[f"crop{k}"] = crop
        for th in (20, 7):
            xs, ys, sc = O.fast(crop, th, True)
            out[f"crop{k}_th{th}"] = np.stack([xs, ys, sc], 1).astype(np.int32)
    p = O.orb_params()
    lev = O.pyramid(p, img)
    out["level3"] = lev[3]
    cx, cy, cr = O.level_candidates(p, lev[3])
    out["level3_candidates"] = np.stack([cx, cy, cr], 1)
    sel = O.octtree(cx, cy, cr, lev[3].shape[1], lev[3].shape[0], int(p.quota[3]))
    out["level3_octtree_sel"] = sel
    out["level3_blur"] = O.blur(lev[3])
    np.savez_compressed(HERE / "g2_fast_octtree.npz", **out)


def g3_match():
    p = O.orb_params()
    f0 = synth.make_frame(7)
    f1, dx, dy = synth.make_shifted(f0, 7)
    k0, d0, _ = O.extract(p, f0)
    k1, d1, _ = O.extract(p, f1)
    n = 400
    k0, d0, k1, d1 = k0[:n], d0[:n], k1[:n], d1[:n]
    D = O.hamming_matrix(d0, d1)
    rng = np.random.default_rng(7)
    z1 = rng.uniform(0.8, 4.5, n).astype(np.float32)
    ur1 = (k1["x"] - np.float32(40.0) / z1).astype(np.float32)
    sf = np.array(list(p.scale)[:8], np.float32)
    q = dict(flags=np.full(n, 3, np.uint8), u=(k0["x"] + dx).astype(np.float32), v=(k0["y"] + dy).astype(np.float32),
             invz=np.full(n, 0.5, np.float32), octave=k0["octave"].astype(np.int32), angle=k0["angle"].astype(np.float32))
    of = O.FrameData(k1["x"], k1["y"], k1["octave"], k1["angle"], ur1, d1)
    assigned = np.full(n, -1, np.int32)
    cnt = O.lib().orc_match_frame_projection(C.byref(of.c), n, q["flags"], q["u"], q["v"], q["invz"], q["octave"],
                                             q["angle"], np.ascontiguousarray(d0), 15.0, 40.0, 0, 1, 8, sf,
                                             np.zeros(n, np.uint8), assigned)
    np.savez_compressed(HERE / "g3_match.npz", d0=d0, d1=d1, D=D, kx=k1["x"], ky=k1["y"], koct=k1["octave"],
                        kang=k1["angle"], ur=ur1, q_u=q["u"], q_v=q["v"], q_oct=q["octave"], q_ang=q["angle"],
                        scale=sf, assigned=assigned, count=np.int32(cnt))


def g6_match_kf():
    """M2 / M5 / M6 / M8 / M9 on g3's frame pair (descriptors etc. are read from g3_match.npz)."""
    g = np.load(HERE / "g3_match.npz")
    n = len(g["d0"])
    p = O.orb_params()
    f0 = synth.make_frame(7)
    f1, dx, dy = synth.make_shifted(f0, 7)
    k0, _, _ = O.extract(p, f0)
    k0 = k0[:n]
    rng = np.random.default_rng(70)
    sf = g["scale"]
    ur0 = np.where(rng.random(n) < 0.5, -1.0, k0["x"] - 12.0).astype(np.float32)
    ur1 = np.where(rng.random(n) < 0.5, -1.0, g["ur"]).astype(np.float32)
    A = O.FrameData(k0["x"], k0["y"], k0["octave"], k0["angle"], ur0, g["d0"])
    B = O.FrameData(g["kx"], g["ky"], g["koct"], g["kang"], ur1, g["d1"])
    node = lambda x, y: (np.floor(x / 64.0).astype(np.int64) + 16 * np.floor(y / 64.0).astype(np.int64)).astype(np.uint32)
    na, nb = node(k0["x"], k0["y"]), node(g["kx"] - dx, g["ky"] - dy)
    ba, bb = O.BowData(na), O.BowData(nb)
    fa, fb = (rng.random(n) < 0.25).astype(np.uint8), (rng.random(n) < 0.25).astype(np.uint8)
    out = dict(ax=k0["x"], ay=k0["y"], aoct=k0["octave"], aang=k0["angle"], aur=ur0, bur=ur1, node_a=na, node_b=nb,
               flag_a=fa, flag_b=fb)
    L = O.lib()
    for mode in (0, 1):
        m = np.full(n, -1, np.int32)
        out[f"bow{mode}_n"] = np.int32(L.orc_match_bow(C.byref(A.c), 1 - fa, C.byref(ba.c), C.byref(B.c), 1 - fb,
                                                       C.byref(bb.c), mode, 0.75, 1, m))
        out[f"bow{mode}"] = m
    F = np.array([[0, 0, dy], [0, 0, -dx], [-dy, dx, 0]], np.float64)
    m = np.full(n, -1, np.int32)
    out["tri_n"] = np.int32(L.orc_match_triangulation(C.byref(A.c), fa, C.byref(ba.c), C.byref(B.c), fb, C.byref(bb.c),
                                                      np.ascontiguousarray(F.reshape(-1)), 300.0, 200.0, sf, 1, m))
    out["tri"], out["F12"] = m, F
    q_lvl = np.clip(k0["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
    q_ur = (g["q_u"] - 15.0).astype(np.float32)
    m = np.full(n, -1, np.int32)
    out["fuse_n"] = np.int32(L.orc_match_fuse(C.byref(B.c), n, 1 - fa, g["q_u"], g["q_v"], q_ur, q_lvl,
                                              np.ascontiguousarray(g["d0"]), 3.0, sf, m))
    out["fuse"], out["q_level"], out["q_ur"] = m, q_lvl, q_ur
    m = np.full(n, -1, np.int32)
    out["kfproj_n"] = np.int32(L.orc_match_frame_keyframe(C.byref(B.c), n, 1 - fa, g["q_u"], g["q_v"], q_lvl, g["q_ang"],
                                                          np.ascontiguousarray(g["d0"]), 10.0, 64.0, 1, sf, fb, m))
    out["kfproj"] = m
    np.savez_compressed(HERE / "g6_match_kf.npz", **out)


def g4_pose():
    pr = synth.make_pose_problem(3, n=300)
    pose, outl, ninl, sums, keep = O.pose_only(pr, trace=True)
    np.savez_compressed(HERE / "g4_pose_only.npz", pts=pr["pts"], obs=pr["obs"], inv_sigma=pr["inv_sigma"], cam=pr["cam"],
                        pose0=pr["pose0"], pose=pose, outlier=outl, n_inlier=np.int32(ninl),
                        iters=np.array([sums[0].iterations, sums[1].iterations], np.int32),
                        accepted=np.array([sums[0].accepted, sums[1].accepted], np.int32),
                        cost0=np.array([keep[0][0][i] for i in range(sums[0].iterations + 1)]),
                        cost1=np.array([keep[1][0][i] for i in range(sums[1].iterations + 1)]),
                        radius0=np.array([keep[0][1][i] for i in range(sums[0].iterations + 1)]))


def g5_lba():
    pr = synth.make_lba_problem(11, n_kf=3, n_pts=50, n_fixed=1)
    deg = np.bincount(pr["e_pt"], minlength=len(pr["points"]))
    active = (deg[pr["e_pt"]] >= 2).astype(np.uint8)
    S, b, cost, nf = O.ba_schur(pr, active=active, point_damping=1e-3)
    poses, pts, erase, sums, rc = O.local_ba(pr)
    arrays = {k: v for k, v in pr.items() if isinstance(v, np.ndarray)}
    np.savez_compressed(HERE / "g5_local_ba.npz", **arrays, schur_active=active, S=S, b=b, cost=cost,
                        out_poses=poses, out_points=pts, edge_erase=erase,
                        iters=np.array([sums[0].iterations, sums[1].iterations], np.int32),
                        final_cost=np.array([sums[0].final_cost, sums[1].final_cost]))


def g8_sim3():
    """solveLoopSim3 (B10/B11) on one synthetic loop candidate + the loop-closure searches on g3's frames"""
    pr = synth.make_sim3_problem(5, n=120, outliers=0.1)
    pose, scale, outl, inl, sums = O.sim3_solve(pr)
    arrays = {k: v for k, v in pr.items() if isinstance(v, np.ndarray)}
    out = dict(**arrays, scale0=np.float64(pr["scale0"]), out_pose=pose, out_scale=np.float64(scale), outlier=outl,
               n_inlier=np.int32(inl), iters=np.array([sums[0].iterations, sums[1].iterations], np.int32),
               final_cost=np.array([sums[0].final_cost, sums[1].final_cost]))
    g = np.load(HERE / "g3_match.npz")
    n = len(g["d0"])
    B = O.FrameData(g["kx"], g["ky"], g["koct"], g["kang"], g["ur"], g["d1"])
    flags = np.ones(n, np.uint8)
    lvl = g["q_oct"].astype(np.int32)
    best = np.full(n, -1, np.int32)
    out["area_n"] = np.int32(O.lib().orc_match_area_best(C.byref(B.c), n, flags, g["q_u"], g["q_v"], lvl,
                                                         np.ascontiguousarray(g["d0"]), 7.5, g["scale"], 100, best))
    out["area_best"] = best
    occ = np.zeros(n, np.uint8)
    occ[:2] = 1
    assigned = np.full(n, -1, np.int32)
    out["sim3proj_n"] = np.int32(O.lib().orc_match_sim3_projection(C.byref(B.c), n, flags, g["q_u"], g["q_v"], lvl,
                                                                   np.ascontiguousarray(g["d0"]), 5, g["scale"], occ, assigned))
    out["sim3proj"], out["sim3proj_occ"] = assigned, occ
    np.savez_compressed(HERE / "g8_sim3.npz", **out)


def g7_se3():
    rng = np.random.default_rng(1)
    xi = np.concatenate([rng.uniform(-2, 2, (12, 6)), np.array([[0.1, 0.2, 0.3, 0, 0, 0], [0.1, 0.2, 0.3, 1e-12, 0, 0],
                                                                 [1, 2, 3, 3.1, 0.0, 0.0], [0, 0, 0, 0, 3.14159, 0]])])
    q, t, tp = np.zeros((len(xi), 4)), np.zeros((len(xi), 3)), np.zeros((len(xi), 3))
    p0 = np.array([0.3, -0.7, 2.5])
    for i, x in enumerate(xi):
        O.lib().orc_se3_exp(np.ascontiguousarray(x), q[i], t[i])
        O.lib().orc_se3_trans_point(np.ascontiguousarray(x), p0, tp[i])
    np.savez_compressed(HERE / "g7_se3.npz", xi=xi, quat_wxyz=q, trans=t, point=p0, transformed=tp)


if __name__ == "__main__":
    for fn in (g1_extract, g2_fast, g3_match, g6_match_kf, g4_pose, g5_lba, g7_se3, g8_sim3):
        fn()
        print("wrote", fn.__name__)
    import os
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, os.path.getsize(f))
