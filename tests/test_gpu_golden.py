"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz) -- no oracle call,
nothing read from outside the repository."""
import pathlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = pathlib.Path(__file__).resolve().parent / "golden"


def test_g1_extract(vo):
    g = np.load(G / "g1_extract.npz")
    for tag in ("vga", "qvga"):
        nf, nl, it, mt = (int(v) for v in g[f"{tag}_params"])
        e = vo.OrbExtractor(nf, 1.2, nl, it, mt)
        kps, desc = e(g[f"{tag}_image"])
        assert list(e.get_level_counts(0)) == list(g[f"{tag}_per_level"])
        e.close()
        exp = g[f"{tag}_kps"]
        assert len(kps) == len(exp)
        for name in kps.dtype.names:
            assert np.array_equal(kps[name], exp[name]), (tag, name)
        assert np.array_equal(desc, g[f"{tag}_desc"])


def test_g3_hamming_and_match(vo):
    g = np.load(G / "g3_match.npz")
    assert np.array_equal(vo.hamming_matrix(g["d0"], g["d1"]), g["D"])
    n = len(g["d0"])
    cur = vo.FrameArrays(g["kx"], g["ky"], g["koct"], g["kang"], g["ur"], g["d1"])
    q = dict(flags=np.full(n, 3, np.uint8), u=g["q_u"], v=g["q_v"], invz=np.full(n, 0.5, np.float32),
             octave=g["q_oct"], angle=g["q_ang"], desc=np.ascontiguousarray(g["d0"]))
    cnt, assigned = vo.Matcher(0.8).searchByProjection_frame(cur, q, 15.0, 40.0, 0, True, g["scale"])
    assert cnt == int(g["count"]) and np.array_equal(assigned, g["assigned"])


def test_g6_keyframe_matchers(vo):
    g3, g = np.load(G / "g3_match.npz"), np.load(G / "g6_match_kf.npz")
    A = vo.FrameArrays(g["ax"], g["ay"], g["aoct"], g["aang"], g["aur"], g3["d0"])
    B = vo.FrameArrays(g3["kx"], g3["ky"], g3["koct"], g3["kang"], g["bur"], g3["d1"])
    na, nb, fa, fb = vo.BowNodes(g["node_a"]), vo.BowNodes(g["node_b"]), g["flag_a"], g["flag_b"]
    m = vo.Matcher(0.75)
    for mode in (0, 1):
        cnt, match = m.searchByBoW(A, 1 - fa, na, B, 1 - fb, nb, bool(mode), True)
        assert cnt == int(g[f"bow{mode}_n"]) and np.array_equal(match, g[f"bow{mode}"])
    cnt, match = m.searchForTriangulation(A, fa, na, B, fb, nb, g["F12"], 300.0, 200.0, g3["scale"], True)
    assert cnt == int(g["tri_n"]) and np.array_equal(match, g["tri"])
    q = dict(flags=(1 - fa).astype(np.uint8), u=g3["q_u"], v=g3["q_v"], ur=g["q_ur"], level=g["q_level"],
             angle=g3["q_ang"], desc=np.ascontiguousarray(g3["d0"]))
    cnt, best = m.fuseMapPoints_match(B, q, 3.0, g3["scale"])
    assert cnt == int(g["fuse_n"]) and np.array_equal(best, g["fuse"])
    cnt, assigned = m.searchByProjection_keyframe(B, q, 10.0, 64.0, True, g3["scale"], np.ascontiguousarray(fb))
    assert cnt == int(g["kfproj_n"]) and np.array_equal(assigned, g["kfproj"])


def test_g4_pose_only(vo):
    g = np.load(G / "g4_pose_only.npz")
    pr = {k: g[k] for k in ("pts", "obs", "inv_sigma", "cam", "pose0")}
    poses, masks, ninl, sums = vo.Optimizer.solvePoseOnlySE3([pr], summaries=True)
    assert ninl[0] == int(g["n_inlier"]) and np.array_equal(masks[0], g["outlier"])
    assert np.abs(poses[0] - g["pose"]).max() < 1e-9
    assert [sums[0].iterations, sums[1].iterations] == g["iters"].tolist()
    assert abs(sums[0].final_cost - g["cost0"][-1]) <= 1e-9 * g["cost0"][-1]


def test_g5_local_ba(vo):
    g = np.load(G / "g5_local_ba.npz")
    pr = {k: g[k] for k in ("poses", "fixed", "points", "e_cam", "e_pt", "e_obs", "e_inv_sigma", "cam")}
    ba = vo.BundleAdjuster(pr)
    erase, sums, rc = ba.local_ba()
    poses, pts = ba.state()
    ba.close()
    assert rc == 0 and np.array_equal(erase, g["edge_erase"])
    assert [sums[0].iterations, sums[1].iterations] == g["iters"].tolist()
    assert np.abs(poses - g["out_poses"]).max() < 1e-7
    assert np.allclose([sums[0].final_cost, sums[1].final_cost], g["final_cost"], rtol=1e-7)


def test_g7_se3(vo):
    g = np.load(G / "g7_se3.npz")
    for i, xi in enumerate(g["xi"]):
        R, t = vo.se3_exp(xi)
        assert np.abs(t - g["trans"][i]).max() < 1e-13
        assert np.abs(R @ g["point"] + t - g["transformed"][i]).max() < 1e-12


def test_g8_sim3_and_loop_searches(vo):
    g, g3 = np.load(G / "g8_sim3.npz"), np.load(G / "g3_match.npz")
    pr = {k: g[k] for k in ("cam_match", "pix_curr", "isig_curr", "cam_curr", "pix_match", "isig_match", "cam", "pose0")}
    pr["scale0"] = float(g["scale0"])
    poses, scales, masks, ninl, sums = vo.Optimizer.solveLoopSim3([pr], summaries=True)
    assert ninl[0] == int(g["n_inlier"]) and np.array_equal(masks[0], g["outlier"])
    assert np.abs(poses[0] - g["out_pose"]).max() < 1e-9 and scales[0] == float(g["out_scale"])
    assert [sums[0].iterations, sums[1].iterations] == g["iters"].tolist()
    assert np.allclose([sums[0].final_cost, sums[1].final_cost], g["final_cost"], rtol=1e-9)
    n = len(g3["d0"])
    kf = vo.FrameArrays(g3["kx"], g3["ky"], g3["koct"], g3["kang"], g3["ur"], g3["d1"])
    q = dict(flags=np.ones(n, np.uint8), u=g3["q_u"], v=g3["q_v"], level=g3["q_oct"].astype(np.int32),
             desc=np.ascontiguousarray(g3["d0"]))
    m = vo.Matcher(0.8)
    cnt, best = m.areaBest(kf, q, 7.5, g3["scale"], 100)
    assert cnt == int(g["area_n"]) and np.array_equal(best, g["area_best"])
    cnt, assigned = m.searchByProjection_sim3(kf, q, 5, g3["scale"], g["sim3proj_occ"])
    assert cnt == int(g["sim3proj_n"]) and np.array_equal(assigned, g["sim3proj"])


@pytest.mark.parametrize("pairs", [0, 1], ids=["pairs-lds", "pairs-lane-per-couple"])
def test_g9_global_ba_config4(vo, pairs):
    """BASELINE config 4 size (500 key-frames, 50 000 points, ~620 k edges, 2994 x 2994 reduced system):
    two LM iterations against the CPU oracle's result (tests/golden/make_g9_global_ba.py), with either gather kernel."""
    from vo_slam_test_amd import synth
    vo.set_option("ba_pairs_kernel", pairs)
    g = np.load(G / "g9_global_ba.npz")
    pr = synth.make_global_ba_problem(0)
    assert len(pr["e_cam"]) == int(g["n_edges"])
    assert pr["e_obs"].sum() + pr["poses"].sum() + pr["points"].sum() == float(g["input_checksum"])
    ba = vo.BundleAdjuster(pr)
    s = ba.solve(float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815))), int(g["iters"]))
    poses, pts = ba.state()
    ba.close()
    vo.set_option("ba_pairs_kernel", 0)
    assert (s.iterations, s.accepted) == (int(g["iters"]), int(g["accepted"])), (s.iterations, s.accepted, s.termination, s.initial_cost, s.final_cost, s.final_radius, vo.lib().vo_last_error())
    assert abs(s.initial_cost - float(g["initial_cost"])) <= 1e-10 * float(g["initial_cost"])
    assert abs(s.final_cost - float(g["final_cost"])) <= 1e-8 * float(g["final_cost"])
    assert np.abs(poses - g["poses"]).max() < 1e-7
    assert np.abs(pts[g["point_idx"]] - g["points"]).max() < 1e-6


def test_g11_pose_graph_config4(vo):
    """BASELINE config 4's reference-defined solver at full size (Optimizer::solvePoseGraphLoop, optimizer_ceres.cpp:1036-1305;
    500 key-frames, 2994 x 2994 normal matrix) against the CPU oracle's whole solve (tests/golden/make_g11_pose_graph.py):
    same LM decisions (the Q-B4 constant makes Ceres' function tolerance fire after a few iterations), poses to 1e-8."""
    from vo_slam_test_amd import synth
    g = np.load(G / "g11_pose_graph.npz")
    pg = synth.make_pose_graph(7, n_kf=500, drift=0.004, extra_edges=4)
    assert len(pg["e_i"]) == int(g["n_edges"])
    assert pg["quats"].sum() + pg["trans"].sum() + pg["q_meas"].sum() + pg["t_meas"].sum() == float(g["input_checksum"])
    q, t, s = vo.Optimizer.solvePoseGraphLoop(pg)
    assert (s.iterations, s.accepted, s.termination) == (int(g["iters"]), int(g["accepted"]), int(g["termination"]))
    assert abs(s.initial_cost - float(g["initial_cost"])) <= 1e-10 * float(g["initial_cost"])
    assert abs(s.final_cost - float(g["final_cost"])) <= 1e-9 * float(g["final_cost"])
    assert np.abs(q - g["quats"]).max() < 1e-8 and np.abs(t - g["trans"]).max() < 1e-7


def test_g10_tracked_frame_and_loop_helpers(vo):
    """Fixture g10: one RGB-D frame through vo_tracker (host image and raw depth in: extraction, undistortion / depth /
    grid, search, solve, culling, isInFrame, search, solve) against the committed vectors; Sim3 hypotheses and the
    median descriptor too."""
    g = np.load(G / "g10_tracking.npz")
    H, W = g["image"].shape
    n_last, n_local = len(g["last_flags"]), len(g["local_valid"])
    trk = vo.Tracker(1, g["cam5"], g["dist"], W, H, max_last=n_last, max_local=n_local, inv_depth_scale=float(g["inv_depth_scale"]))
    trk.set_last_frame(g["Tcw"][None], g["last_points"][None], g["last_flags"][None], g["last_octave"][None],
                       g["last_angle"][None], g["last_desc"][None])
    trk.set_local_map(g["local_points"][None], g["local_normals"][None], g["local_min_dist"][None], g["local_max_dist"][None],
                      g["local_valid"][None], g["local_desc"][None], link=g["local_link"][None])
    trk.track(g["image"][None], g["depth_raw"][None].view(np.uint16))
    r = trk.results()
    fr = trk.download_frame(0)
    n = len(g["kp_x"])
    assert fr["n"] == n
    assert np.array_equal(fr["x"], g["ux"]) and np.array_equal(fr["y"], g["uy"])          # undistorted key-points
    assert np.array_equal(fr["uright"], g["uright"]) and np.array_equal(fr["depth"], g["depth"])
    assert np.array_equal(fr["desc"], g["desc"]) and np.array_equal(fr["octave"], g["kp_octave"])
    assert np.array_equal(trk.get(trk.ASSIGNED_LAST)[0, :n], g["assigned_last"])
    assert int(trk.get(trk.INLIERS_FIRST)[0]) == int(g["inliers_1"])
    assert int(trk.get(trk.OBSERVED_INLIERS_FIRST)[0]) == int(g["observed_inliers_1"])
    assert np.abs(trk.get(trk.POSE_FIRST)[0] - g["pose_1"]).max() < 1e-9
    for what, key in ((trk.LOCAL_FLAGS, "local_flags"), (trk.LOCAL_U, "local_u"), (trk.LOCAL_V, "local_v"),
                      (trk.LOCAL_UR, "local_ur"), (trk.LOCAL_LEVEL, "local_level"), (trk.LOCAL_VIEWCOS, "local_viewcos")):
        assert np.array_equal(trk.get(what)[0, :n_local], g[key]), key
    assert np.array_equal(trk.get(trk.ASSIGNED_LOCAL)[0, :n], g["assigned_local"])
    assert int(r["n_inliers"][0]) == int(g["inliers_2"]) and int(r["n_tracked"][0]) == int(g["n_tracked"])
    assert int(r["n_matches_last"][0]) == int(g["n_last"]) and int(r["n_matches_local"][0]) == int(g["n_local"])
    assert np.abs(r["pose"][0] - g["pose_2"]).max() < 1e-9
    trk.close()
    counts, flags, sims = vo.sim3_ransac_eval(g["s3_pc1"], g["s3_pc2"], g["s3_px1"], g["s3_px2"], g["s3_me1"], g["s3_me2"],
                                              g["s3_cam"], g["s3_tri"], True)
    assert np.array_equal(counts, g["s3_counts"]) and np.array_equal(np.packbits(flags, axis=1), g["s3_flags"])
    assert np.abs(sims - g["s3_sims"]).max() < 1e-11
    best = vo.median_descriptor([g["md_obs"]])
    assert int(best[0]) == int(g["md_best"])
