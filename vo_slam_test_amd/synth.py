"""Seeded synthetic inputs: frames, frame pairs, descriptor sets,
pose-only and local-BA problems.  No TUM data exists in this environment, so every test,
fixture and bench line is driven from here.  Pure numpy; no GPU, no oracle, no reference."""
from __future__ import annotations

import numpy as np

# intrinsics of reference config/example.yaml:20-23,40 (as float32, like Camera::Camera reads them)
CAM = np.array([np.float32(517.306408), np.float32(516.469215), np.float32(318.643040),
                np.float32(255.313989), np.float32(40.0)], dtype=np.float64)
QUOTAS = np.array([217, 181, 151, 126, 105, 87, 73, 60], dtype=np.float64)


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def make_frame(idx: int = 0, w: int = 640, h: int = 480, n_rect: int = 600, n_blob: int = 150,
               noise: int = 6) -> np.ndarray:
    """One textured grey frame: rectangles in painter's order, small blobs, uniform noise."""
    rng = _rng(0x5EED0000 + idx)
    img = np.full((h, w), 110, dtype=np.int16)
    for _ in range(n_rect):
        rw, rh = rng.integers(8, 121, size=2)
        x0 = int(rng.integers(-rw // 2, w - rw // 2))
        y0 = int(rng.integers(-rh // 2, h - rh // 2))
        img[max(y0, 0):max(y0 + rh, 0), max(x0, 0):max(x0 + rw, 0)] = int(rng.integers(20, 236))
    for _ in range(n_blob):
        s = int(rng.integers(3, 8))
        x0 = int(rng.integers(0, w - s))
        y0 = int(rng.integers(0, h - s))
        img[y0:y0 + s, x0:x0 + s] = 240 if rng.integers(0, 2) else 15
    img += rng.integers(-noise, noise + 1, size=img.shape, dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_shifted(frame: np.ndarray, idx: int, max_shift: int = 12, noise: int = 6):
    """Frame k+1 of a pair: integer shift of frame k (edge-replicated) with fresh noise."""
    rng = _rng(0x5EED8000 + idx)
    dx, dy = (int(v) for v in rng.integers(-max_shift, max_shift + 1, size=2))
    h, w = frame.shape
    ys = np.clip(np.arange(h) - dy, 0, h - 1)
    xs = np.clip(np.arange(w) - dx, 0, w - 1)
    out = frame[np.ix_(ys, xs)].astype(np.int16)
    out += rng.integers(-noise, noise + 1, size=out.shape, dtype=np.int16)
    return np.clip(out, 0, 255).astype(np.uint8), dx, dy


# distortion coefficients of reference config/example.yaml:25-29 (k1, k2, p1, p2, k3) as float32
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], dtype=np.float32)
DEPTH_SCALE = 5000.0  # camera_depthScale, example.yaml:31


def make_depth(idx: int = 0, w: int = 640, h: int = 480, holes: float = 0.08) -> np.ndarray:
    """16-bit raw depth image: a tilted plane 0.8 .. 4.5 m plus noise, x 5000, with invalid (0) holes."""
    rng = _rng(0x5EEDD000 + idx)
    yy, xx = np.mgrid[0:h, 0:w]
    a, b = rng.uniform(-1.5, 1.5, 2)
    z = 2.6 + a * (xx / w - 0.5) + b * (yy / h - 0.5) + rng.normal(0, 0.02, (h, w))
    z = np.clip(z, 0.8, 4.5)
    raw = np.round(z * DEPTH_SCALE).astype(np.uint16)
    hole = rng.random((h // 8 + 1, w // 8 + 1)) < holes
    raw[np.kron(hole, np.ones((8, 8), bool))[:h, :w]] = 0
    return raw


def make_frames(n: int, start: int = 0, **kw) -> np.ndarray:
    return np.stack([make_frame(start + i, **kw) for i in range(n)])


def random_descriptors(n: int, seed: int = 0) -> np.ndarray:
    return _rng(0xDE5C0000 + seed).integers(0, 256, size=(n, 32), dtype=np.uint8)


# --------------------------------------------------------------------------- SE3 helpers
def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(xi):
    """Reference tangent order [upsilon; omega] -> (R, t)."""
    ups, om = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th = np.linalg.norm(om)
    Om = _hat(om)
    if th < 1e-10:
        R = np.eye(3) + Om
        V = np.eye(3) + 0.5 * Om
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th**2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * Om + (th - np.sin(th)) / th**3 * Om @ Om
    return R, V @ ups


def se3_log(R, t):
    from scipy.spatial.transform import Rotation
    om = Rotation.from_matrix(R).as_rotvec()
    th = np.linalg.norm(om)
    Om = _hat(om)
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * Om
    else:
        Vinv = np.eye(3) - 0.5 * Om + (1 - th / (2 * np.tan(th / 2))) / th**2 * Om @ Om
    return np.concatenate([Vinv @ t, om])


def project(R, t, P, cam=CAM):
    pc = P @ R.T + t
    u = cam[0] * pc[:, 0] / pc[:, 2] + cam[2]
    v = cam[1] * pc[:, 1] / pc[:, 2] + cam[3]
    ur = u - cam[4] / pc[:, 2]
    return u, v, ur, pc[:, 2]


def _octaves(rng, n):
    return rng.choice(8, size=n, p=QUOTAS / QUOTAS.sum())


def make_pose_problem(i: int = 0, n: int = 1000, outlier_frac: float = 0.10, mono_frac: float = 0.10):
    """Config 2: 1 frame x n observations; true pose identity, perturbed initial guess."""
    rng = _rng(42 + i)
    z = rng.uniform(0.5, 6.0, n)
    u = rng.uniform(20, 620, n)
    v = rng.uniform(20, 460, n)
    P = np.stack([(u - CAM[2]) * z / CAM[0], (v - CAM[3]) * z / CAM[1], z], axis=1)
    octv = _octaves(rng, n)
    sigma = 1.2 ** octv
    obs = np.stack([u, v, u - CAM[4] / z], axis=1) + rng.normal(0, 1, (n, 3)) * sigma[:, None]
    mono = rng.random(n) < mono_frac
    obs[mono, 2] = -1.0
    out = rng.random(n) < outlier_frac
    obs[out, 0] = rng.uniform(0, 640, out.sum())
    obs[out, 1] = rng.uniform(0, 480, out.sum())
    # observations are float32 pixel coordinates in the reference (cv::KeyPoint, uRight_)
    obs = obs.astype(np.float32).astype(np.float64)
    xi0 = np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.05, 0.05, 3)])
    inv_sigma = 1.0 / (np.float32(1.2) ** octv.astype(np.float32)).astype(np.float64)
    return dict(pts=np.ascontiguousarray(P), obs=np.ascontiguousarray(obs), inv_sigma=inv_sigma,
                cam=CAM.copy(), pose0=xi0)


def make_lba_problem(i: int = 0, n_kf: int = 10, n_pts: int = 3000, n_fixed: int = 4,
                     outlier_frac: float = 0.05, fixed_seen: float = 0.30):
    """Config 3: n_kf key-frames on a 1 m arc looking at a box of points, KF 0 constant, plus
    n_fixed extra fixed key-frames each seeing a random 30 % of the points."""
    rng = _rng(1000 + i)
    P = np.stack([rng.uniform(-2, 2, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(2, 5, n_pts)], 1)
    n_cam = n_kf + n_fixed
    poses_true = np.zeros((n_cam, 6))
    Rs, ts = [], []
    for c in range(n_cam):
        if c < n_kf:
            ang = np.deg2rad(-18 + 36.0 * c / max(n_kf - 1, 1))
        else:
            ang = np.deg2rad(rng.uniform(-25, 25))
        centre = np.array([np.sin(ang) * 1.0, rng.uniform(-0.05, 0.05), -np.cos(ang) * 1.0 + 1.0])
        yaw = -ang * 0.6
        Rwc = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        Rcw = Rwc.T
        tcw = -Rcw @ centre
        Rs.append(Rcw), ts.append(tcw)
        poses_true[c] = se3_log(Rcw, tcw)
    e_cam, e_pt, e_obs, e_is = [], [], [], []
    for j in range(n_pts):
        for c in range(n_cam):
            if c >= n_kf and rng.random() > fixed_seen:
                continue
            u, v, ur, z = project(Rs[c], ts[c], P[j:j + 1])
            if z[0] <= 0.2 or not (19 <= u[0] <= 621 and 19 <= v[0] <= 461):
                continue
            octv = int(_octaves(rng, 1)[0])
            sg = 1.2 ** octv
            o = np.array([u[0], v[0], ur[0]]) + rng.normal(0, 1, 3) * sg
            if rng.random() < 0.10:
                o[2] = -1.0
            if rng.random() < outlier_frac:
                o[0], o[1] = rng.uniform(0, 640), rng.uniform(0, 480)
            e_cam.append(c), e_pt.append(j), e_obs.append(o)
            e_is.append(float(np.float64(1.0) / np.float64(np.float32(1.2) ** np.float32(octv))))
    fixed = np.zeros(n_cam, np.uint8)
    fixed[0] = 1
    fixed[n_kf:] = 1
    poses0 = poses_true.copy()
    for c in range(n_cam):
        if not fixed[c]:
            poses0[c, :3] += rng.normal(0, 0.01, 3)
            poses0[c, 3:] += rng.normal(0, np.deg2rad(0.5), 3)
    pts0 = P + rng.normal(0, 0.02, P.shape)
    e_obs = np.asarray(e_obs).astype(np.float32).astype(np.float64)
    return dict(poses=poses0, poses_true=poses_true, fixed=fixed, points=pts0, points_true=P,
                e_cam=np.asarray(e_cam, np.int32), e_pt=np.asarray(e_pt, np.int32),
                e_obs=np.ascontiguousarray(e_obs), e_inv_sigma=np.asarray(e_is, np.float64),
                cam=CAM.copy())


# --------------------------------------------------------------------------- Sim3 (loop closure)
def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3) + _hat(w)
    k = w / th
    K = _hat(k)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_sim3_problem(seed: int, n: int = 150, outliers: float = 0.1, scale: float = 1.0):
    """Matched map points seen from the current and the loop key-frame (solveLoopSim3's inputs,
    optimizer_ceres.cpp:846-878): true Scm = (s, R, t), initial guess perturbed."""
    rng = _rng(0x51300000 + seed)
    fx, fy, cx, cy = CAM[:4]
    w_true = rng.uniform(-0.3, 0.3, 3)
    t_true = rng.uniform(-0.5, 0.5, 3)
    R = _rodrigues(w_true)
    Pm = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)])
    Pc = scale * (Pm @ R.T) + t_true
    ok = Pc[:, 2] > 0.5
    Pm, Pc = Pm[ok], Pc[ok]
    n = len(Pm)
    octs = rng.integers(0, 8, (2, n))
    sig = 1.2 ** octs
    pix_m = np.column_stack([fx * Pm[:, 0] / Pm[:, 2] + cx, fy * Pm[:, 1] / Pm[:, 2] + cy]) + rng.normal(0, 1, (n, 2)) * sig[0][:, None]
    pix_c = np.column_stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy]) + rng.normal(0, 1, (n, 2)) * sig[1][:, None]
    bad = rng.random(n) < outliers
    pix_c[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
    # the stored 3-D points carry some triangulation noise
    Pm_n = Pm + rng.normal(0, 0.01, Pm.shape)
    Pc_n = Pc + rng.normal(0, 0.01, Pc.shape)
    pose0 = np.concatenate([w_true + rng.uniform(-0.03, 0.03, 3), t_true + rng.uniform(-0.05, 0.05, 3)])
    return dict(cam_match=np.ascontiguousarray(Pm_n), pix_curr=np.ascontiguousarray(pix_c),
                isig_curr=np.ascontiguousarray(1.0 / sig[1]), cam_curr=np.ascontiguousarray(Pc_n),
                pix_match=np.ascontiguousarray(pix_m), isig_match=np.ascontiguousarray(1.0 / sig[0]),
                cam=np.array(CAM, np.float64), pose0=pose0, scale0=float(scale), pose_true=np.concatenate([w_true, t_true]),
                is_outlier=bad)


# --------------------------------------------------------------------------- pose graph (loop closure)
def _quat_from_R(R):
    """unit quaternion (x, y, z, w) of a rotation matrix"""
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x, y, z = np.copysign(x, R[2, 1] - R[1, 2]), np.copysign(y, R[0, 2] - R[2, 0]), np.copysign(z, R[1, 0] - R[0, 1])
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def _R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_pose_graph(seed: int, n_kf: int = 40, drift: float = 0.01, extra_edges: int = 2):
    """Key-frames on a closed loop (world -> camera Sim3 with unit scale), odometry edges measured on a
    drifted copy of the trajectory (what the map holds before the loop closes), covisibility edges
    between near neighbours, and one loop edge carrying the true relative pose (the corrected side of
    solvePoseGraphLoop, optimizer_ceres.cpp:1094-1125).  Edge (i -> j) measures S_ji = S_jw * S_wi."""
    rng = _rng(0x9C0000 + seed)
    ang = np.linspace(0, 2 * np.pi, n_kf, endpoint=False)
    Rs, ts = [], []
    for a in ang:
        Rwc = _rodrigues(np.array([0, a + np.pi / 2, 0])) @ _rodrigues(rng.normal(0, 0.02, 3))
        c = np.array([5 * np.cos(a), 0.1 * np.sin(3 * a), 5 * np.sin(a)])
        Rcw = Rwc.T
        Rs.append(Rcw), ts.append(-Rcw @ c)
    # drifted estimate: accumulate small errors along the chain
    Rd, td = [Rs[0]], [ts[0]]
    for i in range(1, n_kf):
        Rrel = Rs[i] @ Rs[i - 1].T
        trel = ts[i] - Rrel @ ts[i - 1]
        Rrel = _rodrigues(rng.normal(0, drift, 3)) @ Rrel
        trel = trel + rng.normal(0, drift, 3)
        Rd.append(Rrel @ Rd[-1]), td.append(Rrel @ td[-1] + trel)

    def rel(Ra, ta, Rb, tb):  # S_ba = S_bw * S_wa
        Rba = Rb @ Ra.T
        return _quat_from_R(Rba), tb - Rba @ ta

    e_i, e_j, qm, tm = [], [], [], []
    for i in range(1, n_kf):                       # spanning tree: child -> parent, measured on the drifted map
        q, t_ = rel(Rd[i], td[i], Rd[i - 1], td[i - 1])
        e_i.append(i), e_j.append(i - 1), qm.append(q), tm.append(t_)
        for k in range(2, 2 + extra_edges):        # covisibility edges to older neighbours
            if i - k >= 0:
                q, t_ = rel(Rd[i], td[i], Rd[i - k], td[i - k])
                e_i.append(i), e_j.append(i - k), qm.append(q), tm.append(t_)
    q, t_ = rel(Rs[n_kf - 1], ts[n_kf - 1], Rs[0], ts[0])   # loop edge: true relative pose
    e_i.append(n_kf - 1), e_j.append(0), qm.append(q), tm.append(t_)
    return dict(quats=np.ascontiguousarray([_quat_from_R(R) for R in Rd]), trans=np.ascontiguousarray(td),
                scales=np.ones(n_kf), fixed=0, e_i=np.array(e_i, np.int32), e_j=np.array(e_j, np.int32),
                q_meas=np.ascontiguousarray(qm), t_meas=np.ascontiguousarray(tm), s_meas=np.ones(len(e_i)),
                true_quats=np.ascontiguousarray([_quat_from_R(R) for R in Rs]), true_trans=np.ascontiguousarray(ts))


# --------------------------------------------------------------------------- global BA (config 4)
def make_global_ba_problem(seed: int = 0, n_kf: int = 500, n_pts: int = 50000, radius: float = 20.0 / (2 * np.pi),
                           view_dist: float = 2.5, noise: bool = True, obs_keep: float = 0.37):
    """BASELINE config 4: key-frames on a closed loop (circumference 2 pi radius = 20 m by default)
    looking outwards at a 3 m shell of points; a point is observed by the key-frames within `view_dist`
    whose frustum contains it (about 12-18 per point at the default sizes); one fixed key-frame.
    Vectorised; edges come sorted by point."""
    rng = _rng(0x6BA0000 + seed)
    fx, fy, cx, cy, bf = CAM
    ang = np.linspace(0, 2 * np.pi, n_kf, endpoint=False)
    centres = np.stack([radius * np.cos(ang), 0.05 * np.sin(5 * ang), radius * np.sin(ang)], 1)
    Rcw = np.zeros((n_kf, 3, 3))
    for c in range(n_kf):  # camera z axis points outwards, y down
        z = np.array([np.cos(ang[c]), 0.0, np.sin(ang[c])])
        y = np.array([0.0, 1.0, 0.0])
        x = np.cross(y, z)
        Rcw[c] = np.stack([x, y, z])
    tcw = -np.einsum("cij,cj->ci", Rcw, centres)
    pa = rng.uniform(0, 2 * np.pi, n_pts)
    pr = radius + rng.uniform(1.0, 4.0, n_pts)
    P = np.stack([pr * np.cos(pa), rng.uniform(-1.0, 1.0, n_pts), pr * np.sin(pa)], 1)
    e_cam, e_pt, e_obs, e_is = [], [], [], []
    order = np.argsort(pa)
    win = max(2, int(np.ceil(view_dist / (2 * np.pi * radius / n_kf))) + 1)
    for c in range(n_kf):
        d = np.abs(((pa - ang[c] + np.pi) % (2 * np.pi)) - np.pi)
        cand = np.nonzero(d < (win * 2 * np.pi / n_kf))[0]
        pc = P[cand] @ Rcw[c].T + tcw[c]
        z = pc[:, 2]
        u, v = fx * pc[:, 0] / z + cx, fy * pc[:, 1] / z + cy
        ok = (z > 0.3) & (u >= 19) & (u <= 621) & (v >= 19) & (v <= 461) & (np.linalg.norm(P[cand] - centres[c], axis=1) < view_dist + 3.0)
        ok &= rng.random(len(cand)) < obs_keep   # a feature is matched in a fraction of the frames that could see it
        idx = cand[ok]
        e_cam.append(np.full(len(idx), c, np.int32)), e_pt.append(idx.astype(np.int32))
        e_obs.append(np.stack([u[ok], v[ok], u[ok] - bf / z[ok]], 1))
    e_cam, e_pt, e_obs = np.concatenate(e_cam), np.concatenate(e_pt), np.concatenate(e_obs)
    octv = rng.integers(0, 8, len(e_cam))
    sg = 1.2 ** octv
    if noise:
        e_obs = e_obs + rng.normal(0, 1, e_obs.shape) * sg[:, None]
    e_obs[rng.random(len(e_cam)) < 0.10, 2] = -1.0
    srt = np.argsort(e_pt, kind="stable")
    e_cam, e_pt, e_obs, sg = e_cam[srt], e_pt[srt], e_obs[srt], sg[srt]
    keep = np.bincount(e_pt, minlength=n_pts) >= 2            # a landmark needs two views
    m = keep[e_pt]
    e_cam, e_pt, e_obs, sg = e_cam[m], e_pt[m], e_obs[m], sg[m]
    poses_true = np.stack([se3_log(Rcw[c], tcw[c]) for c in range(n_kf)])
    fixed = np.zeros(n_kf, np.uint8)
    fixed[0] = 1
    poses0 = poses_true.copy()
    poses0[1:, :3] += rng.normal(0, 0.01, (n_kf - 1, 3))
    poses0[1:, 3:] += rng.normal(0, np.deg2rad(0.3), (n_kf - 1, 3))
    points0 = P + rng.normal(0, 0.02, P.shape)
    return dict(poses=np.ascontiguousarray(poses0), fixed=fixed, points=np.ascontiguousarray(points0),
                e_cam=np.ascontiguousarray(e_cam, np.int32), e_pt=np.ascontiguousarray(e_pt, np.int32),
                e_obs=np.ascontiguousarray(e_obs), e_inv_sigma=np.ascontiguousarray(1.0 / sg),
                cam=np.array(CAM, np.float64), poses_true=poses_true, points_true=P)


# --------------------------------------------------------------------------- vocabulary tree (BoW)
def make_vocabulary(seed: int = 0, k: int = 10, L: int = 4, flip: int = 24):
    """A k-ary tree of depth L over 256-bit descriptors: every child is its parent with `flip` random
    bits toggled (so the descent is meaningful), leaves are words with idf-like weights.  Flat arrays in
    the layout vo_vocab_create takes (node 0 = root, breadth-first)."""
    rng = _rng(0xB0C0000 + seed)
    desc = [rng.integers(0, 256, 32, dtype=np.uint8)]
    child_start, children, level_of = [0], [], [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for node in frontier:
            first = len(desc)
            for _ in range(k):
                bits = np.unpackbits(desc[node])
                idx = rng.choice(256, flip, replace=False)
                bits[idx] ^= 1
                desc.append(np.packbits(bits))
                level_of.append(lvl)
                nxt.append(len(desc) - 1)
            children.extend(range(first, first + k))
        frontier = nxt
    n = len(desc)
    # child_start in node order (breadth-first numbering makes the child lists consecutive)
    cs = np.zeros(n + 1, np.int32)
    inner = [i for i in range(n) if level_of[i] < L]
    for i in inner:
        cs[i + 1] = k
    cs = np.cumsum(cs).astype(np.int32)
    word_id = np.full(n, -1, np.int32)
    leaves = [i for i in range(n) if level_of[i] == L]
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    weight = np.zeros(n)
    weight[leaves] = rng.uniform(0.5, 8.0, len(leaves))
    return dict(k=k, L=L, child_start=cs, children=np.array(children, np.int32), node_desc=np.ascontiguousarray(desc),
                node_weight=weight, word_id=word_id)


# --------------------------------------------------------------------------- tracked-frame map (bench / tests)
def make_tracking_map(kx, ky, octave, angle, desc, depth_m, seed: int, n_local_rep: int = 2, cam=CAM):
    """Synthetic map seen by ONE frame whose (distorted-free) key-points are (kx, ky): the last frame's map points =
    the frame's own features back-projected with their depth (2.5 m where the depth image has a hole), the local
    map = `n_local_rep` noisy copies of them; the current pose estimate is a small perturbation of the true pose
    (identity).  Returns (Tcw12, pose6, last, local) in the layout of tracking.BatchTracker.set_map; projections of
    the local map points are pre-computed in float32 like Frame::isInFrame leaves them (frame.cpp:145-190)."""
    rng = _rng(0x7AC40000 + seed)
    n = len(kx)
    fx, fy, cx, cy, bf = [float(c) for c in cam]
    z = np.where(depth_m > 0, depth_m, 2.5).astype(np.float64)
    P = np.stack([(kx.astype(np.float64) - cx) * z / fx, (ky.astype(np.float64) - cy) * z / fy, z], axis=1)
    xi = np.concatenate([rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.01, 0.01, 3)])
    R, t = se3_exp(xi)
    Tcw12 = np.concatenate([R.reshape(-1), t])
    flags = (1 | ((rng.random(n) < 0.7).astype(np.uint8) << 1)).astype(np.uint8)
    flags[rng.random(n) < 0.05] = 0
    d0 = desc.copy()
    flip = rng.random(d0.shape) < 0.01
    d0[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
    last = dict(points=P + rng.normal(0, 0.005, P.shape), flags=flags, octave=octave.astype(np.int32),
                angle=angle.astype(np.float32), desc=d0)
    idx = np.tile(np.arange(n), n_local_rep)
    rng.shuffle(idx)
    m = len(idx)
    Pl = P[idx] + rng.normal(0, 0.01, (m, 3))
    pc = Pl @ R.T + t
    u = (fx * pc[:, 0] / pc[:, 2] + cx).astype(np.float32)
    v = (fy * pc[:, 1] / pc[:, 2] + cy).astype(np.float32)
    zf = pc[:, 2].astype(np.float32)
    ur = (u - np.float32(bf) / zf).astype(np.float32)
    lflags = np.where(rng.random(m) < 0.6, 3, 1).astype(np.uint8)
    lflags[(u < 0) | (u > 640) | (v < 0) | (v > 480) | (rng.random(m) < 0.05)] = 0
    dl = desc[idx].copy()
    flip = rng.random(dl.shape) < 0.02
    dl[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
    local = dict(points=Pl, flags=lflags, u=u, v=v, ur=ur,
                 level=np.clip(octave[idx] + rng.integers(0, 2, m), 0, 7).astype(np.int32),
                 viewcos=rng.uniform(0.99, 1.0, m).astype(np.float32), desc=dl)
    # what Frame::isInFrame reads of a MapPoint (frame.cpp:145-190): normal vector and distance range as
    # MapPoint::updateNormalAndDepth leaves them for a reference key-frame near the true camera (maxDistance_ = distance
    # times the scale of the feature's level, minDistance_ = maxDistance_ / scale of the last level); `link`: the local
    # point is the same MapPoint as this entry of the last frame's list (-1: it is not in that list); `valid` = flags
    # before the pre-projection above (bit 0 exists / not bad, bit 1 has observations)
    r2 = _rng(0x7AC50000 + seed)
    cref = r2.normal(0, 0.03, (m, 3))
    line = Pl - cref
    dref = np.linalg.norm(line, axis=1)
    sfl = 1.2 ** np.clip(octave[idx] + r2.integers(0, 2, m), 0, 7).astype(np.float64)
    local["normals"] = line / dref[:, None]
    local["max_dist"] = (dref * sfl).astype(np.float32)
    local["min_dist"] = (local["max_dist"] / np.float32(1.2 ** 7)).astype(np.float32)
    local["valid"] = np.where(r2.random(m) < 0.6, 3, 1).astype(np.uint8)
    local["valid"][r2.random(m) < 0.05] = 0
    local["link"] = np.where(r2.random(m) < 0.3, idx, -1).astype(np.int32)
    return Tcw12, se3_log(R, t), last, local
