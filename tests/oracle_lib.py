"""ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (vo_slam_test_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
SO = ORACLE_DIR / "_build" / "liboracle.so"
PATTERN_FILE = ROOT / "tests" / "golden" / "bit_pattern_31.i8"

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("nlevels", C.c_int), ("ini_th", C.c_int), ("min_th", C.c_int),
                ("scale_factor", C.c_float), ("scale", C.c_float * 16), ("inv_scale", C.c_float * 16),
                ("quota", C.c_int * 16), ("umax", C.c_int * 16), ("pattern", C.c_int8 * 1024)]


class LmSummary(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("iterations", C.c_int), ("accepted", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("termination", C.c_int), ("trace_cost", C.POINTER(C.c_double)),
                ("trace_radius", C.POINTER(C.c_double)), ("trace_accepted", C.POINTER(C.c_int))]


class Frame(C.Structure):
    _fields_ = [("n", C.c_int), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("uright", C.c_void_p), ("desc", C.c_void_p),
                ("xmin", C.c_float), ("ymin", C.c_float), ("xmax", C.c_float), ("ymax", C.c_float),
                ("grid_per_px_w", C.c_float), ("grid_per_px_h", C.c_float),
                ("cell_start", C.c_void_p), ("cell_items", C.c_void_p)]


class Bow(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("start", C.c_void_p), ("feat", C.c_void_p)]


class BowData:
    def __init__(self, node_of_feature):
        node_of_feature = np.asarray(node_of_feature)
        order = np.argsort(node_of_feature, kind="stable")
        ids, counts = np.unique(node_of_feature, return_counts=True)
        self.node_id = np.ascontiguousarray(ids, np.uint32)
        self.start = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts)]), np.int32)
        self.feat = np.ascontiguousarray(order, np.uint32)
        b = Bow()
        b.n_nodes = len(self.node_id)
        b.node_id, b.start, b.feat = self.node_id.ctypes.data, self.start.ctypes.data, self.feat.ctypes.data
        self.c = b


def build(force: bool = False) -> pathlib.Path:
    srcs = [ORACLE_DIR / n for n in ("orb_oracle.c", "match_oracle.c", "ba_oracle.c", "frame_oracle.c", "oracle.h")]
    if force or not SO.exists() or any(s.stat().st_mtime > SO.stat().st_mtime for s in srcs if s.exists()):
        if all(s.exists() for s in srcs):
            subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(str(SO))
    L.orc_orb_params_init.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _i8p]
    L.orc_level_size.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_resize_linear_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
    L.orc_gaussian7_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
    L.orc_fast9_16.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int]
    L.orc_fast9_16.restype = C.c_int
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_cv_round_f.argtypes = [C.c_float]
    L.orc_cv_round_f.restype = C.c_int
    L.orc_cos_sin_f.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_cos_sin_f_libm.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_level_candidates.argtypes = [C.POINTER(OrbParams), _u8p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int]
    L.orc_level_candidates.restype = C.c_int
    L.orc_distribute_octtree.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]
    L.orc_distribute_octtree.restype = C.c_int
    L.orc_ic_angle.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i32p]
    L.orc_ic_angle.restype = C.c_float
    L.orc_orb_descriptor.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_float, _i8p, _u8p]
    L.orc_orb_descriptor_libm.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_float, _i8p, _u8p]
    L.orc_orb_extract.argtypes = [C.POINTER(OrbParams), _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p, C.c_int, _i32p]
    L.orc_orb_extract.restype = C.c_int
    L.orc_hamming256.argtypes = [_u8p, _u8p]
    L.orc_hamming256.restype = C.c_int
    L.orc_hamming_matrix.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _u16p]
    L.orc_three_max.argtypes = [_i32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_frame_build_grid.argtypes = [C.POINTER(Frame)]
    L.orc_features_in_area.argtypes = [C.POINTER(Frame), C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _i32p, C.c_int]
    L.orc_features_in_area.restype = C.c_int
    L.orc_match_frame_projection.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _u8p,
                                             C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _f32p, _u8p, _i32p]
    L.orc_match_frame_projection.restype = C.c_int
    L.orc_match_local_map.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _u8p,
                                      C.c_float, C.c_float, _f32p, _u8p, _i32p]
    L.orc_match_local_map.restype = C.c_int
    L.orc_match_frame_keyframe.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _i32p, _f32p, _u8p, C.c_float,
                                           C.c_float, C.c_int, _f32p, _u8p, _i32p]
    L.orc_match_frame_keyframe.restype = C.c_int
    L.orc_match_bow.argtypes = [C.POINTER(Frame), _u8p, C.POINTER(Bow), C.POINTER(Frame), _u8p, C.POINTER(Bow), C.c_int,
                                C.c_float, C.c_int, _i32p]
    L.orc_match_bow.restype = C.c_int
    L.orc_match_triangulation.argtypes = [C.POINTER(Frame), _u8p, C.POINTER(Bow), C.POINTER(Frame), _u8p, C.POINTER(Bow),
                                          _f64p, C.c_float, C.c_float, _f32p, C.c_int, _i32p]
    L.orc_match_triangulation.restype = C.c_int
    L.orc_match_fuse.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _f32p, _i32p, _u8p, C.c_float, _f32p, _i32p]
    L.orc_match_fuse.restype = C.c_int
    L.orc_match_area_best.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _i32p, _u8p, C.c_float, _f32p, C.c_int, _i32p]
    L.orc_match_area_best.restype = C.c_int
    L.orc_match_sim3_projection.argtypes = [C.POINTER(Frame), C.c_int, _u8p, _f32p, _f32p, _i32p, _u8p, C.c_int, _f32p, _u8p, _i32p]
    L.orc_match_sim3_projection.restype = C.c_int
    L.orc_match_sim3_mutual.argtypes = [C.POINTER(Frame), C.POINTER(Frame), _u8p, _f32p, _f32p, _i32p, _u8p, _u8p, _f32p, _f32p,
                                        _i32p, _u8p, C.c_float, _f32p, _f32p, _i32p]
    L.orc_match_sim3_mutual.restype = C.c_int
    L.orc_undistort_points.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_void_p, _f32p, _f32p]
    L.orc_find_depth.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, _f32p]
    L.orc_depth_to_float.argtypes = [_u16p, C.c_int, C.c_float, _f32p]
    L.orc_median_descriptor.argtypes = [_u8p, C.c_int]
    L.orc_is_in_frame.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f32p, _f32p, _u8p, _f32p, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_int, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p]
    L.orc_is_in_frame.restype = None
    L.orc_median_descriptor.restype = C.c_int
    L.orc_sim3_horn.argtypes = [_f64p, _f64p, C.c_int, _f64p, _f64p, C.POINTER(C.c_double)]
    L.orc_sim3_ransac_eval.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f64p, _i32p, _i32p, _f32p, C.c_int, _i32p, C.c_int,
                                       _i32p, _u8p, _f64p]
    L.orc_triangulate.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
    L.orc_triangulate.restype = C.c_int
    L.orc_rgb_to_gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p]
    L.orc_sim3_eval.argtypes = [_f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, C.c_double, _f64p, _f64p, C.c_void_p, _f64p,
                                C.c_void_p]
    L.orc_sim3_solve.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_int, _f64p, _f64p, _u8p,
                                 C.c_void_p]
    L.orc_sim3_solve.restype = C.c_int
    L.orc_quat_plus.argtypes = [_f64p, _f64p, _f64p]
    L.orc_pose_graph_edge.argtypes = [_f64p, _f64p, C.c_double, _f64p, _f64p, C.c_double, _f64p, _f64p, C.c_double, _f64p,
                                      C.c_void_p, C.c_void_p]
    L.orc_pose_graph_solve.argtypes = [C.c_int, _f64p, _f64p, _f64p, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p,
                                       C.c_int, C.c_void_p]
    L.orc_bow_transform.argtypes = [C.c_int, _i32p, _i32p, _u8p, _f64p, _i32p, C.c_int, _u8p, C.c_int, _i32p, _f64p, _i32p]
    L.orc_bow_score.argtypes = [C.c_int, _i32p, _f64p, C.c_int, _i32p, _f64p]
    L.orc_bow_score.restype = C.c_double
    L.orc_se3_exp.argtypes = [_f64p, _f64p, _f64p]
    L.orc_se3_log.argtypes = [_f64p, _f64p, _f64p]
    L.orc_se3_plus.argtypes = [_f64p, _f64p, _f64p]
    L.orc_se3_trans_point.argtypes = [_f64p, _f64p, _f64p]
    L.orc_se3_apply.argtypes = [_f64p, _f64p, _f64p, _f64p]
    L.orc_angle_axis_to_R.argtypes = [_f64p, _f64p]
    L.orc_edge_eval.argtypes = [_f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, C.c_void_p, C.c_void_p]
    L.orc_edge_eval.restype = C.c_int
    L.orc_pose_only_solve.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _u8p, C.c_void_p]
    L.orc_pose_only_solve.restype = C.c_int
    L.orc_ba_lm.argtypes = [C.c_int, _f64p, _u8p, C.c_int, _f64p, C.c_int, _i32p, _i32p, _f64p, _f64p, C.c_void_p,
                            _f64p, C.c_double, C.c_double, C.c_int, C.c_void_p]
    L.orc_local_ba.argtypes = [C.c_int, _f64p, _u8p, C.c_int, _f64p, C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p,
                               C.c_void_p, _u8p, C.c_void_p]
    L.orc_local_ba.restype = C.c_int
    L.orc_ba_schur.argtypes = [C.c_int, _f64p, _u8p, C.c_int, _f64p, C.c_int, _i32p, _i32p, _f64p, _f64p, C.c_void_p,
                               _f64p, C.c_double, C.c_double, C.c_double, _f64p, _f64p, C.POINTER(C.c_double)]
    L.orc_ba_schur.restype = C.c_int
    _lib = L
    return L


def pattern() -> np.ndarray:
    return np.frombuffer(PATTERN_FILE.read_bytes(), dtype=np.int8).copy()


def orb_params(nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7) -> OrbParams:
    p = OrbParams()
    lib().orc_orb_params_init(C.byref(p), nfeatures, np.float32(scale), nlevels, ini_th, min_th, pattern())
    return p


def level_size(p, w, h, level):
    lw, lh = C.c_int(), C.c_int()
    lib().orc_level_size(C.byref(p), w, h, level, C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def resize(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(src, src.shape[1], src.shape[0], src.shape[1], dst, dw, dh, dw)
    return dst


def blur(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().orc_gaussian7_u8(src, src.shape[1], src.shape[0], src.shape[1], dst, src.shape[1])
    return dst


def pyramid(p, img):
    levels = [np.ascontiguousarray(img)]
    for l in range(1, p.nlevels):
        lw, lh = level_size(p, img.shape[1], img.shape[0], l)
        levels.append(resize(levels[-1], lw, lh))
    return levels


def fast(img: np.ndarray, threshold: int, nms: bool = True):
    img = np.ascontiguousarray(img)
    cap = img.size
    xs, ys, sc = (np.empty(cap, np.int32) for _ in range(3))
    n = lib().orc_fast9_16(img, img.shape[1], img.shape[0], img.shape[1], threshold, int(nms), xs, ys, sc, cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def level_candidates(p, img):
    img = np.ascontiguousarray(img)
    cap = img.size
    cx, cy, cr = (np.empty(cap, np.float32) for _ in range(3))
    n = lib().orc_level_candidates(C.byref(p), img, img.shape[1], img.shape[0], img.shape[1], cx, cy, cr, cap)
    return cx[:n].copy(), cy[:n].copy(), cr[:n].copy()


def octtree(cx, cy, cr, w, h, N):
    out = np.empty(max(len(cx), 1) + 8, np.int32)
    n = lib().orc_distribute_octtree(np.ascontiguousarray(cx, np.float32), np.ascontiguousarray(cy, np.float32),
                                     np.ascontiguousarray(cr, np.float32), len(cx), 16, w - 16, 16, h - 16, N,
                                     out, len(out))
    return out[:max(n, 0)].copy()


def extract(p, img, cap=None):
    img = np.ascontiguousarray(img)
    cap = cap or (p.nfeatures + 64)
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    npl = np.zeros(16, np.int32)
    n = lib().orc_orb_extract(C.byref(p), img, img.shape[1], img.shape[0], img.shape[1],
                              kps.ctypes.data, desc, cap, npl)
    return kps[:n].copy(), desc[:n].copy(), npl[:p.nlevels].copy()


def hamming_matrix(A, B):
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    D = np.empty((len(A), len(B)), np.uint16)
    lib().orc_hamming_matrix(A, len(A), B, len(B), D)
    return D


class FrameData:
    """Keeps the numpy arrays alive behind an orc_frame."""

    def __init__(self, x, y, octave, angle, uright, desc, w=640.0, h=480.0):
        self.x = np.ascontiguousarray(x, np.float32)
        self.y = np.ascontiguousarray(y, np.float32)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.angle = np.ascontiguousarray(angle, np.float32)
        self.uright = np.ascontiguousarray(uright, np.float32)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        n = len(self.x)
        self.cell_start = np.zeros(64 * 48 + 1, np.int32)
        self.cell_items = np.zeros(max(n, 1), np.int32)
        f = Frame()
        f.n = n
        f.x, f.y, f.octave = self.x.ctypes.data, self.y.ctypes.data, self.octave.ctypes.data
        f.angle, f.uright, f.desc = self.angle.ctypes.data, self.uright.ctypes.data, self.desc.ctypes.data
        f.xmin, f.ymin, f.xmax, f.ymax = 0.0, 0.0, w, h
        f.grid_per_px_w = np.float32(64.0) / np.float32(w)
        f.grid_per_px_h = np.float32(48.0) / np.float32(h)
        f.cell_start, f.cell_items = self.cell_start.ctypes.data, self.cell_items.ctypes.data
        self.c = f
        lib().orc_frame_build_grid(C.byref(f))


def make_summary(max_it):
    s = LmSummary()
    tc = (C.c_double * (max_it + 1))()
    tr = (C.c_double * (max_it + 1))()
    ta = (C.c_int * (max_it + 1))()
    s.trace_cost, s.trace_radius, s.trace_accepted = tc, tr, ta
    s._keep = (tc, tr, ta)
    return s


def pose_only(prob, trace=False):
    pose = prob["pose0"].copy()
    n = len(prob["pts"])
    outl = np.zeros(n, np.uint8)
    sums = (LmSummary * 2)()
    keep = []
    if trace:
        for k in range(2):
            tc, tr, ta = (C.c_double * 11)(), (C.c_double * 11)(), (C.c_int * 11)()
            sums[k].trace_cost, sums[k].trace_radius, sums[k].trace_accepted = tc, tr, ta
            keep.append((tc, tr, ta))
    ninl = lib().orc_pose_only_solve(n, prob["pts"], prob["obs"], prob["inv_sigma"], prob["cam"], pose, outl,
                                     C.addressof(sums))
    return pose, outl, ninl, sums, keep


def local_ba(prob, stop=None):
    poses, pts = prob["poses"].copy(), prob["points"].copy()
    ne = len(prob["e_cam"])
    erase = np.zeros(ne, np.uint8)
    sums = (LmSummary * 2)()
    rc = lib().orc_local_ba(len(poses), poses, prob["fixed"], len(pts), pts, ne, prob["e_cam"], prob["e_pt"],
                            prob["e_obs"], prob["e_inv_sigma"], prob["cam"], stop, erase, C.addressof(sums))
    return poses, pts, erase, sums, rc


def ba_schur(prob, huber=(0.0, 0.0), point_damping=0.0, active=None):
    nfree = int((prob["fixed"] == 0).sum())
    S = np.zeros((6 * nfree, 6 * nfree))
    b = np.zeros(6 * nfree)
    cost = C.c_double()
    ne = len(prob["e_cam"])
    act = None if active is None else np.ascontiguousarray(active, np.uint8).ctypes.data
    nf = lib().orc_ba_schur(len(prob["poses"]), prob["poses"], prob["fixed"], len(prob["points"]), prob["points"],
                            ne, prob["e_cam"], prob["e_pt"], prob["e_obs"], prob["e_inv_sigma"], act, prob["cam"],
                            huber[0], huber[1], point_damping, S, b, C.byref(cost))
    return S[:6 * nf, :6 * nf], b[:6 * nf], cost.value, nf


def sim3_solve(prob, fix_scale=True, trace=False):
    """Optimizer::solveLoopSim3 on a synth.make_sim3_problem dict -> (pose[6], scale, outlier, inliers, sums)"""
    pose = prob["pose0"].copy()
    scale = np.array([prob["scale0"]], np.float64)
    n = len(prob["cam_match"])
    outl = np.zeros(max(n, 1), np.uint8)
    sums = (LmSummary * 2)()
    inl = lib().orc_sim3_solve(n, prob["cam_match"], prob["pix_curr"], prob["isig_curr"], prob["cam_curr"],
                               prob["pix_match"], prob["isig_match"], prob["cam"][:4].copy(), int(fix_scale), pose, scale,
                               outl, C.cast(sums, C.c_void_p))
    return pose, float(scale[0]), outl[:n], inl, sums


def pose_graph_solve(g, max_iterations=20):
    """Optimizer::solvePoseGraphLoop's solve on a synth.make_pose_graph dict -> (quats, trans, summary)"""
    q, t = g["quats"].copy(), g["trans"].copy()
    s = make_summary(max_iterations)
    lib().orc_pose_graph_solve(len(q), q, t, np.ascontiguousarray(g["scales"], np.float64), int(g["fixed"]), len(g["e_i"]),
                               g["e_i"], g["e_j"], g["q_meas"], g["t_meas"], np.ascontiguousarray(g["s_meas"], np.float64),
                               max_iterations, C.cast(C.pointer(s), C.c_void_p))
    return q, t, s
