"""ctypes binding of libvo_hip.so (include/vo_hip.h).  No CPU fallback: loading or calling
without the HIP library / a gfx950 device raises."""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

PKG = pathlib.Path(__file__).resolve().parent
SO = PKG / "libvo_hip.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class VoError(RuntimeError):
    pass


class LmSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32),
                ("reserved", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("uright", C.c_void_p), ("desc", C.c_void_p),
                ("xmin", C.c_float), ("ymin", C.c_float), ("xmax", C.c_float), ("ymax", C.c_float)]


_lib = None

# every symbol include/vo_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "vo_last_error", "vo_device_count", "vo_version",
    "vo_orb_create", "vo_orb_destroy", "vo_orb_set_stream", "vo_orb_levels", "vo_orb_scale_factor",
    "vo_orb_scale_factors", "vo_orb_features_per_level", "vo_orb_max_keypoints", "vo_orb_extract",
    "vo_orb_extract_batch_dev", "vo_orb_sync", "vo_orb_get_level", "vo_orb_get_candidates",
    "vo_orb_get_level_counts",
    "vo_hamming_matrix_dev", "vo_hamming_matrix_batch_dev", "vo_hamming_matrix",
    "vo_match_frame_projection", "vo_match_local_map",
    "vo_pose_only_solve", "vo_pose_only_solve_dev",
    "vo_ba_create", "vo_ba_destroy", "vo_ba_set_stream", "vo_ba_set_shard", "vo_ba_set_state",
    "vo_ba_get_state", "vo_ba_n_free_cams", "vo_ba_local_ba", "vo_ba_solve", "vo_ba_lm_begin",
    "vo_ba_linearize", "vo_ba_step", "vo_ba_update", "vo_ba_lm_end", "vo_ba_reduced_system",
    "vo_ba_reduced_cost", "vo_ba_debug_schur", "vo_se3_exp", "vo_se3_log",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not SO.exists():
        raise VoError(f"{SO} is missing: run `python -m vo_slam_test_amd.build` (hipcc, gfx950). "
                      "There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
    # process cannot both open the GPU.  Importing torch first makes the loader resolve our
    # NEEDED libamdhip64.so.7 to the copy torch already mapped (same SONAME).  Without torch the
    # library falls back to its RUNPATH (/opt/rocm/lib).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    L = C.CDLL(str(SO))
    L.vo_last_error.restype = C.c_char_p
    L.vo_version.restype = C.c_char_p
    L.vo_orb_scale_factor.restype = C.c_float
    for name in SYMBOLS:
        f = getattr(L, name, None)
        if f is not None and f.restype is C.c_int:
            f.restype = C.c_int
    L.vo_orb_destroy.restype = None
    if hasattr(L, "vo_ba_destroy"):
        L.vo_ba_destroy.restype = None
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        raise VoError(f"{what} failed with status {rc}: {lib().vo_last_error().decode()}")


def _p(a):
    """pointer of a numpy array / torch tensor / int / None as c_void_p"""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class OrbExtractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/myslam/ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self._h = C.c_void_p()
        check(lib().vo_orb_create(C.byref(self._h), int(nfeatures), C.c_float(scaleFactor), int(nlevels),
                                  int(iniThFAST), int(minThFAST)), "vo_orb_create")
        self.nlevels = nlevels

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().vo_orb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def GetLevels(self):
        return lib().vo_orb_levels(self._h)

    def GetScaleFactor(self):
        return lib().vo_orb_scale_factor(self._h)

    def GetScaleFactors(self):
        s = np.zeros(self.nlevels, np.float32)
        check(lib().vo_orb_scale_factors(self._h, _p(s), None))
        return s

    def GetInverseScaleFactors(self):
        s = np.zeros(self.nlevels, np.float32)
        i = np.zeros(self.nlevels, np.float32)
        check(lib().vo_orb_scale_factors(self._h, _p(s), _p(i)))
        return i

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        check(lib().vo_orb_features_per_level(self._h, _p(q)))
        return q

    def max_keypoints(self):
        return lib().vo_orb_max_keypoints(self._h)

    def set_stream(self, stream_ptr: int):
        check(lib().vo_orb_set_stream(self._h, C.c_void_p(stream_ptr)))

    def __call__(self, image: np.ndarray, mask=None):
        """operator()(image, mask, keypoints, descriptors): host image -> (keypoints, descriptors)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        image = np.ascontiguousarray(image, np.uint8)
        cap = self.max_keypoints()
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(lib().vo_orb_extract(self._h, _p(image), image.shape[1], image.shape[0], image.strides[0],
                                   _p(kps), _p(desc), cap, C.byref(n)), "vo_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_dev(self, images, kps, desc, counts):
        """torch uint8 [B,H,W] device tensor -> device outputs (asynchronous)."""
        B, H, W = images.shape
        cap = kps.shape[1]
        check(lib().vo_orb_extract_batch_dev(self._h, _p(images), B, W, H, images.stride(1),
                                             C.c_size_t(images.stride(0)), _p(kps), _p(desc), cap, _p(counts)),
              "vo_orb_extract_batch_dev")

    def sync(self):
        check(lib().vo_orb_sync(self._h), "vo_orb_sync")

    def get_level(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        check(lib().vo_orb_get_level(self._h, frame, level, int(blurred), None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(lib().vo_orb_get_level(self._h, frame, level, int(blurred), _p(out), w.value, C.byref(w), C.byref(h)))
        return out

    def get_candidates(self, frame, level, cap=70000):
        x, y, r = (np.zeros(cap, np.float32) for _ in range(3))
        n = C.c_int()
        check(lib().vo_orb_get_candidates(self._h, frame, level, _p(x), _p(y), _p(r), cap, C.byref(n)))
        return x[:n.value].copy(), y[:n.value].copy(), r[:n.value].copy()

    def get_level_counts(self, frame=0):
        c = np.zeros(self.nlevels, np.int32)
        check(lib().vo_orb_get_level_counts(self._h, frame, _p(c)))
        return c


def hamming_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    d = np.zeros((len(a), len(b)), np.uint16)
    check(lib().vo_hamming_matrix(_p(a), len(a), _p(b), len(b), _p(d)), "vo_hamming_matrix")
    return d


def hamming_matrix_dev(a, b, d, stream=0):
    check(lib().vo_hamming_matrix_dev(_p(a), a.shape[0], _p(b), b.shape[0], _p(d), C.c_void_p(stream)))


def hamming_matrix_batch_dev(a, b, d, stream=0):
    """a [P,na,32], b [P,nb,32], d [P,na,nb] device tensors"""
    P, na, nb = a.shape[0], a.shape[1], b.shape[1]
    check(lib().vo_hamming_matrix_batch_dev(_p(a), na, C.c_size_t(a.stride(0) // 32), _p(b), nb,
                                            C.c_size_t(b.stride(0) // 32), _p(d), C.c_size_t(d.stride(0)), P,
                                            C.c_void_p(stream)))


class FrameArrays:
    def __init__(self, x, y, octave, angle, uright, desc, w=640.0, h=480.0):
        self.x = np.ascontiguousarray(x, np.float32)
        self.y = np.ascontiguousarray(y, np.float32)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.angle = np.ascontiguousarray(angle, np.float32)
        self.uright = np.ascontiguousarray(uright, np.float32)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        v = FrameView()
        v.n = len(self.x)
        v.x, v.y, v.octave = self.x.ctypes.data, self.y.ctypes.data, self.octave.ctypes.data
        v.angle, v.uright, v.desc = self.angle.ctypes.data, self.uright.ctypes.data, self.desc.ctypes.data
        v.xmin, v.ymin, v.xmax, v.ymax = 0.0, 0.0, w, h
        self.view = v


class Matcher:
    """Mirror of myslam::Matcher's projection searches (reference include/myslam/matcher.h:9-45)."""

    def __init__(self, ratio: float = 0.8):
        self.ratio_ = ratio

    def searchByProjection_frame(self, cur: FrameArrays, q, radius, bf, direction, checkRot, scale_factors,
                                 blocked=None):
        nq = len(q["flags"])
        assigned = np.full(cur.view.n, -1, np.int32)
        n = C.c_int()
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(lib().vo_match_frame_projection(
            C.byref(cur.view), nq, _p(q["flags"]), _p(q["u"]), _p(q["v"]), _p(q["invz"]), _p(q["octave"]),
            _p(q["angle"]), _p(q["desc"]), C.c_float(radius), C.c_float(bf), int(direction), int(checkRot),
            len(sf), _p(sf), _p(blocked), _p(assigned), C.byref(n)), "vo_match_frame_projection")
        return n.value, assigned

    def searchByProjection_localmap(self, cur: FrameArrays, q, thRadius, scale_factors, blocked=None):
        nq = len(q["flags"])
        assigned = np.full(cur.view.n, -1, np.int32)
        n = C.c_int()
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(lib().vo_match_local_map(
            C.byref(cur.view), nq, _p(q["flags"]), _p(q["u"]), _p(q["v"]), _p(q["ur"]), _p(q["level"]),
            _p(q["viewcos"]), _p(q["desc"]), C.c_float(thRadius), C.c_float(self.ratio_), _p(sf), _p(blocked),
            _p(assigned), C.byref(n)), "vo_match_local_map")
        return n.value, assigned

    @staticmethod
    def computeDistance(a, b) -> int:
        return int(hamming_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])
