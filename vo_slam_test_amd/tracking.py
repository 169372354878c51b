"""Batched, device-resident tracked-frame path (what VisualOdometry::trackWithMotionModel + trackLocalMap do per
frame, reference src/visualOdometry.cpp:228-251, 745-775): ORB extraction -> Frame post-processing (undistort,
depth, grid) -> searchByProjection against the last frame's map points -> solvePoseOnlySE3 -> searchByProjection
against the local map points -> solvePoseOnlySE3.  Every frame of the batch is an independent tracking problem
(one camera stream each); nothing leaves HBM between the image and the pose.  torch is plumbing only (device
buffers, the stream); all arithmetic is libvo_hip.so through the C-ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as vo


class BatchTracker:
    def __init__(self, batch, extractor, intrinsics5, dist_coef=None, width=640, height=480, n_last=1024, n_local=2048,
                 max_features=None, stream=None, extract_stream=None):
        import torch
        self.torch = torch
        self.B, self.ext = batch, extractor
        self.stream = stream if stream is not None else torch.cuda.current_stream()
        self.st = self.stream.cuda_stream
        # one stream for the whole path (extraction -> frames -> matches -> pose) unless the caller gives the extraction a
        # stream of its own (shared by several trackers: their extractions then take turns on it, in call order)
        self.ext_stream = extract_stream if extract_stream is not None else self.stream
        extractor.set_stream(self.ext_stream.cuda_stream)
        self.build_done = None
        self.kcap = extractor.max_keypoints()
        # feature slots per frame: what the extractor can emit, rounded up (the searches' LDS -- 11 bytes per slot and
        # frame -- and every per-feature buffer scale with it; 2048 slots for 1005 features cost the kernels that run
        # next to the replay a third of their LDS)
        self.cap = max_features or max(256, (self.kcap + 63) // 64 * 64)
        self.W, self.H = width, height
        self.cam5 = np.ascontiguousarray(intrinsics5, np.float32)
        self.frames = vo.Frames(batch, self.cap, self.cam5, dist_coef, float(width), float(height))
        self.sf = extractor.GetScaleFactors()
        self.n_last, self.n_local = n_last, n_local
        dev = "cuda"
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        B, cap = batch, self.cap
        self.kps, self.desc, self.cnt = z((B, self.kcap, 28), torch.uint8), z((B, self.kcap, 32), torch.uint8), z(B, torch.int32)
        # last frame's map points (mode 0 queries; u, v, 1/z, flags come from the projection kernel)
        self.q0 = dict(flags=z((B, n_last), torch.uint8), u=z((B, n_last), torch.float32), v=z((B, n_last), torch.float32),
                       aux=z((B, n_last), torch.float32), level=z((B, n_last), torch.int32), angle=z((B, n_last), torch.float32),
                       desc=z((B, n_last, 32), torch.uint8), n_queries=n_last)
        self.p0, self.pf0 = z((B, n_last, 3), torch.float64), z((B, n_last), torch.uint8)
        # local map points (mode 1 queries, pre-projected by the caller like Frame::isInFrame does)
        self.q1 = dict(flags=z((B, n_local), torch.uint8), u=z((B, n_local), torch.float32), v=z((B, n_local), torch.float32),
                       aux=z((B, n_local), torch.float32), level=z((B, n_local), torch.int32),
                       viewcos=z((B, n_local), torch.float32), desc=z((B, n_local, 32), torch.uint8), n_queries=n_local)
        self.p1 = z((B, n_local, 3), torch.float64)
        self.Tcw, self.pose0 = z((B, 12), torch.float64), z((B, 6), torch.float64)
        self.pose = z((B, 6), torch.float64)
        self.assigned, self.nm = z((B, cap), torch.int32), z(B, torch.int32)
        self.fpoint, self.fhas, self.fobs = z((B, cap, 3), torch.float64), z((B, cap), torch.uint8), z((B, cap), torch.uint8)
        self.pts, self.obs, self.isg = z((B, cap, 3), torch.float64), z((B, cap, 3), torch.float64), z((B, cap), torch.float64)
        self.ranges, self.index = z((B, 2), torch.int32), z((B, cap), torch.int32)
        self.outlier, self.ninl = z((B, cap), torch.uint8), z(B, torch.int32)
        self.cam5d = torch.from_numpy(self.cam5.astype(np.float64)).to(dev)
        self.cam4 = np.ascontiguousarray(self.cam5[:4])
        self.assigned0 = None  # snapshot of the first search (tests)

    def close(self):
        self.frames.close()

    def set_map(self, Tcw12, pose6, last, local):
        """Tcw12 [B,12], pose6 [B,6] = se3 log of the same poses; last: dict(points [B,n,3], flags, octave, angle, desc);
        local: dict(points, flags, u, v, ur, level, viewcos, desc) -- numpy arrays, shorter than the capacity is fine"""
        t = self.torch

        def put(dst, src):
            src = np.asarray(src)
            dst.zero_()
            dst[:, :src.shape[1]].copy_(t.from_numpy(np.ascontiguousarray(src)).to(dst.device))

        self.Tcw.copy_(t.from_numpy(np.ascontiguousarray(Tcw12, np.float64)))
        self.pose0.copy_(t.from_numpy(np.ascontiguousarray(pose6, np.float64)))
        put(self.p0, last["points"]), put(self.pf0, last["flags"]), put(self.q0["level"], last["octave"])
        put(self.q0["angle"], last["angle"]), put(self.q0["desc"], last["desc"])
        put(self.p1, local["points"])
        for k, src in (("flags", "flags"), ("u", "u"), ("v", "v"), ("aux", "ur"), ("level", "level"),
                       ("viewcos", "viewcos"), ("desc", "desc")):
            put(self.q1[k], local[src])
        self.q0["n_queries"], self.q1["n_queries"] = np.asarray(last["flags"]).shape[1], np.asarray(local["flags"]).shape[1]

    def _solve_pose(self):
        L, st = vo.lib(), self.st
        vo.check(L.vo_track_gather_dev(self.frames._h, 0, self.B, vo._p(self.fpoint), vo._p(self.fhas), vo._p(self.sf),
                                       len(self.sf), vo._p(self.pts), vo._p(self.obs), vo._p(self.isg), vo._p(self.ranges),
                                       vo._p(self.index), C.c_void_p(st)), "vo_track_gather_dev")
        vo.check(L.vo_pose_only_solve_ranges_dev(self.B, vo._p(self.ranges), vo._p(self.pts), vo._p(self.obs), vo._p(self.isg),
                                                 vo._p(self.cam5d), vo._p(self.pose), vo._p(self.outlier), vo._p(self.ninl),
                                                 None, C.c_void_p(st)), "vo_pose_only_solve_ranges_dev")

    def track(self, images, depth=None, inv_depth_scale=1.0, radius=15.0, th_radius=3.0, ratio=0.8, direction=0,
              keep_first=False, events=False, after=None):
        """images: uint8 [B,H,W] device tensor, depth: float32 / int16 [B,H,W] device tensor or None.  Asynchronous on the
        tracker's stream; results: self.pose [B,6], self.ninl [B], self.assigned [B,cap].  after: an event the extraction
        waits for; self.extract_done is recorded behind the extraction -- two trackers on two streams chained this way take
        turns on the issue-bound extraction while the other's latency-bound searches and pose solves (one wavefront per
        frame) run next to it."""
        L, st, B = vo.lib(), self.st, self.B
        evs = {}

        def mark(name, end=False):  # HIP events on the launching stream around a stage (bench.py's live stage times)
            if events:
                e = self.torch.cuda.Event(enable_timing=True)
                e.record(self.stream)
                evs.setdefault(name, []).append(e)

        with self.torch.cuda.stream(self.stream):
            if after is not None:
                self.ext_stream.wait_event(after)
            if self.ext_stream is not self.stream and self.build_done is not None:
                self.ext_stream.wait_event(self.build_done)  # the previous batch's key-points have been consumed
            if events:  # on the extraction's stream: the whole extraction of this batch
                e0 = self.torch.cuda.Event(enable_timing=True)
                e0.record(self.ext_stream)
                evs["extract"] = [e0]
            self.ext.extract_batch_dev(images, self.kps, self.desc, self.cnt)
            self.extract_done = self.torch.cuda.Event(enable_timing=events)
            self.extract_done.record(self.ext_stream)
            if events:
                evs["extract"].append(self.extract_done)
            if self.ext_stream is not self.stream:
                self.stream.wait_event(self.extract_done)
            mark("frame_post")
            self.frames.build_dev(self.kps, self.desc, self.cnt, depth, inv_depth_scale, stream=st)
            mark("frame_post")
            if self.ext_stream is not self.stream:
                self.build_done = self.torch.cuda.Event()
                self.build_done.record(self.stream)
            mark("match_last_frame")
            nq0 = self.q0["n_queries"]
            vo.check(L.vo_track_project_dev(B, nq0, self.n_last, vo._p(self.Tcw), vo._p(self.p0), vo._p(self.pf0),
                                            vo._p(self.cam4), 0, int(self.W), 0, int(self.H), vo._p(self.q0["flags"]),
                                            vo._p(self.q0["u"]), vo._p(self.q0["v"]), vo._p(self.q0["aux"]), C.c_void_p(st)),
                     "vo_track_project_dev")
            self.assigned.fill_(-1), self.fhas.zero_(), self.fobs.zero_()
            self.pose.copy_(self.pose0)
            self.frames.match_dev(B, self.q0, vo.Frames.MODE_FRAME, self.sf, radius=radius, bf=float(self.cam5[4]),
                                  direction=direction, check_rot=1, assigned=self.assigned, n_matches=self.nm, stream=st)
            vo.check(L.vo_track_scatter_dev(self.frames._h, 0, B, vo._p(self.assigned), vo._p(self.p0), vo._p(self.q0["flags"]),
                                            self.n_last, vo._p(self.fpoint), vo._p(self.fhas), vo._p(self.fobs), C.c_void_p(st)),
                     "vo_track_scatter_dev")
            mark("match_last_frame")
            mark("pose_only_1")
            self._solve_pose()
            mark("pose_only_1")
            if keep_first:
                self.assigned0, self.pose_first, self.ninl_first = self.assigned.clone(), self.pose.clone(), self.ninl.clone()
            # local map: features that already hold an observed map point are blocked (:314); new claims are added
            mark("match_local_map")
            self.assigned.fill_(-1)
            self.frames.match_dev(B, self.q1, vo.Frames.MODE_LOCAL_MAP, self.sf, radius=th_radius, ratio=ratio,
                                  feature_mask=self.fobs, assigned=self.assigned, n_matches=self.nm, stream=st)
            vo.check(L.vo_track_scatter_dev(self.frames._h, 0, B, vo._p(self.assigned), vo._p(self.p1), vo._p(self.q1["flags"]),
                                            self.n_local, vo._p(self.fpoint), vo._p(self.fhas), vo._p(self.fobs), C.c_void_p(st)),
                     "vo_track_scatter_dev")
            mark("match_local_map")
            mark("pose_only_2")
            self._solve_pose()
            mark("pose_only_2")
        return evs
