#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on synthetic inputs.

One "step" = B synthetic 640x480 frames (already resident in HBM, with their depth images and the map they are tracked
against) through the whole per-frame tracking path: ORB extraction (8-level pyramid, per-cell FAST + NMS, oct-tree,
orientation, blur, steered BRIEF), frame post-processing (undistort, depth, 64x48 grid), searchByProjection against the
last frame's map points, solvePoseOnlySE3, searchByProjection against the local map, solvePoseOnlySE3 -- what
VisualOdometry::trackWithMotionModel + trackLocalMap do per frame (visualOdometry.cpp:228-251, 745-775).
`value` = tracked frames/s over all ranks (weak scaling: every rank owns its own batch of independent camera streams;
no collective on this path).  The same JSON line carries BASELINE configs[1] as written (extract + brute-force
1000x1000 Hamming, `extract_bruteforce_match`), local BA (configs[3]: one problem sharded over the ranks with two
all-reduces per LM iteration, and independent problems per GPU), pose-only BA (configs[2]) and the loop-closure sized
problems (configs[4]).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6  # vector/matrix FP64 peak assumed in BASELINE.md


def level_sizes(w, h, nlevels=8, sf=1.2):
    s = np.float32(1.0)
    out = []
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
    return out


def stage_bytes_per_frame(w, h, nkp, ncand):
    """Algorithmic bytes each stage must move per frame."""
    px = [a * b for a, b in level_sizes(w, h)]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),          # read level l-1, write level l
        "fast": sum(px) + 4 * ncand,                     # read every level once, write candidates
        "octree": 8 * ncand + 4 * nkp,                   # read candidates (+labels), write selection
        "offsets": 64,
        "blur": 2 * sum(px),                             # read level, write blurred level
        "describe": nkp * (749 + 512 + 28 + 32),         # patch + 512 samples in, key-point + descriptor out
        "hamming": 2 * 1000 * 32 + 1000 * 1000 * 2,      # two descriptor sets in, u16 matrix out
    }


def ba_flops_per_iteration(prob):
    """F_ba of SURVEY 8d for one LM iteration of a BA problem dict (synth.make_lba_problem layout)."""
    ne, npts = len(prob["e_cam"]), len(prob["points"])
    nc = int((np.asarray(prob["fixed"]) == 0).sum())
    k = np.bincount(prob["e_pt"], minlength=npts).astype(np.float64)
    m = np.where(prob["e_obs"][:, 2] >= 0, 3, 2).astype(np.float64)
    return float((150 + 108 * m).sum() + (50 + 144 * k + 216 * k * (k + 1) / 2).sum() + (6 * nc) ** 3 / 3 + 2 * (6 * nc) ** 2 + 60 * ne)


def tracking_bytes_per_frame(nkp, n_q0, n_q1, n_obs):
    """Algorithmic bytes of the tracking stages per frame: key-points + descriptors in and the feature store out; per
    query its descriptor and projection plus ~12 gated candidates (feature record 16 B + descriptor 32 B) and the 4-byte
    assignment per feature; pose-only reads 56 B per observation per evaluation."""
    return {
        "frame_post": nkp * (28 + 32 + 2) + nkp * (20 + 32 + 2) + 4 * 3073,
        "match_last_frame": n_q0 * (32 + 16 + 12 * 48) + 8 * nkp,
        "match_local_map": n_q1 * (32 + 20 + 12 * 48) + 8 * nkp,
        "pose_only_1": n_obs * 56, "pose_only_2": n_obs * 56,
    }


def cpu_track_one(orc, p, sf, img, raw_depth, mp, cam5, dist_coef, inv_depth):
    """The oracle's tracked frame (the same stages as vo_tracker, one frame, one thread; tests/track_ref.py): returns the
    second solve's inlier count."""
    from track_ref import track_frame
    H, W = img.shape
    k, d, _ = orc.extract(p, img)
    n = len(k)
    x, y = np.ascontiguousarray(k["x"]), np.ascontiguousarray(k["y"])
    ux, uy = np.zeros(n, np.float32), np.zeros(n, np.float32)
    orc.lib().orc_undistort_points(n, x, y, cam5[:4].copy(), dist_coef.ctypes.data, ux, uy)
    dimg = np.zeros((H, W), np.float32)
    orc.lib().orc_depth_to_float(np.ascontiguousarray(raw_depth).reshape(-1), H * W, inv_depth, dimg.reshape(-1))
    ur, dep = np.zeros(n, np.float32), np.zeros(n, np.float32)
    orc.lib().orc_find_depth(n, x, y, ux, dimg, W, H, W, float(cam5[4]), ur, dep)
    T, pose6, la, lo = mp
    w = track_frame(orc, k, d, ux, uy, ur, T, pose6, la, lo, cam5, sf, W, H)
    return w["inliers_2"], (k, d, ux, uy, dep)


def _cpu_worker(arg):
    """one host core: the oracle's tracked frame on `n` synthetic frames (spawned process)"""
    seed, n = arg
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    from vo_slam_test_amd import synth
    p = orc.orb_params()
    sf = np.array(list(p.scale)[:8], np.float32)
    cam5 = synth.CAM.astype(np.float32)
    inv = float(np.float32(1.0) / np.float32(synth.DEPTH_SCALE))
    frames = synth.make_frames(2, start=seed * 7)
    depth = [synth.make_depth(seed * 7 + i) for i in range(2)]
    maps = []
    for i in range(2):   # first touch (library load, page-in) + the map the frames are tracked against
        k, d, _ = orc.extract(p, frames[i])
        dep = np.where(depth[i][np.clip(k["y"].astype(int), 0, 479), np.clip(k["x"].astype(int), 0, 639)] > 0,
                       depth[i][np.clip(k["y"].astype(int), 0, 479), np.clip(k["x"].astype(int), 0, 639)] * inv, -1).astype(np.float32)
        maps.append(synth.make_tracking_map(k["x"], k["y"], k["octave"], k["angle"], d, dep, seed=i))
    t0 = time.perf_counter()
    for i in range(n):
        cpu_track_one(orc, p, sf, frames[i & 1], depth[i & 1], maps[i & 1], cam5, synth.DIST, inv)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step (resident in HBM; throughput saturates near 1024: "
                                                             "188 k frames/s at 64, 246 k at 256, 268 k at 512, 275 k at 1024, 278 k at 2048)")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight: the issue-bound extraction of one batch runs on a "
                    "shared stream next to the latency-bound searches / pose solves (one wavefront per frame, high-priority "
                    "streams) of the previous ones; 1 = everything on one stream")
    ap.add_argument("--unique-frames", type=int, default=32, help="distinct synthetic frames the batch is built from (repeated to --batch; "
                    "every copy is its own HBM buffer either way).  32 keeps the set-up short; --unique-frames 1024 gives the "
                    "data-dependent kernels -- FAST survivor lists, oct-tree depth, the claim replays -- 1024 different cases "
                    "(set-up takes ~1 min longer; measured once per round, DESIGN.md section 7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-bruteforce", action="store_true", help="skip the extract + all-pairs Hamming measurement (configs[1] as written)")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the one-frame-at-a-time and ingest-inclusive legs (profiling "
                    "runs: their batch-1 launches would be averaged into the per-kernel figures of the 1024-frame step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks "
                    "share one GPU to exercise the N > 1 code path on a single-GPU box)")
    ap.add_argument("--dist-at-one-rank", action="store_true",
                    help="one rank, but a torch.distributed process group of size 1 over --backend and the SHARDED forms of the "
                         "config-3 / config-4 legs on handles with VO_BA_OPT_COLLECTIVES_AT_ONE_RANK: every collective of the "
                         "loop is a real all_reduce (nccl = RCCL) -- brings the multi-GPU code path up on a one-GPU box")
    ap.add_argument("--collective-timeout", type=float, default=180.0,
                    help="seconds without progress at a barrier / all-reduce after which the rank exits with status 70 "
                         "(a hung collective must end the run, not the box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start N fresh ranks (one per GPU) BEFORE anything here touches the GPU -- a process that
        # has initialised HIP must never be replaced by another program -- and relay rank 0's JSON line
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(pathlib.Path(__file__).resolve()), *sys.argv[1:]]
        r = subprocess.run(cmd, env=env)
        raise SystemExit(r.returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_dev = local_rank % max(1, torch.cuda.device_count()) if args.backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dist = None
    dist_one = args.dist_at_one_rank and world == 1
    # Every wait on another rank has a deadline: the process group's own timeout (the NCCL watchdog aborts the process, gloo
    # raises) and, above it, a watchdog thread that ends THIS process with status 70 when no barrier / all-reduce has
    # completed for --collective-timeout seconds.  (os._exit: the process ends, nothing is exec'ed.)
    from vo_slam_test_amd.watchdog import CollectiveWatchdog
    wd = CollectiveWatchdog(args.collective_timeout, rank)   # (started below when there are other ranks to wait for)
    _mark = wd.mark

    if world > 1 or dist_one:
        import datetime
        import threading
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=float(args.collective_timeout))
        if dist_one:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            init = dict(init_method=f"tcp://127.0.0.1:{sk.getsockname()[1]}", rank=0, world_size=1)
            sk.close()
        else:
            init = {}
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), timeout=tmo, **init)
        else:
            dist.init_process_group(args.backend, timeout=tmo, **init)

        wd.start()

    from vo_slam_test_amd import _lib as vo
    from vo_slam_test_amd import synth

    def barrier():
        if dist is not None:
            wd.arm("barrier (waiting)")
            dist.barrier()
            wd.disarm("barrier")
        torch.cuda.synchronize()

    W, H, B = 640, 480, args.batch
    n_unique = max(1, min(B, args.unique_frames))
    uniq = synth.make_frames(n_unique, start=rank * 1000)
    uniq_depth = np.stack([synth.make_depth(rank * 1000 + i) for i in range(n_unique)])
    frames_np = np.stack([uniq[i % n_unique] for i in range(B)])
    inv_depth = float(np.float32(1.0) / np.float32(synth.DEPTH_SCALE))
    cam5 = synth.CAM.astype(np.float32)
    n_pipe = max(1, args.pipeline)
    prio_ext, prio_tail = (int(x) for x in os.environ.get("VO_BENCH_PRIO", "0,-1").split(","))
    stream = torch.cuda.Stream(priority=prio_ext)  # extraction (shared by the batches in flight); also the brute-force leg
    ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)       # set-up and the brute-force leg (the trackers own theirs)
    ext.set_stream(stream.cuda_stream)
    cap = ext.max_keypoints()
    NM = 1000
    from vo_slam_test_amd.tracking import load_maps
    with torch.cuda.stream(stream):
        frames = torch.from_numpy(frames_np).cuda()
        depth = torch.from_numpy(np.stack([uniq_depth[i % n_unique] for i in range(B)]).view(np.int16)).cuda()
        kps0 = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
        desc0 = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        cnt0 = torch.zeros(B, dtype=torch.int32, device="cuda")
    # ---- the map every frame is tracked against (resident in HBM, like the frames): built once from the features of the
    # unique frames -- last frame's map points = the frame's own features back-projected with their depth, local map = two
    # noisy copies of them with the normals / distance ranges Frame::isInFrame reads (synth.make_tracking_map); TUM fr1
    # distortion coefficients (example.yaml:25-29)
    fstore = vo.Frames(n_unique, max(256, (cap + 63) // 64 * 64), cam5, synth.DIST, float(W), float(H))
    with torch.cuda.stream(stream):
        ext.extract_batch_dev(frames[:n_unique], kps0[:n_unique], desc0[:n_unique], cnt0[:n_unique])
        fstore.build_dev(kps0[:n_unique], desc0[:n_unique], cnt0[:n_unique], depth[:n_unique], inv_depth, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    ext.sync()
    maps = []
    for i in range(n_unique):
        fr = fstore.download(i, stream=stream.cuda_stream)
        maps.append(synth.make_tracking_map(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["depth"], seed=i))
    fstore.close()
    # The trackers: vo_tracker runs the whole tracked-frame path behind the C-ABI (csrc/tracker.hip).  n_pipe of them share
    # the extraction stream; their searches and pose solves run on high-priority streams of their own.
    trks = [vo.Tracker(B, cam5, synth.DIST, W, H, max_last=1100, max_local=2200, inv_depth_scale=inv_depth,
                       extract_stream=stream.cuda_stream if (n_pipe > 1 and not os.environ.get("VO_BENCH_OWN_EXT_STREAMS")) else None,
                       single_stream=(n_pipe == 1))
            for _ in range(n_pipe)]
    trk = trks[0]
    exts = [t.extractor() for t in trks]
    if os.environ.get("VO_BENCH_POSE_BLOCK"):   # developer A/B: threads per frame of the pose-only solver (0 auto, 64, 128, 256)
        vo.set_option("pose_block", int(os.environ["VO_BENCH_POSE_BLOCK"]))
    if os.environ.get("VO_BENCH_DESCBLUR"):   # developer A/B: 1 = blurred planes instead of the descriptor kernel's own window blur
        for e in exts:
            e.set_describe_blur(int(os.environ["VO_BENCH_DESCBLUR"]))
        ext.set_describe_blur(int(os.environ["VO_BENCH_DESCBLUR"]))
    if os.environ.get("VO_BENCH_BLUR"):   # developer A/B: 1 = the VALU blur in the trackers' extractors
        for e in exts:
            e.set_blur_kernel(int(os.environ["VO_BENCH_BLUR"]))
    all_maps = [maps[f % n_unique] for f in range(B)]
    for t in trks:
        load_maps(t, all_maps, 1100, 2200)
    n_q0 = float(np.mean([(m[2]["flags"] & 1).sum() for m in maps]))
    n_q1 = float(np.mean([(m[3]["valid"] & 1).sum() for m in maps]))
    n_steps_done = [0]

    def step(only=None):
        """one batch of B frames through the tracked-frame path (ONE C call: vo_tracker_track_dev); consecutive steps
        alternate between the batches in flight"""
        t = trks[n_steps_done[0] % n_pipe] if only is None else only
        n_steps_done[0] += 1
        t.track_dev(frames, depth)

    for _ in range(max(args.warmup, n_pipe)):
        step()
    barrier()
    for t in trks:
        res = t.results()   # synchronises; raises on dropped key-points / exhausted candidate pools
        counts = t.get(t.KEYPOINT_COUNTS)
        assert counts.min() >= (NM if n_unique <= 32 else 900), f"synthetic frames must yield >= {NM} key-points, got {counts.min()}"
        ninl = res["n_inliers"]
        assert ninl.min() >= 100, f"tracking must keep >= 100 pose inliers per frame, got {ninl.min()}"
        assert not res["status"].any(), "every synthetic frame must be tracked (status 0)"
    # FAST candidates per frame: the mean over the batch's distinct frames (the byte counts of the FAST / oct-tree stages use it)
    ncand = float(np.mean([sum(len(exts[0].get_candidates(f, l)[0]) for l in range(8)) for f in range(n_unique)]))
    # reference point: the same step with nothing overlapped -- ONE batch in flight on ONE stream (a tracker of its own in
    # single-stream mode when the timed region pipelines), its steps back to back with one synchronisation at the end.  (Until
    # round 5 this leg synchronised the host after every batch: the clock drops in the idle gaps and every kernel read 5-10 %
    # longer than in the kernel trace of the same launches -- FAST 0.72 against 0.655 ms; profiles/README.md, round 6.)
    barrier()
    if n_pipe == 1:
        trk_one = trk
    else:
        trk_one = vo.Tracker(B, cam5, synth.DIST, W, H, max_last=1100, max_local=2200, inv_depth_scale=inv_depth, single_stream=True)
        load_maps(trk_one, all_maps, 1100, 2200)
    ext_one = trk_one.extractor()
    for _ in range(2):
        trk_one.track_dev(frames, depth)
    trk_one.sync()
    ext_one.set_timing(True)
    trk_one.set_timing(True)
    n_serial = max(4, args.steps // 3)
    ts0 = time.perf_counter()
    for _ in range(n_serial):
        trk_one.track_dev(frames, depth)
    trk_one.sync()
    serial_ms = (time.perf_counter() - ts0) / n_serial * 1e3
    sm, nc = ext_one.get_timing()
    ext_one.set_timing(False)
    serial_stage_ms = {k: v / max(nc, 1) for k, v in sm.items()}
    tm, tc = trk_one.get_timing()
    trk_one.set_timing(False)
    for name in ("frame_post", "match_last_frame", "pose_only_1", "match_local_map", "pose_only_2"):
        serial_stage_ms[name] = tm[name] / max(tc, 1)
    if trk_one is not trk:
        trk_one.close()
    # timed region: instrumented as well (per-kernel HIP events on the launching stream; the un-instrumented extractor,
    # whose blur runs on a side stream, measured the same step time with two batches in flight: 4.74 ms either way)
    for e in exts:
        e.set_timing(True)
    for t in trks:
        t.set_timing(True)
    n_steps_done[0] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    stage_ms, ncalls = {}, 0
    for e in exts:
        sm, nc = e.get_timing()
        e.set_timing(False)
        ncalls += nc
        for k, v in sm.items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    stage_ms = {k: v / max(ncalls, 1) for k, v in stage_ms.items()}
    tsum, tcalls = {}, 0
    for t in trks:
        tm, tc = t.get_timing()
        t.set_timing(False)
        tcalls += tc
        for k, v in tm.items():
            tsum[k] = tsum.get(k, 0.0) + v
    for name in ("extract", "frame_post", "match_last_frame", "pose_only_1", "match_local_map", "pose_only_2"):
        stage_ms[name] = tsum[name] / max(tcalls, 1)
    frames_per_s = world * B * args.steps / elapsed
    res = trk.results()
    n_match0 = float(res["n_matches_last"].mean())
    n_obs2 = float(trk.get(trk.FEATURE_HAS_POINT).sum()) / B

    nkp = int(counts.mean())
    sb = stage_bytes_per_frame(W, H, nkp, ncand)
    sb.update(tracking_bytes_per_frame(nkp, n_q0, n_q1, n_obs2))
    ext_keys = ("pyramid", "fast", "octree", "blur", "describe")
    sb["extract"] = sum(sb[k] for k in ext_keys)
    # With the blur inside the descriptor kernel (VO_ORB_OPT_DESCRIBE_BLUR = 0, the default) there is no blur launch: the `describe`
    # stage then does the work of BOTH rows of SURVEY 8d -- "read for blur + write blurred" (2 x 950 532 B per frame) and the
    # orientation / descriptor gathers + outputs ((749 + 512 + 60) B per key-point) -- and is priced with their sum, so that the
    # stages still add up to B_ext; `describe_gathers_only_frac` below is the same kernel against the gather row alone.
    describe_covers_blur = serial_stage_ms.get("blur", 0.0) <= 0.02
    sb_describe_gathers = sb["describe"]
    if describe_covers_blur:
        sb["describe"] += sb["blur"]
        sb["blur"] = 0
    # The dominant kernel: the longest single-kernel stage of the path in the timed region (HIP events on the launching
    # stream around every kernel; with two batches in flight the other batch's kernels share the CUs with it, which
    # stretches every launch while the step gets shorter -- the same kernel with one batch in flight is under "alone")
    # (a stage that launches nothing -- the blur when the descriptor kernel blurs its own windows, VO_ORB_OPT_DESCRIBE_BLUR = 0 --
    #  reads a few microseconds of event overhead: it is not a kernel)
    kernel_stages = [k for k in serial_stage_ms if k not in ("offsets", "pose_only_1", "pose_only_2") and serial_stage_ms[k] > 0.02]
    # ranked by the durations INSIDE the timed region (the undisturbed ranking picked describe over FAST on a
    # 0.1 % margin); every kernel's fraction is in `per_kernel` below either way
    # (per LAUNCH: the pyramid stage is seven launches of k_resize4, the longest of them 0.14 ms -- as a stage it is stretched
    #  most by the other batch's kernels and would outrank every single kernel)
    launches = {"pyramid": 7}
    dom = max(kernel_stages, key=lambda k: stage_ms.get(k, 0.0) / launches.get(k, 1))
    achieved = sb[dom] * B / (stage_ms[dom] * 1e-3) / 1e9
    traffic = None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            traffic = json.loads(tf.read_text()).get(dom)
        except Exception:
            traffic = None
    alone = sb[dom] * B / (serial_stage_ms[dom] * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "bytes_per_launch": sb[dom] * B, "avg_launch_ms": round(stage_ms[dom], 4),
                "alone": {"avg_launch_ms": round(serial_stage_ms[dom], 4), "achieved": round(alone, 2),
                          "frac": round(alone / HBM_PEAK_GBS, 5), "note": "the same kernel with one batch in flight"}}
    if describe_covers_blur:
        roofline["describe_gather_row_only"] = {
            "bytes_per_launch": sb_describe_gathers * B,
            "alone_frac": round(sb_describe_gathers * B / (serial_stage_ms["describe"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "frac": round(sb_describe_gathers * B / (stage_ms["describe"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        roofline["describe_bytes_note"] = ("describe = k_describe_win, blur included: priced with SURVEY 8d's blur row (2 x 950 532 B per "
                                           "frame) + gather / output row ((749 + 512 + 60) B per key-point); against the gather row alone "
                                           "its alone_frac is " + str(round(sb_describe_gathers * B / (serial_stage_ms["describe"] * 1e-3) / 1e9
                                                                            / HBM_PEAK_GBS, 5)))
    roofline["per_kernel"] = {
        k: {"bytes_per_launch": sb[k] * B, "avg_launch_ms": round(stage_ms[k], 4),
            "frac": round(sb[k] * B / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "alone_ms": round(serial_stage_ms[k], 4),
            "alone_frac": round(sb[k] * B / (serial_stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        for k in kernel_stages if stage_ms.get(k, 0.0) > 0 and serial_stage_ms[k] > 0}
    # The binding unit next to the HBM fraction (VERDICT r5 #1c): vector instructions per launch from the committed PMC pass
    # (profiles/valu_counts.json: SQ_INSTS_VALU per 1024 frames, collected with rocprofv3 --pmc on this workload) x the calibrated
    # issue cost (4.2 cycles per wave64 instruction per SIMD, profiles/r03_valu_issue_calibration.txt) / the launch time measured
    # HERE with one batch in flight.  A kernel near 1 is bound by instruction issue whatever its HBM fraction says.
    vc = None
    try:
        vc = json.loads((ROOT / "profiles" / "valu_counts.json").read_text())
    except Exception:
        vc = None
    if vc:
        cyc, simds, clk = float(vc["cycles_per_valu"]), float(vc["simds"]), float(vc["clock_GHz"]) * 1e9
        for k, d in roofline["per_kernel"].items():
            n = vc["valu_per_1024_frames"].get(k)
            if n:
                d["valu_issue_frac"] = round(n * (B / 1024.0) * cyc / (simds * clk) / (serial_stage_ms[k] * 1e-3), 4)
                d["judged_on"] = "valu_issue" if d["valu_issue_frac"] > d["alone_frac"] else "hbm"
        roofline["valu_issue_note"] = ("valu_issue_frac = SQ_INSTS_VALU (profiles/valu_counts.json, " + vc.get("source", "") + ") x " +
                                       f"{cyc} cycles / ({int(simds)} SIMDs x {vc['clock_GHz']} GHz) / alone_ms")
    stage_gbs = {k: round(sb[k] * B / (serial_stage_ms[k] * 1e-3) / 1e9, 1) for k in kernel_stages if serial_stage_ms[k] > 0}
    stage_gbs_region = {k: round(sb[k] * B / (stage_ms[k] * 1e-3) / 1e9, 1) for k in stage_ms
                        if k in sb and stage_ms[k] > 0.02}
    # pose-only stages are bound by FP64 instruction issue: report their flop rate (~270 flop per observation and iteration)
    po_iters = 20.0
    pose_flops = 270.0 * n_obs2 * po_iters
    ext_ms = stage_ms["extract"]
    match_ms = stage_ms["frame_post"] + stage_ms["match_last_frame"] + stage_ms["match_local_map"]
    em_bytes = sum(sb[k] for k in ext_keys) + sb["frame_post"] + sb["match_last_frame"] + sb["match_local_map"]
    em_gbs = em_bytes * B / ((ext_ms + match_ms) * 1e-3) / 1e9

    out = {
        "metric": "tracked frames/sec + local-BA LM-iters/sec (synthetic 640x480; value = tracked frames/sec: ORB extraction, "
                  "frame post-processing, two guided searches with the culling / isInFrame steps between them and two "
                  "pose-only solves per frame, one vo_tracker C call per batch)",
        "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "collective_backend": (args.backend if world > 1 else None),
        "rccl_note": ("this line's all-reduces went through RCCL (torch.distributed nccl backend)" if (world > 1 and args.backend == "nccl") else
                      "RCCL has not been exercised by this run: the sharded-BA all-reduce path is covered by gloo tests and by an RCCL "
                      "C++ host (examples/rccl_sharded_ba.cpp) that has run the loop's collectives on an MI355X as a single rank only "
                      "(tests/test_gpu_rccl.py); nothing has crossed xGMI and no SCALE record exists yet"),
        "config": {"workload": "tracked frame: ORB extract (640x480, 8-level pyramid, 1000 kpts) + undistort/depth/grid + "
                               "searchByProjection vs last frame + solvePoseOnlySE3 + cullingOutliersBeforeLocalMap + "
                               "isInFrame (refined pose) + searchByProjection vs local map + solvePoseOnlySE3; frames, depth "
                               "and map resident in HBM",
                   "frames_per_gpu_per_step": B, "keypoints_per_frame": float(counts.mean()),
                   "fast_candidates_per_frame": ncand, "last_frame_queries": round(n_q0, 1), "local_map_queries": round(n_q1, 1),
                   "matches_last_frame": round(n_match0, 1), "pose_observations": round(n_obs2, 1),
                   "pose_inliers": float(ninl.mean()), "parallelism": f"frames sharded x{world}, no collective",
                   "batches_in_flight": n_pipe},
        "one_batch_in_flight": {"ms_per_step": round(serial_ms, 4), "frames_per_s": round(B / serial_ms * 1e3, 1),
                                "stage_ms_per_launch": {k: round(v, 4) for k, v in serial_stage_ms.items()},
                                "note": "the same step on ONE stream, batches back to back, one synchronisation at the end (nothing overlapped); "
                                        "stage_ms_per_launch at the top level is measured inside the timed region, i.e. with the "
                                        "other batch's kernels running next to each launch when batches_in_flight > 1"},
        "roofline": roofline,
        "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
        "stage_algorithmic_GBps": stage_gbs_region,
        "kernel_algorithmic_GBps_one_batch_in_flight": stage_gbs,
        "extract_match_algorithmic_GBps_per_gpu": round(em_gbs, 1),
        "extract_match_frac_of_hbm_peak": round(em_gbs / HBM_PEAK_GBS, 4),
        "pose_only_in_path": {"ms_per_launch": round(stage_ms["pose_only_1"] + stage_ms["pose_only_2"], 4),
                              "fp64_GFLOPs": round(2 * pose_flops * B / ((stage_ms["pose_only_1"] + stage_ms["pose_only_2"]) * 1e-3) / 1e9, 1),
                              "bound": "fp64 VALU issue (one wavefront per frame; twice the frames take twice as long: ~5 cycles per FP64 wave-instruction per SIMD at one wavefront per SIMD, profiles/r03_valu_issue_calibration.txt)"},
    }

    # ---- one camera stream, host buffers: Frame construction to pose in ONE call (vo_tracker_track, batch 1: image and raw
    # depth uploaded, pose downloaded) -- the drop-in latency of visualOdometry.cpp:228-251 + 745-775 -- and the same call for
    # a whole batch of host images (PCIe ingest included; never `value`).
    if rank == 0 and not args.no_single_stream:
        t1 = vo.Tracker(1, cam5, synth.DIST, W, H, max_last=1100, max_local=2200, inv_depth_scale=inv_depth, single_stream=True)
        load_maps(t1, [maps[0]], 1100, 2200)
        h_img, h_dep = np.ascontiguousarray(uniq[:1]), np.ascontiguousarray(uniq_depth[:1]).view(np.uint16)
        lat = []
        for i in range(40):
            tq = time.perf_counter()
            t1.track(h_img, h_dep)
            r1 = t1.results()
            lat.append(time.perf_counter() - tq)
        assert r1["n_inliers"][0] >= 100
        t1.close()
        # the floor under a chain of dependent launches on this stack: 200 one-element kernels back to back on one stream
        # (27 launches per tracked frame, most of them at this floor)
        tiny = torch.zeros(64, device="cuda")
        for _ in range(20):
            tiny.add_(1.0)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(200):
            tiny.add_(1.0)
        torch.cuda.synchronize()
        floor_us = (time.perf_counter() - tq) / 200 * 1e6
        out["single_stream"] = {"ms_per_frame": round(float(np.median(lat[5:])) * 1e3, 4),
                                "frames_per_s": round(1.0 / float(np.median(lat[5:])), 1),
                                "launches_per_frame": 27, "dependent_dispatch_floor_us": round(floor_us, 2),
                                "note": "batch 1, host image + raw depth in, pose out, one C call + results (PCIe and "
                                        "launch latency of 27 kernels included; dependent_dispatch_floor_us = one trivial kernel "
                                        "behind another on one stream, host-paired, the least any of the 27 can cost)"}
        h_all = np.ascontiguousarray(frames_np)
        h_dall = np.ascontiguousarray(np.stack([uniq_depth[i % n_unique] for i in range(B)])).view(np.uint16)
        ing = []
        for i in range(4):
            tq = time.perf_counter()
            trk.track(h_all, h_dall)
            trk.results()
            ing.append(time.perf_counter() - tq)
        out["ingest_inclusive"] = {"ms_per_step": round(float(np.median(ing[1:])) * 1e3, 3),
                                   "frames_per_s": round(B / float(np.median(ing[1:])), 1),
                                   "GBps_host_to_device": round(B * W * H * 3 / float(np.median(ing[1:])) / 1e9, 1),
                                   "note": f"{B} host frames (8-bit image + 16-bit depth, pageable memory) uploaded inside "
                                           "the call, one batch in flight"}

    # ---- BASELINE config 1 as written: extract + brute-force 1000 x 1000 Hamming against the next frame (SURVEY 8d bytes)
    if not args.no_bruteforce:
        with torch.cuda.stream(stream):
            bdesc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda")
            dmat = torch.zeros((B, NM, NM), dtype=torch.int16, device="cuda")
        ham_ev = []

        def bf_step():
            with torch.cuda.stream(stream):
                ext.extract_batch_dev(frames, kps0, bdesc[:B], cnt0)
                bdesc[B].copy_(bdesc[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                vo.hamming_matrix_batch_dev(bdesc[:B, :NM], bdesc[1:, :NM], dmat, stream=stream.cuda_stream)
                e1.record(stream)
                ham_ev.append((e0, e1))

        bf_step()
        torch.cuda.synchronize()
        ham_ev.clear()
        ext.set_timing(True)
        tb0 = time.perf_counter()
        nbf = max(3, args.steps // 2)
        for _ in range(nbf):
            bf_step()
        torch.cuda.synchronize()
        tbf = time.perf_counter() - tb0
        bms, bn = ext.get_timing()
        ext.set_timing(False)
        bms = {k: v / max(bn, 1) for k, v in bms.items()}
        bms["hamming"] = float(np.mean([a.elapsed_time(b) for a, b in ham_ev]))
        # the same steps as the library runs them when nobody asks for stage times: the blur on the extractor's side stream next
        # to FAST / oct-tree (it reads the pyramid only) -- the stage times above come from the serial, instrumented order
        ham_ev.clear()
        bf_step()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(nbf):
            bf_step()
        torch.cuda.synchronize()
        tbf_prod = time.perf_counter() - tp0
        ham_ev.clear()
        sb8 = 5123128 + 2064000  # B_ext + B_match per frame
        out["extract_bruteforce_match"] = {
            "workload": "BASELINE configs[1]: extract + all-pairs 1000x1000 Hamming vs next frame (u16 matrix written)",
            "frames_per_s": round(B * nbf / tbf, 1), "ms_per_step": round(tbf / nbf * 1e3, 4),
            "stage_ms_per_launch": {k: round(v, 4) for k, v in bms.items() if k != "offsets"},
            "kernels": {"hamming": "k_hamming_mfma (int8 matrix-core dot products; vo_set_option(VO_OPT_HAMMING_KERNEL, 1) = the VALU form)",
                        "blur": "none: k_describe_win stages the 45 x 45 raw window of a key-point once and blurs it in place (int8 matrix cores; "
                                "VO_ORB_OPT_DESCRIBE_BLUR = 1: blurred planes by k_blur_mfma, VO_ORB_OPT_BLUR_KERNEL = 1: by k_blur_groups)"},
            "hamming_roofline": {"bound": "hbm (write stream)", "bytes_per_launch": 2064000 * B,
                                 "achieved": round(2064000 * B / (bms["hamming"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(2064000 * B / (bms["hamming"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "valu_issue_frac": (round(vc["valu_per_1024_frames"]["hamming"] * (B / 1024.0) * float(vc["cycles_per_valu"]) /
                                                           (float(vc["simds"]) * float(vc["clock_GHz"]) * 1e9) / (bms["hamming"] * 1e-3), 4)
                                                     if vc and vc["valu_per_1024_frames"].get("hamming") else None),
                                 "note": "a linear fill of the same 2.05 GB takes 0.36-0.38 ms on this chip (tools/microbench/write_pattern.hip)"},
            "survey_8d_bytes_per_frame": sb8,
            "end_to_end_algorithmic_GBps_per_gpu": round(sb8 * B * nbf / tbf / 1e9, 1),
            "frac_of_hbm_peak": round(sb8 * B * nbf / tbf / 1e9 / HBM_PEAK_GBS, 4),
            "uninstrumented": {"ms_per_step": round(tbf_prod / nbf * 1e3, 4), "frames_per_s": round(B * nbf / tbf_prod, 1),
                               "frac_of_hbm_peak": round(sb8 * B * nbf / tbf_prod / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "the same steps without per-stage events: the blur overlaps FAST / oct-tree on the extractor's side "
                                       "stream (the library's default order); frac_of_hbm_peak above is the serial, instrumented order"}}
        del dmat, bdesc

    # ------------------------------------------------------------------ BA (configs 2 and 3)
    if not args.no_ba:
        lb = synth.make_lba_problem(0)
        n_edges = len(lb["e_cam"])
        reps = 30   # >= 30 repetitions, median

        class _DevView:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        ar_stats = {"calls": 0, "max_doubles": 0}

        def _allreduce(ptr, n, _stream):
            # the exchange of the sharded LM loop (vo_ba_set_allreduce): RCCL over xGMI with the nccl backend
            ar_stats["calls"] += 1
            ar_stats["max_doubles"] = max(ar_stats["max_doubles"], int(n))
            # (armed: this is called from inside vo_ba_solve's LM loop -- a peer that never arrives must end this rank too)
            wd.arm(f"all-reduce of {int(n)} doubles (waiting)")
            t = torch.as_tensor(_DevView(ptr, n), device="cuda")
            if args.backend == "nccl":
                dist.all_reduce(t)
            else:
                hbuf = t.cpu()
                dist.all_reduce(hbuf)
                t.copy_(hbuf)
            wd.disarm(f"all-reduce of {int(n)} doubles")
            return 0

        def _same_on_all_ranks(v, what):
            chk = torch.tensor([v], dtype=torch.int64, device="cuda")
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert int(lo.item()) == int(hi.item()), f"ranks diverged in the sharded LM loop ({what})"

        shard_opts = {"collectives_at_one_rank": 1} if dist_one else None

        def _sharded_local_ba():
            # one problem, points sharded over the ranks; the LM loop runs inside libvo_hip.so and calls back for its
            # two all-reduces per iteration
            sba = vo.BundleAdjuster(lb, shard=rank, n_shards=world, stream=torch.cuda.current_stream().cuda_stream, options=shard_opts)
            sba.set_allreduce(_allreduce)
            sba.local_ba()
            t_, it_ = [], 0
            ar_stats["calls"] = 0
            for _ in range(reps):
                sba.set_state(lb["poses"], lb["points"])
                barrier()
                t0_ = time.perf_counter()
                _, (s1, s2), _ = sba.local_ba()
                barrier()
                t_.append(time.perf_counter() - t0_)
                it_ += s1.iterations + s2.iterations
            sba.close()
            # every rank took the same decisions: identical iteration counts are part of the contract
            _same_on_all_ranks(it_, "local BA")
            return t_, it_, ar_stats["calls"] / reps

        solve_s = []
        if world == 1:
            ba = vo.BundleAdjuster(lb)
            ba.local_ba()  # warm-up (allocations, code load)
            iters = 0
            for _ in range(reps):
                ba.set_state(lb["poses"], lb["points"])  # reset to the initial guess (not timed)
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                _, sums, _ = ba.local_ba()               # returns after its own final synchronisation
                solve_s.append(time.perf_counter() - tb0)
                iters += sums[0].iterations + sums[1].iterations
            ba.close()
        else:
            solve_s, iters, _ = _sharded_local_ba()
        tb = float(np.median(solve_s)) * reps   # median solve time x repetitions (every solve takes the same LM iterations)
        out["local_ba"] = {"workload": f"10 KF + 4 fixed x 3000 pts, {n_edges} edges, 5 Huber + 10 plain LM iterations",
                           "lm_iters_per_s": round(iters / tb, 1), "ms_per_solve": round(tb / reps * 1e3, 3),
                           "iterations_per_solve": iters / reps, "dtype": "f64",
                           "timing": f"median of {reps} solves (min {min(solve_s) * 1e3:.3f} ms, max {max(solve_s) * 1e3:.3f} ms)",
                           "sharding": f"points % {world}, 2 all-reduces per LM iteration inside the C-ABI (vo_ba_set_allreduce)"
                           if world > 1 else "single GPU"}
        f_lba = ba_flops_per_iteration(lb)
        out["local_ba"]["roofline"] = {"bound": "fp64 (latency-bound: 3 dependent launches per LM iteration)",
                                       "flops_per_iteration": round(f_lba), "achieved": round(f_lba * iters / tb / 1e12, 4),
                                       "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(f_lba * iters / tb / 1e12 / FP64_PEAK_TFLOPS, 5)}
        def _replicas(nconc=8):
            # independent 10-KF problems (several local-mapping sessions / map regions), one handle + stream each; no exchange
            handles = [vo.BundleAdjuster(lb) for _ in range(nconc)]
            for hd in handles:
                hd.local_ba()
            r_iters, r_t = 0, 0.0
            for _ in range(5):
                for hd in handles:
                    hd.set_state(lb["poses"], lb["points"])
                torch.cuda.synchronize()
                tr0 = time.perf_counter()
                for hd in handles:
                    hd.local_ba_enqueue()
                res = [hd.local_ba_finish() for hd in handles]
                r_t += time.perf_counter() - tr0
                r_iters += sum(sr[0].iterations + sr[1].iterations for _, sr in res)
            for hd in handles:
                hd.close()
            return r_iters, r_t

        if world > 1:
            # BA throughput over the node.  First rank 0 ALONE (the others wait at a barrier): the one-GPU reference of this very
            # run; then every rank at once.  `ba_scaling.replicas` = node aggregate / that reference -- the number SCALE asks for.
            if rank == 0:
                i1, t1 = _replicas()
                one_gpu = i1 / t1
            else:
                one_gpu = 0.0
            barrier()
            r_iters, r_t = _replicas()
            barrier()
            tot = torch.tensor([r_iters], dtype=torch.float64, device="cuda")
            tmax = torch.tensor([r_t], dtype=torch.float64, device="cuda")
            ref1 = torch.tensor([one_gpu], dtype=torch.float64, device="cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(ref1, op=dist.ReduceOp.MAX)
            agg = float(tot.item()) / float(tmax.item())
            out["local_ba"]["replicas"] = {"problems_per_gpu": 8, "aggregate_lm_iters_per_s": round(agg, 1),
                                           "one_gpu_alone_lm_iters_per_s": round(float(ref1.item()), 1),
                                           "sharding": "independent problems per GPU, no collective"}
            # the unsharded config-3 solve on rank 0 alone, for the sharded form's ratio
            if rank == 0:
                ub = vo.BundleAdjuster(lb)
                ub.local_ba()
                us = []
                for _ in range(10):
                    ub.set_state(lb["poses"], lb["points"])
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    ub.local_ba()
                    us.append(time.perf_counter() - t0_)
                ub.close()
                out["local_ba"]["one_gpu_unsharded_ms_per_solve"] = round(float(np.median(us)) * 1e3, 3)
            barrier()
        if dist_one:
            t_s, it_s, calls_s = _sharded_local_ba()
            out["local_ba"]["sharded_at_one_rank"] = {
                "ms_per_solve": round(float(np.median(t_s)) * 1e3, 3), "lm_iters_per_s": round(it_s / (float(np.median(t_s)) * reps), 1),
                "allreduce_calls_per_solve": calls_s, "backend": args.backend,
                "note": "ONE rank, process group of size 1: the sharded form of the loop with every collective a real all_reduce "
                        "(VO_BA_OPT_COLLECTIVES_AT_ONE_RANK); the price of the collectives in the dependent chain, not a scaling figure"}
        if world > 1 or dist_one:
            # config 4 sharded: ONE 500-key-frame global BA, points % world per rank, the packed reduced camera system
            # all-reduced once per LM iteration (+ 6 scalars) through the same callback; every rank factors the summed system
            gb = synth.make_global_ba_problem(0)
            hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))

            def _sharded_global_ba(segments):
                # segments: the per-rank segment factorisation (points owned by the rank of their nested-dissection segment,
                # own segments eliminated locally, separator block / extras / step all-reduced) instead of the replicated one
                gsh = vo.BundleAdjuster(gb, shard=rank, n_shards=world, stream=torch.cuda.current_stream().cuda_stream,
                                        options=dict(shard_opts or {}, segments=int(segments)))
                gsh.set_allreduce(_allreduce)
                gsh.solve(hm, hs, 1)                                   # builds the device structures
                ar_stats["calls"], ar_stats["max_doubles"] = 0, 0
                g_s, g_it = [], 0
                for _ in range(5):
                    gsh.set_state(gb["poses"], gb["points"])
                    barrier()
                    tg0 = time.perf_counter()
                    gs = gsh.solve(hm, hs, 10)
                    barrier()
                    g_s.append(time.perf_counter() - tg0)
                    g_it = gs.iterations
                g_order, c0 = gsh.debug_order(), gsh.segment_c0()
                gsh.close()
                _same_on_all_ranks(g_it, "global BA")
                return float(np.median(g_s)), g_it, g_order, c0, ar_stats["max_doubles"] * 8 / 1e6, ar_stats["calls"] / len(g_s)

            tg, g_it, g_order, _, g_mb, g_calls = _sharded_global_ba(False)
            gkey = "global_ba" if world > 1 else "global_ba_sharded_at_one_rank"
            out[gkey] = {"workload": f"{len(gb['poses'])} KF x {len(gb['points'])} pts, {len(gb['e_cam'])} edges, "
                                            f"{6 * (len(gb['poses']) - 1)}-wide reduced system, 10 LM iterations",
                                "lm_iters_per_s": round(g_it / tg, 1), "ms_per_iter": round(tg / max(g_it, 1) * 1e3, 3), "dtype": "f64",
                                "timing": "median of 5 solves",
                                "sharding": f"points % {world}; per LM iteration one all-reduce of the packed reduced system "
                                            f"({g_mb:.1f} MB) and one of 6 scalars; factorisation replicated",
                                "allreduce_payload_MB": round(g_mb, 2),
                                "allreduce_calls_per_solve": g_calls, "key_frame_order": g_order}
            ts, s_it, _, s_c0, s_mb, s_calls = _sharded_global_ba(True)
            out[gkey]["segment_factorisation"] = {
                "lm_iters_per_s": round(s_it / ts, 1), "ms_per_iter": round(ts / max(s_it, 1) * 1e3, 3), "first_separator_tile_column": s_c0,
                "sharding": "points by nested-dissection segment; per LM iteration the camera-block extras, the separator block after the "
                            f"segments' elimination ({s_mb:.1f} MB), the step and 6 scalars are all-reduced; separators factored on every rank",
                "allreduce_payload_MB": round(s_mb, 2), "allreduce_calls_per_solve": s_calls,
                "note": "opt-in (vo_ba_set_option(h, VO_BA_OPT_SEGMENTS, 1)): on one GPU with emulated ranks it computes more per rank than the replicated form "
                        "(profiles/r04_segment_factorisation.txt)"}
            if world > 1:
                # rank 0 alone: the unsharded 500-key-frame solve, the reference of the two sharded forms
                g1 = 0.0
                if rank == 0:
                    ug = vo.BundleAdjuster(gb)
                    ug.solve(hm, hs, 1)
                    ug_s = []
                    for _ in range(3):
                        ug.set_state(gb["poses"], gb["points"])
                        torch.cuda.synchronize()
                        t0_ = time.perf_counter()
                        gs1 = ug.solve(hm, hs, 10)
                        ug_s.append((time.perf_counter() - t0_) / max(gs1.iterations, 1))
                    ug.close()
                    g1 = float(np.median(ug_s)) * 1e3
                barrier()
                lb1 = out["local_ba"].get("one_gpu_unsharded_ms_per_solve", 0.0)
                rp = out["local_ba"]["replicas"]
                out["ba_scaling"] = {
                    "n_gpus": world,
                    "replicas": {"aggregate_lm_iters_per_s": rp["aggregate_lm_iters_per_s"], "one_gpu_lm_iters_per_s": rp["one_gpu_alone_lm_iters_per_s"],
                                 "x_one_gpu": round(rp["aggregate_lm_iters_per_s"] / max(rp["one_gpu_alone_lm_iters_per_s"], 1e-9), 3),
                                 "what": "independent 10-KF / 3000-point local BAs, 8 per GPU, no collective: BA THROUGHPUT of the node"},
                    "sharded_local_ba": {"ms_per_solve": out["local_ba"]["ms_per_solve"], "one_gpu_ms_per_solve": lb1,
                                         "x_one_gpu": round(lb1 / max(out["local_ba"]["ms_per_solve"], 1e-9), 3),
                                         "what": "ONE config-3 problem, points % N, 2 all-reduces per LM iteration (latency-bound: expected < 1)"},
                    "sharded_global_ba_replicated": {"ms_per_iter": out["global_ba"]["ms_per_iter"], "one_gpu_ms_per_iter": round(g1, 3),
                                                     "x_one_gpu": round(g1 / max(out["global_ba"]["ms_per_iter"], 1e-9), 3),
                                                     "what": "ONE config-4 problem, points % N, packed reduced system all-reduced, factorisation on every rank"},
                    "sharded_global_ba_segments": {"ms_per_iter": out["global_ba"]["segment_factorisation"]["ms_per_iter"], "one_gpu_ms_per_iter": round(g1, 3),
                                                   "x_one_gpu": round(g1 / max(out["global_ba"]["segment_factorisation"]["ms_per_iter"], 1e-9), 3),
                                                   "what": "ONE config-4 problem, per-rank segment factorisation, 4 collectives per LM iteration"},
                    "note": "x_one_gpu = speed relative to ONE GPU measured by rank 0 alone inside this run (the other ranks waiting at a barrier). "
                            "north_star's >= 6x BA scaling is answered by `replicas`; a single problem is bound by its chain of dependent tile columns / "
                            "launches and does not scale (DESIGN.md section 6)"}
        if world == 1:
            # What the CALLER of Optimizer::solveLocalBAPoseAndPoint pays per key-frame (localMapping.cpp:38 builds a new problem
            # every time): everything from the host arrays to the written-back state.  (a) create -> local_ba -> get_state ->
            # destroy, the shim's form until round 4; (b) ONE handle re-used through vo_ba_reset, the shim's form now.  The CPU
            # restatement's figure (orc.local_ba) includes its own set-up, so THIS is the like-for-like ratio.
            def _e2e_fresh():
                hd_ = vo.BundleAdjuster(lb)
                _, s_, _ = hd_.local_ba()
                hd_.state()
                hd_.close()
                return s_[0].iterations + s_[1].iterations

            keep = vo.BundleAdjuster(lb)
            keep.local_ba()
            # (b) as the C++ shim calls it: the three C entry points on arguments marshalled ONCE -- a ctypes call builds a dozen
            # pointer objects and numpy allocates the result arrays per call, ~0.03 ms that no C++ caller pays
            L_ = vo.lib()
            a_ = {k_: np.ascontiguousarray(v_) for k_, v_ in lb.items() if isinstance(v_, np.ndarray)}
            rargs_ = (keep._h, len(lb["poses"]), vo._p(a_["poses"]), vo._p(a_["fixed"]), len(lb["points"]), vo._p(a_["points"]),
                      len(lb["e_cam"]), vo._p(a_["e_cam"]), vo._p(a_["e_pt"]), vo._p(a_["e_obs"]), vo._p(a_["e_inv_sigma"]), vo._p(a_["cam"]))
            erase_ = np.zeros(len(lb["e_cam"]), np.uint8)
            sums_ = (vo.LmSummary * 2)()
            po_, px_ = np.zeros((len(lb["poses"]), 6)), np.zeros((len(lb["points"]), 3))
            largs_ = (keep._h, None, vo._p(erase_), ctypes.byref(sums_))
            gargs_ = (keep._h, vo._p(po_), vo._p(px_))

            def _e2e_reuse():
                if L_.vo_ba_reset(*rargs_) != 0 or L_.vo_ba_local_ba(*largs_) != 0 or L_.vo_ba_get_state(*gargs_) != 0:
                    raise RuntimeError(L_.vo_last_error())
                return sums_[0].iterations + sums_[1].iterations

            e2e = {}
            for name, fn in (("create_destroy_per_call", _e2e_fresh), ("one_handle_reset_per_call", _e2e_reuse)):
                fn(), fn()
                ts_, it_ = [], 0
                for _ in range(30):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    it_ = fn()
                    ts_.append(time.perf_counter() - t0_)
                e2e[name] = {"ms_per_call": round(float(np.median(ts_)) * 1e3, 3), "lm_iters_per_s": round(it_ / float(np.median(ts_)), 1),
                             "x_solve_only": round(float(np.median(ts_)) / (tb / reps), 3)}
            keep.close()
            e2e["note"] = ("from host arrays to the written-back state, median of 30; the shim of Optimizer::solveLocalBAPoseAndPoint keeps one "
                           "handle per thread and calls vo_ba_reset (no hipMalloc / hipFree / stream creation per key-frame); "
                           "one_handle_reset_per_call = vo_ba_reset + vo_ba_local_ba + vo_ba_get_state through the C-ABI on arguments "
                           "marshalled once (what the C++ shim does), create_destroy_per_call through the Python wrapper")
            assert np.isfinite(po_).all() and sums_[0].iterations > 0
            out["local_ba"]["end_to_end"] = e2e
            # aggregate throughput: independent problems (one handle + stream each) overlapped on the GPU
            nconc = 8
            handles = [vo.BundleAdjuster(lb) for _ in range(nconc)]
            for hd in handles:
                hd.local_ba()
            agg_iters, tagg = 0, 0.0
            for _ in range(5):
                for hd in handles:
                    hd.set_state(lb["poses"], lb["points"])
                torch.cuda.synchronize()
                ta0 = time.perf_counter()
                for hd in handles:
                    hd.local_ba_enqueue()
                res = [hd.local_ba_finish() for hd in handles]
                tagg += time.perf_counter() - ta0
                agg_iters += sum(s[0].iterations + s[1].iterations for _, s in res)
            for hd in handles:
                hd.close()
            out["local_ba"]["concurrent_problems"] = nconc
            out["local_ba"]["aggregate_lm_iters_per_s"] = round(agg_iters / tagg, 1)
            probs = [synth.make_pose_problem(i) for i in range(64)]
            probs = probs * 16
            vo.Optimizer.solvePoseOnlySE3(probs[:64])
            tp0 = time.perf_counter()
            _, _, ninl, psum = vo.Optimizer.solvePoseOnlySE3(probs, summaries=True)
            tp = time.perf_counter() - tp0
            pit = sum(psum[i].iterations for i in range(2 * len(probs)))
            out["pose_only_ba"] = {"workload": f"{len(probs)} frames x 1000 obs, 2 x <=10 LM iterations, one launch "
                                               "(host buffers in/out, PCIe included)",
                                   "solves_per_s": round(len(probs) / tp, 1), "lm_iters_per_s": round(pit / tp, 1)}
            # the same batch with everything resident in HBM (kernel time between two events)
            offs = np.arange(len(probs) + 1, dtype=np.int32) * 1000
            cat = lambda k: torch.from_numpy(np.ascontiguousarray(np.concatenate([pr[k] for pr in probs]))).cuda()
            d_off, d_pts, d_obs, d_isg = torch.from_numpy(offs).cuda(), cat("pts"), cat("obs"), cat("inv_sigma")
            d_cam = torch.from_numpy(np.ascontiguousarray(probs[0]["cam"], np.float64)).cuda()
            pose0 = torch.from_numpy(np.stack([pr["pose0"] for pr in probs])).cuda()
            d_pose, d_out = pose0.clone(), torch.zeros(len(probs) * 1000, dtype=torch.uint8, device="cuda")
            d_inl = torch.zeros(len(probs), dtype=torch.int32, device="cuda")
            cs = torch.cuda.current_stream()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for rep in range(3):
                d_pose.copy_(pose0)
                if rep == 2:
                    ev[0].record(cs)
                vo.check(vo.lib().vo_pose_only_solve_dev(len(probs), vo._p(d_off), 1000, vo._p(d_pts), vo._p(d_obs), vo._p(d_isg),
                                                         vo._p(d_cam), vo._p(d_pose), vo._p(d_out), vo._p(d_inl), None,
                                                         ctypes.c_void_p(cs.cuda_stream)), "vo_pose_only_solve_dev")
            ev[1].record(cs)
            torch.cuda.synchronize()
            tk = ev[0].elapsed_time(ev[1]) * 1e-3
            f_po = 270.0 * 1000 * pit  # ~150 + 120 flop per observation and LM iteration
            out["pose_only_ba"]["device_resident"] = {"solves_per_s": round(len(probs) / tk, 1), "lm_iters_per_s": round(pit / tk, 1),
                                                      "ms_per_launch": round(tk * 1e3, 3)}
            out["pose_only_ba"]["roofline"] = {"bound": "fp64 VALU issue (one wavefront per frame and SIMD, ~5 cycles per FP64 wave-instruction (profiles/r03_valu_issue_calibration.txt); most instructions are not multiply-adds)",
                                               "achieved": round(f_po / tk / 1e12, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": round(f_po / tk / 1e12 / FP64_PEAK_TFLOPS, 5)}
            pose_probs, pose_iters_dev = probs, pit
            # config 4: loop-closure sized problems (global BA through the large-system path, pose graph, Sim3)
            gb = synth.make_global_ba_problem(0)
            gba = vo.BundleAdjuster(gb)
            hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))
            gba.solve(hm, hs, 1)                                   # builds the device structures
            g_order = gba.debug_order()
            g_s = []
            for _ in range(5):
                gba.set_state(gb["poses"], gb["points"])
                torch.cuda.synchronize()
                tg0 = time.perf_counter()
                gs = gba.solve(hm, hs, 10)
                g_s.append(time.perf_counter() - tg0)
            tg = float(np.median(g_s))
            gba.close()
            out["global_ba"] = {"workload": f"{len(gb['poses'])} KF x {len(gb['points'])} pts, {len(gb['e_cam'])} edges, "
                                            f"{6 * (len(gb['poses']) - 1)}-wide reduced system, 10 LM iterations",
                                "lm_iters_per_s": round(gs.iterations / tg, 1), "ms_per_iter": round(tg / gs.iterations * 1e3, 3),
                                "dtype": "f64", "sharding": "single GPU", "timing": f"median of {len(g_s)} solves"}
            # SURVEY 8d's F_ba prices the reduced system's factorisation dense ((6 nc)^3 / 3: what Ceres' DENSE_SCHUR does);
            # the device factors it on the plan of its tile structure: flops EXECUTED = F_ba - dense term + tile products
            # (2 x 64^3 each) + one 64^3 solve per sub-diagonal tile + 64^3 / 3 per diagonal tile
            nc6 = 6 * int((np.asarray(gb["fixed"]) == 0).sum())
            f_gba_dense = ba_flops_per_iteration(gb)
            f_chol = g_order["tile_products"] * 2 * 64 ** 3 + (g_order["tiles"] - g_order["tile_rows"]) * 64 ** 3 + g_order["tile_rows"] * 64 ** 3 / 3
            f_gba = f_gba_dense - nc6 ** 3 / 3 + f_chol
            out["global_ba"]["key_frame_order"] = g_order
            out["global_ba"]["kernels"] = {"gather": "k_ba_pairs_lds (W / point blocks staged through LDS; vo_set_option(VO_OPT_BA_PAIRS_KERNEL, 1) = "
                                                     "k_ba_pairs, lane = couple)",
                                           "per_iteration_us": "profiles/r06_global_ba_kernels.txt: k_chol_tiles 504, k_ba_pairs_lds 219, k_ba_backsub 96, "
                                                               "k_chol_back 63, k_ba_cams_large 39, seven small kernels 70"}
            out["global_ba"]["roofline"] = {"bound": "fp64 mfma (tile Cholesky on a nested-dissection plan: the serial chain of dependent tile columns bounds it)",
                                            "flops_per_iteration": round(f_gba), "flops_per_iteration_dense_survey_8d": round(f_gba_dense),
                                            "achieved": round(f_gba * gs.iterations / tg / 1e12, 3),
                                            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(f_gba * gs.iterations / tg / 1e12 / FP64_PEAK_TFLOPS, 4)}
            pg = synth.make_pose_graph(7, n_kf=500, drift=0.004, extra_edges=4)
            vo.Optimizer.solvePoseGraphLoop(synth.make_pose_graph(0, n_kf=12))
            tq0 = time.perf_counter()
            _, _, ps = vo.Optimizer.solvePoseGraphLoop(pg)
            tq = time.perf_counter() - tq0
            out["pose_graph"] = {"workload": f"500 KF, {len(pg['e_i'])} Sim3 edges, scales fixed", "ms_per_solve": round(tq * 1e3, 2),
                                 "lm_iterations": ps.iterations, "ms_per_iter": round(tq / max(ps.iterations, 1) * 1e3, 2)}
            sp = [synth.make_sim3_problem(i, n=200, outliers=0.1) for i in range(256)]
            vo.Optimizer.solveLoopSim3(sp[:8])
            ts0 = time.perf_counter()
            vo.Optimizer.solveLoopSim3(sp)
            ts = time.perf_counter() - ts0
            out["sim3"] = {"workload": "256 loop candidates x 200 matches, one launch (host buffers in/out)",
                           "solves_per_s": round(len(sp) / ts, 1)}

    # ------------------------------------------------------------------ CPU baseline (oracle)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib as orc
        p = orc.orb_params()
        sfo = np.array(list(p.scale)[:8], np.float32)
        cpu_track_one(orc, p, sfo, uniq[0], uniq_depth[0], maps[0], cam5, synth.DIST, inv_depth)   # first touch
        tc0 = time.perf_counter()
        nfr, cpu_inl = 0, 0
        while time.perf_counter() - tc0 < args.cpu_seconds * 0.6 and nfr < 400:
            ni, _ = cpu_track_one(orc, p, sfo, uniq[nfr % n_unique], uniq_depth[nfr % n_unique], maps[nfr % n_unique], cam5,
                                  synth.DIST, inv_depth)
            cpu_inl += ni
            nfr += 1
        tc = time.perf_counter() - tc0
        cpu = {"value": round(nfr / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{nfr} tracked frames of the same synthetic workload (extract, undistort/depth/grid, two guided "
                         f"searches, two pose-only solves; {cpu_inl / max(nfr, 1):.0f} pose inliers per frame), "
                         f"{tc:.1f} s, oracle/ C restatement, gcc -O3 -ffp-contract=off, 1 thread"}
        if not args.no_ba:
            tl0 = time.perf_counter()
            its, nsol = 0, 0
            while time.perf_counter() - tl0 < args.cpu_seconds * 0.4 and nsol < 40:
                _, _, _, osums, _ = orc.local_ba(lb)
                its += osums[0].iterations + osums[1].iterations
                nsol += 1
            tl = time.perf_counter() - tl0
            cpu["local_ba_lm_iters_per_s"] = round(its / tl, 2)
            cpu["local_ba_sample"] = f"{nsol} solves of the same 10-KF/3000-pt problem, {tl:.1f} s, 1 thread"
            out["local_ba"]["speedup_vs_cpu_port"] = round(out["local_ba"]["lm_iters_per_s"] / (its / tl), 1)
            if "end_to_end" in out["local_ba"]:  # the like-for-like ratio: both sides from host arrays to the final state
                for k_ in ("create_destroy_per_call", "one_handle_reset_per_call"):
                    out["local_ba"]["end_to_end"][k_]["speedup_vs_cpu_port"] = round(
                        out["local_ba"]["end_to_end"][k_]["lm_iters_per_s"] / (its / tl), 1)
            if "aggregate_lm_iters_per_s" in out["local_ba"]:
                out["local_ba"]["aggregate_speedup_vs_cpu_port"] = round(
                    out["local_ba"]["aggregate_lm_iters_per_s"] / (its / tl), 1)
            if "pose_only_ba" in out:
                tq0 = time.perf_counter()
                nps, pits = 0, 0
                while time.perf_counter() - tq0 < 2.0 and nps < 64:
                    _, _, _, ps_, _ = orc.pose_only(pose_probs[nps])
                    pits += ps_[0].iterations + ps_[1].iterations
                    nps += 1
                tq = time.perf_counter() - tq0
                cpu["pose_only_solves_per_s"] = round(nps / tq, 1)
                cpu["pose_only_lm_iters_per_s"] = round(pits / tq, 1)
                cpu["pose_only_sample"] = f"{nps} of the same 1000-observation problems, {tq:.1f} s, 1 thread"
                out["pose_only_ba"]["speedup_vs_cpu_port"] = round(out["pose_only_ba"]["device_resident"]["solves_per_s"] / (nps / tq), 1)
            if "global_ba" in out:
                gp, gpt = gb["poses"].copy(), gb["points"].copy()
                gsum = orc.make_summary(8)
                tgc0 = time.perf_counter()
                n_git = 4  # (one call: the linearisation of the start point is paid once, as in the device's 10-iteration solves)
                orc.lib().orc_ba_lm(len(gp), gp, gb["fixed"], len(gpt), gpt, len(gb["e_cam"]), gb["e_cam"], gb["e_pt"],
                                    gb["e_obs"], gb["e_inv_sigma"], None, gb["cam"], hm, hs, n_git,
                                    ctypes.cast(ctypes.pointer(gsum), ctypes.c_void_p))
                tgc = (time.perf_counter() - tgc0) / max(int(gsum.iterations), 1)
                cpu["global_ba_lm_iters_per_s"] = round(1.0 / tgc, 3)
                cpu["global_ba_sample"] = (f"{int(gsum.iterations)} LM iterations of the same 500-KF problem, {tgc * int(gsum.iterations):.1f} s, "
                                           "1 thread")
                out["global_ba"]["speedup_vs_cpu_port"] = round(out["global_ba"]["lm_iters_per_s"] * tgc, 1)
        # the same port on all host cores at once, one frame stream per core (so the 1-core figure is no strawman)
        try:
            import multiprocessing as mp
            ncore = len(os.sched_getaffinity(0))
            try:  # a container's CPU quota (cgroup v2), e.g. "1600000 100000" = 16 cores
                q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    ncore = min(ncore, max(1, int(q) // int(per_us)))
            except Exception:
                pass
            ncore = max(1, min(ncore, 128))
            per = 12
            with mp.get_context("spawn").Pool(ncore) as pool:
                ta0 = time.perf_counter()
                times = pool.map(_cpu_worker, [(i, per) for i in range(ncore)])
                ta = time.perf_counter() - ta0
            cpu["all_cores"] = {"value": round(ncore * per / max(times), 1), "unit": "frames/s", "cores": ncore,
                                "sample": f"{per} frames on each of {ncore} processes, slowest {max(times):.1f} s "
                                          f"(pool wall time {ta:.1f} s incl. start-up)"}
        except Exception as exc:  # the baseline must never break the benchmark line
            cpu["all_cores"] = {"error": repr(exc)}
        cpu["host"] = {"cpu_count": os.cpu_count()}
        out["cpu_baseline"] = cpu
    for t in trks:
        t.close()
    for e in exts:
        e.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
