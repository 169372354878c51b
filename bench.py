#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on synthetic inputs.

One "step" = one pass of the ORB front end over a batch of B synthetic 640x480 frames that is
already resident in HBM: 8-level pyramid, per-cell FAST + NMS, oct-tree, orientation, blur,
steered BRIEF (vo_orb_extract_batch_dev) followed by the all-pairs 1000x1000 Hamming matrix of
every frame against its successor (vo_hamming_matrix_batch_dev) -- BASELINE.json configs[1].
`value` = frames/s over all ranks (weak scaling: every rank owns its own batch; no collective on
this path).  After the timed region the same process measures local BA (configs[3], sharded over
the ranks with two all-reduces per LM iteration) and batched pose-only BA (configs[2]) and
reports them as extra keys of the same JSON line.

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


def _cpu_worker(arg):
    """one host core: the oracle's extract + 1000x1000 Hamming on `n` synthetic frames (spawned process)"""
    seed, n = arg
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    from vo_slam_test_amd import synth
    p = orc.orb_params()
    frames = synth.make_frames(2, start=seed * 7)
    orc.extract(p, frames[0])                      # first touch: library load, page-in
    t0 = time.perf_counter()
    prev = None
    for i in range(n):
        k, d, _ = orc.extract(p, frames[i & 1])
        orc.hamming_matrix((prev if prev is not None else d)[:1000], d[:1000])
        prev = d
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step (resident in HBM; throughput saturates near 1024: "
                                                             "188 k frames/s at 64, 246 k at 256, 268 k at 512, 275 k at 1024, 278 k at 2048)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks "
                    "share one GPU to exercise the N > 1 code path on a single-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_dev = local_rank % max(1, torch.cuda.device_count()) if args.backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(args.backend)

    from vo_slam_test_amd import _lib as vo
    from vo_slam_test_amd import synth

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    W, H, B = 640, 480, args.batch
    n_unique = min(B, 32)
    uniq = synth.make_frames(n_unique, start=rank * 1000)
    frames_np = np.stack([uniq[i % n_unique] for i in range(B)])
    stream = torch.cuda.Stream()
    ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
    ext.set_stream(stream.cuda_stream)
    cap = ext.max_keypoints()
    NM = 1000
    with torch.cuda.stream(stream):
        frames = torch.from_numpy(frames_np).cuda()
        kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
        desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        dmat = torch.zeros((B, NM, NM), dtype=torch.int16, device="cuda")
    ham_ev = []

    def step(timed=False):
        with torch.cuda.stream(stream):
            ext.extract_batch_dev(frames, kps, desc[:B], cnt)
            desc[B].copy_(desc[0])
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            vo.hamming_matrix_batch_dev(desc[:B, :NM], desc[1:, :NM], dmat, stream=stream.cuda_stream)
            if timed:
                e1.record(stream)
                ham_ev.append((e0, e1))

    for _ in range(args.warmup):
        step()
    barrier()
    ext.sync()
    counts = cnt.cpu().numpy()
    assert counts.min() >= NM, f"synthetic frames must yield >= {NM} key-points, got {counts.min()}"
    ncand = sum(len(ext.get_candidates(0, l)[0]) for l in range(8))
    ext.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    stage_ms, ncalls = ext.get_timing()
    ext.set_timing(False)
    stage_ms = {k: v / max(ncalls, 1) for k, v in stage_ms.items()}
    stage_ms["hamming"] = float(np.mean([a.elapsed_time(b) for a, b in ham_ev]))
    frames_per_s = world * B * args.steps / elapsed

    sb = stage_bytes_per_frame(W, H, int(counts.mean()), ncand)
    dom = max((k for k in stage_ms if k != "offsets"), key=lambda k: stage_ms[k])
    achieved = sb[dom] * B / (stage_ms[dom] * 1e-3) / 1e9
    traffic = None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            traffic = json.loads(tf.read_text()).get(dom)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "bytes_per_launch": sb[dom] * B, "avg_launch_ms": round(stage_ms[dom], 4)}
    stage_gbs = {k: round(sb[k] * B / (stage_ms[k] * 1e-3) / 1e9, 1) for k in stage_ms if stage_ms[k] > 0}
    total_alg = sum(sb[k] for k in sb if k != "offsets")
    e2e_gbs = total_alg * frames_per_s / world / 1e9

    out = {
        "metric": "tracked frames/sec + local-BA LM-iters/sec (synthetic 640x480; value = ORB extract+match frames/sec)",
        "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "ORB extract+match: 640x480, 8-level pyramid, 1000 kpts/frame, all-pairs "
                               "1000x1000 Hamming vs next frame; frames resident in HBM",
                   "frames_per_gpu_per_step": B, "keypoints_per_frame": float(counts.mean()),
                   "fast_candidates_per_frame": ncand, "parallelism": f"frames sharded x{world}, no collective"},
        "roofline": roofline,
        "stage_ms_per_launch": {k: round(v, 4) for k, v in stage_ms.items()},
        "stage_algorithmic_GBps": stage_gbs,
        "end_to_end_algorithmic_GBps_per_gpu": round(e2e_gbs, 1),
    }

    # ------------------------------------------------------------------ BA (configs 2 and 3)
    if not args.no_ba:
        lb = synth.make_lba_problem(0)
        n_edges = len(lb["e_cam"])
        reps = 20
        if world == 1:
            ba = vo.BundleAdjuster(lb)
            ba.local_ba()  # warm-up (allocations, code load)
            iters, tb = 0, 0.0
            for _ in range(reps):
                ba.set_state(lb["poses"], lb["points"])  # reset to the initial guess (not timed)
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                _, sums, _ = ba.local_ba()               # returns after its own final synchronisation
                tb += time.perf_counter() - tb0
                iters += sums[0].iterations + sums[1].iterations
            ba.close()
        else:
            from vo_slam_test_amd.dist_ba import ShardedBundleAdjuster
            sba = ShardedBundleAdjuster(lb, rank, world)
            sba.local_ba()
            iters, tb = 0, 0.0
            for _ in range(reps):
                sba.ba.set_state(lb["poses"], lb["points"])
                barrier()
                tb0 = time.perf_counter()
                _, _, _, (s1, s2) = sba.local_ba()
                barrier()
                tb += time.perf_counter() - tb0
                iters += s1.iterations + s2.iterations
            sba.close()
            # every rank took the same decisions: identical iteration counts are part of the contract
            chk = torch.tensor([iters], dtype=torch.int64, device="cuda")
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert int(lo.item()) == int(hi.item()), "ranks diverged in the sharded LM loop"
        out["local_ba"] = {"workload": f"10 KF + 4 fixed x 3000 pts, {n_edges} edges, 5 Huber + 10 plain LM iterations",
                           "lm_iters_per_s": round(iters / tb, 1), "ms_per_solve": round(tb / reps * 1e3, 3),
                           "iterations_per_solve": iters / reps, "dtype": "f64",
                           "sharding": f"points % {world}, 2 all-reduces per LM iteration" if world > 1 else "single GPU"}
        if world > 1:
            # replicas: every GPU works on its own set of independent 10-KF problems (several local-mapping
            # sessions / map regions); no exchange.  Aggregate LM-iterations/s over the node.
            nconc = 8
            handles = [vo.BundleAdjuster(lb) for _ in range(nconc)]
            for hd in handles:
                hd.local_ba()
            r_iters, r_t = 0, 0.0
            for _ in range(5):
                for hd in handles:
                    hd.set_state(lb["poses"], lb["points"])
                barrier()
                tr0 = time.perf_counter()
                for hd in handles:
                    hd.local_ba_enqueue()
                res = [hd.local_ba_finish() for hd in handles]
                barrier()
                r_t += time.perf_counter() - tr0
                r_iters += sum(sr[0].iterations + sr[1].iterations for _, sr in res)
            for hd in handles:
                hd.close()
            tot = torch.tensor([r_iters], dtype=torch.float64, device="cuda")
            tmax = torch.tensor([r_t], dtype=torch.float64, device="cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            out["local_ba"]["replicas"] = {"problems_per_gpu": nconc, "aggregate_lm_iters_per_s": round(float(tot.item()) / float(tmax.item()), 1),
                                           "sharding": "independent problems per GPU, no collective"}
        if world == 1:
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
            # config 4: loop-closure sized problems (global BA through the large-system path, pose graph, Sim3)
            gb = synth.make_global_ba_problem(0)
            gba = vo.BundleAdjuster(gb)
            hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))
            gba.solve(hm, hs, 1)                                   # builds the device structures
            gba.set_state(gb["poses"], gb["points"])
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            gs = gba.solve(hm, hs, 10)
            tg = time.perf_counter() - tg0
            gba.close()
            out["global_ba"] = {"workload": f"{len(gb['poses'])} KF x {len(gb['points'])} pts, {len(gb['e_cam'])} edges, "
                                            f"{6 * (len(gb['poses']) - 1)}-wide reduced system, 10 LM iterations",
                                "lm_iters_per_s": round(gs.iterations / tg, 1), "ms_per_iter": round(tg / gs.iterations * 1e3, 3),
                                "dtype": "f64", "sharding": "single GPU"}
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
        tc0 = time.perf_counter()
        nfr = 0
        prev = None
        while time.perf_counter() - tc0 < args.cpu_seconds * 0.6 and nfr < 400:
            k, d, _ = orc.extract(p, uniq[nfr % n_unique])
            if prev is not None:
                orc.hamming_matrix(prev[:NM], d[:NM])
            else:
                orc.hamming_matrix(d[:NM], d[:NM])
            prev = d
            nfr += 1
        tc = time.perf_counter() - tc0
        cpu = {"value": round(nfr / tc, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{nfr} frames of the same synthetic workload (extract + 1000x1000 Hamming), "
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
            if "aggregate_lm_iters_per_s" in out["local_ba"]:
                out["local_ba"]["aggregate_speedup_vs_cpu_port"] = round(
                    out["local_ba"]["aggregate_lm_iters_per_s"] / (its / tl), 1)
            if "global_ba" in out:
                gp, gpt = gb["poses"].copy(), gb["points"].copy()
                gsum = orc.make_summary(1)
                tgc0 = time.perf_counter()
                orc.lib().orc_ba_lm(len(gp), gp, gb["fixed"], len(gpt), gpt, len(gb["e_cam"]), gb["e_cam"], gb["e_pt"],
                                    gb["e_obs"], gb["e_inv_sigma"], None, gb["cam"], hm, hs, 1,
                                    ctypes.cast(ctypes.pointer(gsum), ctypes.c_void_p))
                tgc = time.perf_counter() - tgc0
                cpu["global_ba_lm_iters_per_s"] = round(1.0 / tgc, 3)
                cpu["global_ba_sample"] = f"1 LM iteration of the same 500-KF problem, {tgc:.1f} s, 1 thread"
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
    ext.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
