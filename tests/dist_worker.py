"""Worker of tests/test_gpu_dist.py: one rank of a 2-process sharded bundle adjustment through the C-ABI.

Launched by torch.distributed.run (gloo on a one-GPU box: both ranks share GPU 0; nccl = RCCL when every rank has
its own GPU).  The LM loop runs INSIDE libvo_hip.so (vo_ba_local_ba / vo_ba_solve on a sharded handle); this
process only supplies the all-reduce callback (vo_ba_set_allreduce), exactly what a C++ host would do with
ncclAllReduce.  Each rank checks the sharded result against the unsharded device solve of the same problem."""
import argparse
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class _DevView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def main():
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--case", default="lba", choices=["lba", "gba500"])
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from vo_slam_test_amd import _lib as vo
    from vo_slam_test_amd import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(rank % max(ndev, 1))
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    calls = {"n": 0, "max_doubles": 0}

    def allreduce(ptr, n, stream):
        calls["n"] += 1
        calls["max_doubles"] = max(calls["max_doubles"], int(n))
        t = torch.as_tensor(_DevView(ptr, n), device="cuda")
        if a.backend == "gloo":
            h = t.cpu()  # synchronises the (current = handle) stream
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)
        return 0

    st = torch.cuda.current_stream().cuda_stream
    ok = True
    if a.case == "gba500":
        # BASELINE config 4's shape: 500 key-frames -> a 3000 x 3000 reduced camera system, factored on the nested-
        # dissection plan every shard derives from the whole covisibility graph (the summed matrix must mean the same
        # rows on every rank); the payload of the collective is the Cholesky storage itself
        prob = synth.make_global_ba_problem(0, n_kf=500, n_pts=12000)
        hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))
        ref = vo.BundleAdjuster(prob, stream=st)
        s0 = ref.solve(hm, hs, 3)
        p0, x0 = ref.state()
        o0 = ref.debug_order()
        ref.close()
        good_all = True
        # (a) per-rank segment factorisation: points owned by the rank of their nested-dissection segment; per LM iteration
        #     the camera-block extras, the separator block after the segments' elimination, the step and 6 scalars are summed;
        # (b) the default: points p % world, the packed reduced system summed, every rank factors all of it
        for mode in ("segments", "replicated"):
            calls["n"], calls["max_doubles"] = 0, 0
            sh = vo.BundleAdjuster(prob, shard=rank, n_shards=world, stream=st, options={"segments": int(mode == "segments")})
            sh.set_allreduce(allreduce)
            s1 = sh.solve(hm, hs, 3)
            p1, x1 = sh.state()
            o1 = sh.debug_order()
            c0 = sh.segment_c0()
            sh.close()
            dp, dx = np.abs(p0 - p1).max(), np.abs(x0 - x1).max()
            # the collective carries the tiles of the matrix that exist + rhs + extras, not the 3008 x 3072 Cholesky storage (74 MB)
            payload_mb = calls["max_doubles"] * 8 / 1e6
            per_it = 4 if mode == "segments" else 2
            good = ((s0.iterations, s0.accepted, s0.termination) == (s1.iterations, s1.accepted, s1.termination) and o0 == o1
                    and o0["parts"] > 1 and abs(s0.final_cost - s1.final_cost) <= 1e-9 * s0.final_cost and dp < 1e-8 and dx < 1e-6
                    and payload_mb < 15.0 and (c0 > 0) == (mode == "segments") and calls["n"] >= per_it * s1.iterations)
            print(f"rank {rank} gba500 {mode}: order {o1} first separator tile column {c0} iterations {s1.iterations} accepted {s1.accepted} "
                  f"collectives {calls['n']} (largest payload {payload_mb:.1f} MB) "
                  f"cost {s1.initial_cost:.6g} -> {s1.final_cost:.6g} dpose {dp:.2e} dpoint {dx:.2e} -> {'OK' if good else 'MISMATCH'}", flush=True)
            good_all = good_all and good
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if good_all else 1)
    for name, prob, tol in (("lds-path", synth.make_lba_problem(3, n_kf=6, n_pts=800, n_fixed=2), 1e-9),
                            ("large-path", synth.make_lba_problem(4, n_kf=26, n_pts=1500, n_fixed=2), 1e-8)):
        ref = vo.BundleAdjuster(prob, stream=st)
        e0, s0, rc0 = ref.local_ba()
        p0, x0 = ref.state()
        ref.close()
        sh = vo.BundleAdjuster(prob, shard=rank, n_shards=world, stream=st)
        # without the callback the single-call entry points must refuse a sharded handle
        scratch = np.zeros(sh.n_edges, np.uint8)  # (kept alive across the call: _p only takes its address)
        rc_bad = vo.lib().vo_ba_local_ba(sh._h, None, vo._p(scratch), None)
        assert rc_bad == -1, rc_bad
        sh.set_allreduce(allreduce)
        calls["n"] = 0
        e1, s1, rc1 = sh.local_ba()
        p1, x1 = sh.state()
        sh.close()
        its = [(s.iterations, s.accepted, s.termination) for s in s0], [(s.iterations, s.accepted, s.termination) for s in s1]
        dp, dx = np.abs(p0 - p1).max(), np.abs(x0 - x1).max()
        good = rc0 == rc1 == 0 and its[0] == its[1] and np.array_equal(e0, e1) and dp < tol and dx < 100 * tol
        n_it = sum(s.iterations for s in s1)
        print(f"rank {rank} {name}: iterations {its[1]} collectives {calls['n']} (2 per LM iteration + merges) "
              f"dpose {dp:.2e} dpoint {dx:.2e} erase {int(e1.sum())} -> {'OK' if good else 'MISMATCH'}", flush=True)
        ok = ok and good and calls["n"] >= 2 * n_it
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
