"""Developer tool: the per-rank segment factorisation of BASELINE config 4 (500 key-frames x 50k points), its ranks emulated
as threads of one process on ONE GPU (tests/thread_ranks.py, serial mode: the ranks take turns between collectives, so each
stretch of a rank's GPU work runs alone and is timed on the host, launch overheads included).
  per LM iteration and rank:  A linearisation -> [extras]  B assemble + own segments (phase 1) -> [separator block]
                              C separators + back-substitution (phases 2, 3) -> [step]  D step, points, cost -> [6 scalars]
Nothing here is a multi-GPU measurement: it says what each rank computes between the collectives; the collectives are
priced from their payloads in DESIGN.md section 6.
usage: tools/gba_seg_run.py [world ...]   (default 2 4 8)"""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import thread_ranks  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
ITS = 10
pr = synth.make_global_ba_problem(0, n_kf=500, n_pts=50000)
hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))
ba = vo.BundleAdjuster(pr)
ba.solve(hm, hs, 1)
times = []
for _ in range(5):
    ba.set_state(pr["poses"], pr["points"])
    t0 = time.perf_counter()
    s0 = ba.solve(hm, hs, ITS)
    times.append((time.perf_counter() - t0) * 1e3 / s0.iterations)
print(f"one GPU, unsharded: {np.median(times):.3f} ms per LM iteration ({s0.iterations} iterations, cost {s0.initial_cost:.6g} -> {s0.final_cost:.6g}); order {ba.debug_order()}")
ba.close()
out = {"unsharded_ms_per_iter": float(np.median(times)), "worlds": {}}
import os  # noqa: E402
for world in worlds:
    for mode in ("segments", "replicated"):
        ncol = 4 if mode == "segments" else 2

        def solve(h, rank):
            h.solve(hm, hs, 1)  # builds the device structures
            h.set_state(pr["poses"], pr["points"])
            s = h.solve(hm, hs, ITS)
            return s.iterations, s.final_cost, h.segment_c0()

        res, stats = thread_ranks.run_ranks(vo, pr, world, solve, serial=True)
        assert all(r[0] == s0.iterations and abs(r[1] - s0.final_cost) <= 1e-9 * s0.final_cost for r in res), res
        sizes = stats["sizes"]
        per = {}
        for rank in range(world):
            st = np.array(stats["stretch_ms"][rank])
            # the timed solve's stretches: the last ncol * ITS collectives (state() is not called here)
            body = st[-ncol * ITS:].reshape(ITS, ncol)[1:]  # (the first stretch of the solve contains the set-up kernels)
            per[rank] = np.median(body, axis=0)
        tab = np.array([per[r] for r in range(world)])
        pay = np.array(sizes[-ncol * ITS:]).reshape(ITS, ncol)[0] * 8 / 1e6
        crit = tab.max(axis=0)
        if mode == "segments":
            print(f"world {world} segments: first separator tile column {res[0][2]}; payloads MB extras {pay[0]:.3f} separator {pay[1]:.2f} step {pay[2]:.3f} scalars {pay[3]:.6f}")
            print("   rank   A linearise   B assemble+segments   C separators+backsub   D step+points   sum (ms)")
            for r in range(world):
                print(f"   {r:4d}   {tab[r, 0]:9.3f}   {tab[r, 1]:17.3f}   {tab[r, 2]:18.3f}   {tab[r, 3]:11.3f}   {tab[r].sum():7.3f}")
        else:
            print(f"world {world} replicated: payloads MB packed system {pay[0]:.2f} scalars {pay[1]:.6f}")
            print("   rank   A linearise+pack   B unpack+assemble+factor+step+points   sum (ms)")
            for r in range(world):
                print(f"   {r:4d}   {tab[r, 0]:14.3f}   {tab[r, 1]:30.3f}   {tab[r].sum():13.3f}")
        print(f"   slowest rank per stretch: {crit.round(3).tolist()} -> {crit.sum():.3f} ms of compute per LM iteration + {ncol} collectives")
        out["worlds"][f"{world}_{mode}"] = {"stretch_ms_by_rank": tab.round(4).tolist(), "critical_ms": crit.round(4).tolist(), "payload_MB": pay.round(4).tolist()}
print(json.dumps(out))
