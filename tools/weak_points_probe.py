"""Developer tool (GPU box): local BA, device vs oracle, deviation of the written-back map points by the number of key-frames
that see them (every local point is written back, optimizer_ceres.cpp:793-803; which bound holds for the weakly
observed ones?)."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import torch  # noqa: F401,E402
import oracle_lib as orc  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

worst = {}
probs = [synth.make_lba_problem(s) for s in (0, 1, 2, 3)]
rng = np.random.default_rng(11)
for k in range(20):
    probs.append(synth.make_lba_problem(100 + k, n_kf=int(rng.integers(3, 9)), n_pts=int(rng.integers(80, 500)), n_fixed=int(rng.integers(0, 3))))
for pr in probs:
    oposes, opts, oerase, osums, rc = orc.local_ba(pr)
    ba = vo.BundleAdjuster(pr)
    erase, sums, rc2 = ba.local_ba()
    gposes, gpts = ba.state()
    ba.close()
    deg = np.bincount(pr["e_pt"], minlength=len(opts))
    # observations that survive the erase mask
    live = np.bincount(pr["e_pt"][oerase == 0], minlength=len(opts))
    d = np.abs(gpts - opts).max(axis=1)
    scale = np.maximum(np.linalg.norm(opts, axis=1), 1.0)
    moved = np.abs(opts - pr["points"]).max(axis=1)
    for g in range(0, 8):
        m = deg == g if g < 7 else deg >= 7
        if m.any():
            w = worst.setdefault(g, [0, 0.0, 0.0, 0.0])
            w[0] += int(m.sum()); w[1] = max(w[1], float(d[m].max())); w[2] = max(w[2], float((d[m] / scale[m]).max()))
            w[3] = max(w[3], float(moved[m].max()))
for g in sorted(worst):
    n, a, r, mv = worst[g]
    print(f"degree {g if g < 7 else '>=7'}: {n} points, max |device - oracle| {a:.3e} (relative to max(|X|, 1): {r:.3e}); the solve moved them by up to {mv:.3e}")
