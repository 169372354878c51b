"""Developer tool: repeat the config-4 global BA (2 LM iterations, as tests/golden g9) and report every run whose
iteration / acceptance counts differ -- looks for rare failures of the persistent Cholesky kernel.  usage: gba_soak.py [runs]"""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa
from vo_slam_test_amd import _lib, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pr = synth.make_global_ba_problem(0)
hm, hs = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))
ba = _lib.BundleAdjuster(pr)
ref = None
bad = 0
for r in range(runs):
    ba.set_state(pr["poses"], pr["points"])
    t0 = time.perf_counter()
    s = ba.solve(hm, hs, 2)
    dt = time.perf_counter() - t0
    key = (s.iterations, s.accepted, s.termination, round(s.final_cost, 6))
    if ref is None:
        ref = key
    if key != ref:
        bad += 1
        print("run", r, "differs:", key, "expected", ref, "last error:", _lib.lib().vo_last_error(), f"{dt*1e3:.1f} ms", flush=True)
print(f"{runs} runs, {bad} differing; reference {ref}")
ba.close()
