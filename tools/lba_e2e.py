"""Developer tool: what a local-BA call costs the CALLER (localMapping.cpp:38 builds a new problem per key-frame): handle
create + solve + state download + destroy from host arrays, against the solve alone on a pre-created handle; and, where the
library has it, the same through ONE re-used handle (vo_ba_reset)."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

lb = synth.make_lba_problem(0)
def med(f, n=30):
    f(); f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t)), float(np.min(t))
ba = vo.BundleAdjuster(lb)
ba.local_ba()
def solve_only():
    ba.set_state(lb["poses"], lb["points"]); 
t_set = med(lambda: ba.set_state(lb["poses"], lb["points"]))
def so():
    ba.set_state(lb["poses"], lb["points"]); ba.local_ba()
t_so = med(so)
def e2e():
    h = vo.BundleAdjuster(lb); h.local_ba(); h.state(); h.close()
t_e2e = med(e2e)
def create_only():
    h = vo.BundleAdjuster(lb); h.close()
t_c = med(create_only)
print(f"set_state {t_set[0]:.3f} ms; set_state + solve {t_so[0]:.3f} ms (solve alone ~{t_so[0]-t_set[0]:.3f}); create+destroy (no device build) {t_c[0]:.3f} ms")
print(f"end to end create -> local_ba -> get_state -> destroy: median {t_e2e[0]:.3f} ms, min {t_e2e[1]:.3f} ms")
if hasattr(ba, "reset"):
    def reuse():
        ba.reset(lb); ba.local_ba(); ba.state()
    t_r = med(reuse)
    print(f"one handle re-used (vo_ba_reset -> local_ba -> get_state): median {t_r[0]:.3f} ms, min {t_r[1]:.3f} ms")
ba.close()
# breakdown of the re-used-handle call sequence
ba = vo.BundleAdjuster(lb); ba.local_ba()
tt = np.zeros(3)
for _ in range(30):
    t0 = time.perf_counter(); ba.reset(lb); t1 = time.perf_counter(); ba.local_ba(); t2 = time.perf_counter(); ba.state(); t3 = time.perf_counter()
    tt += [t1 - t0, t2 - t1, t3 - t2]
print("breakdown (ms): reset %.3f  local_ba (device build + solve + results) %.3f  get_state %.3f" % tuple(tt / 30 * 1e3))
ba.close()
