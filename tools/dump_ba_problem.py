"""Write a synthetic BA problem in the layout examples/rccl_sharded_ba.cpp reads.
usage: python tools/dump_ba_problem.py out.bin [local|global] [seed]"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
from vo_slam_test_amd import synth
out, kind, seed = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "local"), int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = synth.make_lba_problem(seed) if kind == "local" else synth.make_global_ba_problem(seed)
with open(out, "wb") as f:
    f.write(np.array([len(p["poses"]), len(p["points"]), len(p["e_cam"])], np.int32).tobytes())
    for key, dt in (("poses", np.float64), ("fixed", np.uint8), ("points", np.float64), ("e_cam", np.int32), ("e_pt", np.int32),
                    ("e_obs", np.float64), ("e_inv_sigma", np.float64), ("cam", np.float64)):
        f.write(np.ascontiguousarray(p[key], dt).tobytes())
print("wrote", out, {k: np.asarray(v).shape for k, v in p.items() if hasattr(v, "__len__")})
