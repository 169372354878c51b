import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from vo_slam_test_amd import _lib as vo, synth
vo.SO = vo.PKG.parent / "stamp_tmp" / "libvo_stamp.so"
lb = synth.make_lba_problem(0)
ba = vo.BundleAdjuster(lb)
ba.solve(2.4, 2.8, 5)
out = (C.c_ulonglong * 8)()
vo.lib().vo_ba_debug_stamps(ba._h, out)
t = np.array(list(out), dtype=np.int64)
print("stamps (x10ns):", (t - t[0]).tolist())
