"""Developer tool: where the fused level pass differs from the oracle (pyramid / blur / candidates per level)."""
import sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402
import oracle_lib as orc  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
img = synth.make_frame(0, w=w, h=h) if (w, h) != (640, 480) else synth.make_frame(0)
e = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
p = orc.orb_params()
kps, desc = e(img)
lev = orc.pyramid(p, img)
for l in range(8):
    g = e.get_level(0, l)
    d = np.argwhere(g != lev[l])
    print(f"level {l} {lev[l].shape}: pyramid mismatches {len(d)}", d[:5].tolist() if len(d) else "")
    b, ob = e.get_level(0, l, blurred=True), orc.blur(lev[l])
    d = np.argwhere(b != ob)
    if len(d):
        ys, xs = d[:, 0], d[:, 1]
        print(f"   blur mismatches {len(d)}: rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; first {d[:8].tolist()}")
        print("   distinct cols", sorted(set(xs.tolist()))[:40], "distinct rows", sorted(set(ys.tolist()))[:40])
    cx, cy, cr = orc.level_candidates(p, lev[l])
    gx, gy, gr = e.get_candidates(0, l)
    same = len(gx) == len(cx) and np.array_equal(gx, cx) and np.array_equal(gy, cy) and np.array_equal(gr, cr)
    print(f"   candidates {len(gx)} vs {len(cx)}: {'ok' if same else 'DIFFER'}")
