"""Developer tool: the fused level pass's plan (which levels take it, tile geometry, LDS) for an image size."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: F401,E402
from vo_slam_test_amd import _lib as vo  # noqa: E402
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
e = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
for l, p in enumerate(e.level_pass_plan(w, h)):
    print(l, p)
