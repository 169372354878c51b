import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as orc
from vo_slam_test_amd import _lib as vo, synth
from vo_slam_test_amd.tracking import BatchTracker
B, W, H = 1, 640, 480
imgs = synth.make_frames(B, start=60)
raw = np.stack([synth.make_depth(60 + i) for i in range(B)])
inv = np.float32(1.0) / np.float32(synth.DEPTH_SCALE)
cam5 = synth.CAM.astype(np.float32)
for distorted in (False, True):
    dist = synth.DIST if distorted else None
    ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
    trk = BatchTracker(B, ext, cam5, dist, W, H, n_last=1100, n_local=2200)
    t_img = torch.from_numpy(imgs).cuda(); t_dep = torch.from_numpy(raw.view(np.int16)).cuda()
    # map from the device's own frame
    trk.ext.extract_batch_dev(t_img, trk.kps, trk.desc, trk.cnt)
    trk.frames.build_dev(trk.kps, trk.desc, trk.cnt, t_dep, float(inv))
    torch.cuda.synchronize()
    fr = trk.frames.download(0)
    print("distorted", distorted, "n", fr["n"], "x range", fr["x"].min(), fr["x"].max(), "cells", fr["cell_start"][-1])
    T, pose6, la, lo = synth.make_tracking_map(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["depth"], seed=0)
    trk.set_map(T[None], pose6[None], {k: v[None] for k, v in la.items()}, {k: v[None] for k, v in lo.items()})
    trk.track(t_img, t_dep, float(inv), keep_first=True)
    torch.cuda.synchronize()
    print(" q flags valid", int((trk.q0["flags"][0] & 1).sum()), "u", trk.q0["u"][0, :4].cpu().numpy(), "nm", trk.nm.cpu().numpy(),
          "asg0>=0", int((trk.assigned0[0] >= 0).sum()), "asg1>=0", int((trk.assigned[0] >= 0).sum()), "ninl", trk.ninl_first.cpu().numpy(), trk.ninl.cpu().numpy())
    try:
        trk.frames.match_status()
    except Exception as e:
        print(" status", e)
    trk.close(); ext.close()
