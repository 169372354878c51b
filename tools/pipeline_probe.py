"""Two trackers on two streams: does the latency-bound tail (searches, pose solves) hide behind the other batch's
extraction?  python tools/pipeline_probe.py [batch]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from vo_slam_test_amd import _lib as vo, synth
from vo_slam_test_amd.tracking import BatchTracker

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = 640, 480
NU = 32
uniq = synth.make_frames(NU)
udep = np.stack([synth.make_depth(i) for i in range(NU)])
frames = torch.from_numpy(np.stack([uniq[i % NU] for i in range(B)])).cuda()
depth = torch.from_numpy(np.stack([udep[i % NU] for i in range(B)]).view(np.int16)).cuda()
inv_depth = float(np.float32(1.0) / np.float32(synth.DEPTH_SCALE))
cam5 = synth.CAM.astype(np.float32)


def make(bsz, es=None, prio=0):
    st = torch.cuda.Stream(priority=prio)
    ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
    trk = BatchTracker(bsz, ext, cam5, synth.DIST, W, H, n_last=1100, n_local=2200, stream=st, extract_stream=es)
    with torch.cuda.stream(trk.ext_stream):
        ext.extract_batch_dev(frames[:NU], trk.kps[:NU], trk.desc[:NU], trk.cnt[:NU])
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        trk.frames.build_dev(trk.kps[:NU], trk.desc[:NU], trk.cnt[:NU], depth[:NU], inv_depth, stream=st.cuda_stream)
    torch.cuda.synchronize(); ext.sync()
    maps = []
    for i in range(NU):
        fr = trk.frames.download(i, stream=st.cuda_stream)
        maps.append(synth.make_tracking_map(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["depth"], seed=i))

    def stack(which, key, n, tail=()):
        o = np.zeros((bsz, n) + tail, maps[0][which][key].dtype)
        for f in range(bsz):
            a = maps[f % NU][which][key]
            o[f, :len(a)] = a[:n]
        return o
    last = dict(points=stack(2, "points", 1100, (3,)), flags=stack(2, "flags", 1100), octave=stack(2, "octave", 1100),
                angle=stack(2, "angle", 1100), desc=stack(2, "desc", 1100, (32,)))
    local = {k: stack(3, k, 2200, (3,) if k == "points" else (32,) if k == "desc" else ())
             for k in ("points", "flags", "u", "v", "ur", "level", "viewcos", "desc")}
    with torch.cuda.stream(st):
        trk.set_map(np.stack([maps[f % NU][0] for f in range(bsz)]), np.stack([maps[f % NU][1] for f in range(bsz)]), last, local)
    torch.cuda.synchronize()
    return trk


def run(name, trackers, bsz, chained, steps=20):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None
        for i in range(steps):
            t = trackers[i % len(trackers)]
            t.track(frames[:bsz], depth[:bsz], inv_depth, after=prev if chained else None)
            prev = t.extract_done
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{name}: {bsz * steps / dt / 1e3:.1f} k frames/s, {dt / steps * 1e3:.3f} ms per batch of {bsz}", flush=True)
    return [t.ninl.cpu().numpy().copy() for t in trackers]


es = torch.cuda.Stream()
shared = [make(B, es, -1) for _ in range(3)]
own = [make(B, torch.cuda.Stream(), -1) for _ in range(3)]
r0 = run("shared extraction stream + 2 high-priority tail streams", shared[:2], B, False, steps=24)[0]
for label, tk in (("shared extraction stream", shared), ("one extraction stream per batch", own)):
    for n in (2, 3):
        r = run(f"{label} + {n} high-priority tail streams", tk[:n], B, False, steps=24)
        assert all(np.array_equal(r0, x) for x in r)
print("inlier counts identical in every mode")
