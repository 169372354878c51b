"""Developer tool (GPU box): BASELINE configs[1] as written -- extract + all-pairs 1000 x 1000 Hamming, 1024 frames per step --
for every combination of the process-wide / per-handle options named on the command line, interleaved and repeated so that
clock drift and box-to-box differences cancel.  Prints ms per step (uninstrumented = the library's default launch order; and the
instrumented per-stage times) and the share of the 8 TB/s HBM peak that the SURVEY 8d bytes (7 187 128 per frame) make.
usage: [VO_HIP_LIB=...] python tools/ebm_probe.py [ham=0,1] [blur=0,1] [db=0,1] [reps=3] [stages=1]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
hams = [int(x) for x in kw.get("ham", "0,1").split(",")]
blurs = [int(x) for x in kw.get("blur", "0").split(",")]
dbs = [int(x) for x in kw.get("db", "0").split(",")]  # VO_ORB_OPT_DESCRIBE_BLUR: 0 on demand, 1 blurred planes
reps = int(kw.get("reps", "3"))
B, NM = 1024, 1000
stream = torch.cuda.Stream()
ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
ext.set_stream(stream.cuda_stream)
cap = ext.max_keypoints()
with torch.cuda.stream(stream):
    frames = torch.from_numpy(synth.make_frames(32)).cuda().repeat(B // 32, 1, 1).contiguous()
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    dmat = torch.zeros((B, NM, NM), dtype=torch.int16, device="cuda")
ham_ev = []


def step():
    with torch.cuda.stream(stream):
        ext.extract_batch_dev(frames, kps, desc[:B], cnt)
        desc[B].copy_(desc[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        vo.hamming_matrix_batch_dev(desc[:B, :NM], desc[1:, :NM], dmat, stream=stream.cuda_stream)
        e1.record(stream)
        ham_ev.append((e0, e1))


def timed(n=10):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ham_ev.clear()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(np.mean([a.elapsed_time(b) for a, b in ham_ev]))


SB8 = 5123128 + 2064000
for rep in range(reps):
    for ham, blur, db in [(a, b, c) for a in hams for b in blurs for c in dbs]:
        vo.set_option("hamming_kernel", ham)
        ext.set_blur_kernel(blur)
        ext.set_describe_blur(db)
        ext.set_timing(False)
        t, th = timed()
        line = f"ham={ham} blur={blur} db={db}: default order {t:.3f} ms = {SB8 * B / t / 1e6 / 8000 * 100:.1f} % of HBM peak (hamming in it {th:.3f})"
        if kw.get("stages", "1") == "1":
            ext.set_timing(True)
            ti, thi = timed()
            ms, n = ext.get_timing()
            ext.set_timing(False)
            line += f"; instrumented {ti:.3f} ms = {SB8 * B / ti / 1e6 / 8000 * 100:.1f} %, stages " + \
                    " ".join(f"{k}={v / max(n, 1):.3f}" for k, v in ms.items() if k != "offsets") + f" hamming={thi:.3f}"
        print(line, flush=True)
