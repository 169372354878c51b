"""Developer tool (GPU box): does the all-pairs Hamming kernel of batch A (matrix cores + HBM writes) overlap the extraction
of batch B (VALU issue) when they are launched on two streams?  Times, per 1024 frames: extraction alone, Hamming alone, both
launched together (Hamming first / extraction first), interleaved and repeated.
usage: python tools/overlap_probe.py [ham=0|1] [prio=0|1]  (prio=1: the Hamming stream gets the higher stream priority)"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
vo.set_option("hamming_kernel", int(kw.get("ham", "0")))
B, NM = 1024, 1000
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream(priority=-1 if kw.get("prio", "0") == "1" else 0)
ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
ext.set_stream(s1.cuda_stream)
cap = ext.max_keypoints()
with torch.cuda.stream(s1):
    frames = torch.from_numpy(synth.make_frames(32)).cuda().repeat(B // 32, 1, 1).contiguous()
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda")
    descA = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    dmat = torch.zeros((B, NM, NM), dtype=torch.int16, device="cuda")
    ext.extract_batch_dev(frames, kps, descA[:B], cnt)
    descA[B].copy_(descA[0])
torch.cuda.synchronize()


def run(mode, n=8):
    def once():
        if mode in ("ham", "ham+ext"):
            vo.hamming_matrix_batch_dev(descA[:B, :NM], descA[1:, :NM], dmat, stream=s2.cuda_stream)
        if mode in ("ext", "ham+ext", "ext+ham"):
            ext.extract_batch_dev(frames, kps, desc[:B], cnt)
        if mode == "ext+ham":
            vo.hamming_matrix_batch_dev(descA[:B, :NM], descA[1:, :NM], dmat, stream=s2.cuda_stream)
        torch.cuda.synchronize()
    once()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        once()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for rep in range(3):
    print("  ".join(f"{m}: {run(m):.3f} ms" for m in ("ext", "ham", "ham+ext", "ext+ham")), flush=True)
