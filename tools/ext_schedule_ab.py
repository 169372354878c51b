"""Developer tool (GPU box): the extractor's launch order outside the instrumented mode -- level 0's FAST cells and blur next to
the resize chain (VO_ORB_OPT_EARLY_LEVEL0 = 1) against the default order (0) -- interleaved; extraction alone and
extraction + all-pairs Hamming, 1024 frames per step.  Also checks that both orders give identical key-points / descriptors."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

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


def step(match):
    with torch.cuda.stream(stream):
        ext.extract_batch_dev(frames, kps, desc[:B], cnt)
        if match:
            desc[B].copy_(desc[0])
            vo.hamming_matrix_batch_dev(desc[:B, :NM], desc[1:, :NM], dmat, stream=stream.cuda_stream)


def timed(match, n=12):
    for _ in range(2):
        step(match)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(match)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = None
for early in (1, 0):
    ext.set_early_level0(bool(early))
    step(False)
    torch.cuda.synchronize()
    got = (kps.cpu().numpy().copy(), desc[:B].cpu().numpy().copy(), cnt.cpu().numpy().copy())
    if ref is None:
        ref = got
    else:
        assert all(np.array_equal(a, b) for a, b in zip(ref, got)), "the two launch orders disagree"
print("identical outputs in both orders")
for rep in range(3):
    for early in (1, 0):
        ext.set_early_level0(bool(early))
        print(f"early_level0={early}: extraction {timed(False):.3f} ms, extraction + Hamming {timed(True):.3f} ms per {B} frames")
