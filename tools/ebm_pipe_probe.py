"""Developer tool (GPU box): extract + all-pairs Hamming (BASELINE configs[1]) with the 1024-frame step cut into P parts that
run on P streams (an extractor handle per stream, its Hamming launch on the same stream), against the one-stream order.
usage: python tools/ebm_pipe_probe.py [parts=1,2,4] [ham=0] [blur=0] [reps=3]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
parts_list = [int(x) for x in kw.get("parts", "1,2,4").split(",")]
vo.set_option("hamming_kernel", int(kw.get("ham", "0")))
B, NM = 1024, 1000
SB8 = 5123128 + 2064000
frames_all = torch.from_numpy(synth.make_frames(32)).cuda().repeat(B // 32, 1, 1).contiguous()


class Part:
    def __init__(self, n, off):
        self.n, self.off = n, off
        self.stream = torch.cuda.Stream()
        self.ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
        self.ext.set_stream(self.stream.cuda_stream)
        self.ext.set_blur_kernel(int(kw.get("blur", "0")))
        cap = self.ext.max_keypoints()
        with torch.cuda.stream(self.stream):
            self.frames = frames_all[off:off + n]
            self.kps = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
            self.desc = torch.zeros((n + 1, cap, 32), dtype=torch.uint8, device="cuda")
            self.cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
            self.dmat = torch.zeros((n, NM, NM), dtype=torch.int16, device="cuda")

    def step(self):
        with torch.cuda.stream(self.stream):
            self.ext.extract_batch_dev(self.frames, self.kps, self.desc[:self.n], self.cnt)
            self.desc[self.n].copy_(self.desc[0])
            vo.hamming_matrix_batch_dev(self.desc[:self.n, :NM], self.desc[1:, :NM], self.dmat, stream=self.stream.cuda_stream)


def timed(parts, n=10):
    for _ in range(2):
        for p in parts:
            p.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for p in parts:
            p.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


sets = {P: [Part(B // P, i * (B // P)) for i in range(P)] for P in parts_list}
for rep in range(int(kw.get("reps", "3"))):
    print("  ".join(f"{P} part(s): {timed(sets[P]):.3f} ms = {SB8 * B / timed(sets[P]) / 1e6 / 8000 * 100:.1f} %" for P in parts_list), flush=True)
