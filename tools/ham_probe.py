"""Developer tool (GPU box): time of the batched all-pairs Hamming kernel (1024 pairs of 1000 x 1000 descriptors, u16 matrix
written) and a check against numpy on the first pair.  usage: [VO_HIP_LIB=...] python tools/ham_probe.py [pairs] [kernel: 0 matrix cores / 1 VALU]"""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(os.environ.get("VO_HAM_N", "1000"))
KERNEL = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if hasattr(vo, "set_option"):
    try:
        vo.set_option("hamming_kernel", KERNEL)
    except Exception as e:  # an older library
        print("(no hamming_kernel option:", e, ")")
g = torch.Generator(device="cuda").manual_seed(1)
desc = torch.randint(0, 256, (B + 1, max(N, 1024), 32), dtype=torch.uint8, device="cuda", generator=g)
dmat = torch.zeros((B, N, N), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream()
ts = []
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    vo.hamming_matrix_batch_dev(desc[:B, :N], desc[1:, :N], dmat, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
CHECK = not os.environ.get("VO_HAM_NOCHECK")  # ablated developer builds
if not CHECK:
    print(f"k_hamming[{KERNEL}] (unchecked): {[round(x, 4) for x in ts]} median {float(np.median(ts[2:])):.4f} ms"); sys.exit(0)
a, b = desc[0, :N].cpu().numpy(), desc[1, :N].cpu().numpy()
want = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2).astype(np.int16)
assert np.array_equal(dmat[0].cpu().numpy(), want), "Hamming matrix mismatch"
a, b = desc[B - 1, :N].cpu().numpy(), desc[B, :N].cpu().numpy()
want = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2).astype(np.int16)
assert np.array_equal(dmat[B - 1].cpu().numpy(), want), "Hamming matrix mismatch (last pair)"
t = float(np.median(ts[2:]))
print(f"k_hamming[{KERNEL}], {B} pairs of {N} x {N}: {[round(x, 4) for x in ts]} median {t:.4f} ms = {B * N * N * 2 / t / 1e9:.2f} TB/s of matrix written; exact")
