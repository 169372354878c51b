"""Developer tool (GPU box): what a pure write stream reaches on this chip (torch fill of 2 GB, 16 bytes per lane), next to a
copy and a pure read (sum) of the same size -- the bound the all-pairs Hamming matrix (2 bytes written per distance, 64 KB read
per 2 MB written) is priced against."""
import torch

n = 1024 * 1000 * 1000  # int16 elements = 2.048 GB
x = torch.empty(n, dtype=torch.int16, device="cuda")
y = torch.empty(n, dtype=torch.int16, device="cuda")


def t(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts


for name, f, b in (("fill  ", lambda: x.fill_(7), 2 * n), ("zero  ", lambda: x.zero_(), 2 * n), ("copy  ", lambda: y.copy_(x), 4 * n),
                   ("read  ", lambda: x.view(torch.int32).sum(), 2 * n)):
    ts = t(f)
    m = sorted(ts)[len(ts) // 2]
    print(f"{name} {2 * n / 1e9:.2f} GB: {[round(v, 3) for v in ts]} median {m:.3f} ms = {b / m / 1e9:.2f} TB/s")
