"""Developer tool: achievable device copy bandwidth at a few sizes (calibrates the HBM roofline)."""
import torch

for mb in (16, 80, 256, 1024):
    x = torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"copy {mb:5d} MB: {ms * 1e3:8.1f} us  ->  {2 * mb / 1024 / (ms / 1e3) / 1e3 * 1.048576:6.2f} TB/s (read+write)")
