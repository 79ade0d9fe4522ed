#!/usr/bin/env python3
"""Fill and copy rates of torch's own kernels on this GPU, next to the engine's ONE_VALUE / UNCOMPRESSED columns."""
import torch
def rate(fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
for gib in (1, 2):
    n = gib << 28
    x = torch.empty(n, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
    print(f"{gib} GiB fill  {rate(lambda: x.fill_(7), x.numel()*4):8.0f} GB/s (bytes written)")
    print(f"{gib} GiB zero  {rate(lambda: x.zero_(), x.numel()*4):8.0f} GB/s")
    print(f"{gib} GiB copy  {rate(lambda: y.copy_(x), x.numel()*8):8.0f} GB/s (read + written)")
