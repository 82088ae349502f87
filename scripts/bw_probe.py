#!/usr/bin/env python3
"""Achievable copy bandwidth at the UNet's tensor sizes (tuning aid): torch copy_ vs the GroupNorm / LayerNorm kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"


def timeit(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for rows, c in ((65536, 320), (16384, 640), (4096, 1280)):
    x = torch.randn(rows, c, device=dev).half()
    y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x))
    mb = 2 * rows * c * 2 / 1e6
    line = f"rows={rows:8d} C={c:5d} ({mb:7.1f} MB r+w) copy_ {us:7.1f} us {mb / us / 1e3:5.2f} TB/s"
    g, b = torch.randn(c, device=dev).half(), torch.randn(c, device=dev).half()
    us = timeit(lambda: ops.layernorm(x, g, b, y, rows=rows, c=c, eps=1e-5))
    line += f" | layernorm {us:7.1f} us {mb / us / 1e3:5.2f} TB/s"
    B = 16 if rows <= 65536 else 8
    hw = rows // B
    stats = torch.empty(B * 32 * 2 + 64, dtype=torch.float32, device=dev)
    us = timeit(lambda: ops.groupnorm(x, g, b, y, stats, batch=B, hw=hw, c0=c, eps=1e-5, silu=True))
    line += f" | groupnorm+silu (own stats) {us:7.1f} us"
    print(line, flush=True)
print("groupnorm apply with producer-side column statistics (the in-graph path):")
for B, hw, c in ((16, 4096, 320), (16, 1024, 640), (16, 256, 1280), (16, 4096, 640)):
    rows = B * hw
    x = torch.randn(rows, c, device=dev).half(); y = torch.empty_like(x)
    g, b = torch.randn(c, device=dev).half(), torch.randn(c, device=dev).half()
    stats = torch.empty(1 << 20, dtype=torch.float32, device=dev)
    cs = torch.randn(rows // 32, 2, c, device=dev).abs().float()
    us = timeit(lambda: ops.groupnorm_colstats(x, g, b, y, stats, cs, batch=B, hw=hw, c0=c, eps=1e-5, silu=True))
    mb = 2 * rows * c * 2 / 1e6
    usc = timeit(lambda: y.copy_(x))
    print(f"B={B} hw={hw} C={c}: groupnorm(colstats)+silu {us:6.1f} us ({mb / us:5.2f} GB/ms... {mb / us / 1e3 * 1e3:.0f} GB/s x1e-3) copy {usc:5.1f} us", flush=True)
