#!/usr/bin/env python3
"""Split-K factor of sd_conv_gemm_f16 on the small-M launches of the UNet at one image per call (UNet batch 2): the launch rule's choice
(knob 0) against forced factors 1 .. 12 (tuning knob bits 24-27 of `epi`, see include/sd_hip.h).  Minimum of two rounds per factor."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from coma_amd.sd import ops  # noqa: E402

dev = "cuda:0"
WS = torch.empty(16 << 20, dtype=torch.float32, device=dev)          # the LaunchGraph's 64 MiB
# M, N, K, taps, hw, residual
SHAPES = [(512, 1280, 1280, 1, None, True), (2048, 640, 640, 1, None, True), (128, 1280, 1280, 1, None, True), (512, 1280, 5120, 1, None, True),
          (2048, 640, 2560, 1, None, True), (512, 3840, 1280, 1, None, False), (2048, 1920, 640, 1, None, False), (128, 1280, 11520, 9, 64, True),
          (512, 1280, 11520, 9, 256, True), (2048, 640, 5760, 9, 1024, True), (8192, 320, 2880, 9, 4096, True), (8192, 320, 640, 1, None, True)]
FACTORS = [0, 1, 2, 3, 4, 6, 8, 10, 12, 15]
for M, N, K, taps, hw, res in SHAPES:
    C = K // taps
    if taps == 9:
        B, H = M // hw, int(hw ** 0.5)
        kw = dict(batch=B, in_h=H, in_w=H, c0=C, n=N, taps=9)
    else:
        kw = dict(batch=M, in_h=1, in_w=1, c0=K, n=N)
    x = torch.randn(M, C, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    best = {f: 1e30 for f in FACTORS}
    for rnd in range(3):
        for f in FACTORS:
            for _ in range(3):
                ops.conv_gemm(x, w, out, bias=b, res=r, epi=f << 24, workspace=WS, **kw)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                ops.conv_gemm(x, w, out, bias=b, res=r, epi=f << 24, workspace=WS, **kw)
            e.record()
            torch.cuda.synchronize()
            if rnd:
                best[f] = min(best[f], a.elapsed_time(e) / 30)
    print(f"M={M:5d} N={N:5d} K={K:6d} taps={taps}: " + "  ".join(f"{'rule' if f == 0 else f}:{best[f] * 1e3:6.1f}" for f in FACTORS) + "  us", flush=True)
