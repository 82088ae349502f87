#!/usr/bin/env python3
"""Time single conv_gemm shapes (tuning aid): python scripts/time_gemm.py [knob_bits ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
WS = torch.empty(96 << 20, dtype=torch.float32, device=dev)


def bench_many(M, N, K, knobs, taps=1, base_epi=0, res=False, reps=20, hw=None, rounds=2):
    """One set of operands, every knob timed `rounds` times in turn (A B A B): the minimum per knob is reported, so that clock
    ramp-up / cold caches of the first timed launches of a process cannot favour whichever knob happens to be measured second."""
    C = K // taps
    if taps == 9:
        B, H = M // hw, int(hw ** 0.5)
        x = torch.randn(M, C, device=dev).half()
        kw = dict(batch=B, in_h=H, in_w=H, c0=C, n=N, taps=9)
    else:
        x = torch.randn(M, K, device=dev).half()
        kw = dict(batch=M, in_h=1, in_w=1, c0=K, n=N)
    w = torch.randn(N, K, device=dev).half() * K ** -0.5
    b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if (res and M < 1000000) else None
    out = torch.empty(M, N if not (base_epi & 1) else N // 2, device=dev, dtype=torch.float16)
    kw["workspace"] = WS
    best = [1e30] * len(knobs)
    for rnd in range(rounds + 1):                     # round 0 = warm-up, discarded
        for i, knob in enumerate(knobs):
            epi = base_epi | knob
            for _ in range(3):
                ops.conv_gemm(x, w, out, bias=b, res=r, epi=epi, **kw)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                ops.conv_gemm(x, w, out, bias=b, res=r, epi=epi, **kw)
            e.record(); torch.cuda.synchronize()
            if rnd:
                best[i] = min(best[i], a.elapsed_time(e) / reps)
    return [(ms, 2 * M * N * K / ms / 1e9) for ms in best]


def bench(M, N, K, taps=1, epi=0, res=False, reps=20, hw=None):
    return bench_many(M, N, K, [0], taps=taps, base_epi=epi, res=res, reps=reps, hw=hw)[0]


if os.environ.get("SMALL"):
    SHAPES_OVERRIDE = [(1024, 1280, 11520, 9, 64), (1024, 1280, 23040, 9, 64), (4096, 1280, 1280, 1, None), (4096, 1280, 11520, 9, 256), (4096, 1280, 23040, 9, 256),
                       (4096, 640, 5760, 9, 256), (1024, 1280, 1280, 1, None)]
if os.environ.get("OVH"):
    SHAPES_OVERRIDE = [(16384, 640, 64, 1, None), (16384, 640, 320, 1, None), (16384, 640, 640, 1, None), (16384, 640, 1280, 1, None),
                       (4096, 1280, 64, 1, None), (4096, 1280, 640, 1, None), (4096, 1280, 1280, 1, None), (4096, 1280, 2560, 1, None),
                       (65536, 320, 64, 1, None), (65536, 320, 320, 1, None)]
if os.environ.get("ONLY"):
    SHAPES_OVERRIDE = [(65536, 640, 5760, 9, 4096), (65536, 320, 2880, 9, 4096), (65536, 320, 1280, 1, None)]
    pass
if os.environ.get("VAE"):
    SHAPES_OVERRIDE = [(2097152, 128, 1152, 9, 262144), (2097152, 128, 2304, 9, 262144), (524288, 256, 2304, 9, 65536),
                       (131072, 512, 4608, 9, 16384), (2097152, 64, 1152, 9, 262144)]
SHAPES = [(65536, 320, 320, 1, None), (65536, 320, 2880, 9, 4096), (65536, 320, 1280, 1, None), (65536, 640, 5760, 9, 4096),
          (16384, 640, 640, 1, None), (16384, 640, 5760, 9, 1024), (16384, 1280, 11520, 9, 1024),
          (4096, 1280, 1280, 1, None), (4096, 1280, 5120, 1, None), (4096, 1280, 11520, 9, 256), (4096, 1280, 23040, 9, 256),
          (1024, 1280, 11520, 9, 64), (1024, 1280, 23040, 9, 64)]
GEGLU = [] if (os.environ.get("ONLY") or os.environ.get("SMALL")) else [(65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280)]
knobs = [0] + [sum(1 << int(b) for b in a.split("+")) for a in sys.argv[1:]]
for (M, N, K, taps, hw) in (SHAPES_OVERRIDE if (os.environ.get('SMALL') or os.environ.get('VAE') or os.environ.get('OVH') or os.environ.get('ONLY')) else SHAPES):
    row = [f"{ms*1e3:7.1f} us {tf:6.1f} TF" for ms, tf in bench_many(M, N, K, knobs, taps=taps, hw=hw, res=not os.environ.get('NORES'))]
    print(f"M={M:6d} N={N:5d} K={K:6d} taps={taps} | " + " | ".join(row))
for (M, N, K) in GEGLU:
    row = [f"{ms*1e3:7.1f} us {tf:6.1f} TF" for ms, tf in bench_many(M, N, K, knobs, base_epi=1)]
    print(f"GEGLU M={M:6d} N={N:5d} K={K:6d} | " + " | ".join(row))
