#!/usr/bin/env python3
"""sd_xattn_chain_f16 at the UNet's 64 x 64 level (16 x 4096 rows): total time and cumulative time up to each debug stage
(1: h1, 2: n2, 3: q2, 4: a2), next to the six launches it replaces."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops

dev = "cuda:0"
B, RPS, C, LK = 16, 4096, 320, 77
M = B * RPS
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).half()
a, h = r(M, C), r(M, C)
wo1, wq, wo2 = r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5)
bo1, bo2, g2, b2, g3, b3 = r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1)
k2 = r(B * LK, C)
vt2 = ops.perm16_columns(r(B, C, LK))
h2, n3, dbg = (torch.empty(M, C, dtype=torch.float16, device=dev) for _ in range(3))
args = (a, h, wo1, bo1, g2, b2, wq, k2, vt2, wo2, bo2, g3, b3, h2, n3)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3


for stage in (1, 2, 3, 4, 0):
    t = timeit(lambda: ops.xattn_chain(*args, rows=M, rows_per_sample=RPS, lk=LK, ldv2=80, debug_out=dbg if stage else None, debug_stage=stage))
    print(f"fused, up to stage {stage or 'end'}: {t:7.1f} us")
h1, n2, q2, a2 = (torch.empty(M, C, dtype=torch.float16, device=dev) for _ in range(4))


def unfused():
    ops.conv_gemm(a, wo1, h1, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=bo1, res=h)
    ops.layernorm(h1, g2, b2, n2, rows=M, c=C)
    ops.conv_gemm(n2, wq, q2, batch=M, in_h=1, in_w=1, c0=C, n=C)
    ops.attention(q2, k2, vt2, a2, batch=B, heads=8, lq=RPS, lk=LK, d=40, ldq=C, ldk=C, ldv=80, ldo=C, scale=40 ** -0.5, vt_perm16=True)
    ops.conv_gemm(a2, wo2, h2, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=bo2, res=h1)
    ops.layernorm(h2, g3, b3, n3, rows=M, c=C)


print(f"unfused chain (6 launches): {timeit(unfused):7.1f} us")
