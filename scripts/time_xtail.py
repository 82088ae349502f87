#!/usr/bin/env python3
"""Time sd_xtail_f16 at the 64 x 64 level (16 x 4096 rows, C = 320) against the three launches it replaces: python scripts/time_xtail.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
from coma_amd.sd.weights import geglu_interleave

dev = "cuda:0"
M, C = 65536, 320
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device=dev) * scale).half()
n3, h2, x = r(M, C), r(M, C), r(M, C)
w1, b1 = geglu_interleave(r(8 * C, C, scale=C**-0.5), r(8 * C, scale=0.1))
w2, b2, wpo, bpo = r(C, 4 * C, scale=(4 * C)**-0.5), r(C, scale=0.1), r(C, C, scale=C**-0.5), r(C, scale=0.1)
out = torch.empty(M, C, dtype=torch.float16, device=dev)
cs = torch.zeros(M // 32, 2, C, dtype=torch.float32, device=dev)
f, h3, out2 = torch.empty(M, 4 * C, dtype=torch.float16, device=dev), torch.empty(M, C, dtype=torch.float16, device=dev), torch.empty(M, C, dtype=torch.float16, device=dev)


def fused():
    ops.xtail(n3, h2, x, w1, b1, w2, b2, wpo, bpo, out, cs, rows=M)


def unfused():
    ops.conv_gemm(n3, w1, f, batch=M, in_h=1, in_w=1, c0=C, n=8 * C, taps=1, bias=b1, epi=ops.EPI_GEGLU)
    ops.conv_gemm(f, w2, h3, batch=M, in_h=1, in_w=1, c0=4 * C, n=C, taps=1, bias=b2, res=h2)
    ops.conv_gemm(h3, wpo, out2, batch=M, in_h=1, in_w=1, c0=C, n=C, taps=1, bias=bpo, res=x)


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


fl = 2 * M * C * 13 * C
for name, fn in (("xtail", fused), ("three launches", unfused), ("xtail", fused), ("three launches", unfused)):
    ms = t(fn)
    print(f"{name:15s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF/s")
print("max |fused - unfused| =", float((out.float() - out2.float()).abs().max()), " max |out| =", float(out2.float().abs().max()))
