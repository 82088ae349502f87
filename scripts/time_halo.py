#!/usr/bin/env python3
"""One 128-channel 3x3 layer of the VAE at 512 x 512, batch 8 (measure one layer first): GroupNorm (statistics from the
producer's column sums) + SiLU + conv3x3 128 -> 128 (+ bias, residual, column sums for the next GroupNorm)
  A: sd_groupnorm_colstats_f16 (finalise + apply pass) -> sd_conv_gemm_f16 (implicit GEMM, 256 x 128 four-wave tile)
  B: sd_groupnorm_table_f16 -> sd_conv3x3_halo_f16 (halo-patch convolution, affine + SiLU on the way into LDS)
A B A B, HIP events, min per arm.   python scripts/time_halo.py [C_in]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
B, H, W, n = 8, 512, 512, 128
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, hw = B * H * W, H * W
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, generator=g, device=dev).half()
w = (torch.randn(n, 9 * C, generator=g, device=dev) * (9 * C) ** -0.5).half()
bias = torch.randn(n, generator=g, device=dev).half()
res = torch.randn(M, n, generator=g, device=dev).half()
ga, be = (torch.rand(C, generator=g, device=dev) + 0.5).half(), (torch.randn(C, generator=g, device=dev) * 0.1).half()
cs_in = torch.zeros(M // 32, 2, C, dtype=torch.float32, device=dev)
xf = x.float().reshape(M // 32, 32, C)
cs_in[:, 0], cs_in[:, 1] = xf.sum(1), (xf * xf).sum(1)
del xf
stats = torch.empty(1 << 20, dtype=torch.float32, device=dev)
norm = torch.empty(M, C, dtype=torch.float16, device=dev)
out_a, out_b = torch.empty(M, n, dtype=torch.float16, device=dev), torch.empty(M, n, dtype=torch.float16, device=dev)
cs_a, cs_b = torch.zeros(M // 32, 2, n, dtype=torch.float32, device=dev), torch.zeros(M // 256, 2, n, dtype=torch.float32, device=dev)
ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)


def arm_a():
    ops.groupnorm_colstats(x, ga, be, norm, stats, cs_in, batch=B, hw=hw, c0=C, eps=1e-6, silu=True)
    ops.conv_gemm(norm, w, out_a, batch=B, in_h=H, in_w=W, c0=C, n=n, taps=9, bias=bias, res=res, colstats=cs_a, workspace=ws)


def arm_b():
    ops.groupnorm_table(x, ga, be, stats, batch=B, hw=hw, c0=C, eps=1e-6, colstats0=cs_in)
    ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, gn_affine=stats, silu=True, colstats=cs_b)


def conv_only_a():
    ops.conv_gemm(norm, w, out_a, batch=B, in_h=H, in_w=W, c0=C, n=n, taps=9, bias=bias, res=res, colstats=cs_a, workspace=ws)


def conv_only_b():
    ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, gn_affine=stats, silu=True, colstats=cs_b)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


arm_a(); arm_b(); torch.cuda.synchronize()
err = float((out_a.float() - out_b.float()).abs().max()) / float(out_a.float().abs().max())
cerr = float((cs_a[:, 0].reshape(B, -1, n).sum(1) - cs_b[:, 0].reshape(B, -1, n).sum(1)).abs().max())
print(f"C={C}: max |A - B| / max|A| = {err:.2e}; per-sample column-sum difference {cerr:.3g}")
best = {}
for rnd in range(3):
    for name, fn in (("A norm+gemm", arm_a), ("B table+halo", arm_b), ("A gemm only", conv_only_a), ("B halo only", conv_only_b)):
        t = timeit(fn)
        if rnd:
            best[name] = min(best.get(name, 1e30), t)
fl = 2 * M * n * 9 * C
for k, v in best.items():
    print(f"  {k:14s} {v:8.1f} us  {fl / v / 1e6:7.1f} TF/s")
if "--ablate" in sys.argv:
    var = {
        "halo full": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, gn_affine=stats, silu=True, colstats=cs_b),
        "no silu": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, gn_affine=stats, silu=False, colstats=cs_b),
        "no affine": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, colstats=cs_b),
        "no affine/res/stats": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias),
        "no res/stats": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, gn_affine=stats, silu=True),
    }
    best = {}
    for rnd in range(3):
        for name, fn in var.items():
            t = timeit(fn)
            if rnd:
                best[name] = min(best.get(name, 1e30), t)
    for k, v in best.items():
        print(f"  {k:22s} {v:8.1f} us  {fl / v / 1e6:7.1f} TF/s")
    var2 = {
        "res only": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, res=res, gn_affine=stats, silu=True),
        "stats only": lambda: ops.conv3x3_halo(x, w, out_b, batch=B, h=H, w_=W, c=C, bias=bias, gn_affine=stats, silu=True, colstats=cs_b),
    }
    for name, fn in var2.items():
        t = min(timeit(fn) for _ in range(3))
        print(f"  {name:22s} {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s")
