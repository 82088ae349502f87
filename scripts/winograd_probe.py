#!/usr/bin/env python3
"""is Winograd F(2x2,3x3) worth building?  For every 3x3 shape of the UNet / VAE: the shipped implicit-GEMM
launch against  input transform + 16 plane GEMMs (sd_conv_gemm_f16, nbatch_z = 16) + output transform, timed separately and as a
chain, results checked against the direct convolution.   python scripts/winograd_probe.py [vae]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
WS = torch.empty(96 << 20, dtype=torch.float32, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / reps)
    return best * 1e3


def probe(B, H, Cin, Cout):
    M, T = B * H * H, B * H * H // 4
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, Cin, device=dev, generator=g).half()
    w = (torch.randn(Cout, 9, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).half()
    bias = torch.randn(Cout, device=dev, generator=g).half()
    res = torch.randn(M, Cout, device=dev, generator=g).half() if M * Cout < (1 << 29) else None
    out_d = torch.empty(M, Cout, device=dev, dtype=torch.float16)
    out_w = torch.empty_like(out_d)
    V = torch.empty(16, T, Cin, device=dev, dtype=torch.float16)
    U = torch.empty(16, Cout, Cin, device=dev, dtype=torch.float16)
    Mp = torch.empty(16, T, Cout, device=dev, dtype=torch.float16)
    ops.winograd_weight(w, U, n=Cout, c=Cin)

    def direct():
        ops.conv_gemm(x, w.reshape(Cout, -1), out_d, batch=B, in_h=H, in_w=H, c0=Cin, n=Cout, taps=9, bias=bias, res=res, workspace=WS)

    def xin():
        ops.winograd_input(x, V, batch=B, h=H, w=H, c0=Cin)

    def gemm():
        ops.conv_gemm(V, U, Mp, batch=T, in_h=1, in_w=1, c0=Cin, n=Cout, nbatch_z=16, stride_a=T * Cin, stride_w=Cout * Cin, stride_out=T * Cout)

    def xout():
        ops.winograd_output(Mp, out_w, batch=B, h=H, w=H, n=Cout, bias=bias, res=res)

    def chain():
        xin(); gemm(); xout()

    direct(); chain(); torch.cuda.synchronize()
    ref = out_d.float()
    err = float((out_w.float() - ref).abs().max() / ref.abs().max())
    td, ti, tg, to, tc = timeit(direct), timeit(xin), timeit(gemm), timeit(xout), timeit(chain)
    fl = 2 * M * Cout * 9 * Cin
    print(f"B={B:3d} {H:3d}x{H:<3d} Cin={Cin:5d} Cout={Cout:5d} | direct {td:7.1f} us {fl/td/1e6:6.0f} TF | in {ti:6.1f} gemm {tg:7.1f} "
          f"({fl/2.25/tg/1e6:5.0f} TF) out {to:6.1f} chain {tc:7.1f} us | direct/chain {td/tc:4.2f} direct/gemm {td/tg:4.2f} | max err vs direct {err:.1e}",
          flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "vae":
    SHAPES = [(8, 512, 128, 128), (8, 256, 256, 256), (8, 128, 512, 512), (8, 64, 512, 512)]
else:
    SHAPES = [(16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 1920, 640),
              (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280)]
for s in SHAPES:
    probe(*s)
