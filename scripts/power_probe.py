#!/usr/bin/env python3
"""Sample rocm-smi (power, shader clock) while one kernel class runs back to back (tuning aid): is the chip power-capped?"""
import subprocess, sys, os, time, threading, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
WS = torch.empty(96 << 20, dtype=torch.float32, device=dev)


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    cap = re.findall(r"Max Graphics Package Power \(W\): ([\d.]+)", out)
    return (pw[0] if pw else "?"), (sclk[0] if sclk else "?"), (cap[0] if cap else "?")


def loop(name, fn, seconds=4.0):
    stop = [False]
    samples = []

    def sampler():
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.3)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    print(f"{name:34s} {dt / n * 1e6:8.1f} us/launch | power W {[s[0] for s in samples[2:8]]} sclk MHz {[s[1] for s in samples[2:8]]} cap {samples[0][2]}", flush=True)


def gemm(M, N, K, taps=1, hw=None, epi=0):
    C = K // taps
    if taps == 9:
        B, H = M // hw, int(hw ** 0.5)
        kw = dict(batch=B, in_h=H, in_w=H, c0=C, n=N, taps=9)
    else:
        kw = dict(batch=M, in_h=1, in_w=1, c0=K, n=N)
    x = torch.randn(M, C, device=dev).half(); w = torch.randn(N, K, device=dev).half() * K ** -0.5; b = torch.randn(N, device=dev).half()
    out = torch.empty(M, N // 2 if epi & 1 else N, device=dev, dtype=torch.float16)
    return lambda: ops.conv_gemm(x, w, out, bias=b, epi=epi, workspace=WS, **kw)


print("idle", smi())
loop("conv 64x64 N=640 K=5760", gemm(65536, 640, 5760, 9, 4096))
loop("conv 32x32 N=1280 K=11520", gemm(16384, 1280, 11520, 9, 1024))
loop("GEGLU M=65536 N=2560 K=320", gemm(65536, 2560, 320, epi=1))
loop("linear M=65536 N=320 K=320", gemm(65536, 320, 320))
zeros = torch.zeros(65536, 640, device=dev).half()
f = gemm(65536, 640, 5760, 9, 4096)
print("(same conv, all-zero operands would show the data-dependent power; skipped)")
