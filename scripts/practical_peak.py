#!/usr/bin/env python3
"""What the 1400 W package cap leaves of the 2.5 PF dense-fp16 MFMA peak (tuning aid): torch.matmul (hipBLASLt) on N(0,1) operands,
on all-zero operands (no data toggling), and this library's 3x3 convolution on both, with rocm-smi power / shader clock next to each."""
import os, sys, time, re, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
WS = torch.empty(96 << 20, dtype=torch.float32, device=dev)


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    pw = re.findall(r"Graphics Package Power \(W\): ([\d.]+)", out)
    sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return float(pw[0]) if pw else 0.0, int(sclk[0]) if sclk else 0


def loop(name, fn, flops, seconds=3.0):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append(smi()); time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n
    stop[0] = True; th.join()
    s = samples[2:] or samples
    print(f"{name:58s} {dt * 1e6:8.1f} us {flops / dt / 1e12:7.1f} TFLOP/s | {sum(p for p, _ in s) / len(s):6.0f} W {sum(c for _, c in s) / len(s):5.0f} MHz", flush=True)


for label, gen in (("N(0,1)", lambda *s: torch.randn(*s, device=dev).half()), ("zeros", lambda *s: torch.zeros(*s, device=dev).half())):
    for n in (8192,):
        a, b = gen(n, n), gen(n, n)
        loop(f"torch.matmul {n}^3 fp16, {label}", lambda: torch.matmul(a, b), 2 * n ** 3)
    M, N, K = 65536, 640, 5760
    x, w, bias = gen(M, K // 9), gen(N, K), gen(N)
    if label == "N(0,1)":
        w = w * K ** -0.5
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    loop(f"sd_conv_gemm_f16 3x3 M=65536 N=640 K=5760, {label}", lambda: ops.conv_gemm(x, w, out, bias=bias, batch=16, in_h=64, in_w=64, c0=640, n=N, taps=9, workspace=WS), 2 * M * N * K)
