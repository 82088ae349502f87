#!/usr/bin/env python3
"""Per-phase cycle breakdown of conv_gemm launches (tuning aid): SD_GEMM_DBG=1 python scripts/phase_gemm.py
For each shape: kernel time (HIP events), and per workgroup the shader-clock cycles spent in
prologue (entry -> first K tile landed), K loop, epilogue; plus the effective clock = cycles(first entry -> last exit) / time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SD_GEMM_DBG", "1")
import numpy as np
import torch
from coma_amd import _lib
from coma_amd.sd import ops
dev = "cuda:0"
WS = torch.empty(96 << 20, dtype=torch.float32, device=dev)


def run(M, N, K, taps=1, hw=None, epi=0, res=True, knob=0):
    C_ = K // taps
    if taps == 9:
        B, H = M // hw, int(hw ** 0.5)
        kw = dict(batch=B, in_h=H, in_w=H, c0=C_, n=N, taps=9)
    else:
        kw = dict(batch=M, in_h=1, in_w=1, c0=K, n=N)
    x = torch.randn(M, C_, device=dev).half()
    w = torch.randn(N, K, device=dev).half() * K ** -0.5
    b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res and not (epi & 1) else None
    out = torch.empty(M, N // 2 if epi & 1 else N, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.conv_gemm(x, w, out, bias=b, res=r, epi=epi | knob, workspace=WS, **kw)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    reps = 10
    for _ in range(reps):
        ops.conv_gemm(x, w, out, bias=b, res=r, epi=epi | knob, workspace=WS, **kw)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / reps * 1e3
    nb = 4096
    st = np.zeros((nb, 4), dtype=np.uint64)
    _lib.check(_lib.lib().sd_debug_timestamps(st.ctypes.data_as(C.c_void_p), nb), "dbg")
    st = st.astype(np.int64)
    ok = (st[:, 3] > st[:, 0]) & (st[:, 0] > 0)
    # only blocks of the last launch: their entry stamps lie within one kernel duration of the newest exit
    t_end = st[ok, 3].max()
    ok &= st[:, 0] > t_end - int(us * 2.6e3 * 1.5)
    s = st[ok]
    span = s[:, 3].max() - s[:, 0].min()
    pro, loop, epi_c = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2])
    tf = 2 * M * N * K / us / 1e6
    # how phase-locked are the workgroups?  fraction of resident blocks that sit in their epilogue, sampled over the launch
    ts = np.linspace(s[:, 0].min(), s[:, 3].max(), 400)
    act = ((s[:, 0][None] <= ts[:, None]) & (ts[:, None] < s[:, 3][None]))
    inep = ((s[:, 2][None] <= ts[:, None]) & (ts[:, None] < s[:, 3][None]))
    frac = inep.sum(1) / np.maximum(act.sum(1), 1)
    lock = f"in-epilogue share of resident blocks: mean {frac.mean():.2f} p90 {np.quantile(frac, 0.9):.2f} max {frac.max():.2f}"
    print(f"M={M:6d} N={N:5d} K={K:6d} t={taps} epi={epi} | {us:7.1f} us {tf:6.0f} TF | blocks {len(s):4d} span {span / 1e3:6.1f} kcyc clk {span / us / 1e3:4.2f} GHz | "
          f"prologue {np.median(pro) / 1e3:5.1f}k  loop {np.median(loop) / 1e3:6.1f}k  epilogue {np.median(epi_c) / 1e3:5.1f}k (max {epi_c.max() / 1e3:5.1f}k) | {lock}")


SH = [(65536, 320, 320, 1, None, 0), (65536, 320, 1280, 1, None, 0), (65536, 320, 2880, 9, 4096, 0), (65536, 640, 5760, 9, 4096, 0),
      (16384, 640, 640, 1, None, 0), (16384, 640, 5760, 9, 1024, 0), (16384, 1280, 11520, 9, 1024, 0),
      (4096, 1280, 1280, 1, None, 0), (4096, 1280, 11520, 9, 256, 0), (1024, 1280, 11520, 9, 64, 0),
      (65536, 2560, 320, 1, None, 1), (16384, 5120, 640, 1, None, 1), (4096, 10240, 1280, 1, None, 1)]
if os.environ.get("TAIL"):      # the short-K launches of the C = 640 / 1280 transformer blocks
    SH = [(16384, 640, 640, 1, None, 0), (4096, 1280, 1280, 1, None, 0), (16384, 1920, 640, 1, None, 0), (4096, 3840, 1280, 1, None, 0),
          (16384, 640, 2560, 1, None, 0), (4096, 1280, 5120, 1, None, 0), (16384, 5120, 640, 1, None, 1), (4096, 10240, 1280, 1, None, 1)]
knob = sum(1 << int(b) for b in sys.argv[1].split("+")) if len(sys.argv) > 1 else 0
for sh in SH:
    if os.environ.get("GEGLU_ONLY") and not sh[5]:
        continue
    run(*sh, knob=knob)
