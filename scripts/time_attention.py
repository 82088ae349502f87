#!/usr/bin/env python3
"""Time sd_attention_f16 at the UNet's self-attention shapes and check it against torch fp32 (tuning aid).
   python scripts/time_attention.py            # env SD_ATTN_V=1|2 picks the kernel generation where both apply"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
from oracle import sd_oracle as so

dev = "cuda:0"
SHAPES = [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 8, 4096, 77, 40)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
for (B, H, lq, lk, d) in SHAPES:
    g = torch.Generator().manual_seed(0)
    C = H * d
    q = torch.randn(B, lq, C, generator=g).half().to(dev)
    k = torch.randn(B, lk, C, generator=g).half().to(dev)
    v = torch.randn(B, lk, C, generator=g).half().to(dev)
    ldv = (lk + 15) // 16 * 16
    vt = ops.perm16_columns(v.transpose(1, 2).contiguous())              # [B, C, ldv], key-permuted as the projection GEMM writes it
    out = torch.empty(B, lq, C, dtype=torch.float16, device=dev)
    kw = dict(batch=B, heads=H, lq=lq, lk=lk, d=d, ldq=C, ldk=C, ldv=ldv, ldo=C, scale=d ** -0.5, vt_perm16=True)
    for _ in range(3):
        ops.attention(q, k, vt, out, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.attention(q, k, vt, out, **kw)
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / 10)
    nb = min(B, 2)
    ref = so.attention_ref(q[:nb].float().cpu(), k[:nb].float().cpu(), v[:nb].float().cpu(), H, d ** -0.5)
    err = float((out[:nb].float().cpu() - ref).abs().max() / ref.abs().max())
    print(f"B={B} h={H} lq={lq} lk={lk} d={d}: {best * 1e3:8.1f} us  {4 * B * H * lq * lk * d / best / 1e9:7.1f} TF/s   max err / max|ref| = {err:.2e}", flush=True)
