#!/usr/bin/env python3
"""Time sd_attention_f16 on the UNet's self/cross-attention shapes (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import ops
dev = "cuda:0"
for (B, H, lq, lk, d) in [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 8, 4096, 77, 40)]:
    C = H * d
    q = torch.randn(B, lq, C, device=dev).half()
    k = torch.randn(B, lk, C, device=dev).half()
    ldv = (lk + 7) // 8 * 8
    vt = torch.randn(B, C, ldv, device=dev).half()
    out = torch.empty(B, lq, C, device=dev, dtype=torch.float16)
    ldv = (lk + 15) // 16 * 16
    vt = torch.randn(B, C, ldv, device=dev).half()
    row = []
    for perm in (False, True):
        for _ in range(2):
            ops.attention(q, k, vt, out, batch=B, heads=H, lq=lq, lk=lk, d=d, ldq=C, ldk=C, ldv=ldv, ldo=C, scale=d ** -0.5, vt_perm16=perm)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.attention(q, k, vt, out, batch=B, heads=H, lq=lq, lk=lk, d=d, ldq=C, ldk=C, ldv=ldv, ldo=C, scale=d ** -0.5, vt_perm16=perm)
        e.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(e) / 10
        row.append(f"{'perm16' if perm else 'plain '} {ms*1e3:8.1f} us {4*B*H*lq*lk*d/ms/1e9:7.1f} TF/s")
    print(f"attention B={B} H={H} lq={lq} lk={lk} d={d}: " + " | ".join(row))
