#!/usr/bin/env python3
"""One layer shape of the segmentation network at a time, every tile the fp32 GEMM has: python scripts/time_seg_gemm.py
(M, N, K = kh*kh*C, residual) at batch 8, 512 x 512 -> 800 x 800; HIP events over 20 launches; which tile / split the launch rule would
pick is the `tile=0` row.  Tuning aid for the dispatch rule in seg_conv_gemm_f32."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coma_amd.seg import ops  # noqa: E402

dev = torch.device("cuda:0")
# name, B, H, W, C, N, kh, stride, residual
SHAPES = [
    ("res2.conv3", 8, 200, 200, 64, 256, 1, 1, True),
    ("res2.conv1", 8, 200, 200, 256, 64, 1, 1, False),
    ("res2.conv2", 8, 200, 200, 64, 64, 3, 1, False),
    ("res3.conv3", 8, 100, 100, 128, 512, 1, 1, True),
    ("res3.conv1", 8, 100, 100, 512, 128, 1, 1, False),
    ("res3.conv2", 8, 100, 100, 128, 128, 3, 1, False),
    ("res4.conv3", 8, 50, 50, 256, 1024, 1, 1, True),
    ("res4.conv1", 8, 50, 50, 1024, 256, 1, 1, False),
    ("res4.conv2", 8, 50, 50, 256, 256, 3, 1, False),
    ("res5.conv3", 8, 25, 25, 512, 2048, 1, 1, True),
    ("res5.conv2", 8, 25, 25, 512, 512, 3, 1, False),
    ("fpn_lat2", 8, 200, 200, 256, 256, 1, 1, False),
    ("fpn_out3", 8, 100, 100, 256, 256, 3, 1, False),
    ("fpn_out2", 8, 200, 200, 256, 256, 3, 1, False),
    ("box_fc1", 8000, 1, 1, 12544, 1024, 1, 1, False),
    ("point_fc", 25088, 1, 1, 352, 256, 1, 1, False),
]
only = [a for a in sys.argv[1:] if not a.startswith("--")]
ws = torch.empty((256 << 20) if "--splits" in sys.argv else (16 << 20), dtype=torch.float32, device=dev)      # 64 MiB = the plan's; forced factors need more
for name, B, H, W, C, N, kh, stride, has_res in SHAPES:
    if only and not any(o in name for o in only):
        continue
    pad = kh // 2
    oh, ow = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    M, K = B * oh * ow, kh * kh * C
    x = torch.randn(B * H * W, C, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if has_res else None
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    byt = 4.0 * (B * H * W * C + N * K + M * N * (2 if has_res else 1))
    for tile in (0, 1, 2, 3):
        if tile == 3 and N > 64:
            continue
        for split in ((0, -1) if tile == 0 else ((-1, 2, 3, 4) if "--splits" in sys.argv else (-1,))):
            if split > 1 and K // 32 // split < 2:
                continue
            def run():
                ops.conv_gemm(x, w, out, batch=B, in_h=H, in_w=W, c=C, n=N, kh=kh, kw=kh, stride=stride, pad=pad, bias=bias, res=res,
                              res_mode=1 if has_res else 0, relu=True, tile=tile, workspace=ws, split_k=split)
            run()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            e.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(e) / 20
            print(f"{name:11s} M={M:7d} N={N:5d} K={K:6d} tile={tile} split={'rule' if split == 0 else ('off' if split < 0 else str(split)):4s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TF/s  "
                  f"{byt / ms / 1e6:6.0f} GB/s algorithmic", flush=True)
