#!/usr/bin/env python3
"""Categorised per-launch time of one eager UNet forward (batch 16): python scripts/unet_breakdown.py"""
import sys, os, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import weights
from coma_amd.sd.unet import HipUNet2DConditionModel
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
state = weights.random_state(weights.unet_shapes(), seed=0, device=dev)
unet = HipUNet2DConditionModel(state, batch=B, height=64, width=64, device=dev, use_graph=True)
g = torch.Generator(device=dev).manual_seed(0)
unet.set_context(torch.randn(B, 77, 768, generator=g, device=dev))
unet.x_in.copy_(torch.randn(unet.x_in.shape, generator=g, device=dev).half())
unet.timesteps.fill_(961.0)
unet.forward_static(); torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
for tag, fl, ms in unet.g.profile(reps=3):
    m = re.match(r"gemm M=(\d+) N=(\d+) K=(\d+) taps=(\d) z=(\d+)", tag)
    if m:
        M, N, K, taps, z = map(int, m.groups())
        if taps == 9:
            cat = f"conv3x3 M={M}"
        elif "winograd planes" in tag:
            cat = f"winograd plane products (conv3x3 M={4 * M})"
        elif z > 1:
            cat = "batched V^T projection"
        elif N >= 2 * K and N >= 2560:
            cat = "GEGLU ff1"
        elif K <= 1280 and N <= 1280:
            cat = "transformer / 1x1 linears K<=1280"
        else:
            cat = "ff2 and skip 1x1 (K>1280)"
    else:
        cat = "winograd transforms (+ folded GroupNorm)" if "winograd" in tag else (tag.split() or ["other"])[0].split("(")[0]
    acc[cat][0] += ms; acc[cat][1] += 1; acc[cat][2] += fl
tot = sum(v[0] for v in acc.values())
print(f"eager per-launch total {tot:.2f} ms")
for cat, (ms, n, fl) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms:7.3f} ms {100 * ms / tot:5.1f}%  n={n:3d}  {fl / ms / 1e9 if ms else 0:7.1f} TF/s  {cat}")
if "--shapes" in sys.argv:
    per = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for tag, fl, ms in unet.g.profile(reps=3):
        per[tag][0] += ms; per[tag][1] += 1; per[tag][2] += fl
    print("per launch tag (top 45 by time):")
    for tag, (ms, n, fl) in sorted(per.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"{ms:7.3f} ms n={n:3d} {ms / n * 1e3:7.1f} us each {fl / ms / 1e9 if ms else 0:7.1f} TF/s  {tag}")
