#!/usr/bin/env python3
"""Time the UNet launch graph at BASELINE config 2 shape (batch 16 = 8 images x CFG, 64x64 latents)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import weights
from coma_amd.sd.unet import HipUNet2DConditionModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
state = weights.random_state(weights.unet_shapes(), seed=0, device=dev)
unet = HipUNet2DConditionModel(state, batch=B, height=64, width=64, device=dev, use_graph="--eager" not in sys.argv, cfg_shared_prefix="--shared" in sys.argv,
                               fuse_xchain="--no-xchain" not in sys.argv, fuse_xfront="--no-xfront" not in sys.argv, fuse_xtail="--no-xtail" not in sys.argv, fuse_qkv="--no-qkv" not in sys.argv,
                               winograd_min_batch=int(next((a.split("=")[1] for a in sys.argv if a.startswith("--wino-min-batch=")), 8)),
                               **({"xtail_min_rows": 0} if "--xtail-always" in sys.argv else {}))
g = torch.Generator(device=dev).manual_seed(0)
unet.set_context(torch.randn(B, 77, 768, generator=g, device=dev))
unet.x_in.copy_(torch.randn(unet.x_in.shape, generator=g, device=dev).half())
unet.x_in[:, :, 9:] = 0
unet.x_in[B // 2:] = unet.x_in[:B // 2]          # the two CFG halves carry the same sample
unet.timesteps.fill_(961.0)
unet.forward_static(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    unet.forward_static()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"UNet B={B}: {dt*1e3:.2f} ms/forward, {unet.g.flops/1e12:.2f} TFLOP/forward -> {unet.g.flops/dt/1e12:.1f} TFLOP/s "
      f"({len(unet.g.launches)} launches); eps finite={bool(torch.isfinite(unet.eps.float()).all())}")
if "--profile" in sys.argv:
    import collections
    prof = unet.g.profile()
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for tag, fl, ms in prof:
        acc[tag][0] += ms; acc[tag][1] += 1; acc[tag][2] += fl
    tot = sum(v[0] for v in acc.values())
    print(f"eager per-launch total {tot:.2f} ms")
    for tag, (ms, n, fl) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{ms:8.3f} ms {100*ms/tot:5.1f}%  n={n:3d}  {fl/ms/1e9 if ms else 0:7.1f} TF/s  {tag}")
