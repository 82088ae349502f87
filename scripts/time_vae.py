#!/usr/bin/env python3
"""Time the VAE decoder / encoder graphs at batch 8, 512x512 (tuning aid)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd import weights
from coma_amd.sd import vae as vae_mod
from coma_amd.sd.vae import HipAutoencoderKL
if "--unfused" in sys.argv:
    vae_mod._VaeBase.fused_attention = False          # A/B: QK^T GEMM -> softmax -> PV GEMM through memory
if "--gemm-conv-out" in sys.argv:
    vae_mod._VaeBase.fused_conv_out = False           # A/B: GroupNorm kernel + 64-column implicit-GEMM tile for the decoder's conv_out
if "--halo128" in sys.argv:
    vae_mod._VaeBase.halo_widths = (128,)             # A/B: halo convolutions only for 128 output channels
if "--halo256" in sys.argv:
    vae_mod._VaeBase.halo_widths = (128, 256)
if "--packed-conv-in" in sys.argv:
    vae_mod._VaeBase.direct_conv_in = False           # A/B: im2col pass + K = 32 product for the encoder's conv_in
if "--no-halo" in sys.argv:
    vae_mod._VaeBase.halo_conv = False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
vae = HipAutoencoderKL(weights.random_state(weights.vae_shapes(), seed=1), batch=B, device=dev)
vae.dec.z.copy_(torch.randn(vae.dec.z.shape, device=dev).half()); vae.dec.z[:, :, 4:] = 0
vae.enc.x.copy_(torch.randn(vae.enc.x.shape, device=dev).half()); vae.enc.x[:, :, 3:] = 0
for name, m, run in (("decoder", vae.dec, vae.dec.decode_static), ("encoder", vae.enc, vae.enc.encode_static)):
    run(); run(); torch.cuda.synchronize()       # record + eager run, then the graph capture
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name} B={B}: {dt*1e3:.2f} ms, {m.g.flops/1e12:.2f} TFLOP -> {m.g.flops/dt/1e12:.1f} TFLOP/s ({len(m.g.launches)} launches)")
    if "--profile" in sys.argv:
        acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
        for tag, fl, ms in m.g.profile(reps=2):
            acc[tag][0] += ms; acc[tag][1] += 1; acc[tag][2] += fl
        for tag, (ms, n, fl) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:(40 if "--all" in sys.argv else 12)]:
            print(f"   {ms:8.3f} ms n={n:3d} {fl/ms/1e9 if ms else 0:7.1f} TF/s  {tag}")
