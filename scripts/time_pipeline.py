#!/usr/bin/env python3
"""Where one fixed-mask pipeline call (config 2: batch 8, 50 steps) spends its time: python scripts/time_pipeline.py [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
from coma_amd.sd import ops

dev = torch.device("cuda:0")
B = 8
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=B, height=512, width=512, device=dev, seed=0)
g = torch.Generator().manual_seed(100)
image = torch.rand(B, 3, 512, 512, generator=g) * 2 - 1
mask = torch.zeros(B, 1, 512, 512)
mask[:, :, 128:384, 128:384] = 1
pe, ne = torch.randn(B, 77, 768, generator=g), torch.randn(B, 77, 768, generator=g)
gens = torch.Generator(device=dev)


def call(img, msk, p, n, steps=50):
    gens.manual_seed(1)
    return pipe(image=img, default_mask_image=msk, prompt_embeds=p, negative_prompt_embeds=n, num_inference_steps=steps, strength=1.0,
                guidance_scale=11.0, generator=gens, output_type="u8", use_adaptive_mask=False).images


def wall(fn):
    fn(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print(f"pipe(), host inputs   : {wall(lambda: call(image, mask, pe, ne)):8.2f} ms")
d = [t.to(dev) for t in (image, mask, pe, ne)]
print(f"pipe(), device inputs : {wall(lambda: call(*d)):8.2f} ms")
print(f"pipe(), 1 step        : {wall(lambda: call(*d, steps=1)):8.2f} ms  (everything but 49 loop iterations)")
u = pipe.unet
print(f"50 bare UNet replays  : {wall(lambda: [u.forward_static() for _ in range(50)]):8.2f} ms")
hw = 64 * 64
lat = torch.zeros(B, hw, 4, device=dev); x0 = torch.zeros_like(lat)
ml = torch.zeros(B, hw, dtype=torch.float16, device=dev); mk = torch.zeros(B, hw, 4, dtype=torch.float16, device=dev)


def loop():
    for i in range(50):
        u.timesteps.fill_(float(981 - 20 * i))
        eps = u.forward_static()
        ops.cfg_ddim_step(eps, 64, lat, x0, ml, mk, u.x_in, batch=B, hw=hw, guidance=11.0, alpha_t=0.5, alpha_prev=0.6)


print(f"50 loop iterations    : {wall(loop):8.2f} ms  (fill_ + replay + cfg/ddim kernel)")
print(f"VAE decode / encode   : {wall(lambda: pipe.vae.dec.g.replay()):8.2f} / {wall(lambda: pipe.vae.enc.g.replay()):8.2f} ms")
