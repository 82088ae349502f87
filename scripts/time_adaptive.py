#!/usr/bin/env python3
"""Where the adaptive-mask loop spends its time (batch of AB images: python scripts/time_adaptive.py [AB]): plug-in host time vs everything else."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline, SyntheticHumanMaskPredictor, default_adaptive_mask_settings
dev = "cuda:0"
AB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=AB, height=512, width=512, device=dev, seed=0)


class Timed(SyntheticHumanMaskPredictor):
    t = 0.0
    n = 0

    def __call__(self, image_u8):
        t0 = time.perf_counter()
        r = super().__call__(image_u8)
        Timed.t += time.perf_counter() - t0
        Timed.n += 1
        return r


pipe.register_adaptive_mask_model(Timed())
pipe.register_adaptive_mask_settings(default_adaptive_mask_settings(50, "p"))
g = torch.Generator().manual_seed(5)
image = torch.rand(AB, 3, 512, 512, generator=g) * 2 - 1
mask = torch.zeros(AB, 1, 512, 512)
mask[:, :, 100:420, 150:400] = 1
pe, ne = torch.randn(AB, 77, 768, generator=g), torch.randn(AB, 77, 768, generator=g)
gen = torch.Generator(device=dev)


def one(seed, adaptive=True):
    gen.manual_seed(seed)
    return pipe(image=image, default_mask_image=mask, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=50,
                strength=0.98, guidance_scale=11.0, generator=gen, output_type="u8", use_adaptive_mask=adaptive,
                enforce_full_mask_ratio=0.0, human_detection_thres=0.015).images


one(0)
torch.cuda.synchronize()
Timed.t, Timed.n = 0.0, 0
t0 = time.perf_counter()
one(1)
torch.cuda.synchronize()
ta = time.perf_counter() - t0
t0 = time.perf_counter()
one(1, adaptive=False)
torch.cuda.synchronize()
tf = time.perf_counter() - t0
print(f"adaptive loop {ta*1e3:.0f} ms per batch of {AB}; same schedule without adaptation {tf*1e3:.0f} ms; "
      f"plug-in (host, {Timed.n} calls) {Timed.t*1e3:.0f} ms; adaptation overhead excluding the plug-in {(ta-tf-Timed.t)*1e3:.0f} ms "
      f"(21 x [VAE decode + u8 + D2H + mask glue + VAE encode])")
