"""Shared by tests/test_sd_adaptive_gpu.py and scripts/adaptive_check.py: seeded inputs of the adaptive loop, a deterministic mask
plug-in that runs identically on NumPy arrays (the reference's contract) and on device tensors, and the two runners."""
import numpy as np
import torch

from coma_amd.sd import weights
from coma_amd.sd.pipeline import SyntheticHumanMaskPredictor, default_adaptive_mask_settings
from oracle import sd_oracle as so


class BlockLumaPredictor:
    """human = 16 x 16 blocks whose summed luminance exceeds the image's mean block, inside an ellipse.  Integer arithmetic only,
    so the NumPy path (oracle, reference contract) and the device path give the same mask for the same uint8 image."""
    use_visualizer = False
    accepts_device_tensor = True
    BS = 16

    def _ellipse(self, H, W):
        yy, xx = np.mgrid[0:H, 0:W]
        return ((yy - H / 2) / (H * 0.34)) ** 2 + ((xx - W / 2) / (W * 0.22)) ** 2 <= 1.0

    def __call__(self, image_u8):
        H, W = image_u8.shape[:2]
        bs = self.BS
        if isinstance(image_u8, torch.Tensor):
            s3 = image_u8.to(torch.int64).sum(-1)
            blk = s3.reshape(H // bs, bs, W // bs, bs).sum((1, 3))
            hot = (blk * blk.numel() > blk.sum()).repeat_interleave(bs, 0).repeat_interleave(bs, 1)
            ell = torch.from_numpy(self._ellipse(H, W)).to(image_u8.device)
            return {"mask": (hot & ell).to(torch.uint8), "vis": None, "asset_mask": None}
        s3 = image_u8.astype(np.int64).sum(-1)
        blk = s3.reshape(H // bs, bs, W // bs, bs).sum((1, 3))
        hot = np.repeat(np.repeat(blk * blk.size > blk.sum(), bs, 0), bs, 1)
        return {"mask": (hot & self._ellipse(H, W)).astype(np.uint8), "vis": None, "asset_mask": None}


def make_plugin(kind="block"):
    return BlockLumaPredictor() if kind == "block" else SyntheticHumanMaskPredictor()


def iou(a, b):
    a, b = np.asarray(a) > 0, np.asarray(b) > 0
    u = np.logical_or(a, b).sum()
    return 1.0 if u == 0 else float(np.logical_and(a, b).sum() / u)


def make_inputs(B, HW=512, seed=5, guidance=11.0, ratio=0.5, thres=0.008):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, 3, HW, HW, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, HW, HW)
    for b in range(B):
        mask[b, :, HW // 8 + 4 * b:HW - HW // 8, HW // 6:HW - HW // 6 - 3 * b] = 1
    pe = torch.randn(B, 77, 768, generator=g).half().float()
    ne = torch.randn(B, 77, 768, generator=g).half().float()
    lat0 = torch.randn(B, 4, HW // 8, HW // 8, generator=g).half().float()
    return dict(B=B, HW=HW, image=image, mask=mask, pe=pe, ne=ne, lat0=lat0, guidance=guidance, ratio=ratio, thres=thres,
                settings=default_adaptive_mask_settings(50, "p"), default_np=(mask[:, 0].numpy() >= 0.5).astype(np.uint8))


def take(inp, idx):
    """The inputs of images `idx` only (a batch-1 run of image b of a batch-8 run)."""
    out = dict(inp)
    for k in ("image", "mask", "pe", "ne", "lat0", "default_np"):
        out[k] = inp[k][idx]
    out["B"] = len(idx)
    return out


def run_hip(pipe, inp, plugin, *, strength, seeds=None, dev="cuda:0", use_adaptive_mask=True, steps=50):
    B = inp["B"]
    pipe.register_adaptive_mask_model(plugin)
    pipe.register_adaptive_mask_settings(inp["settings"])
    seeds = seeds if seeds is not None else [100 + b for b in range(B)]
    gens = [torch.Generator(device=dev).manual_seed(s) for s in seeds]
    trace = []

    def on_trace(d):
        t = d["t"]
        use_default = (t < 1000 * inp["ratio"]) if inp["ratio"] > 0 else False
        img = d["image_u8"]
        trace.append(dict(i=d["i"], t=t, use_default=use_default, x0=d["x0"].clone().reshape(B, inp["HW"] // 8, inp["HW"] // 8, 4).permute(0, 3, 1, 2).cpu(),
                          image_u8=(img.cpu().numpy() if isinstance(img, torch.Tensor) else np.array(img)), seg=d["seg"].cpu().numpy(),
                          mask=d["mask"].cpu().numpy(), mask_lat=d["mask_lat"].float().cpu().numpy(), area=d["area"].cpu().numpy(),
                          masked_lat=d["masked_lat"].float().reshape(B, inp["HW"] // 8, inp["HW"] // 8, 4).permute(0, 3, 1, 2).cpu()))

    pipe._noise_log, pipe._trace = [], on_trace
    try:
        out = pipe(image=inp["image"], default_mask_image=inp["mask"], prompt_embeds=inp["pe"], negative_prompt_embeds=inp["ne"],
                   num_inference_steps=steps, strength=strength, guidance_scale=inp["guidance"], generator=gens, latents=inp["lat0"],
                   output_type="latent", use_adaptive_mask=use_adaptive_mask, enforce_full_mask_ratio=inp["ratio"],
                   human_detection_thres=inp["thres"]).images
        L = inp["HW"] // 8
        noises = [n.reshape(B, L, L, 4).permute(0, 3, 1, 2).cpu() for n in pipe._noise_log]
    finally:
        pipe._noise_log = pipe._trace = None
    return dict(latents=out.float().cpu(), trace=trace, noises=noises, last_mask=pipe.last_mask_image_np)


def run_ref(inp, noises, plugin, *, strength, device="cpu", use_adaptive_mask=True, steps=50, seed=0, dtype_flow="fp32"):
    plugin = type(plugin)()
    plugin.accepts_device_tensor = False
    ref = so.AdaptiveLoopRef(weights.random_state(weights.unet_shapes(), seed=seed), weights.random_state(weights.vae_shapes(), seed=seed + 1),
                             weights.UNET_CFG, weights.VAE_CFG, image=inp["image"], default_mask=inp["mask"], ctx_uncond=inp["ne"],
                             ctx_cond=inp["pe"], lat0=inp["lat0"], plugin=plugin, settings=inp["settings"], num_inference_steps=steps,
                             strength=strength, guidance=inp["guidance"], enforce_full_mask_ratio=inp["ratio"],
                             human_detection_thres=inp["thres"], use_adaptive_mask=use_adaptive_mask, device=device, dtype_flow=dtype_flow)
    trace = []
    with torch.no_grad():
        lat = ref.run(noises, on_adapt=lambda d: trace.append(dict(i=d["i"], t=d["t"], x0=d["x0"].float().cpu(), image_u8=d["image_u8"],
                                                                  seg=d["seg"], mask=d["mask"].astype(np.uint8),
                                                                  masked_lat=d["masked_lat"].float().cpu())))
    return dict(latents=lat.float().cpu(), trace=trace)
