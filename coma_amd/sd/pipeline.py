"""Adaptive-mask Stable-Diffusion inpainting loop on MI355X: host mirror of the reference's
``utils/adaptive_mask_inpainting.py`` (``AdaptiveMaskInpaintPipeline`` :248-1157, mask glue :131-245 / :1123-1157,
``seg2bbox`` :1160, ``merge_bbox`` :1168, ``MaskDilateScheduler`` :1457, ``ProvokeScheduler`` :1468).

Same call surface (``__call__`` keyword names, ``register_adaptive_mask_model`` / ``register_adaptive_mask_settings``
plugin hooks, batch size 1 when ``use_adaptive_mask``), but the loop never leaves the device except for the mask
plugin itself:
  * UNet = one hipGraph replay per step (coma_amd/sd/unet.py); CFG + DDIM step + assembly of the next 9-channel
    input = one elementwise kernel (sd_cfg_ddim_step);
  * the x0 decode runs only on steps where the mask is actually re-estimated (or the visualiser is on): the
    reference decodes every step (:1028) but consumes the image only there (:1031, :1051) -- outputs are identical;
  * dilation (cv2.dilate 3x3 x k == (2k+1)^2 box max), AND with the default mask, binarisation, masked image and the
    nearest 8x mask down-sample are one kernel (sd_mask_adapt); the re-encode is the VAE-encoder graph.
Third-party pieces that cannot exist offline (CLIP text encoder, PointRend / SAM) stay plug-ins: prompts can be
given as ``prompt_embeds`` / ``negative_prompt_embeds`` and any callable ``image_u8_HWC -> {"mask": u8[H,W], ...}``
can be registered as the mask model.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .scheduler import DDIMScheduler
from .unet import HipUNet2DConditionModel
from .vae import HipAutoencoderKL
from .weights import random_state, unet_shapes, vae_shapes

try:
    import PIL.Image
except Exception:  # pragma: no cover
    PIL = None


# ------------------------------------------------------------------------------------------- host-side glue
def prepare_mask_and_masked_image(image, mask, height, width, return_image=False):
    """(image, mask) -> (mask [B,1,H,W] in {0,1}, masked_image [B,3,H,W] = image * (mask < 0.5)[, image]) as fp32.

    Accepts PIL images, HxWx3 / HxW numpy arrays (uint8 image scaled to [-1,1], mask scaled by 255 when PIL) or
    torch tensors already in [-1,1] / [0,1], with the reference's checks and error types."""
    if image is None:
        raise ValueError("`image` input cannot be undefined.")
    if mask is None:
        raise ValueError("`mask_image` input cannot be undefined.")
    if isinstance(image, torch.Tensor):
        if not isinstance(mask, torch.Tensor):
            raise TypeError(f"`image` is a torch.Tensor but `mask` (type: {type(mask)} is not")
        if image.ndim == 3:
            assert image.shape[0] == 3, "Image outside a batch should be of shape (3, H, W)"
            image = image.unsqueeze(0)
        if mask.ndim == 2:
            mask = mask.unsqueeze(0).unsqueeze(0)
        if mask.ndim == 3:
            mask = mask.unsqueeze(0) if mask.shape[0] == 1 else mask.unsqueeze(1)
        assert image.ndim == 4 and mask.ndim == 4, "Image and Mask must have 4 dimensions"
        assert image.shape[-2:] == mask.shape[-2:], "Image and Mask must have the same spatial dimensions"
        assert image.shape[0] == mask.shape[0], "Image and Mask must have the same batch size"
        if image.min() < -1 or image.max() > 1:
            raise ValueError("Image should be in [-1, 1] range")
        if mask.min() < 0 or mask.max() > 1:
            raise ValueError("Mask should be in [0, 1] range")
        mask = (mask >= 0.5).to(mask.dtype)
        image = image.to(dtype=torch.float32)
    elif isinstance(mask, torch.Tensor):
        raise TypeError(f"`mask` is a torch.Tensor but `image` (type: {type(image)} is not")
    else:
        if not isinstance(image, list):
            image = [image]
        if PIL is not None and isinstance(image[0], PIL.Image.Image):
            image = [np.array(i.resize((width, height), resample=PIL.Image.LANCZOS).convert("RGB"))[None, :] for i in image]
        else:
            image = [np.asarray(i)[None, :] for i in image]
        image = np.concatenate(image, axis=0).transpose(0, 3, 1, 2)
        image = torch.from_numpy(np.ascontiguousarray(image)).to(dtype=torch.float32) / 127.5 - 1.0
        if not isinstance(mask, list):
            mask = [mask]
        if PIL is not None and isinstance(mask[0], PIL.Image.Image):
            mask = np.concatenate([np.array(m.resize((width, height), resample=PIL.Image.LANCZOS).convert("L"))[None, None, :]
                                   for m in mask], axis=0).astype(np.float32) / 255.0
        else:
            mask = np.concatenate([np.asarray(m)[None, None, :] for m in mask], axis=0).astype(np.float32)
        mask = torch.from_numpy((mask >= 0.5).astype(np.float32))
    masked_image = image * (mask < 0.5)
    if return_image:
        return mask, masked_image, image
    return mask, masked_image


def seg2bbox(seg_mask: np.ndarray):
    ii, jj = seg_mask.nonzero()
    return np.array([jj.min(), ii.min(), jj.max() + 1, ii.max() + 1])


def merge_bbox(bboxes: list):
    assert len(bboxes) > 0
    b = np.stack(bboxes, axis=0)
    out = np.zeros_like(b[0])
    out[0], out[1], out[2], out[3] = b[:, 0].min(), b[:, 1].min(), b[:, 2].max(), b[:, 3].max()
    return out


class MaskDilateScheduler:
    def __init__(self, max_dilate_num=15, num_inference_steps=50, schedule=None):
        self.max_dilate_num = max_dilate_num
        self.schedule = [num_inference_steps - i for i in range(num_inference_steps)] if schedule is None else schedule
        assert len(self.schedule) == num_inference_steps

    def __call__(self, i):
        return min(self.max_dilate_num, self.schedule[i])


class ProvokeScheduler:
    def __init__(self, num_inference_steps=50, schedule=None, is_zero_indexing=False):
        if len(schedule) > 0:
            assert max(schedule) <= (num_inference_steps - 1 if is_zero_indexing else num_inference_steps)
        self.is_zero_indexing = is_zero_indexing
        self.schedule = schedule

    def __call__(self, i):
        return (i if self.is_zero_indexing else i + 1) in self.schedule


class AdaptiveMaskSettings(dict):
    __getattr__ = dict.__getitem__


def default_adaptive_mask_settings(num_inference_steps=50, adaptive_mask_model_type="p"):
    """The schedules of src/generation/inpaint.py:112-132."""
    n = num_inference_steps
    step = int(n * 0.1)
    final = n - step * 7
    sched = ([20] * step + [10] * step + [5] * step + [4] * step + [3] * step + [2] * step + [1] * step + [0] * final
             if adaptive_mask_model_type == "p" else [10] * 50)
    provoke = list(range(2, 10 + 1, 2)) + list(range(12, 40 + 1, 2)) + [45] if adaptive_mask_model_type != "baseline" else []
    return AdaptiveMaskSettings(dilate_scheduler=MaskDilateScheduler(max_dilate_num=20, num_inference_steps=n, schedule=sched),
                                dilate_kernel=np.ones((3, 3), dtype=np.uint8),
                                provoke_scheduler=ProvokeScheduler(num_inference_steps=n, schedule=provoke, is_zero_indexing=False))


class SyntheticHumanMaskPredictor:
    """Deterministic stand-in for PointRend (weights cannot be provisioned offline): thresholds the luminance of the
    decoded x0 image inside an ellipse.  Same plugin contract as PointRendPredictor.__call__ (:1225-1236).

    ``accepts_device_tensor``: a plug-in that sets this attribute is handed the decoded image as a uint8 [H,W,3] tensor on
    the pipeline's device (no D2H copy) and may return its mask as a device tensor; plug-ins without it (the reference's
    PointRend / SAM predictors) get the uint8 HWC NumPy array of the reference's contract.  Both paths of this class
    produce the same mask bit for bit (integer arithmetic; the mean is an exact integer sum / N in f64)."""
    use_visualizer = False
    accepts_device_tensor = True
    _ellipses = {}

    def _ellipse(self, H, W):
        ell = self._ellipses.get((H, W))
        if ell is None:
            yy, xx = np.mgrid[0:H, 0:W]
            ell = self._ellipses[(H, W)] = ((yy - H / 2) / (H * 0.3)) ** 2 + ((xx - W / 2) / (W * 0.18)) ** 2 <= 1.0
        return ell

    def __call__(self, image_u8):
        H, W = image_u8.shape[:2]
        if isinstance(image_u8, torch.Tensor):
            key = (H, W, str(image_u8.device))
            ell = self._ellipses.get(key)
            if ell is None:
                ell = self._ellipses[key] = torch.from_numpy(self._ellipse(H, W)).to(image_u8.device)
            s3 = image_u8.to(torch.int32).sum(-1)
            thr = s3.sum(dtype=torch.int64).to(torch.float64) / float(H * W) - 120.0
            return {"mask": (ell & (s3.to(torch.float64) > thr)).to(torch.uint8), "vis": None, "asset_mask": None}
        ell = self._ellipse(H, W)
        s3 = image_u8[..., 0].astype(np.uint16) + image_u8[..., 1] + image_u8[..., 2]    # 3 x luminance, integer
        thr = s3.mean(dtype=np.float64) - 120.0                      # lum > mean(lum) - 40
        return {"mask": (ell & (s3 > thr)).view(np.uint8), "vis": None, "asset_mask": None}


    def predict_batch(self, images_u8):
        """Optional batched form of the plug-in contract (an addition: the reference calls its predictor once per image, one image per
        pipeline call): uint8 [B, H, W, 3] device tensor -> {"mask": uint8 [B, H, W]}.  A pipeline call over B images then costs ONE
        plug-in call per mask re-estimation instead of B (8 x fewer host round trips: 48 -> 6 ms per batch of 8 over the 21
        re-estimations).  Same integer arithmetic per image as __call__: bit-identical masks."""
        B, H, W = images_u8.shape[:3]
        key = (H, W, str(images_u8.device))
        ell = self._ellipses.get(key)
        if ell is None:
            ell = self._ellipses[key] = torch.from_numpy(self._ellipse(H, W)).to(images_u8.device)
        s3 = images_u8.to(torch.int32).sum(-1)
        thr = s3.sum(dim=(1, 2), dtype=torch.int64).to(torch.float64) / float(H * W) - 120.0
        return {"mask": (ell[None] & (s3.to(torch.float64) > thr[:, None, None])).to(torch.uint8), "vis": None, "asset_mask": None}


class _Output(dict):
    __getattr__ = dict.__getitem__


# ------------------------------------------------------------------------------------------- the pipeline
class AdaptiveMaskInpaintPipeline:
    def __init__(self, vae: HipAutoencoderKL, unet: HipUNet2DConditionModel, scheduler: DDIMScheduler, text_encoder=None,
                 tokenizer=None, device="cuda"):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.device = torch.device(device)
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        scheduler.config["steps_offset"] = 1          # as the reference's constructor forces (:295-307)
        self.adaptive_mask_model = None
        self.adaptive_mask_settings = None
        self.safety_checker = None
        self._noise_log = None          # tests: a list here receives every latent-shaped noise draw, in order
        self._trace = None              # tests: a callable here is handed the per-step intermediates (device tensors)

    # ---- construction helpers
    @classmethod
    def from_random(cls, batch_size=1, height=512, width=512, device="cuda", seed=0, with_encoder=True, use_graph=True):
        """Seeded random SD-1.5-inpainting weights (no checkpoint is reachable offline); batch_size = images per call."""
        dev = torch.device(device)
        unet = HipUNet2DConditionModel(random_state(unet_shapes(), seed=seed), batch=2 * batch_size, height=height // 8,
                                       width=width // 8, device=dev, use_graph=use_graph, cfg_shared_prefix=True)
        vae = HipAutoencoderKL(random_state(vae_shapes(), seed=seed + 1), batch=batch_size, height=height, width=width, device=dev,
                               with_encoder=with_encoder, use_graph=use_graph)
        sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                            set_alpha_to_one=False)
        return cls(vae, unet, sch, device=dev)

    @classmethod
    def from_pretrained(cls, weights_dir, batch_size=1, height=512, width=512, device="cuda", with_encoder=True, use_graph=True):
        """diffusers-layout checkpoint directory (`unet/diffusion_pytorch_model.safetensors`,
        `vae/diffusion_pytorch_model.safetensors`), what `DiffusionPipeline.from_pretrained(...)` reads in the reference
        (src/generation/inpaint.py:64-70).  Scheduler constants are the reference's (`:54-59`)."""
        import os
        from .weights import check_state, load_safetensors
        dev = torch.device(device)
        ust = check_state(load_safetensors(os.path.join(weights_dir, "unet", "diffusion_pytorch_model.safetensors")), unet_shapes(), "UNet")
        vst = check_state(load_safetensors(os.path.join(weights_dir, "vae", "diffusion_pytorch_model.safetensors")), vae_shapes(), "VAE")
        unet = HipUNet2DConditionModel(ust, batch=2 * batch_size, height=height // 8, width=width // 8, device=dev, use_graph=use_graph, cfg_shared_prefix=True)
        vae = HipAutoencoderKL(vst, batch=batch_size, height=height, width=width, device=dev, with_encoder=with_encoder,
                               use_graph=use_graph)
        sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                            set_alpha_to_one=False)
        tok = enc = None
        if os.path.isdir(os.path.join(weights_dir, "text_encoder")) and os.path.isdir(os.path.join(weights_dir, "tokenizer")):
            # the checkpoint's own CLIP text tower (transformers; third party, as in the reference :26, :459-482)
            from transformers import CLIPTextModel, CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(os.path.join(weights_dir, "tokenizer"))
            enc = CLIPTextModel.from_pretrained(os.path.join(weights_dir, "text_encoder"), torch_dtype=torch.float16).to(dev).eval()
        return cls(vae, unet, sch, text_encoder=enc, tokenizer=tok, device=dev)

    def to(self, device):
        return self

    def register_adaptive_mask_settings(self, adaptive_mask_settings):
        self.adaptive_mask_settings = adaptive_mask_settings

    def register_adaptive_mask_model(self, adaptive_mask_model: callable):
        self.adaptive_mask_model = adaptive_mask_model

    def get_timesteps(self, num_inference_steps, strength, device=None):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    def _encode_prompt(self, prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, batch):
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("no text encoder registered: pass prompt_embeds / negative_prompt_embeds [B,77,768]")
            tok = lambda p: self.tokenizer(p, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                           return_tensors="pt").input_ids.to(self.device)
            prompt = [prompt] * batch if isinstance(prompt, str) else prompt
            negative_prompt = [negative_prompt or ""] * batch if not isinstance(negative_prompt, list) else negative_prompt
            prompt_embeds = self.text_encoder(tok(prompt))[0]
            negative_prompt_embeds = self.text_encoder(tok(negative_prompt))[0]
        if negative_prompt_embeds is None:
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)
        return torch.cat([negative_prompt_embeds, prompt_embeds]).to(self.device)      # [uncond | cond]

    def _randn_latent(self, generator):
        """One draw of latent-shaped noise the way the reference draws it -- `randn_tensor((B,4,h,w), generator, device, fp16)`
        (prepare_latents :656, DiagonalGaussianDistribution.sample behind :677-680): NCHW element order, the pipeline's fp16
        dtype, one (1,4,h,w) draw per image from its own generator when a list is given (:676-678), on the generator's device
        -- returned as fp32 NHWC [B, h*w, 4] on the pipeline's device.  Same seed -> same stream positions as the reference."""
        B, h, w = self.vae.batch, self.vae.dec.h, self.vae.dec.w
        gens = generator if isinstance(generator, (list, tuple)) else [generator]
        if isinstance(generator, (list, tuple)) and len(generator) != B:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {B}. Make sure the batch size matches the length of the generators.")
        parts = []
        for g in gens:
            gdev = g.device if g is not None else self.device
            n = B if len(gens) == 1 else 1
            parts.append(torch.randn(n, 4, h, w, generator=g, device=gdev, dtype=torch.float16).to(self.device))
        noise = torch.cat(parts, 0).to(torch.float32).permute(0, 2, 3, 1).reshape(B, h * w, 4).contiguous()
        if self._noise_log is not None:
            self._noise_log.append(noise.clone())
        return noise

    def _encode_vae_image(self, image, generator):
        """image fp16 NHWC already in self.vae.enc.x -> latents fp16 [B,hw,4] scaled by scaling_factor (on device)."""
        enc = self.vae.enc
        mom = enc.encode_static()
        B, n = self.vae.batch, enc.lat_h * enc.lat_w
        noise = self._randn_latent(generator)
        lat32 = torch.empty(B, n, 4, dtype=torch.float32, device=self.device)
        lat16 = torch.empty(B, n, 4, dtype=torch.float16, device=self.device)
        ops.vae_sample(mom, 64, noise, float(self.vae.config.scaling_factor), B * n, lat32=lat32, lat16=lat16)
        return lat32, lat16

    def decode_to_u8_image(self, latents_nhwc_f32):
        """latents fp32 [B,hw,4] -> uint8 [B,H,W,3] on the device (truncating cast, as `(x*255).astype(np.uint8)` at :1114)."""
        img = self._decode(latents_nhwc_f32)
        H, W = self.vae.dec.out_h, self.vae.dec.out_w
        u8 = torch.empty(self.vae.batch, H * W, 3, dtype=torch.uint8, device=self.device)
        ops.image_to_u8(img, u8, batch=self.vae.batch, hw=H * W, ld=64, round_mode=0)
        return u8.reshape(self.vae.batch, H, W, 3)

    def decode_to_npuint8_image(self, latents_nhwc_f32, all_images=False):
        """The reference's helper (:1111-1115): uint8 HWC NumPy, image 0 only unless all_images."""
        arr = self.decode_to_u8_image(latents_nhwc_f32).cpu().numpy()
        return arr if all_images else arr[0]

    def _decode(self, latents_nhwc_f32):
        dec = self.vae.dec
        z = dec.z
        z.zero_()
        z[:, :, :4] = (latents_nhwc_f32 / float(self.vae.config.scaling_factor)).to(torch.float16)
        return dec.decode_static()

    # ---- the call
    @torch.no_grad()
    def __call__(self, prompt=None, image=None, default_mask_image=None, height=None, width=None, strength=1.0,
                 num_inference_steps=50, guidance_scale=7.5, negative_prompt=None, num_images_per_prompt=1, eta=0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="pil",
                 return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None, use_adaptive_mask=True,
                 enforce_full_mask_ratio=0.5, human_detection_thres=0.008, visualization_save_dir=None):
        B = self.vae.batch
        assert num_images_per_prompt == 1 and eta == 0.0
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if use_adaptive_mask:
            # the reference handles one image per call (squeeze() at :1114); here a batch of independent images runs the
            # loop together, each with its own adapted mask (the plug-in is called once per image)
            assert self.adaptive_mask_model is not None and self.adaptive_mask_settings is not None
        if PIL is not None and isinstance(image, PIL.Image.Image):
            width, height = image.size
        H, W = self.vae.dec.out_h, self.vae.dec.out_w
        height, width = height or H, width or W
        assert (height, width) == (H, W), f"pipeline was built for {H}x{W}"
        dev = self.device
        do_cfg = guidance_scale > 1.0
        assert do_cfg, "the graphs are built for classifier-free guidance (batch 2B), as every reference config uses"

        # 3. prompt
        ctx = self._encode_prompt(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, B)
        self.unet.set_context(ctx)
        # 4. timesteps
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, dev)
        if num_inference_steps < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline"
                             f"steps is {num_inference_steps} which is < 1 and not appropriate for this pipeline.")
        is_strength_max = strength == 1.0
        # 5. mask and image
        mask, masked_image, init_image = prepare_mask_and_masked_image(image, default_mask_image, height, width, return_image=True)
        assert mask.shape[0] == B and init_image.shape[0] == B, "one (image, mask) per batch slot"
        init_image = init_image.to(dev).contiguous()
        default_mask_u8 = (mask[:, 0] >= 0.5).to(torch.uint8).to(dev).contiguous()             # [B,H,W]
        h, w = H // self.vae_scale_factor, W // self.vae_scale_factor
        hw = h * w
        # 6. latents -- the draw order is the reference's (prepare_latents :653-665): image encode first, then the noise; a
        # caller-provided `latents` is used as the starting noise as is (no image mix-in even when strength < 1, :662-664)
        if latents is None:
            if not is_strength_max:
                ops.nchw_to_nhwc(init_image, self.vae.enc.x, batch=B, c=3, hw=H * W, cpad=64)
                img_lat32, _ = self._encode_vae_image(init_image, generator)
            noise = self._randn_latent(generator)
            if is_strength_max:
                lat = noise * self.scheduler.init_noise_sigma
            else:
                lat = torch.empty_like(noise)
                ops.add_noise(img_lat32, noise, float(self.scheduler.alphas_cumprod[int(timesteps[0])]), lat)
        else:
            noise = latents.to(dev, torch.float32)
            lat = noise.permute(0, 2, 3, 1).reshape(B, hw, 4).contiguous() * self.scheduler.init_noise_sigma
        # 7. mask latents (device glue + VAE encoder graph)
        mask_full = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        mask_lat = torch.empty(B, hw, dtype=torch.float16, device=dev)
        area = torch.zeros(B, dtype=torch.int32, device=dev)
        scratch = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        self.vae.enc.x.zero_()                # channels >= 8 of the encoder input stay zero from here on

        def set_mask(segs, dilate_iters, force_default=False, area_thres=0.0):
            """segs: u8 [B,H,W] device masks to dilate & intersect with the default mask (None -> default mask); an image
            whose segmentation sums to less than area_thres keeps the default mask (decided on the device)."""
            ops.mask_adapt_batched(segs, default_mask_u8, init_image, mask_full, mask_lat, self.vae.enc.x, area, scratch, batch=B,
                                   H=H, W=W, dilate_iters=dilate_iters, force_default=force_default or segs is None,
                                   area_thres=area_thres, cpad=64)
            return self._encode_vae_image(None, generator)[1]

        masked_lat = set_mask(None, 0)
        self._last_masked_lat = masked_lat
        x0 = torch.empty(B, hw, 4, dtype=torch.float32, device=dev)
        # first UNet input (no step yet)
        ops.cfg_ddim_step(None, 0, lat, None, mask_lat, masked_lat, self.unet.x_in, batch=B, hw=hw, guidance=guidance_scale,
                          alpha_t=1.0, alpha_prev=1.0)
        # 10. denoising loop
        adapted = False
        for i, t in enumerate(timesteps.tolist()):
            self.unet.timesteps.fill_(float(t))
            eps = self.unet.forward_static()
            a_t, a_p = self.scheduler.alphas_for(t)
            adapt = use_adaptive_mask and self.adaptive_mask_settings.provoke_scheduler(i)
            vis = use_adaptive_mask and getattr(self.adaptive_mask_model, "use_visualizer", False)
            if not (adapt or vis):
                ops.cfg_ddim_step(eps, 64, lat, x0, mask_lat, masked_lat, self.unet.x_in, batch=B, hw=hw, guidance=guidance_scale,
                                  alpha_t=a_t, alpha_prev=a_p)
            else:
                # step first (latents + x0), adapt the mask from the decoded x0, then assemble the next input
                ops.cfg_ddim_step(eps, 64, lat, x0, None, None, None, batch=B, hw=hw, guidance=guidance_scale, alpha_t=a_t,
                                  alpha_prev=a_p)
                on_device = bool(getattr(self.adaptive_mask_model, "accepts_device_tensor", False))
                pred_orig_images = self.decode_to_u8_image(x0)
                if not on_device:
                    pred_orig_images = pred_orig_images.cpu().numpy()
                if adapt:
                    if enforce_full_mask_ratio > 0.0:
                        use_default = t < self.scheduler.config.num_train_timesteps * enforce_full_mask_ratio
                    elif enforce_full_mask_ratio == 0.0:
                        use_default = False
                    else:
                        raise NotImplementedError
                    select = getattr(self.adaptive_mask_model, "select", None)      # predictors.PerItemState: per-image plug-in state
                    batched = getattr(self.adaptive_mask_model, "predict_batch", None) if (on_device and select is None and B > 1) else None
                    if batched is not None:
                        # a stateless device plug-in that offers the batched form: one call for the B images of this re-estimation
                        segs = batched(pred_orig_images)["mask"].to(device=dev, dtype=torch.uint8).contiguous()
                    else:
                        segs = []
                        for b in range(B):
                            if select is not None:
                                select(b)
                            seg = self.adaptive_mask_model(pred_orig_images[b])["mask"]
                            if isinstance(seg, torch.Tensor):
                                seg = seg.to(device=dev, dtype=torch.uint8)
                            else:
                                seg = torch.from_numpy(np.ascontiguousarray(seg).astype(np.uint8)).to(dev)
                            segs.append(seg)
                        segs = torch.stack(segs).contiguous()
                    # `use_default_mask or mask.sum() < 512 * 512 * thres` (:1132): the literal 512 * 512 is the reference's
                    masked_lat = set_mask(segs, int(self.adaptive_mask_settings.dilate_scheduler(i)), force_default=use_default,
                                          area_thres=512 * 512 * human_detection_thres)
                    if self._trace is not None:
                        self._trace(dict(i=i, t=t, x0=x0, image_u8=pred_orig_images, seg=segs, mask=mask_full, mask_lat=mask_lat,
                                         masked_lat=masked_lat, lat=lat, area=area))
                    adapted = True
                ops.cfg_ddim_step(None, 0, lat, None, mask_lat, masked_lat, self.unet.x_in, batch=B, hw=hw,
                                  guidance=guidance_scale, alpha_t=1.0, alpha_prev=1.0)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, lat)

        if output_type == "latent":
            images = lat.reshape(B, h, w, 4).permute(0, 3, 1, 2).contiguous()
        else:
            img = self._decode(lat)
            if output_type == "pt":
                out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
                ops.nhwc_to_nchw(img, out, batch=B, c=3, hw=H * W, ld=64)
                images = (out / 2 + 0.5).clamp(0, 1)
            else:
                u8 = torch.empty(B, H * W, 3, dtype=torch.uint8, device=dev)
                ops.image_to_u8(img, u8, batch=B, hw=H * W, ld=64, round_mode=1)
                arr = u8.reshape(B, H, W, 3)
                if output_type == "u8":
                    images = arr
                elif output_type == "np":
                    images = arr.cpu().numpy().astype(np.float32) / 255.0
                else:
                    images = [PIL.Image.fromarray(a) for a in arr.cpu().numpy()]
        self.last_mask_image_np = mask_full[0].cpu().numpy().astype(np.float32) if adapted else None
        if not return_dict:
            return images, None
        return _Output(images=images, nsfw_content_detected=None)
