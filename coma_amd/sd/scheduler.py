"""DDIM scheduler (eta = 0) with the constants the reference builds it with
(src/generation/inpaint.py:54-59: beta 0.00085 -> 0.012 "scaled_linear", clip_sample=False, set_alpha_to_one=False;
steps_offset forced to 1 by the pipeline constructor, utils/adaptive_mask_inpainting.py:295-307).  Restates the
parts of diffusers' DDIMScheduler the pipeline touches (set_timesteps, timesteps, order, scale_model_input, step,
add_noise, init_noise_sigma, config.num_train_timesteps) -- diffusers itself is third party and absent here.
The arithmetic of ``step`` / ``add_noise`` runs in the HIP kernels sd_cfg_ddim_step / sd_add_noise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", num_train_timesteps=1000, clip_sample=True,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", timestep_spacing="leading"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if clip_sample or prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only the configuration the reference uses is implemented")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset, beta_start=beta_start,
                           beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                           set_alpha_to_one=set_alpha_to_one)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)          # kept on the host: they only parameterise kernel launches

    def scale_model_input(self, sample, timestep=None):
        return sample

    def alphas_for(self, timestep):
        """(alpha_t, alpha_prev) as Python floats for one step."""
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True, **kw):
        """model_output / sample: NCHW [B,4,h,w] on the HIP device -> prev_sample, pred_original_sample (fp32 NCHW)."""
        assert eta == 0.0, "eta != 0 is not on the reference's path"
        B, C, h, w = sample.shape
        dev = sample.device
        a_t, a_p = self.alphas_for(timestep)
        eps = model_output.to(torch.float32).permute(0, 2, 3, 1).reshape(B, h * w, C)
        eps16 = torch.zeros(2 * B, h * w, 8, dtype=torch.float16, device=dev)
        eps16[:B, :, :C] = eps
        eps16[B:, :, :C] = eps
        lat = sample.to(torch.float32).permute(0, 2, 3, 1).reshape(B, h * w, C).contiguous()
        x0 = torch.empty_like(lat)
        ops.cfg_ddim_step(eps16, 8, lat, x0, None, None, None, batch=B, hw=h * w, guidance=1.0, alpha_t=a_t, alpha_prev=a_p)
        back = lambda z: z.reshape(B, h, w, C).permute(0, 3, 1, 2).contiguous()
        out = _Cfg(prev_sample=back(lat), pred_original_sample=back(x0))
        return out if return_dict else (out.prev_sample,)

    def add_noise(self, original_samples, noise, timesteps):
        t = int(torch.as_tensor(timesteps).reshape(-1)[0])
        x0 = original_samples.to(torch.float32).contiguous()
        out = torch.empty_like(x0)
        ops.add_noise(x0, noise.to(torch.float32).contiguous(), float(self.alphas_cumprod[t]), out)
        return out
