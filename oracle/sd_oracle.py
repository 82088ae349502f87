"""CPU reference for the diffusion operators  --  TEST INFRASTRUCTURE ONLY (see oracle/coma_oracle.py header).

"parity unpinned": the arithmetic of this half of the path lives in third-party packages that are absent from
/root/reference and from this image (diffusers==0.20.2 UNet2DConditionModel / AutoencoderKL / DDIMScheduler,
INSTALL.md:31; call sites utils/adaptive_mask_inpainting.py:1001-1017, 1086, 1112, 680), and no weights can be
downloaded.  The substitute oracle is plain torch fp32 functional ops (F.conv2d, F.group_norm, F.layer_norm,
softmax(QK^T)V, exact GELU) on the same seeded random weights, composed into the published SD-1.5-inpainting
architecture, plus the closed-form DDIM update.  Tolerances are stated in the tests.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def nhwc_to_nchw(x, b, h, w):
    return x.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def conv_ref(x_nhwc, w, *, batch, h, w_, taps=1, stride=1, upsample=False, bias=None, bias_bn=None, res=None, silu=False):
    """x_nhwc: [B*H*W, Cin] fp32 (already concatenated); w: [Cout, taps, Cin] -> [B*Ho*Wo, Cout] fp32."""
    cin = x_nhwc.shape[-1]
    x = nhwc_to_nchw(x_nhwc.float(), batch, h, w_)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    k = 3 if taps == 9 else 1
    wt = w.float().reshape(w.shape[0], k, k, cin).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(x, wt, None, stride=stride, padding=1 if taps == 9 else 0)
    y = nchw_to_nhwc(y).reshape(-1, w.shape[0])
    rows_per_batch = y.shape[0] // batch
    if bias is not None:
        y = y + bias.float()[None]
    if bias_bn is not None:
        y = y + bias_bn.float().repeat_interleave(rows_per_batch, dim=0)
    if silu:
        y = F.silu(y)
    if res is not None:
        y = y + res.float()
    return y


def geglu_ref(x, w, b):
    """diffusers GEGLU: proj -> chunk(2) -> hidden * gelu(gate) (exact erf GELU)."""
    y = x.float() @ w.float().t() + b.float()[None]
    h, g = y.chunk(2, dim=-1)
    return h * F.gelu(g)


def groupnorm_ref(x_nhwc, gamma, beta, *, batch, hw, groups=32, eps=1e-5, silu=True):
    c = x_nhwc.shape[-1]
    x = x_nhwc.float().reshape(batch, hw, c).permute(0, 2, 1)
    y = F.group_norm(x, groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    return y.permute(0, 2, 1).reshape(batch * hw, c)


def attention_ref(q, k, v, heads, scale):
    """q [B,Lq,H*d], k/v [B,Lk,H*d] fp32 -> [B,Lq,H*d]."""
    B, Lq, C = q.shape
    d = C // heads
    qh = q.float().reshape(B, Lq, heads, d).transpose(1, 2)
    kh = k.float().reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.float().reshape(B, -1, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, C)


def timestep_embedding_ref(t, dim):
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)      # flip_sin_to_cos=True, shift 0


# ------------------------------------------------------------------ DDIM (closed form)
def ddim_alphas(num_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train, dtype=torch.float64) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps=50, num_train=1000, steps_offset=1):
    ratio = num_train // num_inference_steps
    return [int(i * ratio) + steps_offset for i in range(num_inference_steps)][::-1]


def ddim_step_ref(eps, t, x, alphas, num_inference_steps=50, num_train=1000):
    prev_t = t - num_train // num_inference_steps
    a_t = alphas[t]
    a_p = alphas[prev_t] if prev_t >= 0 else alphas[0]           # set_alpha_to_one=False
    x0 = (x.double() - (1 - a_t).sqrt() * eps.double()) / a_t.sqrt()
    prev = a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps.double()
    return prev, x0
