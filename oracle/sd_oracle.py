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


# ------------------------------------------------------------------ UNet2DConditionModel (fp32, NCHW, diffusers semantics)
def _gn(x, s, p, eps, groups=32):
    return F.group_norm(x, groups, s[p + ".weight"].float(), s[p + ".bias"].float(), eps)


def _conv(x, s, p, stride=1, padding=1):
    return F.conv2d(x, s[p + ".weight"].float(), s[p + ".bias"].float(), stride=stride, padding=padding)


def _lin(x, s, p, bias=True):
    return F.linear(x, s[p + ".weight"].float(), s[p + ".bias"].float() if bias else None)


def resnet_ref(x, semb, s, p, eps=1e-5):
    h = _conv(F.silu(_gn(x, s, p + ".norm1", eps)), s, p + ".conv1")
    if semb is not None:
        h = h + _lin(semb, s, p + ".time_emb_proj")[:, :, None, None]
    h = _conv(F.silu(_gn(h, s, p + ".norm2", eps)), s, p + ".conv2")
    if p + ".conv_shortcut.weight" in s:
        x = _conv(x, s, p + ".conv_shortcut", padding=0)
    return x + h


def _attn(xq, xkv, s, p, heads):
    q = _lin(xq, s, p + ".to_q", bias=False)
    k = _lin(xkv, s, p + ".to_k", bias=False)
    v = _lin(xkv, s, p + ".to_v", bias=False)
    d = q.shape[-1] // heads
    return _lin(attention_ref(q, k, v, heads, d**-0.5), s, p + ".to_out.0")


def transformer_ref(x, ctx, s, p, heads):
    B, C, H, W = x.shape
    h = _conv(_gn(x, s, p + ".norm", 1e-6), s, p + ".proj_in", padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = p + ".transformer_blocks.0"
    ln = lambda z, n: F.layer_norm(z, (C,), s[f"{t}.{n}.weight"].float(), s[f"{t}.{n}.bias"].float(), 1e-5)
    n1 = ln(h, "norm1")
    h = _attn(n1, n1, s, t + ".attn1", heads) + h
    h = _attn(ln(h, "norm2"), ctx, s, t + ".attn2", heads) + h
    y = _lin(ln(h, "norm3"), s, t + ".ff.net.0.proj")
    a, gate = y.chunk(2, dim=-1)
    h = _lin(a * F.gelu(gate), s, t + ".ff.net.2") + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _conv(h, s, p + ".proj_out", padding=0) + x


def unet_ref(state, sample, timesteps, ctx, cfg):
    """sample [B,9,H,W], timesteps [B], ctx [B,77,768] -> noise prediction [B,4,H,W]; everything fp32."""
    s = state
    ch = cfg["block_out_channels"]
    heads = cfg["heads"]
    x = sample.float()
    ctx = ctx.float()
    emb = timestep_embedding_ref(timesteps, ch[0])
    emb = _lin(F.silu(_lin(emb, s, "time_embedding.linear_1")), s, "time_embedding.linear_2")
    semb = F.silu(emb)
    h = _conv(x, s, "conv_in")
    skips = [h]
    for i in range(len(ch)):
        for j in range(cfg["layers_per_block"]):
            h = resnet_ref(h, semb, s, f"down_blocks.{i}.resnets.{j}")
            if cfg["down_has_attn"][i]:
                h = transformer_ref(h, ctx, s, f"down_blocks.{i}.attentions.{j}", heads)
            skips.append(h)
        if i < len(ch) - 1:
            h = _conv(h, s, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
            skips.append(h)
    h = resnet_ref(h, semb, s, "mid_block.resnets.0")
    h = transformer_ref(h, ctx, s, "mid_block.attentions.0", heads)
    h = resnet_ref(h, semb, s, "mid_block.resnets.1")
    for i in range(len(ch)):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet_ref(torch.cat([h, skips.pop()], dim=1), semb, s, f"up_blocks.{i}.resnets.{j}")
            if cfg["up_has_attn"][i]:
                h = transformer_ref(h, ctx, s, f"up_blocks.{i}.attentions.{j}", heads)
        if i < len(ch) - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), s, f"up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(h, s, "conv_norm_out", 1e-5)), s, "conv_out")


# ------------------------------------------------------------------ AutoencoderKL (fp32, NCHW, diffusers semantics)
def _vae_attn_ref(x, s, p):
    B, C, H, W = x.shape
    h = _gn(x, s, p + ".group_norm", 1e-6).reshape(B, C, H * W).transpose(1, 2)
    q, k, v = _lin(h, s, p + ".to_q"), _lin(h, s, p + ".to_k"), _lin(h, s, p + ".to_v")
    a = torch.softmax(q @ k.transpose(1, 2) * C**-0.5, dim=-1) @ v
    a = _lin(a, s, p + ".to_out.0").transpose(1, 2).reshape(B, C, H, W)
    return a + x


def vae_decode_ref(state, z, cfg):
    s, ch = state, cfg["block_out_channels"]
    x = _conv(z.float(), s, "post_quant_conv", padding=0)
    x = _conv(x, s, "decoder.conv_in")
    x = resnet_ref(x, None, s, "decoder.mid_block.resnets.0", eps=1e-6)
    x = _vae_attn_ref(x, s, "decoder.mid_block.attentions.0")
    x = resnet_ref(x, None, s, "decoder.mid_block.resnets.1", eps=1e-6)
    for i in range(len(ch)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet_ref(x, None, s, f"decoder.up_blocks.{i}.resnets.{j}", eps=1e-6)
        if i < len(ch) - 1:
            x = _conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), s, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(x, s, "decoder.conv_norm_out", 1e-6)), s, "decoder.conv_out")


def vae_encode_ref(state, img, cfg):
    """-> moments [B, 8, H/8, W/8] = (mean | logvar)."""
    s, ch = state, cfg["block_out_channels"]
    x = _conv(img.float(), s, "encoder.conv_in")
    for i in range(len(ch)):
        for j in range(cfg["layers_per_block"]):
            x = resnet_ref(x, None, s, f"encoder.down_blocks.{i}.resnets.{j}", eps=1e-6)
        if i < len(ch) - 1:
            x = _conv(F.pad(x, (0, 1, 0, 1)), s, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    x = resnet_ref(x, None, s, "encoder.mid_block.resnets.0", eps=1e-6)
    x = _vae_attn_ref(x, s, "encoder.mid_block.attentions.0")
    x = resnet_ref(x, None, s, "encoder.mid_block.resnets.1", eps=1e-6)
    x = _conv(F.silu(_gn(x, s, "encoder.conv_norm_out", 1e-6)), s, "encoder.conv_out")
    return _conv(x, s, "quant_conv", padding=0)
