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
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)      # flip_sin_to_cos=True, shift 0


# ------------------------------------------------------------------ DDIM (closed form)
def ddim_alphas(num_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train, dtype=torch.float64) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps=50, num_train=1000, steps_offset=1):
    ratio = num_train // num_inference_steps
    return [int(i * ratio) + steps_offset for i in range(num_inference_steps)][::-1]


def r16(x):
    """Round to fp16 and come back: one fp16 tensor op of the reference's half-precision pipeline (torch computes an fp16 elementwise
    op in fp32 and rounds the result once, python / 0-dim fp32 scalars stay fp32)."""
    return x.to(torch.float16).to(torch.float32)


def ddim_step_ref(eps, t, x, alphas, num_inference_steps=50, num_train=1000, dtype_flow="fp32"):
    """DDIMScheduler.step with eta = 0 (:1015-1017).  dtype_flow "fp32": the closed form in f64.  dtype_flow "fp16": op by op the way
    diffusers 0.20.2 evaluates it on the fp16 tensors of the reference's pipeline (src/generation/inpaint.py:64-70 loads it with
    torch_dtype=float16): `pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5`,
    `pred_sample_direction = (1 - alpha_prod_t_prev) ** 0.5 * model_output`,
    `prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction` -- the alphas are fp32 scalars, every
    tensor result is rounded to fp16."""
    prev_t = t - num_train // num_inference_steps
    a_t = alphas[t]
    a_p = alphas[prev_t] if prev_t >= 0 else alphas[0]           # set_alpha_to_one=False
    if dtype_flow == "fp16":
        a_t32, a_p32 = a_t.float(), a_p.float()                  # alphas_cumprod is an fp32 tensor in the scheduler
        e, xs = r16(eps.float()), r16(x.float())
        x0 = r16(r16(xs - r16((1 - a_t32) ** 0.5 * e)) / a_t32 ** 0.5)
        direction = r16((1 - a_p32) ** 0.5 * e)
        prev = r16(r16(a_p32 ** 0.5 * x0) + direction)
        return prev, x0
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


# ------------------------------------------------------------------ the adaptive-mask branch of the loop (BASELINE config 3)
# fp32 restatement of utils/adaptive_mask_inpainting.py:988-1076 (loop), :1111-1115 (decode_to_npuint8_image),
# :1123-1157 (adapt_mask), :131-245 (prepare_mask_and_masked_image, tensor branch), :675-694 (VAE encode + mask
# interpolation).  The reference handles ONE image per call when the mask adapts (squeeze() at :1114); `AdaptiveLoopRef`
# carries B independent images side by side so that a batch-8 run of the HIP loop can be checked image by image.
# cv2 is absent from this image: cv2.dilate(mask, ones((3,3)), iterations=k) (zero border, k = 0 copies) is restated
# with scipy.ndimage.grey_dilation, k times.  "parity unpinned" as the rest of this file (diffusers absent).
import numpy as np


def dilate_ref(mask_u8, iterations):
    from scipy.ndimage import grey_dilation
    out = np.asarray(mask_u8).astype(np.uint8)
    for _ in range(int(iterations)):
        out = grey_dilation(out, size=(3, 3), mode="constant", cval=0)
    return out


def vae_sample_ref(moments, noise, scaling):
    """DiagonalGaussianDistribution.sample(generator) * scaling_factor (:675-684) with the noise given."""
    mean, logvar = moments.float().chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return (mean + std * noise.to(mean)) * scaling


def decode_to_npuint8_ref(vstate, vcfg, latents):
    """:1111-1115 -- decode(latents / scaling_factor), (x / 2 + 0.5).clamp(0, 1), HWC, * 255, TRUNCATING cast. -> u8 [B,H,W,3]"""
    img = vae_decode_ref(vstate, latents.float() / vcfg["scaling_factor"], vcfg)
    img = (img / 2 + 0.5).clamp(0, 1)
    return (img.permute(0, 2, 3, 1).detach().cpu().numpy() * 255).astype(np.uint8)


def adapt_mask_ref(seg, default_mask_np, dilate_num, use_default_mask, human_detection_thres):
    """:1130-1141 up to the tensor mask: seg = plug-in output [H,W]; -> binary float32 [H,W]."""
    seg = np.asarray(seg)
    if use_default_mask or seg.sum() < 512 * 512 * human_detection_thres:
        mask = default_mask_np
    else:
        mask = np.logical_and(dilate_ref(seg, dilate_num), default_mask_np)
    return (np.asarray(mask, dtype=np.float32) >= 0.5).astype(np.float32)


class AdaptiveLoopRef:
    def __init__(self, ustate, vstate, ucfg, vcfg, *, image, default_mask, ctx_uncond, ctx_cond, lat0, plugin, settings,
                 num_inference_steps=50, strength=1.0, guidance=7.5, enforce_full_mask_ratio=0.5, human_detection_thres=0.008,
                 use_adaptive_mask=True, device="cpu", dtype_flow="fp32"):
        """dtype_flow "fp32": everything outside the networks in fp32 / f64 (what the product does: fp32 latents).  "fp16": the
        tensors the reference's fp16 pipeline holds in half precision are rounded where it rounds them -- latents, the UNet's
        output, the CFG combination op by op (:1010-1012), the scheduler step (ddim_step_ref), the VAE moments / posterior sample
        and the decoder input.  The networks themselves stay fp32 restatements in both flows."""
        assert dtype_flow in ("fp32", "fp16")
        self.flow = dtype_flow
        dev = torch.device(device)
        self.dev = dev
        self.us = {k: v.to(dev, torch.float32) for k, v in ustate.items()}
        self.vs = {k: v.to(dev, torch.float32) for k, v in vstate.items()}
        self.ucfg, self.vcfg = ucfg, vcfg
        self.image = image.float().to(dev)                                     # [B,3,H,W] in [-1,1]
        self.default_np = (default_mask.float().cpu().numpy()[:, 0] >= 0.5).astype(np.float32)     # [B,H,W]
        self.ctx = torch.cat([ctx_uncond, ctx_cond]).float().to(dev)
        self.B = image.shape[0]
        self.plugin, self.settings = plugin, settings
        self.N, self.guidance = num_inference_steps, guidance
        self.ratio, self.thres, self.adaptive = enforce_full_mask_ratio, human_detection_thres, use_adaptive_mask
        self.alphas = ddim_alphas()
        ts = ddim_timesteps(num_inference_steps)
        init = min(int(num_inference_steps * strength), num_inference_steps)
        self.timesteps = ts[max(num_inference_steps - init, 0):]               # get_timesteps (:622-628)
        self.lat = lat0.double().to(dev)                                       # caller-provided `latents` (:662-664), sigma = 1
        if self.flow == "fp16":
            self.lat = r16(self.lat.float())
        self.mask_np = self.default_np.copy()
        self.mask_lat = self.masked_lat = None

    # -- :686-694 + :675-684
    def set_mask(self, mask_np, noise):
        """mask_np [B,H,W] in {0,1} -> mask latents (nearest, source pixel (8y, 8x)) + masked-image latents."""
        m = torch.from_numpy(np.ascontiguousarray(mask_np)).to(self.dev)[:, None]
        masked = self.image * (m < 0.5)
        self.mask_np = mask_np
        self.mask_lat = F.interpolate(m, size=(m.shape[-2] // 8, m.shape[-1] // 8))
        mom = vae_encode_ref(self.vs, masked, self.vcfg)
        if self.flow == "fp16":          # moments, std, std * noise, mean + ., . * scaling_factor: five fp16 tensor results (:675-684)
            mean, logvar = r16(mom.float()).chunk(2, dim=1)
            std = r16(torch.exp(r16(0.5 * logvar.clamp(-30.0, 20.0))))
            self.masked_lat = r16(r16(mean + r16(std * r16(noise.to(self.dev).float()))) * self.vcfg["scaling_factor"])
        else:
            self.masked_lat = vae_sample_ref(mom, noise.to(self.dev), self.vcfg["scaling_factor"])
        return self.masked_lat

    def unet_eps(self, t):
        B = self.B
        inp = torch.cat([self.lat.float(), self.mask_lat, self.masked_lat], dim=1)
        eps = unet_ref(self.us, torch.cat([inp, inp]), torch.full((2 * B,), float(t), device=self.dev), self.ctx, self.ucfg)
        if self.flow == "fp16":          # noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond) on fp16 tensors
            eps = r16(eps)
            return r16(eps[:B] + r16(self.guidance * r16(eps[B:] - eps[:B])))
        return eps[:B] + self.guidance * (eps[B:] - eps[:B])

    def segment(self, x0):
        """decode x0 and run the plug-in per image (its NumPy contract) -> images u8 [B,H,W,3], segs u8 [B,H,W]."""
        imgs = decode_to_npuint8_ref(self.vs, self.vcfg, r16(r16(x0.float()) / self.vcfg["scaling_factor"]) * self.vcfg["scaling_factor"]
                                     if self.flow == "fp16" else x0)
        segs = np.stack([np.asarray(self.plugin(imgs[b])["mask"]).astype(np.uint8) for b in range(self.B)])
        return imgs, segs

    def adapt(self, i, t, segs):
        if self.ratio > 0.0:
            use_default = t < 1000 * self.ratio
        elif self.ratio == 0.0:
            use_default = False
        else:
            raise NotImplementedError
        k = self.settings.dilate_scheduler(i)
        return np.stack([adapt_mask_ref(segs[b], self.default_np[b], k, use_default, self.thres) for b in range(self.B)])

    def run(self, noises, on_adapt=None):
        """noises: iterable of [B,4,h,w] draws, consumed one per VAE sample in the reference's order.  -> final latents."""
        it = iter(noises)
        self.set_mask(self.default_np, next(it))
        for i, t in enumerate(self.timesteps):
            e = self.unet_eps(t)
            self.lat, x0 = ddim_step_ref(e, t, self.lat, self.alphas.to(self.dev), num_inference_steps=self.N, dtype_flow=self.flow)
            if self.adaptive and self.settings.provoke_scheduler(i):
                imgs, segs = self.segment(x0)
                mask = self.adapt(i, t, segs)
                self.set_mask(mask, next(it))
                if on_adapt is not None:
                    on_adapt(dict(i=i, t=t, x0=x0, image_u8=imgs, seg=segs, mask=mask, masked_lat=self.masked_lat, lat=self.lat))
        return self.lat
