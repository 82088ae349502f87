"""Parameter tables of SD-1.5-inpainting (UNet2DConditionModel + AutoencoderKL) in diffusers' key naming, seeded
random initialisation (no checkpoint can be downloaded here), and the re-layouts the HIP kernels want.

Architecture source: the public SD-1.5 config (third party -- diffusers is not under the
reference tree).  Keys follow diffusers so that a real `unet/diffusion_pytorch_model.safetensors` loads unchanged
through :func:`load_safetensors`.
"""
from __future__ import annotations

import math

import torch

UNET_CFG = dict(in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                heads=8, cross_attention_dim=768, groups=32,
                down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True))
VAE_CFG = dict(latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32,
               scaling_factor=0.18215)


# ------------------------------------------------------------------ shape tables
def _resnet(p, cin, cout, temb=1280):
    s = {f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,), f"{p}.conv1.weight": (cout, cin, 3, 3),
         f"{p}.conv1.bias": (cout,), f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,),
         f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,)}
    if temb:
        s[f"{p}.time_emb_proj.weight"] = (cout, temb)
        s[f"{p}.time_emb_proj.bias"] = (cout,)
    if cin != cout:
        s[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[f"{p}.conv_shortcut.bias"] = (cout,)
    return s


def _transformer(p, c, ctx=768):
    t = f"{p}.transformer_blocks.0"
    s = {f"{p}.norm.weight": (c,), f"{p}.norm.bias": (c,), f"{p}.proj_in.weight": (c, c, 1, 1), f"{p}.proj_in.bias": (c,),
         f"{p}.proj_out.weight": (c, c, 1, 1), f"{p}.proj_out.bias": (c,)}
    for n in ("norm1", "norm2", "norm3"):
        s[f"{t}.{n}.weight"] = (c,)
        s[f"{t}.{n}.bias"] = (c,)
    for a, kdim in (("attn1", c), ("attn2", ctx)):
        s[f"{t}.{a}.to_q.weight"] = (c, c)
        s[f"{t}.{a}.to_k.weight"] = (c, kdim)
        s[f"{t}.{a}.to_v.weight"] = (c, kdim)
        s[f"{t}.{a}.to_out.0.weight"] = (c, c)
        s[f"{t}.{a}.to_out.0.bias"] = (c,)
    s[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
    s[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    s[f"{t}.ff.net.2.weight"] = (c, 4 * c)
    s[f"{t}.ff.net.2.bias"] = (c,)
    return s


def unet_shapes(cfg=UNET_CFG):
    ch = cfg["block_out_channels"]
    s = {"conv_in.weight": (ch[0], cfg["in_channels"], 3, 3), "conv_in.bias": (ch[0],),
         "time_embedding.linear_1.weight": (4 * ch[0], ch[0]), "time_embedding.linear_1.bias": (4 * ch[0],),
         "time_embedding.linear_2.weight": (4 * ch[0], 4 * ch[0]), "time_embedding.linear_2.bias": (4 * ch[0],)}
    temb = 4 * ch[0]
    skips = [ch[0]]
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg["layers_per_block"]):
            s.update(_resnet(f"down_blocks.{i}.resnets.{j}", cin, cout, temb))
            if cfg["down_has_attn"][i]:
                s.update(_transformer(f"down_blocks.{i}.attentions.{j}", cout, cfg["cross_attention_dim"]))
            cin = cout
            skips.append(cout)
        if i < len(ch) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
            skips.append(cout)
    s.update(_resnet("mid_block.resnets.0", ch[-1], ch[-1], temb))
    s.update(_transformer("mid_block.attentions.0", ch[-1], cfg["cross_attention_dim"]))
    s.update(_resnet("mid_block.resnets.1", ch[-1], ch[-1], temb))
    rev = list(reversed(ch))
    cin = ch[-1]
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            skip = skips.pop()
            s.update(_resnet(f"up_blocks.{i}.resnets.{j}", cin + skip, cout, temb))
            if cfg["up_has_attn"][i]:
                s.update(_transformer(f"up_blocks.{i}.attentions.{j}", cout, cfg["cross_attention_dim"]))
            cin = cout
        if i < len(ch) - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    s["conv_norm_out.weight"] = (ch[0],)
    s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    s["conv_out.bias"] = (cfg["out_channels"],)
    return s


def _vae_attn(p, c):
    s = {f"{p}.group_norm.weight": (c,), f"{p}.group_norm.bias": (c,)}
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{p}.{n}.weight"] = (c, c)
        s[f"{p}.{n}.bias"] = (c,)
    return s


def vae_shapes(cfg=VAE_CFG):
    ch = cfg["block_out_channels"]
    lc = cfg["latent_channels"]
    s = {"post_quant_conv.weight": (lc, lc, 1, 1), "post_quant_conv.bias": (lc,),
         "quant_conv.weight": (2 * lc, 2 * lc, 1, 1), "quant_conv.bias": (2 * lc,),
         "decoder.conv_in.weight": (ch[-1], lc, 3, 3), "decoder.conv_in.bias": (ch[-1],),
         "encoder.conv_in.weight": (ch[0], 3, 3, 3), "encoder.conv_in.bias": (ch[0],)}
    for side in ("decoder", "encoder"):
        s.update(_resnet(f"{side}.mid_block.resnets.0", ch[-1], ch[-1], 0))
        s.update(_vae_attn(f"{side}.mid_block.attentions.0", ch[-1]))
        s.update(_resnet(f"{side}.mid_block.resnets.1", ch[-1], ch[-1], 0))
    cin = ch[-1]
    for i, cout in enumerate(reversed(ch)):
        for j in range(cfg["layers_per_block"] + 1):
            s.update(_resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin, cout, 0))
            cin = cout
        if i < len(ch) - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    s["decoder.conv_norm_out.weight"] = (ch[0],)
    s["decoder.conv_norm_out.bias"] = (ch[0],)
    s["decoder.conv_out.weight"] = (3, ch[0], 3, 3)
    s["decoder.conv_out.bias"] = (3,)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg["layers_per_block"]):
            s.update(_resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin, cout, 0))
            cin = cout
        if i < len(ch) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
    s["encoder.conv_norm_out.weight"] = (ch[-1],)
    s["encoder.conv_norm_out.bias"] = (ch[-1],)
    s["encoder.conv_out.weight"] = (2 * lc, ch[-1], 3, 3)
    s["encoder.conv_out.bias"] = (2 * lc,)
    return s


# ------------------------------------------------------------------ seeded random parameters
def random_state(shapes, seed=0, dtype=torch.float16, device="cpu", gain=1.0):
    """Variance-preserving random init (std = gain/sqrt(fan_in)); norm scales ~ 1, biases small.  Values are
    rounded to fp16 so that the fp32 oracle and the fp16 kernels see identical parameters."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("weight") and len(shp) >= 2:
            fan_in = math.prod(shp[1:])
            t = torch.randn(shp, generator=g) * (gain / math.sqrt(fan_in))
        elif name.endswith("weight"):          # norm scale
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            t = 0.05 * torch.randn(shp, generator=g)
        out[name] = t.to(torch.float16).to(dtype).to(device)
    return out


def load_safetensors(path, device="cpu", dtype=torch.float16):
    from safetensors.torch import load_file
    return {k: v.to(dtype).to(device) for k, v in load_file(path).items()}


_LEGACY_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def remap_legacy_attention(state):
    """Older diffusers VAE checkpoints (e.g. the Oct-2022 `runwayml/stable-diffusion-inpainting` export) name the mid-block
    attention `query / key / value / proj_attn`, sometimes as 1x1 convolutions [C,C,1,1]; diffusers converts them on load
    (deprecated-attention path), which the reference relies on.  Same conversion here: rename, squeeze the 1x1 dims."""
    out = {}
    for k, v in state.items():
        parts = k.split(".")
        if "attentions" in parts and len(parts) >= 2 and parts[-2] in _LEGACY_VAE_ATTN:
            parts[-2] = _LEGACY_VAE_ATTN[parts[-2]]
            k = ".".join(parts)
            if v.dim() == 4 and v.shape[-2:] == (1, 1):
                v = v.reshape(v.shape[0], v.shape[1])
        out[k] = v
    return out


def check_state(state, shapes, what="model"):
    """A checkpoint must carry exactly the tensors of the architecture the kernels were laid out for (diffusers' key
    names); a wrong family (e.g. a text-to-image UNet with a 4-channel conv_in) is reported before anything is launched."""
    state = remap_legacy_attention(state)
    missing = sorted(k for k in shapes if k not in state)
    wrong = sorted(f"{k}: {tuple(state[k].shape)} != {tuple(shapes[k])}" for k in shapes if k in state and tuple(state[k].shape) != tuple(shapes[k]))
    if missing or wrong:
        raise ValueError(f"{what} checkpoint does not match the SD-1.5-inpainting layout: {len(missing)} missing "
                         f"(first: {missing[:3]}), {len(wrong)} with other shapes (first: {wrong[:3]})")
    return {k: state[k] for k in shapes}          # extra tensors (e.g. position ids) are ignored


# ------------------------------------------------------------------ kernel layouts
def conv_weight(w, cin_pad=None, cout_pad=None):
    """[Cout, Cin, kh, kw] -> [Cout(_pad), kh*kw*Cin(_pad)] fp16, K ordered (ky, kx, ci)."""
    co, ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)                     # [Cout, kh, kw, Cin]
    if cin_pad and cin_pad > ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
    w = w.reshape(co, -1)
    if cout_pad and cout_pad > co:
        w = torch.nn.functional.pad(w, (0, 0, 0, cout_pad - co))
    return w.contiguous()


def upsample_phase_weights(w):
    """[Cout, Cin, 3, 3] -> four [Cout, 4 * Cin] fp16 matrices, one per output parity (a, b) of `conv3x3(nearest-upsample-x2(x))`: output
    pixel (2y + a, 2x + b) only sees the 2 x 2 source pixels (y - 1 + a .. y + a) x (x - 1 + b .. x + b), each weighted by the sum of the
    3x3 taps that land on it (summed in fp32, rounded once).  K order (dy, dx, ci), what sd_conv_gemm_f16(taps = 4, phase = 1 + 2a + b) reads."""
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}                 # parity -> taps folded onto window position 0 / 1
    w32 = w.float()
    out = []
    for a in (0, 1):
        for b in (0, 1):
            taps = [sum(w32[:, :, ky, kx] for ky in sets[a][dy] for kx in sets[b][dx]) for dy in (0, 1) for dx in (0, 1)]      # each [Cout, Cin]
            out.append(torch.stack(taps, dim=1).reshape(w.shape[0], -1).to(w.dtype).contiguous())
    return out


def pad_vec(b, n):
    return torch.nn.functional.pad(b, (0, n - b.shape[0])).contiguous() if n > b.shape[0] else b.contiguous()


def geglu_interleave(w, b):
    """Rows [value(0..inner) | gate(0..inner)] -> per 32 output columns: [32 value rows | 32 gate rows] so that a
    wave's two MFMA column tiles hold value and gate of the same outputs (sd_conv_gemm_f16, SD_EPI_GEGLU)."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0
    idx = torch.arange(inner).reshape(-1, 32)
    perm = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    return w[perm].contiguous(), b[perm].contiguous()
