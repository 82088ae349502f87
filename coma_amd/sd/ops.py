"""Thin Python bindings of the sd_* C ABI (include/sd_hip.h).  Activations are NHWC fp16 tensors on a HIP device;
every function launches on torch's current stream and raises ComaHipError on failure.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib

EPI_NONE, EPI_GEGLU, EPI_SILU, EPI_BIAS_ROWS, EPI_PERM16_N, EPI_PERM32_N = 0, 1, 2, 4, 8, 16
F16 = torch.float16


class ConvGemmDesc(C.Structure):
    _fields_ = [("a0", C.c_void_p), ("a1", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int),
                ("batch", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
                ("taps", C.c_int), ("stride", C.c_int), ("upsample", C.c_int), ("pad", C.c_int), ("n", C.c_int),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("bias_bn", C.c_void_p), ("ldbb", C.c_int), ("res", C.c_void_p), ("ldr", C.c_int),
                ("out", C.c_void_p), ("ldo", C.c_int), ("epi", C.c_int), ("nbatch_z", C.c_int),
                ("stride_a", C.c_int64), ("stride_w", C.c_int64), ("stride_out", C.c_int64), ("stride_res", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("colstats", C.c_void_p),
                ("out_t", C.c_void_p), ("n_split", C.c_int), ("ldo_t", C.c_int), ("rows_per_sample", C.c_int), ("phase", C.c_int)]


def _p(t, name="tensor", dtype=F16):
    if t is None:
        return None
    from . import model
    rec = model.recording()
    if rec is not None:                         # this thread is recording a plan: whatever it points into belongs to the model
        rec.register(t, model.BUF_IF_NEW)       # scratch of LaunchGraph.buf stays scratch; a constant seen for the first time is PERSISTENT
    return _lib.ptr(t, dtype, name).value


def _stream(t):
    return _lib.stream_ptr(t.device)


def conv_gemm(a0, w, out, *, batch, in_h, in_w, out_h=None, out_w=None, c0, n, a1=None, c1=0, taps=1, stride=1, upsample=0,
              pad=1, bias=None, bias_bn=None, ldbb=0, res=None, ldr=0, ldo=0, epi=EPI_NONE, nbatch_z=1, stride_a=0, stride_w=0,
              stride_out=0, stride_res=0, workspace=None, colstats=None,
              out_t=None, n_split=0, ldo_t=0, rows_per_sample=0, phase=0):
    d = ConvGemmDesc()
    d.phase = phase
    d.a0, d.a1, d.c0, d.c1 = _p(a0, "a0"), _p(a1, "a1"), c0, c1
    d.batch, d.in_h, d.in_w = batch, in_h, in_w
    d.out_h = out_h if out_h is not None else in_h
    d.out_w = out_w if out_w is not None else in_w
    d.taps, d.stride, d.upsample, d.pad, d.n = taps, stride, upsample, pad, n
    d.ldbb = ldbb
    d.w, d.bias, d.bias_bn, d.res, d.ldr = _p(w, "w"), _p(bias, "bias"), _p(bias_bn, "bias_bn"), _p(res, "res"), ldr
    d.out, d.ldo, d.epi, d.nbatch_z = _p(out, "out"), ldo, epi, nbatch_z
    d.stride_a, d.stride_w, d.stride_out, d.stride_res = stride_a, stride_w, stride_out, stride_res
    if workspace is not None:
        d.workspace, d.workspace_bytes = _p(workspace, "workspace", torch.float32), workspace.numel() * 4
    if colstats is not None:
        d.colstats = _p(colstats, "colstats", torch.float32)
    if out_t is not None:          # columns [n_split, n) leave transposed per sample in the PERM16 key order (to_q | to_k | to_v in one launch)
        d.out_t, d.n_split, d.ldo_t, d.rows_per_sample = _p(out_t, "out_t"), n_split, ldo_t, rows_per_sample
    _lib.check(_lib.lib().sd_conv_gemm_f16(C.byref(d), _stream(out)), "sd_conv_gemm_f16")
    return out


def linear(x, w, out, *, rows, k, n, bias=None, res=None, epi=EPI_NONE, ldo=0):
    """out[rows, n] = x[rows, k] @ w[n, k]^T (+bias)(+res)."""
    return conv_gemm(x, w, out, batch=rows, in_h=1, in_w=1, c0=k, n=n, bias=bias, res=res, epi=epi, ldo=ldo)


def groupnorm(x0, gamma, beta, out, stats, *, batch, hw, c0, x1=None, c1=0, groups=32, eps=1e-5, silu=True):
    rc = _lib.lib().sd_groupnorm_f16(_p(x0, "x0"), _p(x1, "x1"), c0, c1, batch, hw, groups, eps, _p(gamma), _p(beta),
                                     1 if silu else 0, _p(out, "out"), _p(stats, "stats", torch.float32), _stream(out))
    _lib.check(rc, "sd_groupnorm_f16")
    return out


def groupnorm_colstats(x0, gamma, beta, out, stats, colstats0, *, batch, hw, c0, x1=None, c1=0, colstats1=None, groups=32, eps=1e-5,
                       silu=True):
    f32 = torch.float32
    rc = _lib.lib().sd_groupnorm_colstats_f16(_p(x0, "x0"), _p(x1, "x1"), c0, c1, batch, hw, groups, eps, _p(gamma), _p(beta),
                                              1 if silu else 0, _p(out, "out"), _p(stats, "stats", f32),
                                              _p(colstats0, "colstats0", f32), _p(colstats1, "colstats1", f32), _stream(out))
    _lib.check(rc, "sd_groupnorm_colstats_f16")
    return out


def gn_scratch_floats(batch, hw, groups=32, channels=2560):
    return batch * channels * 2 + batch * groups * 2 * ((hw + 63) // 64)


def layernorm(x, gamma, beta, out, *, rows, c, eps=1e-5):
    _lib.check(_lib.lib().sd_layernorm_f16(_p(x), rows, c, eps, _p(gamma), _p(beta), _p(out), _stream(out)), "sd_layernorm_f16")
    return out


def attention(q, k, vt, out, *, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, scale, vt_perm16=False, pipelined=None):
    """pipelined: None = the library's own choice; True / False force / forbid the software-pipelined d = 40 kernel (tests, A-B)."""
    flags = (1 if vt_perm16 else 0) | (2 if pipelined is True else 0) | (4 if pipelined is False else 0)
    rc = _lib.lib().sd_attention_f16(_p(q, "q"), _p(k, "k"), _p(vt, "vt"), _p(out, "out"), batch, heads, lq, lk, d, ldq, ldk,
                                     ldv, ldo, scale, flags, _stream(out))
    _lib.check(rc, "sd_attention_f16")
    return out


def perm16_columns(x):
    """[..., n] -> [..., roundup16(n)] with every group of 16 columns in the order (0-3, 8-11, 4-7, 12-15): what SD_EPI_PERM16_N
    produces and sd_attention_f16(vt_perm16=1) reads (pad columns zero).  Host / test helper."""
    n = x.shape[-1]
    n16 = (n + 15) // 16 * 16
    y = torch.zeros(*x.shape[:-1], n16, dtype=x.dtype, device=x.device)
    y[..., :n] = x
    j = torch.arange(n16, device=x.device)
    src = (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)
    return y[..., src].contiguous()


def attention_wide(q, k, vt, out, *, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, scale):
    """Wide heads (d = 128 / 256 / 512), lk % 64 == 0, V^T in the EPI_PERM32_N key order (perm32_columns)."""
    rc = _lib.lib().sd_attention_wide_f16(_p(q, "q"), _p(k, "k"), _p(vt, "vt"), _p(out, "out"), batch, heads, lq, lk, d, ldq, ldk, ldv, ldo,
                                          scale, _stream(out))
    _lib.check(rc, "sd_attention_wide_f16")
    return out


def perm32_columns(x):
    """[..., n] (n % 32 == 0) -> the column order SD_EPI_PERM32_N produces: position 8g + e of every 32 holds column 16 (e >> 2) + 4g + (e & 3)."""
    n = x.shape[-1]
    assert n % 32 == 0
    p = torch.arange(n, device=x.device)
    src = (p & ~28) | (((p >> 3) & 3) << 2) | (((p >> 2) & 1) << 4)
    return x[..., src].contiguous()


def xattn_chain(a, h, wo1, bo1, g2, b2, wq, k2, vt2, wo2, bo2, g3, b3, h2, n3, *, rows, rows_per_sample, lk, ldv2, eps=1e-5, debug_out=None,
                debug_stage=0):
    """attn1.to_out + residual -> LayerNorm2 -> attn2 (to_q, 77-key cross attention, to_out + residual) -> LayerNorm3 in one launch
    (C = 320, 8 heads of 40); see sd_xattn_chain_f16 in include/sd_hip.h."""
    args = [_p(t, n) for t, n in ((a, "attn1_out"), (h, "h"), (wo1, "wo1"), (bo1, "bo1"), (g2, "gamma2"), (b2, "beta2"), (wq, "wq2"), (k2, "k2"),
                                  (vt2, "vt2"), (wo2, "wo2"), (bo2, "bo2"), (g3, "gamma3"), (b3, "beta3"), (h2, "h2"), (n3, "n3"))]
    rc = _lib.lib().sd_xattn_chain_f16(*args, rows, rows_per_sample, lk, ldv2, eps, _p(debug_out, "debug_out"), debug_stage, _stream(h2))
    _lib.check(rc, "sd_xattn_chain_f16")


def groupnorm_table(x0, gamma, beta, stats, *, batch, hw, c0, groups=32, eps=1e-5, colstats0=None, rows_per_slot=32):
    """(scale, shift) per (sample, channel) -> stats[: batch * c0 * 2] (fp32), nothing applied; colstats0 fp32 [batch * hw / rows_per_slot][2][c0]."""
    rc = _lib.lib().sd_groupnorm_table_f16(_p(x0, "x0"), c0, batch, hw, groups, eps, _p(gamma), _p(beta), _p(stats, "stats", torch.float32),
                                           _p(colstats0, "colstats0", torch.float32), rows_per_slot, _stream(x0))
    _lib.check(rc, "sd_groupnorm_table_f16")
    return stats


def xfront(x, gn_affine, wpi, bpi, g1, b1, wqk, wv, h, qk, vt, *, rows, rows_per_sample, ldv, eps=1e-5):
    """GroupNorm affine -> proj_in -> LayerNorm1 -> q | k -> V^T (perm16) in one launch (C = 320); see sd_xfront_f16."""
    rc = _lib.lib().sd_xfront_f16(_p(x, "x"), _p(gn_affine, "gn_affine", torch.float32), _p(wpi), _p(bpi), _p(g1), _p(b1), _p(wqk), _p(wv),
                                  _p(h, "h"), _p(qk, "qk"), _p(vt, "vt"), rows, rows_per_sample, ldv, eps, _stream(h))
    _lib.check(rc, "sd_xfront_f16")


def xtail(n3, h2, x, w1, b1, w2, b2, wpo, bpo, out, colstats=None, *, rows):
    """ff.net.0 (GEGLU) -> ff.net.2 + residual -> proj_out + residual (+ GroupNorm column sums of the result) in one launch (C = 320)."""
    rc = _lib.lib().sd_xtail_f16(_p(n3, "n3"), _p(h2, "h2"), _p(x, "x"), _p(w1), _p(b1), _p(w2), _p(b2), _p(wpo), _p(bpo), _p(out, "out"),
                                 _p(colstats, "colstats", torch.float32), rows, _stream(out))
    _lib.check(rc, "sd_xtail_f16")


def softmax_(x, *, rows, n, ld, scale):
    _lib.check(_lib.lib().sd_softmax_f16(_p(x), rows, n, ld, scale, _stream(x)), "sd_softmax_f16")
    return x


def cfg_ddim_step(eps_uc, eps_ld, latents, x0_out, mask, masked_latents, unet_in, *, batch, hw, guidance, alpha_t, alpha_prev,
                  write_latents=True):
    f32 = torch.float32
    rc = _lib.lib().sd_cfg_ddim_step(_p(eps_uc, "eps"), eps_ld, _p(latents, "latents", f32), _p(x0_out, "x0", f32), _p(mask),
                                     _p(masked_latents), _p(unet_in), batch, hw, guidance, alpha_t, alpha_prev,
                                     1 if write_latents else 0, _stream(latents))
    _lib.check(rc, "sd_cfg_ddim_step")


def timestep_embedding(t, out, *, batch, dim):
    _lib.check(_lib.lib().sd_timestep_embedding_f16(_p(t, "t", torch.float32), batch, dim, _p(out), _stream(out)),
               "sd_timestep_embedding_f16")
    return out


def nchw_to_nhwc(x, out, *, batch, c, hw, cpad):
    _lib.check(_lib.lib().sd_nchw_to_nhwc_f16(_p(x, "x", torch.float32), batch, c, hw, cpad, _p(out), _stream(out)),
               "sd_nchw_to_nhwc_f16")
    return out


def nhwc_to_nchw(x, out, *, batch, c, hw, ld):
    _lib.check(_lib.lib().sd_nhwc_to_nchw_f32(_p(x), batch, c, hw, ld, _p(out, "out", torch.float32), _stream(out)),
               "sd_nhwc_to_nchw_f32")
    return out


def image_to_u8(x, out, *, batch, hw, ld, round_mode=0):
    _lib.check(_lib.lib().sd_image_to_u8(_p(x), batch, hw, ld, round_mode, _p(out, "out", torch.uint8), _stream(out)),
               "sd_image_to_u8")
    return out


def vae_sample(moments, ld, noise, scale, npix, lat32=None, lat16=None):
    rc = _lib.lib().sd_vae_sample(_p(moments), ld, _p(noise, "noise", torch.float32), scale, npix, _p(lat32, "lat32", torch.float32),
                                  _p(lat16), _stream(moments))
    _lib.check(rc, "sd_vae_sample")


def add_noise(x0, noise, alpha, out):
    f32 = torch.float32
    _lib.check(_lib.lib().sd_add_noise(_p(x0, "x0", f32), _p(noise, "noise", f32), alpha, x0.numel(), _p(out, "out", f32),
                                       _stream(out)), "sd_add_noise")
    return out


def mask_adapt(seg, default_mask, image_nchw, mask_full, mask_latent, masked_image, *, H, W, dilate_iters, use_default, cpad):
    u8 = torch.uint8
    rc = _lib.lib().sd_mask_adapt(_p(seg, "seg", u8), _p(default_mask, "default", u8), H, W, dilate_iters, 1 if use_default else 0,
                                  _p(image_nchw, "image", torch.float32), cpad, _p(mask_full, "mask_full", u8), _p(mask_latent),
                                  _p(masked_image), _stream(mask_full))
    _lib.check(rc, "sd_mask_adapt")


def mask_adapt_batched(seg, default_mask, image_nchw, mask_full, mask_latent, masked_image, area, scratch, *, batch, H, W,
                       dilate_iters, force_default, area_thres, cpad, write_pad=False):
    u8 = torch.uint8
    rc = _lib.lib().sd_mask_adapt_batched(_p(seg, "seg", u8), _p(default_mask, "default", u8), batch, H, W, dilate_iters,
                                          1 if force_default else 0, float(area_thres), _p(image_nchw, "image", torch.float32), cpad,
                                          1 if write_pad else 0, _p(mask_full, "mask_full", u8), _p(mask_latent), _p(masked_image),
                                          _p(area, "area", torch.int32), _p(scratch, "scratch", u8), _stream(mask_full))
    _lib.check(rc, "sd_mask_adapt_batched")


def copy_d2d(dst, src):
    """dst <- src (same byte count), device to device on the current stream; recordable into a plan."""
    n = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == n and dst.is_contiguous() and src.is_contiguous()
    _p(dst, "dst", dst.dtype), _p(src, "src", src.dtype)
    _lib.check(_lib.lib().sd_copy_d2d(dst.data_ptr(), src.data_ptr(), n, _stream(dst)), "sd_copy_d2d")


def groupnorm_table_cat(gamma, beta, stats, colstats0, colstats1, *, batch, hw, c0, c1, groups=32, eps=1e-5):
    """(scale, shift) per (sample, channel) of a GroupNorm over [x0 | x1], from the two producers' column sums only."""
    f32 = torch.float32
    rc = _lib.lib().sd_groupnorm_table_cat_f16(c0, c1, batch, hw, groups, eps, _p(gamma), _p(beta), _p(stats, "stats", f32),
                                               _p(colstats0, "colstats0", f32), _p(colstats1, "colstats1", f32), _stream(stats))
    _lib.check(rc, "sd_groupnorm_table_cat_f16")
    return stats


def winograd_input(x0, v, *, batch, h, w, c0, x1=None, c1=0, upsample=False, gn_affine=None, silu=False, vscale=1.0):
    """v fp16 [16][batch*h/2*w/2][c0+c1] = B^T d B of every 4x4 patch (F(2x2,3x3), zero pad 1); upsample: [h, w] is the nearest-x2
    upsampling of the [h/2, w/2] sources; vscale: power-of-two scale of the stored planes (fp16 headroom, undone by the output transform)."""
    _lib.check(_lib.lib().sd_winograd_input_f16(_p(x0, "x0"), _p(x1, "x1"), c0, c1, batch, h, w, 1 if upsample else 0,
                                                _p(gn_affine, "gn_affine", torch.float32), 1 if silu else 0, vscale, _p(v, "v"), _stream(v)),
               "sd_winograd_input_f16")
    return v


def winograd_weight(w, u, *, n, c, uscale=1.0):
    """u fp16 [16][n][c] = uscale * G g G^T of w fp16 [n][9][c]."""
    _lib.check(_lib.lib().sd_winograd_weight_f16(_p(w, "w"), n, c, uscale, _p(u, "u"), _stream(u)), "sd_winograd_weight_f16")
    return u


def winograd_output(m, out, *, batch, h, w, n, ldm=0, bias=None, bias_bn=None, ldbb=0, res=None, ldr=0, ldo=0, silu=False, colstats=None,
                    mscale=1.0):
    """out fp16 [batch*h*w, ldo] = mscale * A^T m A of m fp16 [16][T][ldm] (+ bias, per-sample bias, SiLU, residual)."""
    rc = _lib.lib().sd_winograd_output_f16(_p(m, "m"), ldm or n, batch, h, w, n, _p(bias, "bias"), _p(bias_bn, "bias_bn"), ldbb,
                                           _p(res, "res"), ldr, _p(out, "out"), ldo, 1 if silu else 0, mscale, _p(colstats, "colstats", torch.float32),
                                           _stream(out))
    _lib.check(rc, "sd_winograd_output_f16")
    return out


def conv3x3_small_n(x, w, out, *, batch, h, w_, c, n, bias=None, gn_affine=None, silu=False, ldo=64):
    """out[:, 0:n] = conv3x3(act(x * scale + shift)) + bias for n <= 4 output channels (VAE decoder conv_norm_out + SiLU + conv_out)."""
    rc = _lib.lib().sd_conv3x3_small_n_f16(_p(x, "x"), _p(gn_affine, "gn_affine", torch.float32), 1 if silu else 0, _p(w, "w"), _p(bias, "bias"),
                                           batch, h, w_, c, n, _p(out, "out"), ldo, _stream(out))
    _lib.check(rc, "sd_conv3x3_small_n_f16")
    return out


def conv3x3_halo(x, w, out, *, batch, h, w_, c, n=128, bias=None, res=None, gn_affine=None, silu=False, colstats=None, ldo=0, ldr=0):
    """out[:, 0:n] = conv3x3(act(x * scale + shift)) + bias (+ res): halo-patch convolution (sd_haloconv.hip); n in {128, 256, 384, 512},
    c % 64 == 0 and c <= 512, h and w multiples of 16, one input sample below 2 GiB (refused otherwise)."""
    rc = _lib.lib().sd_conv3x3_halo_f16(_p(x, "x"), c, _p(gn_affine, "gn_affine", torch.float32), 1 if silu else 0, _p(w, "w"), _p(bias, "bias"),
                                        _p(res, "res"), ldr, batch, h, w_, n, _p(out, "out"), ldo, _p(colstats, "colstats", torch.float32),
                                        _stream(out))
    _lib.check(rc, "sd_conv3x3_halo_f16")
    return out


def gn_winograd_input(v, gamma, beta, *, batch, h, w, c0, x0=None, x1=None, c1=0, m=None, ldm=0, bias=None, bias_bn=None, ldbb=0, groups=32,
                      eps=1e-5, silu=True, mscale=1.0):
    """v = B^T act(GroupNorm(source)) B, source = [x0 | x1] or the output transform of the plane products m (+ bias, per-sample bias)."""
    rc = _lib.lib().sd_gn_winograd_input_f16(_p(x0, "x0"), _p(x1, "x1"), c0, c1, _p(m, "m"), ldm or c0, _p(bias, "bias"), _p(bias_bn, "bias_bn"),
                                             ldbb, batch, h, w, groups, eps, _p(gamma), _p(beta), 1 if silu else 0, mscale, _p(v, "v"), _stream(v))
    _lib.check(rc, "sd_gn_winograd_input_f16")
    return v


def im2col3x3_c3(x, out, *, batch, h, w, ldx):
    """out[m][3 * tap + ch] = the 3x3x3 neighbourhood of pixel m of a 3-channel NHWC image (zero pad; columns 27..31 zero)."""
    _lib.check(_lib.lib().sd_im2col3x3_c3_f16(_p(x, "x"), ldx, batch, h, w, _p(out, "out"), _stream(out)), "sd_im2col3x3_c3_f16")
    return out


def conv3x3_c3(x, w32, out, *, batch, h, w, ldx, n=128, bias=None, colstats=None, ldo=0):
    """3x3 / pad 1 convolution of a 3-channel NHWC image into n = 128 channels in one launch (w32 = [n][ky][kx][c] padded to 32 halfs);
    colstats fp32 [batch*h*w/256][2][n]: per-tile column sums of the stored output (rows_per_slot = 256 for sd_groupnorm_table_f16)."""
    rc = _lib.lib().sd_conv3x3_c3_f16(_p(x, "x"), ldx, _p(w32), _p(bias), batch, h, w, n, _p(out, "out"), ldo or out.shape[-1],
                                      _p(colstats, "colstats", torch.float32), _stream(out))
    _lib.check(rc, "sd_conv3x3_c3_f16")
    return out
