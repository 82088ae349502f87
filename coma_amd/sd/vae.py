"""AutoencoderKL (SD-1.5 VAE) on MI355X as two static launch graphs (decoder, encoder) over NHWC fp16 buffers.

The reference pipeline uses exactly: ``vae.decode(z / scaling_factor, return_dict=False)[0]``
(utils/adaptive_mask_inpainting.py:1086, :1112), ``vae.encode(img).latent_dist.sample(generator)`` (:677-680) and
``vae.config.{scaling_factor, latent_channels, block_out_channels}`` (:371, :682, :927).  Architecture: public
SD-1.5 VAE (diffusers is third party).  GroupNorm eps 1e-6, no time embedding, single-head
attention over 4096 tokens at C = 512 in the mid block (sd_attention_wide_f16: 16 queries per wave on 16x16x32 MFMAs).
"""
from __future__ import annotations

import torch

from . import ops
from .graph import F16, LaunchGraph
from .weights import VAE_CFG, conv_weight, pad_vec


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class _VaeBase:
    upsample_phases = True       # class-level switch (tests / A-B): False = Upsample2D convolutions as 3x3 over the upsampled tensor
    direct_conv_in = True        # class-level switch (tests / A-B): False = im2col pass + K = 32 product (packed_conv_in) for the encoder's conv_in
    packed_conv_in = True        # class-level switch (tests / A-B): False = encoder conv_in as a K = 576 implicit GEMM over the 64-channel padded input
    fused_conv_out = True        # class-level switch (tests / A-B): False = GroupNorm kernel + 64-column implicit-GEMM tile for conv_out
    halo_conv = True             # class-level switch (tests / A-B): False = GroupNorm kernel + implicit GEMM for the 128-channel 3x3 convolutions
    halo_min_tiles = 512         # ... from this many 16 x 16 tiles on (two per CU)
    halo_widths = (128, 256, 512)     # ... for these output widths (n / 128 workgroups per tile, 128 channels each)
    fused_attention = True       # class-level switch (tests / A-B): False = QK^T GEMM -> softmax -> PV GEMM through memory

    def __init__(self, state, batch, device, cfg, use_graph=True, plan="decode"):
        self.use_graph = use_graph
        self.cfgd = cfg
        self.device = torch.device(device)
        self.batch = batch
        self.s = {k: v.to(self.device, F16) for k, v in state.items()}
        self.g = LaunchGraph(self.device, plan=plan)

    def _halo(self, cin, cout, H, W):
        """GroupNorm + SiLU + conv3x3 as ONE halo-patch convolution (sd_conv3x3_halo_f16) where it was built and measured: 128 / 256 /
        512 output channels (n / 128 workgroups per tile), input channels a multiple of 64 up to 512, one input sample below 2 GiB (the
        kernel's 32-bit offsets), feature maps in whole 16 x 16 tiles, at least halo_min_tiles workgroups (two per CU)."""
        return (self.halo_conv and cout in self.halo_widths and cin % 64 == 0 and cin <= 512 and H % 16 == 0 and W % 16 == 0
                and H * W * cin * 2 < 0x7fffffff and self.batch * (H // 16) * (W // 16) * (cout // 128) >= self.halo_min_tiles)

    def _resnet(self, p, x, cin, cout, H, W):
        g, s, B = self.g, self.s, self.batch
        M = B * H * W
        h = g.buf(M, cout)
        if self._halo(cin, cout, H, W):
            # (per-tile column sums only when their one consumer, norm2 -> conv2, reads them through the affine table: a halo convolution too)
            g.gn_silu_conv3x3_halo(x, s[p + ".norm1.weight"], s[p + ".norm1.bias"], conv_weight(s[p + ".conv1.weight"]), s[p + ".conv1.bias"], h,
                                   batch=B, h=H, w_=W, c=cin, n=cout, eps=1e-6, stats=self._halo(cout, cout, H, W))
        else:
            n1 = g.buf(M, cin)
            g.groupnorm(x, s[p + ".norm1.weight"], s[p + ".norm1.bias"], n1, batch=B, hw=H * W, c0=cin, eps=1e-6, silu=True)
            g.conv(n1, conv_weight(s[p + ".conv1.weight"]), h, batch=B, in_h=H, in_w=W, c0=cin, n=cout, taps=9, bias=s[p + ".conv1.bias"],
                   stats=True)
        if p + ".conv_shortcut.weight" in s:
            sc = g.buf(M, cout)
            g.conv(x, conv_weight(s[p + ".conv_shortcut.weight"]), sc, batch=B, in_h=H, in_w=W, c0=cin, n=cout,
                   bias=s[p + ".conv_shortcut.bias"])
        else:
            sc = x
        out = g.buf(M, cout)
        if self._halo(cout, cout, H, W):
            g.gn_silu_conv3x3_halo(h, s[p + ".norm2.weight"], s[p + ".norm2.bias"], conv_weight(s[p + ".conv2.weight"]), s[p + ".conv2.bias"], out,
                                   batch=B, h=H, w_=W, c=cout, n=cout, eps=1e-6, res=sc, stats=True)
        else:
            n2 = g.buf(M, cout)
            g.groupnorm(h, s[p + ".norm2.weight"], s[p + ".norm2.bias"], n2, batch=B, hw=H * W, c0=cout, eps=1e-6, silu=True)
            g.conv(n2, conv_weight(s[p + ".conv2.weight"]), out, batch=B, in_h=H, in_w=W, c0=cout, n=cout, taps=9,
                   bias=s[p + ".conv2.bias"], res=sc, stats=True)
        return out

    def _attention(self, p, x, C, H, W):
        """Single-head attention with q/k/v/out biases and a residual.  Fused (sd_attention_wide_f16) when the head is 128 / 256 /
        512 wide and the token count a multiple of 64 -- the score matrix (4096 x 4096 per image at 512 x 512) never exists;
        otherwise S = QK^T, row softmax, O = P V through memory."""
        g, s, B = self.g, self.s, self.batch
        L, M = H * W, B * H * W
        gn = g.buf(M, C)
        g.groupnorm(x, s[p + ".group_norm.weight"], s[p + ".group_norm.bias"], gn, batch=B, hw=L, c0=C, eps=1e-6, silu=False)
        q, k = g.buf(M, C), g.buf(M, C)
        g.conv(gn, s[p + ".to_q.weight"], q, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=s[p + ".to_q.bias"])
        g.conv(gn, s[p + ".to_k.weight"], k, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=s[p + ".to_k.bias"])
        fused = self.fused_attention and C in (128, 256, 512) and L % 64 == 0
        vt = g.buf(B, C, L)                                  # V^T[b] = Wv . X_b^T + bv (bias per row)
        g.conv(s[p + ".to_v.weight"], gn, vt, batch=C, in_h=1, in_w=1, c0=C, n=L, bias=s[p + ".to_v.bias"],
               epi=ops.EPI_BIAS_ROWS | (ops.EPI_PERM32_N if fused else 0), nbatch_z=B, stride_w=L * C, stride_out=C * L)
        a = g.buf(M, C)
        if fused:
            g.attention_wide(q, k, vt, a, batch=B, heads=1, lq=L, lk=L, d=C, ldq=C, ldk=C, ldv=L, ldo=C)
        else:
            sc = g.buf(B, L, L)
            g.conv(q, k, sc, batch=L, in_h=1, in_w=1, c0=C, n=L, nbatch_z=B, stride_a=L * C, stride_w=L * C, stride_out=L * L)
            g.add(lambda: ops.softmax_(sc, rows=B * L, n=L, ld=L, scale=C ** -0.5), tag=f"softmax rows={B * L} n={L}")
            g.conv(sc, vt, a, batch=L, in_h=1, in_w=1, c0=L, n=C, nbatch_z=B, stride_a=L * L, stride_w=C * L, stride_out=L * C)
        out = g.buf(M, C)
        g.conv(a, s[p + ".to_out.0.weight"], out, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=s[p + ".to_out.0.bias"], res=x)   # (M tokens as batch: no stats)
        return out

    def _replay(self):
        if not self.use_graph:
            return self.g.run()
        self.g.replay()

    def save(self, path):
        """The model file sd_model_load + sd_vae_decode / sd_vae_encode run without Python."""
        self.g.capture()
        self.g.model.save(path)


class HipVaeDecoder(_VaeBase):
    """z (fp16 NHWC [B, h*w, 64], 4 valid channels, ALREADY divided by scaling_factor) -> image NHWC [B, 64*h*w, 64]."""

    def __init__(self, state, batch, latent_h=64, latent_w=64, device="cuda", cfg=VAE_CFG):
        super().__init__(state, batch, device, cfg, plan="decode")
        g, s, B = self.g, self.s, batch
        ch = cfg["block_out_channels"]
        H, W = latent_h, latent_w
        self.h, self.w = H, W
        self.z = g.buf(B, H * W, 64, zero=True)
        pq = g.buf(B * H * W, 64, zero=True)
        g.conv(self.z, conv_weight(s["post_quant_conv.weight"], cin_pad=64, cout_pad=64), pq, batch=B * H * W, in_h=1, in_w=1,
               c0=64, n=64, bias=pad_vec(s["post_quant_conv.bias"], 64))
        x = g.buf(B * H * W, ch[-1])
        g.conv(pq, conv_weight(s["decoder.conv_in.weight"], cin_pad=64), x, batch=B, in_h=H, in_w=W, c0=64, n=ch[-1], taps=9,
               bias=s["decoder.conv_in.bias"])
        x = self._resnet("decoder.mid_block.resnets.0", x, ch[-1], ch[-1], H, W)
        x = self._attention("decoder.mid_block.attentions.0", x, ch[-1], H, W)
        x = self._resnet("decoder.mid_block.resnets.1", x, ch[-1], ch[-1], H, W)
        cin = ch[-1]
        for i, cout in enumerate(reversed(ch)):
            for j in range(cfg["layers_per_block"] + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, cin, cout, H, W)
                cin = cout
            if i < len(ch) - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                o = g.buf(B * 4 * H * W, cout)
                if self.upsample_phases and W & (W - 1) == 0 and (H * W) % 32 == 0 and 4 * B * H * W * cout * 2 < (1 << 31) - (1 << 22):
                    # four sub-pixel phase products over the source instead of a 3x3 convolution over the upsampled tensor (2.25 x fewer
                    # multiplies, exact: sd_conv_gemm_desc.phase)
                    g.conv3x3_upsampled(x, s[p + ".weight"], o, batch=B, in_h=H, in_w=W, c0=cout, n=cout, bias=s[p + ".bias"], stats=True)
                else:
                    g.conv(x, conv_weight(s[p + ".weight"]), o, batch=B, in_h=H, in_w=W, out_h=2 * H, out_w=2 * W, c0=cout, n=cout,
                           taps=9, upsample=1, bias=s[p + ".bias"], stats=True)
                x, H, W = o, 2 * H, 2 * W
        self.image = g.buf(B * H * W, 64, zero=True)          # 3 valid channels
        nout = s["decoder.conv_out.weight"].shape[0]
        if self.fused_conv_out and cin == 128 and nout <= 4:
            # conv_norm_out -> SiLU -> conv_out as ONE pass over the last feature map (sd_conv3x3_small_n_f16): no normalised copy of
            # the 0.5 GB tensor, no 64-column GEMM tile for 3 channels
            g.gn_silu_conv3x3_small_n(x, s["decoder.conv_norm_out.weight"], s["decoder.conv_norm_out.bias"],
                                      conv_weight(s["decoder.conv_out.weight"]), s["decoder.conv_out.bias"].contiguous(), self.image,
                                      batch=B, h=H, w_=W, c=cin, n=nout, eps=1e-6, silu=True)
        else:
            gn = g.buf(B * H * W, cin)
            g.groupnorm(x, s["decoder.conv_norm_out.weight"], s["decoder.conv_norm_out.bias"], gn, batch=B, hw=H * W, c0=cin,
                        eps=1e-6, silu=True)
            g.conv(gn, conv_weight(s["decoder.conv_out.weight"], cout_pad=64), self.image, batch=B, in_h=H, in_w=W, c0=cin, n=64,
                   taps=9, bias=pad_vec(s["decoder.conv_out.bias"], 64))
        self.out_h, self.out_w = H, W
        g.model.bind("z", self.z)
        g.model.bind("image", self.image)

    def decode_static(self):
        self._replay()
        return self.image


class HipVaeEncoder(_VaeBase):
    """image (fp16 NHWC [B, H*W, 64], 3 valid channels in [-1,1]) -> moments NHWC [B, H/8*W/8, 64] (mean 4 | logvar 4)."""

    def __init__(self, state, batch, height=512, width=512, device="cuda", cfg=VAE_CFG):
        super().__init__(state, batch, device, cfg, plan="encode")
        g, s, B = self.g, self.s, batch
        ch = cfg["block_out_channels"]
        H, W = height, width
        self.x = g.buf(B, H * W, 64, zero=True)
        x = g.buf(B * H * W, ch[0])
        if self.direct_conv_in and s["encoder.conv_in.weight"].shape[1] == 3 and ch[0] == 128 and H % 16 == 0 and W % 16 == 0:
            # 3 -> 128 channels in one launch: the halo patch of a 16 x 16 tile in LDS, K = 32 operands built there, per-tile column sums for the
            # first ResNet's GroupNorm table (sd_conv3x3_c3_f16)
            w27 = torch.nn.functional.pad(conv_weight(s["encoder.conv_in.weight"]), (0, 5)).contiguous()       # [n][ky][kx][c] -> [n][32]
            cs = g.buf(B * H * W // 256, 2, ch[0], dtype=torch.float32, zero=True)
            g._colstats_tile[x.data_ptr()] = cs
            g.add(lambda h0=H, w0=W, xo=x: ops.conv3x3_c3(self.x, w27, xo, batch=B, h=h0, w=w0, ldx=64, n=ch[0], bias=s["encoder.conv_in.bias"], colstats=cs),     # (x, H, W are rebound below)
                  flops=2 * B * H * W * ch[0] * 32, alg_flops=2 * B * H * W * ch[0] * 27, tag=f"conv3x3(c3) B={B} {H}x{W} n={ch[0]}",
                  nbytes=2 * B * H * W * (4 + ch[0]))
        elif self.packed_conv_in and s["encoder.conv_in.weight"].shape[1] == 3:
            # 3 input channels: one elementwise pass packs every pixel's 3x3x3 neighbourhood into 32 halfs and conv_in is a plain K = 32 product
            # (the implicit GEMM multiplied a 64-channel padded input: K = 576 for 27 real products)
            xp = g.buf(B * H * W, 32)
            g.add(lambda h0=H, w0=W: ops.im2col3x3_c3(self.x, xp, batch=B, h=h0, w=w0, ldx=64),     # (H, W are rebound by the level loop below)
                  tag=f"im2col 3x3x3 B={B} {H}x{W}", nbytes=2 * B * H * W * (4 + 32))
            w27 = torch.nn.functional.pad(conv_weight(s["encoder.conv_in.weight"]), (0, 5)).contiguous()       # [n][ky][kx][c] -> [n][32]
            g.conv(xp, w27, x, batch=B, in_h=H, in_w=W, c0=32, n=ch[0], bias=s["encoder.conv_in.bias"], alg_flops=2 * B * H * W * ch[0] * 27,
                   tag_note=" (conv_in, packed 3x3x3)")        # (a 1x1 over [B, H, W]: the kernel keeps the sample index in 16 bits)
        else:
            g.conv(self.x, conv_weight(s["encoder.conv_in.weight"], cin_pad=64), x, batch=B, in_h=H, in_w=W, c0=64, n=ch[0], taps=9,
                   bias=s["encoder.conv_in.bias"])
        cin = ch[0]
        for i, cout in enumerate(ch):
            for j in range(cfg["layers_per_block"]):
                x = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", x, cin, cout, H, W)
                cin = cout
            if i < len(ch) - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                o = g.buf(B * (H // 2) * (W // 2), cout)
                # F.pad(x, (0,1,0,1)) + conv stride 2 padding 0  ==  low-side pad 0, high side bounds-checked
                g.conv(x, conv_weight(s[p + ".weight"]), o, batch=B, in_h=H, in_w=W, out_h=H // 2, out_w=W // 2, c0=cout, n=cout,
                       taps=9, stride=2, pad=0, bias=s[p + ".bias"], stats=True)
                x, H, W = o, H // 2, W // 2
        x = self._resnet("encoder.mid_block.resnets.0", x, cin, cin, H, W)
        x = self._attention("encoder.mid_block.attentions.0", x, cin, H, W)
        x = self._resnet("encoder.mid_block.resnets.1", x, cin, cin, H, W)
        gn = g.buf(B * H * W, cin)
        g.groupnorm(x, s["encoder.conv_norm_out.weight"], s["encoder.conv_norm_out.bias"], gn, batch=B, hw=H * W, c0=cin,
                    eps=1e-6, silu=True)
        mo = g.buf(B * H * W, 64, zero=True)
        g.conv(gn, conv_weight(s["encoder.conv_out.weight"], cout_pad=64), mo, batch=B, in_h=H, in_w=W, c0=cin, n=64, taps=9,
               bias=pad_vec(s["encoder.conv_out.bias"], 64))
        self.moments = g.buf(B * H * W, 64, zero=True)
        g.conv(mo, conv_weight(s["quant_conv.weight"], cin_pad=64, cout_pad=64), self.moments, batch=B * H * W, in_h=1, in_w=1,
               c0=64, n=64, bias=pad_vec(s["quant_conv.bias"], 64))
        self.lat_h, self.lat_w = H, W
        g.model.bind("x", self.x)
        g.model.bind("moments", self.moments)

    def encode_static(self):
        self._replay()
        return self.moments


class _LatentDist:
    def __init__(self, vae, moments, npix):
        self.vae, self.moments, self.npix = vae, moments, npix

    def sample(self, generator=None):
        B, h, w = self.vae.batch, self.vae.enc.lat_h, self.vae.enc.lat_w
        # drawn as diffusers draws it (randn_tensor(mean.shape, generator, dtype=fp16): NCHW element order), consumed NHWC
        gdev = generator.device if generator is not None else self.vae.device
        noise = torch.randn(B, 4, h, w, generator=generator, device=gdev, dtype=torch.float16).to(self.vae.device, torch.float32)
        return self._to_nchw(noise.permute(0, 2, 3, 1).reshape(B, h * w, 4).contiguous())

    def mode(self):
        return self._to_nchw(None)

    def _to_nchw(self, noise):
        B, h, w = self.vae.batch, self.vae.enc.lat_h, self.vae.enc.lat_w
        lat = torch.empty(B, h * w, 4, dtype=torch.float32, device=self.vae.device)
        ops.vae_sample(self.moments, 64, noise, 1.0, self.npix, lat32=lat)
        return lat.reshape(B, h, w, 4).permute(0, 3, 1, 2).contiguous()


class HipAutoencoderKL:
    """diffusers-shaped facade over the two graphs (NCHW tensors at the boundary, as the reference pipeline passes)."""

    def __init__(self, state, batch, height=512, width=512, device="cuda", cfg=VAE_CFG, with_encoder=True, use_graph=True):
        self.config = _Cfg(scaling_factor=cfg["scaling_factor"], latent_channels=cfg["latent_channels"],
                           block_out_channels=list(cfg["block_out_channels"]))
        self.device, self.batch, self.dtype = torch.device(device), batch, F16
        self.dec = HipVaeDecoder(state, batch, height // 8, width // 8, device, cfg)
        self.enc = HipVaeEncoder(state, batch, height, width, device, cfg) if with_encoder else None
        self.dec.use_graph = use_graph
        if self.enc is not None:
            self.enc.use_graph = use_graph

    def decode(self, z, return_dict=False, **kw):
        B, hw = self.batch, self.dec.h * self.dec.w
        ops.nchw_to_nhwc(z.to(self.device, torch.float32).contiguous(), self.dec.z, batch=B, c=4, hw=hw, cpad=64)
        img = self.dec.decode_static()
        out = torch.empty(B, 3, self.dec.out_h, self.dec.out_w, dtype=torch.float32, device=self.device)
        ops.nhwc_to_nchw(img, out, batch=B, c=3, hw=self.dec.out_h * self.dec.out_w, ld=64)
        return _Cfg(sample=out) if return_dict else (out,)

    def encode(self, image):
        B, H, W = self.batch, image.shape[-2], image.shape[-1]
        ops.nchw_to_nhwc(image.to(self.device, torch.float32).contiguous(), self.enc.x, batch=B, c=3, hw=H * W, cpad=64)
        mom = self.enc.encode_static()
        return _Cfg(latent_dist=_LatentDist(self, mom, B * self.enc.lat_h * self.enc.lat_w))
