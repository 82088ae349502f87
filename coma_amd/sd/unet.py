"""SD-1.5-inpainting UNet on MI355X: the module the reference pipeline calls as
``self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=None,
return_dict=False)[0]`` (utils/adaptive_mask_inpainting.py:1001-1007), rebuilt as a static launch graph of
hand-written gfx950 kernels (coma_amd/csrc/sd_*.hip) over NHWC fp16 buffers.

Graph (public SD-1.5 architecture; diffusers itself is third party):
conv_in -> 3 x [Res, Transformer] x2 + Down -> [Res x2] -> mid [Res, Transformer, Res] -> up blocks with skip
concatenation -> GroupNorm/SiLU -> conv_out.  Fusions done here, none of which changes results beyond fp16 rounding:
  * torch.cat of skip connections is never materialised (two-source GroupNorm / conv / shortcut GEMM);
  * time-embedding projections of all 22 ResNet blocks are ONE GEMM; SiLU(temb) is the epilogue of linear_2;
  * time-embedding bias, conv bias and the residual add are conv epilogues; GEGLU is the epilogue of ff.net.0;
  * Q and K projections are one GEMM; V is produced transposed by swapping the GEMM operands (what the
    attention kernel wants), so there is no transpose or head-split copy anywhere;
  * nearest-x2 upsampling and stride-2 downsampling are index arithmetic inside the conv's operand gather;
  * cross-attention K/V of the text context are computed once per prompt, not once per step.
Measured and removed (A/B history in profiles/r02_notes.md 6, r04_notes.md 4-7, r05_notes.md): the LayerNorms folded algebraically into
their consumer GEMMs (0.2-0.7 ms SLOWER per forward in three rounds of A/B), Upsample2D as four sub-pixel phase products inside the UNet
(slower; the VAE decoder uses them), and the run-time switches of the Winograd sub-features, whose winning settings are now code.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .graph import F16, LaunchGraph
from .weights import UNET_CFG, conv_weight, geglu_interleave, pad_vec


class _Cfg(dict):
    __getattr__ = dict.__getitem__


GN_WINOGRAD_MIN_CG = 40     # GroupNorm folded into the Winograd input transform only for groups of >= 40 channels: a workgroup owns one (sample, group)
                            # slice, and with 20-channel groups (C = 640) its 40-byte pieces of every (plane, tile) row waste most of each memory
                            # transaction (48-114 us per launch at the 32 x 32 level against 18 us at C = 1280, profiles/r04_notes.md 4)
XTAIL_MIN_ROWS = 32768      # the fused feed-forward tail (sd_xtail_f16, 128-row tiles) from one workgroup per CU on: at UNet batch 2 (8192 rows, 64
                            # workgroups) the three separate GEMMs are 2.1 % of a forward faster (7.13 -> 6.97 ms, profiles/r06_notes.md 6), at batch 16 the
                            # fused launch wins (r5)
WINOGRAD_MAX_H = 32         # ResNet 3x3 convolutions of feature maps up to 32 x 32 with >= 640 channels run as Winograd F(2x2,3x3).  Measured inside
                            # the captured batch-16 forward (A B A B on one box, profiles/r04_notes.md 4): 19.02 ms direct, 18.18 ms with the
                            # 16 x 16 / 8 x 8 levels, 18.01 ms with the 32 x 32 level as well; the 64 x 64 level (C = 320) loses.


class HipUNet2DConditionModel:
    def __init__(self, state, batch, height=64, width=64, ctx_len=77, device="cuda", cfg=UNET_CFG, use_graph=True,
                 cfg_shared_prefix=False, fuse_xchain=True, fuse_xfront=True, fuse_xtail=True, fuse_qkv=True, winograd_max_h=None,
                 winograd_min_batch=8, xtail_min_rows=XTAIL_MIN_ROWS):
        """cfg_shared_prefix: the caller guarantees that the two halves of the batch carry IDENTICAL sample and timestep
        (classifier-free guidance: [uncond | cond] differ only in the text context, utils/adaptive_mask_inpainting.py:990).
        Everything before the first cross-attention (conv_in, the first ResNet block, the first self-attention) then
        sees identical inputs in both halves; it is computed once on batch/2 and duplicated -- bit-identical outputs."""
        self.cfgd = cfg
        self.config = _Cfg(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], sample_size=height)
        self.device = torch.device(device)
        self.batch, self.H, self.W, self.ctx_len = batch, height, width, ctx_len
        self.heads = cfg["heads"]
        self.ctx_dim = cfg["cross_attention_dim"]
        self.use_graph = use_graph
        self.fuse_xchain = fuse_xchain      # C = 320 blocks: attn1.to_out ... norm3 in one launch (sd_xattn_chain_f16)
        self.fuse_xfront = fuse_xfront      # C = 320 blocks: norm, proj_in, norm1, to_q | to_k, to_v^T in one launch (sd_xfront_f16)
        self.fuse_qkv = fuse_qkv            # C = 640 / 1280 blocks: to_q | to_k | to_v one GEMM, V^T written transposed by its epilogue (sd_conv_gemm_desc.out_t)
        self.fuse_xtail = fuse_xtail        # C = 320 blocks: ff (GEGLU, Linear) + residual, proj_out + residual in one launch (sd_xtail_f16)
        # ResNet 3x3 convolutions of feature maps up to this edge run as Winograd F(2x2,3x3) (0 = never; SD_WINOGRAD=<edge> for A/B runs), from
        # this UNet batch on (at batch 2 -- one image per call -- the plane products are a few tiles each and the direct form is 0.7 % faster)
        self.winograd_max_h = int(os.environ.get("SD_WINOGRAD", WINOGRAD_MAX_H)) if winograd_max_h is None else int(winograd_max_h)
        self.winograd_min_batch = int(winograd_min_batch)
        self.xtail_min_rows = int(xtail_min_rows)
        self.cfg_shared_prefix = bool(cfg_shared_prefix) and batch % 2 == 0 and cfg["down_has_attn"][0]
        self._B = batch                              # batch the block builders currently emit launches for
        self.dtype = F16
        self.s = {k: v.to(self.device, F16) for k, v in state.items()}
        self.g = LaunchGraph(self.device, plan="step")                              # per-step graph
        self.gc = LaunchGraph(self.device, model=self.g.model, plan="context")      # per-prompt graph (cross-attention K / V^T of the text context)
        B, HW = batch, height * width
        # static inputs / outputs
        self.x_in = self.g.buf(B, HW, 64, zero=True)              # 9 valid channels (latents|mask|masked latents)
        self.timesteps = self.g.buf(B, dtype=torch.float32, zero=True)
        self.ctx = self.g.buf(B, ctx_len, self.ctx_dim, zero=True)
        self.eps = None
        self._build()
        # the names sd_unet_forward / sd_unet_set_context look up (include/sd_hip.h)
        for name, t in (("x_in", self.x_in), ("timesteps", self.timesteps), ("ctx", self.ctx), ("eps", self.eps)):
            self.g.model.bind(name, t)

    # ------------------------------------------------------------------ graph construction
    def _build(self):
        g, s, B = self.g, self.s, self.batch
        ch = self.cfgd["block_out_channels"]
        temb_dim = 4 * ch[0]
        # --- time embedding: sinusoid -> linear_1 + SiLU -> linear_2 (+ SiLU, the only way temb is consumed)
        t_sin = g.buf(B, ch[0])
        g.add(lambda: ops.timestep_embedding(self.timesteps, t_sin, batch=B, dim=ch[0]))
        t1 = g.buf(B, temb_dim)
        g.conv(t_sin, s["time_embedding.linear_1.weight"], t1, batch=B, in_h=1, in_w=1, c0=ch[0], n=temb_dim,
               bias=s["time_embedding.linear_1.bias"], epi=ops.EPI_SILU)
        semb = g.buf(B, temb_dim)
        g.conv(t1, s["time_embedding.linear_2.weight"], semb, batch=B, in_h=1, in_w=1, c0=temb_dim, n=temb_dim,
               bias=s["time_embedding.linear_2.bias"], epi=ops.EPI_SILU)
        # --- all ResNet time projections in one GEMM
        pref = sorted(k[:-len(".time_emb_proj.weight")] for k in s if k.endswith(".time_emb_proj.weight"))
        self._tb_off, off = {}, 0
        for p in pref:
            self._tb_off[p] = off
            off += s[p + ".time_emb_proj.weight"].shape[0]
        self._tb_ld = off
        w_all = torch.cat([s[p + ".time_emb_proj.weight"] for p in pref]).contiguous()
        b_all = torch.cat([s[p + ".time_emb_proj.bias"] for p in pref]).contiguous()
        self._tb = g.buf(B, off)
        g.conv(semb, w_all, self._tb, batch=B, in_h=1, in_w=1, c0=temb_dim, n=off, bias=b_all)

        # --- conv_in (9 -> 64 padded input channels)
        H, W = self.H, self.W
        w_in = conv_weight(s["conv_in.weight"], cin_pad=64)
        if self.cfg_shared_prefix:
            self._B = B // 2                         # shared CFG prefix: first half of x_in / timesteps only
        h = g.buf(self._B * H * W, ch[0])
        g.conv(self.x_in, w_in, h, batch=self._B, in_h=H, in_w=W, c0=64, n=ch[0], taps=9, bias=s["conv_in.bias"], stats=True)
        skips = [(g.dup(h, g.buf(B * H * W, ch[0])) if self.cfg_shared_prefix else h, ch[0], H, W)]
        cin = ch[0]
        for i, cout in enumerate(ch):
            for j in range(self.cfgd["layers_per_block"]):
                h = self._resnet(f"down_blocks.{i}.resnets.{j}", h, cin, None, 0, cout, H, W)
                if self.cfgd["down_has_attn"][i]:
                    h = self._transformer(f"down_blocks.{i}.attentions.{j}", h, cout, H, W)   # returns at the full batch
                cin = cout
                skips.append((h, cout, H, W))
            if i < len(ch) - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                o = g.buf(B * (H // 2) * (W // 2), cout)
                g.conv(h, conv_weight(s[p + ".weight"]), o, batch=B, in_h=H, in_w=W, out_h=H // 2, out_w=W // 2, c0=cout,
                       n=cout, taps=9, stride=2, bias=s[p + ".bias"], stats=True)
                h, H, W = o, H // 2, W // 2
                skips.append((h, cout, H, W))
        h = self._resnet("mid_block.resnets.0", h, cin, None, 0, cin, H, W)
        h = self._transformer("mid_block.attentions.0", h, cin, H, W)
        h = self._resnet("mid_block.resnets.1", h, cin, None, 0, cin, H, W)
        for i, cout in enumerate(reversed(ch)):
            for j in range(self.cfgd["layers_per_block"] + 1):
                sk, sc, sh, sw = skips.pop()
                assert (sh, sw) == (H, W)
                h = self._resnet(f"up_blocks.{i}.resnets.{j}", h, cin, sk, sc, cout, H, W)
                if self.cfgd["up_has_attn"][i]:
                    h = self._transformer(f"up_blocks.{i}.attentions.{j}", h, cout, H, W)
                cin = cout
            if i < len(ch) - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                o = g.buf(B * 4 * H * W, cout)
                if self.winograd_max_h and 2 * max(H, W) <= self.winograd_max_h and cout >= 1280 and B >= self.winograd_min_batch:
                    # Upsample2D + conv at the deep levels: the input transform reads the nearest-x2 upsampling in place.  (The four sub-pixel
                    # phase products the VAE decoder uses lose here: 17.55 -> 17.67 ms per forward at M = 1024 ... 16384 source rows.)
                    g.conv3x3_winograd(h, conv_weight(s[p + ".weight"]), o, batch=B, in_h=2 * H, in_w=2 * W, c0=cout, n=cout, bias=s[p + ".bias"],
                                       upsample=True, stats=True)
                else:
                    g.conv(h, conv_weight(s[p + ".weight"]), o, batch=B, in_h=H, in_w=W, out_h=2 * H, out_w=2 * W, c0=cout,
                           n=cout, taps=9, upsample=1, bias=s[p + ".bias"], stats=True)
                h, H, W = o, 2 * H, 2 * W
        self.eps = g.buf(B * H * W, 64, zero=True)                # 4 valid output channels
        nout = s["conv_out.weight"].shape[0]
        if cin == 320 and nout <= 4:
            # conv_norm_out -> SiLU -> conv_out in one pass over the last feature map (sd_conv3x3_small_n_f16): no normalised copy, no
            # 64-column GEMM tile for 4 channels
            g.gn_silu_conv3x3_small_n(h, s["conv_norm_out.weight"], s["conv_norm_out.bias"], conv_weight(s["conv_out.weight"]),
                                      s["conv_out.bias"].contiguous(), self.eps, batch=B, h=H, w_=W, c=cin, n=nout, eps=1e-5, silu=True)
        else:
            gn = g.buf(B * H * W, cin)
            g.groupnorm(h, s["conv_norm_out.weight"], s["conv_norm_out.bias"], gn, batch=B, hw=H * W, c0=cin, eps=1e-5, silu=True)
            g.conv(gn, conv_weight(s["conv_out.weight"], cout_pad=64), self.eps, batch=B, in_h=H, in_w=W, c0=cin, n=64, taps=9,
                   bias=pad_vec(s["conv_out.bias"], 64))

    def _resnet(self, p, x0, c0, x1, c1, cout, H, W):
        g, s, B = self.g, self.s, self._B
        M, cin = B * H * W, c0 + c1
        off = self._tb_off[p]
        # (not below UNet batch 8: at batch 2 -- one image per call -- the plane products are a few tiles each and the direct form is 0.7 % faster)
        wino = self.winograd_max_h and max(H, W) <= self.winograd_max_h and H % 2 == 0 and W % 2 == 0 and min(cin, cout) >= 640 and B >= self.winograd_min_batch
        # deep levels: Winograd F(2x2,3x3), 2.25 x fewer MFMA flops where the transformed tensors stay in cache (profiles/r04_notes.md 1, 4);
        # with the GroupNorms folded into the transforms a block is five launches: [norm1 + B^T d B] -> planes -> [A^T m A + bias + temb,
        # norm2, B^T d B] -> planes -> [A^T m A + bias + shortcut]
        fused_gn = (wino and H * W * (max(cin, cout) // 32) <= g.GN_WINO_MAX_SLICE and cin % 128 == 0 and cout % 128 == 0
                    and min(cin, cout) // 32 >= GN_WINOGRAD_MIN_CG)
        T = B * (H // 2) * (W // 2)
        if fused_gn:
            V1 = g.gn_winograd_input(s[p + ".norm1.weight"], s[p + ".norm1.bias"], batch=B, h=H, w=W, c0=c0, x0=x0, x1=x1, c1=c1, eps=1e-5)
            P1 = g.winograd_planes(V1, g.winograd_weight(conv_weight(s[p + ".conv1.weight"]), n=cout, c=cin), tiles=T, c=cin, n=cout)
            V2 = g.gn_winograd_input(s[p + ".norm2.weight"], s[p + ".norm2.bias"], batch=B, h=H, w=W, c0=cout, m=P1, bias=s[p + ".conv1.bias"],
                                     bias_bn=self._tb.view(-1)[off:], ldbb=self._tb_ld, eps=1e-5)
        elif wino:
            # slices too large (or groups too narrow) for the LDS-resident fusion: GroupNorm through the affine table inside the input
            # transform when the producers left their column sums (the Winograd output transform does, at the 32 x 32 level)
            U1 = g.winograd_weight(conv_weight(s[p + ".conv1.weight"]), n=cout, c=cin)
            V1 = g.gn_table_winograd_input(x0, s[p + ".norm1.weight"], s[p + ".norm1.bias"], batch=B, h=H, w=W, c0=c0, x1=x1, c1=c1, eps=1e-5)
            if V1 is None:
                n1 = g.buf(M, cin)
                g.groupnorm(x0, s[p + ".norm1.weight"], s[p + ".norm1.bias"], n1, batch=B, hw=H * W, c0=c0, x1=x1, c1=c1, eps=1e-5, silu=True)
                V1 = g.winograd_input(n1, batch=B, h=H, w=W, c0=cin)
            h = g.buf(M, cout)
            g.winograd_output(g.winograd_planes(V1, U1, tiles=T, c=cin, n=cout), h, batch=B, h=H, w=W, n=cout, bias=s[p + ".conv1.bias"],
                              bias_bn=self._tb.view(-1)[off:], ldbb=self._tb_ld, stats=True)
            V2 = g.gn_table_winograd_input(h, s[p + ".norm2.weight"], s[p + ".norm2.bias"], batch=B, h=H, w=W, c0=cout, eps=1e-5)
            if V2 is None:
                n2 = g.buf(M, cout)
                g.groupnorm(h, s[p + ".norm2.weight"], s[p + ".norm2.bias"], n2, batch=B, hw=H * W, c0=cout, eps=1e-5, silu=True)
                V2 = g.winograd_input(n2, batch=B, h=H, w=W, c0=cout)
        else:
            n1 = g.buf(M, cin)
            g.groupnorm(x0, s[p + ".norm1.weight"], s[p + ".norm1.bias"], n1, batch=B, hw=H * W, c0=c0, x1=x1, c1=c1, eps=1e-5,
                        silu=True)
            h = g.buf(M, cout)
            g.conv(n1, conv_weight(s[p + ".conv1.weight"]), h, batch=B, in_h=H, in_w=W, c0=cin, n=cout, taps=9,
                   bias=s[p + ".conv1.bias"], bias_bn=self._tb.view(-1)[off:], ldbb=self._tb_ld, stats=True)
            n2 = g.buf(M, cout)
            g.groupnorm(h, s[p + ".norm2.weight"], s[p + ".norm2.bias"], n2, batch=B, hw=H * W, c0=cout, eps=1e-5, silu=True)
        if p + ".conv_shortcut.weight" in s:
            sc = g.buf(M, cout)
            g.conv(x0, conv_weight(s[p + ".conv_shortcut.weight"]), sc, batch=B, in_h=H, in_w=W, c0=c0, n=cout, a1=x1, c1=c1,
                   bias=s[p + ".conv_shortcut.bias"])
        else:
            assert x1 is None and c0 == cout
            sc = x0
        out = g.buf(M, cout)
        if wino:
            P2 = g.winograd_planes(V2, g.winograd_weight(conv_weight(s[p + ".conv2.weight"]), n=cout, c=cout), tiles=T, c=cout, n=cout)
            g.winograd_output(P2, out, batch=B, h=H, w=W, n=cout, bias=s[p + ".conv2.bias"], res=sc, stats=True)
        else:
            g.conv(n2, conv_weight(s[p + ".conv2.weight"]), out, batch=B, in_h=H, in_w=W, c0=cout, n=cout, taps=9,
                   bias=s[p + ".conv2.bias"], res=sc, stats=True)
        return out

    def _transformer(self, p, x, C, H, W):
        g, s, B, heads = self.g, self.s, self._B, self.heads
        L, M, d = H * W, B * H * W, C // heads
        t = p + ".transformer_blocks.0"
        wqk = torch.cat([s[t + ".attn1.to_q.weight"], s[t + ".attn1.to_k.weight"]]).contiguous()
        wv = s[t + ".attn1.to_v.weight"]
        h = g.buf(M, C)
        qk = g.buf(M, 2 * C)
        ldv = (L + 15) // 16 * 16
        vt = g.buf(B, C, ldv, zero=True)         # V^T with the keys of every 16 in the order the attention kernel's MFMA operand wants
        if self.fuse_xfront and C == 320 and L % 64 == 0:
            # everything before the self-attention is local to a token row once the GroupNorm statistics exist: one launch
            g.xfront(x, s[p + ".norm.weight"], s[p + ".norm.bias"], conv_weight(s[p + ".proj_in.weight"]), s[p + ".proj_in.bias"],
                     s[t + ".norm1.weight"], s[t + ".norm1.bias"], wqk, wv, h, qk, vt, batch=B, hw=L, gn_eps=1e-6)
        else:
            gn = g.buf(M, C)
            g.groupnorm(x, s[p + ".norm.weight"], s[p + ".norm.bias"], gn, batch=B, hw=L, c0=C, eps=1e-6, silu=False)
            g.conv(gn, conv_weight(s[p + ".proj_in.weight"]), h, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=s[p + ".proj_in.bias"])
            # ---- self attention
            n1 = g.buf(M, C)
            g.layernorm(h, s[t + ".norm1.weight"], s[t + ".norm1.bias"], n1, rows=M, c=C)
            if self.fuse_qkv and C % 640 == 0 and L % 32 == 0:
                # to_q | to_k | to_v in one launch: the V columns leave transposed per sample in the key order of the attention kernel
                g.conv(n1, torch.cat([wqk, wv]).contiguous(), qk, batch=M, in_h=1, in_w=1, c0=C, n=3 * C, ldo=2 * C, out_t=vt, n_split=2 * C,
                       ldo_t=ldv, rows_per_sample=L)
            else:
                g.conv(n1, wqk, qk, batch=M, in_h=1, in_w=1, c0=C, n=2 * C)
                g.conv(wv, n1, vt, batch=C, in_h=1, in_w=1, c0=C, n=L, ldo=ldv, nbatch_z=B, stride_w=L * C, stride_out=C * ldv,
                       epi=ops.EPI_PERM16_N)
        a = g.buf(M, C)
        g.attention(qk, qk.view(-1)[C:], vt, a, batch=B, heads=heads, lq=L, lk=L, d=d, ldq=2 * C, ldk=2 * C, ldv=ldv, ldo=C, vt_perm16=True)
        Lk, cd = self.ctx_len, self.ctx_dim
        ldv2 = (Lk + 15) // 16 * 16
        xchain = self.fuse_xchain and C == 320 and L % 64 == 0 and Lk <= 96

        def context_kv(Bf):
            """K / V^T of the text context for this block (per-prompt graph)."""
            k2 = self.gc.buf(Bf * Lk, C)
            self.gc.conv(self.ctx, s[t + ".attn2.to_k.weight"], k2, batch=Bf * Lk, in_h=1, in_w=1, c0=cd, n=C)
            vt2 = self.gc.buf(Bf, C, ldv2, zero=True)
            self.gc.conv(s[t + ".attn2.to_v.weight"], self.ctx, vt2, batch=C, in_h=1, in_w=1, c0=cd, n=Lk, ldo=ldv2, nbatch_z=Bf,
                         stride_w=Lk * cd, stride_out=C * ldv2, epi=ops.EPI_PERM16_N)
            return k2, vt2

        if xchain:
            # everything between the self-attention output and the feed-forward is local to a token row: one launch
            if B != self.batch:
                # end of the shared CFG prefix: the two halves meet different text contexts from here on
                B = self._B = self.batch
                M = B * L
                a = g.dup(a, g.buf(M, C))
                h = g.dup(h, g.buf(M, C))
                x = g.dup(x, g.buf(M, C))
            k2, vt2 = context_kv(B)
            h2, n3 = g.buf(M, C), g.buf(M, C)
            g.xattn_chain(a, h, s[t + ".attn1.to_out.0.weight"], s[t + ".attn1.to_out.0.bias"], s[t + ".norm2.weight"], s[t + ".norm2.bias"],
                          s[t + ".attn2.to_q.weight"], k2, vt2, s[t + ".attn2.to_out.0.weight"], s[t + ".attn2.to_out.0.bias"],
                          s[t + ".norm3.weight"], s[t + ".norm3.bias"], h2, n3, rows=M, rows_per_sample=L, lk=Lk, ldv2=ldv2)
        else:
            h1 = g.buf(M, C)
            g.conv(a, s[t + ".attn1.to_out.0.weight"], h1, batch=M, in_h=1, in_w=1, c0=C, n=C, bias=s[t + ".attn1.to_out.0.bias"],
                   res=h)
            if B != self.batch:
                # end of the shared CFG prefix: from the first cross-attention on the two halves differ
                B = self._B = self.batch
                M = B * L
                h1 = g.dup(h1, g.buf(M, C))
                x = g.dup(x, g.buf(M, C))
            # ---- cross attention (K, V^T of the context live in the per-prompt graph)
            q2 = g.buf(M, C)
            n2 = g.buf(M, C)
            g.layernorm(h1, s[t + ".norm2.weight"], s[t + ".norm2.bias"], n2, rows=M, c=C)
            g.conv(n2, s[t + ".attn2.to_q.weight"], q2, batch=M, in_h=1, in_w=1, c0=C, n=C)
            k2, vt2 = context_kv(B)
            a2 = g.buf(M, C)
            g.attention(q2, k2, vt2, a2, batch=B, heads=heads, lq=L, lk=Lk, d=d, ldq=C, ldk=C, ldv=ldv2, ldo=C, vt_perm16=True)
            h2 = g.buf(M, C)
            g.conv(a2, s[t + ".attn2.to_out.0.weight"], h2, batch=M, in_h=1, in_w=1, c0=C, n=C,
                   bias=s[t + ".attn2.to_out.0.bias"], res=h1)
        # ---- feed-forward (GEGLU)
        wff, bff = geglu_interleave(s[t + ".ff.net.0.proj.weight"], s[t + ".ff.net.0.proj.bias"])
        if xchain and self.fuse_xtail and M % 128 == 0 and M >= self.xtail_min_rows:
            # ... and so is everything after it: feed-forward + residual + proj_out + residual in one launch, the hidden tensor never exists
            out = g.buf(M, C)
            g.xtail(n3, h2, x, wff, bff, s[t + ".ff.net.2.weight"], s[t + ".ff.net.2.bias"], conv_weight(s[p + ".proj_out.weight"]),
                    s[p + ".proj_out.bias"], out, rows=M)
            return out
        f = g.buf(M, 4 * C)
        if xchain:
            g.conv(n3, wff, f, batch=M, in_h=1, in_w=1, c0=C, n=8 * C, bias=bff, epi=ops.EPI_GEGLU)
        else:
            n3 = g.buf(M, C)
            g.layernorm(h2, s[t + ".norm3.weight"], s[t + ".norm3.bias"], n3, rows=M, c=C)
            g.conv(n3, wff, f, batch=M, in_h=1, in_w=1, c0=C, n=8 * C, bias=bff, epi=ops.EPI_GEGLU)
        h3 = g.buf(M, C)
        g.conv(f, s[t + ".ff.net.2.weight"], h3, batch=M, in_h=1, in_w=1, c0=4 * C, n=C, bias=s[t + ".ff.net.2.bias"], res=h2)
        out = g.buf(M, C)
        g.conv(h3, conv_weight(s[p + ".proj_out.weight"]), out, batch=M, in_h=1, in_w=1, c0=C, n=C,
               bias=s[p + ".proj_out.bias"], res=x, stats=True)
        return out

    # ------------------------------------------------------------------ execution
    def set_context(self, encoder_hidden_states):
        """[batch, ctx_len, 768]; recomputes the cross-attention K / V^T of every transformer block."""
        self.ctx.copy_(encoder_hidden_states.to(self.device, F16).reshape(self.ctx.shape))
        if self.use_graph:
            self.gc.replay()
        else:
            self.gc.run()

    def forward_static(self):
        """x_in / timesteps already written into the static buffers; result lands in self.eps ([B*HW, 64])."""
        if self.use_graph:
            self.g.replay()                  # sd_model_replay: the library's hipGraph of the recorded launch list
        else:
            self.g.run()
        return self.eps

    def save(self, path):
        """Write the model file (registry + bindings + both plans + weights) that sd_model_load / sd_unet_forward run without Python."""
        self.g.capture()
        self.gc.capture()
        self.g.model.save(path)

    def __call__(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False, **kw):
        """diffusers-compatible call: sample NCHW [batch, 9, H, W] -> (noise_pred NCHW [batch, 4, H, W],)."""
        B, HW = self.batch, self.H * self.W
        assert tuple(sample.shape) == (B, self.config.in_channels, self.H, self.W), sample.shape
        if encoder_hidden_states is not None:
            self.set_context(encoder_hidden_states)
        x = sample.to(self.device, torch.float32).contiguous()
        ops.nchw_to_nhwc(x, self.x_in, batch=B, c=self.config.in_channels, hw=HW, cpad=64)
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        self.timesteps.copy_(t.expand(B) if t.numel() == 1 else t)
        self.forward_static()
        out = torch.empty(B, self.config.out_channels, self.H, self.W, dtype=torch.float32, device=self.device)
        ops.nhwc_to_nchw(self.eps, out, batch=B, c=self.config.out_channels, hw=HW, ld=64)
        out = out.to(sample.dtype) if sample.dtype in (torch.float16, torch.float32) else out
        if return_dict:
            return _Cfg(sample=out)
        return (out,)
