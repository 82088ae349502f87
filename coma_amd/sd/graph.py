"""Static launch graphs: a network is compiled ONCE into a flat list of kernel launches over pre-allocated NHWC fp16 buffers (no
allocation, no host sync inside).  The list is RECORDED into a library-owned model (coma_amd/csrc/sd_plan.hip, sd/model.py): the
launch list and the hipGraph captured from it live in libcoma_hip.so, a denoising step is one sd_model_replay, and the model can
be saved to a file a caller without Python loads and runs (sd_model_load / sd_unet_forward).  The Python closures are kept for
per-launch profiling (`run`, `profile`).  This replaces the per-op Python dispatch of the reference's diffusers modules -- the
MI355X-first equivalent of a tracing compiler is "hipGraph over hand-written kernels", not op-by-op eager execution.
"""
from __future__ import annotations

import torch

from . import ops
from .model import BUF_ZEROED, SdModel

F16 = torch.float16
_TORCH_GRAPH = __import__("os").environ.get("SD_TORCH_GRAPH") == "1"
# GroupNorm statistics ride on the producer's epilogue from this many output rows on (below, the GEMMs are split-K launches whose reduce pass
# owns the epilogue, and a one-launch GroupNorm is as cheap as the finalize + apply pair); SD_GN_STATS_MIN_M=<rows> for A/B runs
GN_STATS_MIN_M = int(__import__("os").environ.get("SD_GN_STATS_MIN_M", 16384))


class LaunchGraph:
    def __init__(self, device, model=None, plan="step"):
        self.device = torch.device(device)
        self.model = model if model is not None else SdModel(device)     # several graphs may share one model (UNet: step + context)
        self.plan = plan
        self._recorded = False
        self.launches = []          # zero-argument closures
        self.tags = []              # (description, flops) per launch, for profiling
        self.alg_bytes = []         # algorithmic HBM bytes per launch (inputs read once + output written once)
        self.flops = 0              # ALGORITHMIC flops per run: 2*M*N*K of every GEMM-shaped operator (a Winograd convolution counts as the
                                    # 3x3 convolution it computes, 2 * 9 * M * N * C_in)
        self.exec_flops = 0         # MFMA flops actually issued (a Winograd convolution: 16 plane products = 4/9 of the above)
        self.exec_tags = []         # executed flops per launch, parallel to `tags`
        self._gn_stats = None
        self._ws = None             # split-K workspace shared by every GEMM of the graph (launches are serial)
        self._colstats = {}         # data_ptr of a GEMM output -> its [M/32][2][N] column-sum buffer (GroupNorm statistics)
        self._colstats_tile = {}    # data_ptr of a halo-convolution output -> its [M/256][2][N] per-tile column sums (table consumers only)
        self.fuse_gn_stats = True

    # ---- memory
    def buf(self, *shape, dtype=F16, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.device)
        self.model.register(t, BUF_ZEROED if zero else 0)                # scratch: not part of a saved model's contents
        return t

    def gn_scratch(self, batch, hw):
        n = ops.gn_scratch_floats(batch, hw)
        if self._gn_stats is None or self._gn_stats.numel() < n:
            self._gn_stats = self.buf(max(n, 1 << 16), dtype=torch.float32)
        return self._gn_stats

    # ---- recording
    def add(self, fn, flops=0, tag="", nbytes=0, alg_flops=None):
        alg = flops if alg_flops is None else alg_flops
        self.launches.append(fn)
        self.tags.append((tag, alg))
        self.exec_tags.append(flops)
        self.alg_bytes.append(nbytes)
        self.flops += alg
        self.exec_flops += flops

    def conv(self, a0, w, out, *, batch, in_h, in_w, c0, n, out_h=None, out_w=None, a1=None, c1=0, taps=1, **kw):
        oh = out_h if out_h is not None else in_h
        ow = out_w if out_w is not None else in_w
        z = kw.get("nbatch_z", 1)
        if self._ws is None:
            self._ws = self.buf(16 << 20, dtype=torch.float32)   # 64 MiB
        kw.setdefault("workspace", self._ws)
        # GroupNorm statistics of the consumer come for free from the epilogue of large, never-split GEMMs
        M = batch * oh * ow
        alg_flops = kw.pop("alg_flops", None)
        tag_note = kw.pop("tag_note", "")
        if kw.pop("stats", False) and self.fuse_gn_stats and z == 1 and M >= GN_STATS_MIN_M and M % 32 == 0:
            cs = self.buf(M // 32, 2, n, dtype=torch.float32, zero=True)
            kw["colstats"] = cs
            self._colstats[out.data_ptr()] = cs
        self.add(lambda: ops.conv_gemm(a0, w, out, batch=batch, in_h=in_h, in_w=in_w, out_h=oh, out_w=ow, c0=c0, n=n, a1=a1,
                                       c1=c1, taps=taps, **kw),
                 flops=2 * batch * oh * ow * n * taps * (c0 + c1) * z, alg_flops=alg_flops,
                 # unique bytes: the input activation(s), the weights (shared over z unless strided), bias / residual, the output
                 nbytes=2 * (z * batch * in_h * in_w * (c0 + c1) + (z if kw.get("stride_w") else 1) * n * taps * (c0 + c1)
                             + z * M * (n // 2 if kw.get("epi", 0) & ops.EPI_GEGLU else n) * (2 if kw.get("res") is not None else 1)
                             + (n if kw.get("bias") is not None else 0) + (batch * n if kw.get("bias_bn") is not None else 0)),
                 tag=f"gemm M={batch * oh * ow} N={n} K={taps * (c0 + c1)} taps={taps} z={z}" + tag_note)
        return out

    def conv3x3_upsampled(self, a0, w_raw, out, *, batch, in_h, in_w, c0, n, bias=None, stats=False):
        """diffusers Upsample2D: conv3x3(nearest-upsample-x2(a0)) as FOUR sub-pixel phase products over the source (sd_conv_gemm_desc.phase):
        16 multiplies per output 2 x 2 block instead of 36, exact (the folded weights are sums of the 3x3 taps), no transformed tensors.
        w_raw: the torch-layout weight [n, c0, 3, 3]; out: [batch * 2 in_h * 2 in_w, n]."""
        from .weights import upsample_phase_weights
        assert in_w & (in_w - 1) == 0, "phase launches need a power-of-two source width"
        M = batch * in_h * in_w
        ws = upsample_phase_weights(w_raw)
        cs = None
        if stats and self.fuse_gn_stats and 4 * M >= GN_STATS_MIN_M and (in_h * in_w) % 32 == 0:
            cs = self.buf(4 * M // 32, 2, n, dtype=torch.float32, zero=True)          # 4 M / 32 slots: phase p of sample b owns a quarter of b's range
            self._colstats[out.data_ptr()] = cs
        for ph in range(4):
            self.conv(a0, ws[ph], out, batch=batch, in_h=in_h, in_w=in_w, c0=c0, n=n, taps=4, phase=ph + 1, bias=bias, colstats=cs,
                      alg_flops=2 * M * n * 9 * c0, tag_note=f" (upsample phase {ph})")
        return out

    # ---- Winograd F(2x2,3x3) for the deep ResNet levels (profiles/r04_notes.md 1, 4): input transform -> 16 plane products (the 1x1 GEMM
    # path, nbatch_z = 16) -> output transform with the epilogue.  2.25 x fewer MFMA flops; only worth it where the 4 x larger transformed
    # tensors stay in the Infinity Cache and K = C_in is long -- the 16 x 16 / 8 x 8 levels of the UNet.
    # fp16 headroom of the STORED planes (sd_hip.h "fp16 range"): U carries 1/4, and V another 1/4 where the input is an un-normalised
    # tensor (the Upsample2D convolutions: V alone reaches 4 max|d|); the output transforms multiply by 4 / 16 in fp32 -- exact.
    WINO_USCALE = 0.25
    WINO_VSCALE_RAW = 0.25

    def winograd_weight(self, w9, *, n, c):
        """w9 fp16 [n][9 * c] (the direct kernel's layout) -> U fp16 [16][n][c] = WINO_USCALE * G g G^T, computed once, here."""
        U = torch.empty(16, n, c, dtype=torch.float16, device=self.device)        # constants: registered PERSISTENT when first recorded
        ops.winograd_weight(w9, U, n=n, c=c, uscale=self.WINO_USCALE)
        return U

    def winograd_planes(self, V, U, *, tiles, c, n):
        P = self.buf(16, tiles, n)
        self.conv(V, U, P, batch=tiles, in_h=1, in_w=1, c0=c, n=n, nbatch_z=16, stride_a=tiles * c, stride_w=n * c, stride_out=tiles * n,
                  alg_flops=2 * 9 * (4 * tiles) * n * c, tag_note=" (winograd planes)")      # counted as the 3x3 convolution these 16 products compute
        return P

    def winograd_input(self, a0, *, batch, h, w, c0, a1=None, c1=0, upsample=False, vscale=1.0):
        C, T = c0 + c1, batch * (h // 2) * (w // 2)
        V = self.buf(16, T, C)
        self.add(lambda: ops.winograd_input(a0, V, batch=batch, h=h, w=w, c0=c0, x1=a1, c1=c1, upsample=upsample, vscale=vscale),
                 tag=f"winograd input{' (upsampled)' if upsample else ''} B={batch} {h}x{w} C={C}", nbytes=2 * 5 * batch * h * w * C)
        return V

    GN_WINO_MAX_SLICE = 20480

    def gn_winograd_input(self, gamma, beta, *, batch, h, w, c0, x0=None, x1=None, c1=0, m=None, bias=None, bias_bn=None, ldbb=0, eps, silu=True,
                          groups=32, vscale_of_m=1.0):
        """GroupNorm (+ SiLU) -> Winograd input transform in ONE launch (sd_gn_winograd_input_f16); the source is [x0 | x1] or the plane
        products `m` of the previous Winograd convolution (its output transform, bias and per-sample bias happen here)."""
        C, T = c0 + c1, batch * (h // 2) * (w // 2)
        assert h * w * (C // groups) <= self.GN_WINO_MAX_SLICE
        V = self.buf(16, T, C)
        self.add(lambda: ops.gn_winograd_input(V, gamma, beta, batch=batch, h=h, w=w, c0=c0, x0=x0, x1=x1, c1=c1, m=m, ldm=c0, bias=bias,
                                               bias_bn=bias_bn, ldbb=ldbb, groups=groups, eps=eps, silu=silu, mscale=1.0 / (self.WINO_USCALE * vscale_of_m)),
                 tag=f"groupnorm + winograd input{' (from planes)' if m is not None else ''} B={batch} {h}x{w} C={C}",
                 nbytes=2 * (5 + (16 if m is not None else 1)) * batch * h * w * C // (4 if m is not None else 1))
        return V

    def winograd_output(self, P, out, *, batch, h, w, n, bias=None, bias_bn=None, ldbb=0, res=None, stats=False, vscale=1.0):
        """stats: also leave the column sums of `out` for the consumer's GroupNorm (w = 32 only: one image row = one 32-row slot).
        vscale: the scale the producer of V applied (the planes are WINO_USCALE * vscale times the true products)."""
        mscale = 1.0 / (self.WINO_USCALE * vscale)
        cs, M = None, batch * h * w
        if stats and self.fuse_gn_stats and w == 32 and n % 128 == 0 and M >= GN_STATS_MIN_M:
            cs = self.buf(M // 32, 2, n, dtype=torch.float32, zero=True)
            self._colstats[out.data_ptr()] = cs
        self.add(lambda: ops.winograd_output(P, out, batch=batch, h=h, w=w, n=n, bias=bias, bias_bn=bias_bn, ldbb=ldbb, res=res, colstats=cs, mscale=mscale),
                 tag=f"winograd output{' (+ colstats)' if cs is not None else ''} B={batch} {h}x{w} N={n}",
                 nbytes=2 * (5 + (1 if res is not None else 0)) * batch * h * w * n)
        return out

    def gn_table_winograd_input(self, x0, gamma, beta, *, batch, h, w, c0, x1=None, c1=0, eps, silu=True):
        """GroupNorm (+ SiLU) folded into the input transform through the per-(sample, channel) affine table, for slices too large for
        gn_winograd_input: needs the column sums of every source (left by its producer); returns None when one is missing.  Two launches
        (table, transform), and the normalised tensor is never written."""
        hw = h * w
        cs0 = self._colstats.get(x0.data_ptr()) if hw % 32 == 0 else None
        cs1 = self._colstats.get(x1.data_ptr()) if (x1 is not None and hw % 32 == 0) else None
        if cs0 is None or (x1 is not None and cs1 is None):
            return None
        table = self.gn_scratch(batch, hw)
        C, T = c0 + c1, batch * (h // 2) * (w // 2)
        self.add(lambda: ops.groupnorm_table_cat(gamma, beta, table, cs0, cs1, batch=batch, hw=hw, c0=c0, c1=c1, eps=eps),
                 tag=f"groupnorm(table) B={batch} hw={hw} C={C}")
        V = self.buf(16, T, C)
        self.add(lambda: ops.winograd_input(x0, V, batch=batch, h=h, w=w, c0=c0, x1=x1, c1=c1, gn_affine=table, silu=silu),
                 tag=f"groupnorm apply + winograd input B={batch} {h}x{w} C={C}", nbytes=2 * 5 * batch * h * w * C)
        return V

    def conv3x3_winograd(self, a0, w9, out, *, batch, in_h, in_w, c0, n, a1=None, c1=0, bias=None, bias_bn=None, ldbb=0, res=None, upsample=False,
                         stats=False):
        """The unfused chain: input transform -> plane products -> output transform (+ bias, per-sample bias, residual).  in_h, in_w are
        the convolution's own (= output) resolution; with upsample the sources are [in_h / 2, in_w / 2]."""
        C, T = c0 + c1, batch * (in_h // 2) * (in_w // 2)
        vs = self.WINO_VSCALE_RAW       # the caller's tensor is not known to be normalised (the UNet's Upsample2D convolutions use this chain)
        V = self.winograd_input(a0, batch=batch, h=in_h, w=in_w, c0=c0, a1=a1, c1=c1, upsample=upsample, vscale=vs)
        P = self.winograd_planes(V, self.winograd_weight(w9, n=n, c=C), tiles=T, c=C, n=n)
        return self.winograd_output(P, out, batch=batch, h=in_h, w=in_w, n=n, bias=bias, bias_bn=bias_bn, ldbb=ldbb, res=res, stats=stats, vscale=vs)

    def dup(self, src, dst):
        """dst = [src | src] along the batch axis (two identical CFG halves); the GroupNorm column sums of src follow."""
        assert dst.numel() == 2 * src.numel()
        self._dup(src, dst, tag=f"dup {src.numel() * 2 >> 20} MiB")
        cs = self._colstats.get(src.data_ptr())
        if cs is not None:
            cs2 = self.buf(2 * cs.shape[0], *cs.shape[1:], dtype=cs.dtype, zero=True)
            self._dup(cs, cs2, tag="dup colstats")
            self._colstats[dst.data_ptr()] = cs2
        return dst

    def _dup(self, src, dst, tag):
        """dst = [src | src]: two device-to-device copies (recordable, unlike a torch copy_)."""
        half = dst.view(2, -1)
        self.add(lambda: (ops.copy_d2d(half[0], src.view(-1)), ops.copy_d2d(half[1], src.view(-1))), tag=tag)

    def groupnorm(self, x0, gamma, beta, out, *, batch, hw, c0, x1=None, c1=0, eps, silu):
        stats = self.gn_scratch(batch, hw)
        cs0 = self._colstats.get(x0.data_ptr())
        cs1 = self._colstats.get(x1.data_ptr()) if x1 is not None else None
        if cs0 is not None and (x1 is None or cs1 is not None) and hw % 32 == 0:
            self.add(lambda: ops.groupnorm_colstats(x0, gamma, beta, out, stats, cs0, batch=batch, hw=hw, c0=c0, x1=x1, c1=c1,
                                                    colstats1=cs1, eps=eps, silu=silu),
                     tag=f"groupnorm(colstats) B={batch} hw={hw} C={c0 + c1}")
        else:
            self.add(lambda: ops.groupnorm(x0, gamma, beta, out, stats, batch=batch, hw=hw, c0=c0, x1=x1, c1=c1, eps=eps, silu=silu),
                     tag=f"groupnorm B={batch} hw={hw} C={c0 + c1}")
        return out

    def layernorm(self, x, gamma, beta, out, *, rows, c):
        self.add(lambda: ops.layernorm(x, gamma, beta, out, rows=rows, c=c), tag=f"layernorm rows={rows} C={c}")
        return out

    def attention(self, q, k, vt, out, *, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, vt_perm16=False):
        self.add(lambda: ops.attention(q, k, vt, out, batch=batch, heads=heads, lq=lq, lk=lk, d=d, ldq=ldq, ldk=ldk, ldv=ldv,
                                       ldo=ldo, scale=d ** -0.5, vt_perm16=vt_perm16),
                 flops=4 * batch * heads * lq * lk * d, tag=f"attention B={batch} h={heads} lq={lq} lk={lk} d={d}")
        return out

    def xattn_chain(self, a, h, wo1, bo1, g2, b2, wq, k2, vt2, wo2, bo2, g3, b3, h2, n3, *, rows, rows_per_sample, lk, ldv2):
        c = 320
        self.add(lambda: ops.xattn_chain(a, h, wo1, bo1, g2, b2, wq, k2, vt2, wo2, bo2, g3, b3, h2, n3, rows=rows, rows_per_sample=rows_per_sample,
                                         lk=lk, ldv2=ldv2),
                 flops=3 * 2 * rows * c * c + 4 * rows * lk * c, tag=f"xchain rows={rows} C={c} lk={lk}", nbytes=2 * (4 * rows * c + 3 * c * c))
        return h2, n3

    def xfront(self, x, gamma, beta, wpi, bpi, g1, b1, wqk, wv, h, qk, vt, *, batch, hw, gn_eps):
        """GroupNorm table (from the producer's column sums when it left them) + the fused front of a C = 320 transformer block."""
        c = 320
        stats = self.gn_scratch(batch, hw)
        cs0 = self._colstats.get(x.data_ptr()) if hw % 32 == 0 else None
        self.add(lambda: ops.groupnorm_table(x, gamma, beta, stats, batch=batch, hw=hw, c0=c, eps=gn_eps, colstats0=cs0),
                 tag=f"groupnorm(table) B={batch} hw={hw} C={c}")
        rows = batch * hw
        self.add(lambda: ops.xfront(x, stats, wpi, bpi, g1, b1, wqk, wv, h, qk, vt, rows=rows, rows_per_sample=hw, ldv=vt.shape[-1]),
                 flops=4 * 2 * rows * c * c, tag=f"xfront rows={rows} C={c}", nbytes=2 * (5 * rows * c + 4 * c * c))
        return h, qk, vt

    def xtail(self, n3, h2, x, w1, b1, w2, b2, wpo, bpo, out, *, rows):
        c = 320
        cs = None
        if self.fuse_gn_stats and rows >= GN_STATS_MIN_M:        # the next GroupNorm takes its statistics from these column sums
            cs = self.buf(rows // 32, 2, c, dtype=torch.float32, zero=True)
            self._colstats[out.data_ptr()] = cs
        self.add(lambda: ops.xtail(n3, h2, x, w1, b1, w2, b2, wpo, bpo, out, cs, rows=rows),
                 flops=2 * rows * c * (8 * c + 4 * c + c), tag=f"xtail rows={rows} C={c}", nbytes=2 * (4 * rows * c + 13 * c * c))
        return out

    def gn_silu_conv3x3_small_n(self, x, gamma, beta, w, bias, out, *, batch, h, w_, c, n, eps, silu=True):
        """GroupNorm (+ SiLU) -> 3x3 convolution with n <= 4 output channels in one pass over x: only the per-(sample, channel) affine
        table is computed (statistics from the producer's column sums when it left them), the normalised tensor is never written."""
        hw = h * w_
        stats = self._table(x, gamma, beta, batch=batch, hw=hw, c=c, eps=eps)
        self.add(lambda: ops.conv3x3_small_n(x, w, out, batch=batch, h=h, w_=w_, c=c, n=n, bias=bias, gn_affine=stats, silu=silu,
                                             ldo=out.shape[-1]),
                 flops=2 * batch * hw * n * 9 * c, tag=f"conv3x3(small n) B={batch} {h}x{w_} C={c} n={n}", nbytes=2 * batch * hw * (c + 8))
        return out

    def _table(self, x, gamma, beta, *, batch, hw, c, eps):
        """The (scale, shift) table of a GroupNorm over x, from whatever column sums its producer left (32-row slots of a GEMM epilogue,
        per-tile slots of a halo convolution) or from a statistics pass."""
        table = self.gn_scratch(batch, hw)
        cs0, rps = self._colstats_tile.get(x.data_ptr()), 256
        if cs0 is None:
            cs0, rps = (self._colstats.get(x.data_ptr()) if hw % 32 == 0 else None), 32
        self.add(lambda: ops.groupnorm_table(x, gamma, beta, table, batch=batch, hw=hw, c0=c, eps=eps, colstats0=cs0, rows_per_slot=rps),
                 tag=f"groupnorm(table) B={batch} hw={hw} C={c}")
        return table

    def gn_silu_conv3x3_halo(self, x, gamma, beta, w, bias, out, *, batch, h, w_, c, n, eps, silu=True, res=None, stats=False):
        """GroupNorm (+ SiLU) -> 3x3 convolution with 128 output channels as a halo-patch convolution (sd_conv3x3_halo_f16): only the
        per-(sample, channel) affine table is computed (statistics from the producer's column sums when it left them), the normalised
        tensor is never written; stats: leave the per-tile column sums of `out` for the next GroupNorm table."""
        hw = h * w_
        table = self._table(x, gamma, beta, batch=batch, hw=hw, c=c, eps=eps)
        cs = None
        if stats and self.fuse_gn_stats:
            cs = self.buf(batch * hw // 256, 2, n, dtype=torch.float32, zero=True)
            self._colstats_tile[out.data_ptr()] = cs
        self.add(lambda: ops.conv3x3_halo(x, w, out, batch=batch, h=h, w_=w_, c=c, n=n, bias=bias, res=res, gn_affine=table, silu=silu,
                                          colstats=cs, ldo=out.shape[-1]),
                 flops=2 * batch * hw * n * 9 * c, tag=f"conv3x3(halo) B={batch} {h}x{w_} C={c} n={n}",
                 nbytes=2 * batch * hw * (c + n * (2 if res is not None else 1)) + 2 * n * 9 * c)
        return out

    def attention_wide(self, q, k, vt, out, *, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo):
        self.add(lambda: ops.attention_wide(q, k, vt, out, batch=batch, heads=heads, lq=lq, lk=lk, d=d, ldq=ldq, ldk=ldk, ldv=ldv, ldo=ldo,
                                            scale=d ** -0.5),
                 flops=4 * batch * heads * lq * lk * d, tag=f"attention(wide) B={batch} h={heads} lq={lq} lk={lk} d={d}")
        return out

    # ---- execution
    def run(self):
        for fn in self.launches:
            fn()

    def profile(self, reps=3):
        """Eager per-launch timing with HIP events -> list of (tag, flops, ms); for tuning only."""
        dev = self.device
        self.run()
        torch.cuda.synchronize(dev)
        out = []
        for fn, (tag, fl) in zip(self.launches, self.tags):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize(dev)
            out.append((tag, fl, a.elapsed_time(b) / reps))
        return out

    def capture(self):
        """Record the launch list into the library-owned plan (once), run it eagerly once (module loading and argument checks happen
        outside any capture); the library captures its hipGraph on the first replay."""
        if not self._recorded:
            def record_checked():
                # every closure must end up in the plan: one that calls no sd_* entry point (a stray torch op) would execute now,
                # during recording, and be missing from every replay and from a saved model
                for fn, (tag, _) in zip(self.launches, self.tags):
                    n0 = self.model.num_launches(self.plan)
                    fn()
                    if self.model.num_launches(self.plan) <= n0:
                        raise RuntimeError(f"launch '{tag}' of plan '{self.plan}' recorded nothing: only sd_* entry points may be added to a LaunchGraph")
            assert len(self.tags) == len(self.launches)
            self.model.record(self.plan, record_checked)
            self._recorded = True
            self.model.run(self.plan)
            torch.cuda.synchronize(self.device)
        return self.model

    def run_recorded(self):
        """The recorded list launched natively one by one (no Python per launch, no graph)."""
        self.capture()
        self.model.run(self.plan)

    def replay(self):
        if _TORCH_GRAPH:                          # A/B aid (SD_TORCH_GRAPH=1): torch.cuda.CUDAGraph over the Python closures
            if getattr(self, "_tg", None) is None:
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    self.run()
                torch.cuda.current_stream(self.device).wait_stream(s)
                torch.cuda.synchronize(self.device)
                self._tg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._tg):
                    self.run()
            return self._tg.replay()
        if not self._recorded:
            self.capture()            # recorded, then run eagerly once: that run IS this call's execution (no second pass over the step)
            self.model.prepare(self.plan)      # the hipGraph is captured and instantiated now (nothing executes), so the next call only launches
            return
        self.model.replay(self.plan)
