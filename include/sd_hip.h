/*
 * sd_hip.h -- C ABI of the diffusion operators in libcoma_hip.so (MI355X / gfx950, fp16 storage, fp32 accumulate).
 *
 * The reference drives Stable-Diffusion-1.5-inpainting through diffusers modules
 * (utils/adaptive_mask_inpainting.py:1001-1007 `self.unet(...)`, :1086/:1112 `self.vae.decode`, :677-680
 * `self.vae.encode`, :1015-1017 `self.scheduler.step`); diffusers itself is a third-party dependency that is
 * not under the reference tree (pinned diffusers==0.20.2, INSTALL.md:31).  These entry points are the
 * operators those modules decompose into; the Python host mirror (coma_amd/sd) builds the UNet / VAE graphs
 * out of them behind the same call signatures the reference pipeline uses.
 *
 * Layout: every activation is NHWC fp16, i.e. a row-major [batch*H*W, C] matrix, so a 1x1 convolution, an
 * nn.Linear and a token sequence [B, HW, C] are the same buffer.  Weights are [C_out][taps][C_in] fp16
 * (K-contiguous).  Same conventions as coma_hip.h: int return codes, coma_last_error(), caller-owned device
 * buffers, explicit hipStream_t.
 */
#ifndef SD_HIP_H
#define SD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* epilogue flags of sd_conv_gemm_f16 */
#define SD_EPI_NONE 0
#define SD_EPI_GEGLU 1      /* out[:, j] = v_j * gelu(g_j); weight rows pre-interleaved per 64 columns: [32 v | 32 g] */
#define SD_EPI_SILU 2       /* out = silu(acc + bias) */
#define SD_EPI_BIAS_ROWS 4  /* bias indexed by output row instead of column (A = weights, W = activations) */
#define SD_EPI_PERM16_N 8   /* output column j holds the product with W row kappa(j) = j with bits 2 and 3 swapped, i.e. every group of
                               16 columns is stored in the order (0-3, 8-11, 4-7, 12-15): the V^T layout sd_attention_f16 reads with one
                               16-byte LDS load per MFMA operand (vt_perm16 = 1).  n is rounded up to 16 columns (ldo must cover them);
                               columns whose source row is >= n hold a clamped finite row */
#define SD_EPI_PERM32_N 16  /* position p = 8g + e of every group of 32 output columns holds the product with W row 16 (e >> 2) + 4g + (e & 3):
                               the V^T layout sd_attention_wide_f16 reads (one 16-byte LDS load per 16x16x32 MFMA operand).  n % 32 == 0 */
/* bits 20..27 select kernel variants for tuning runs (scripts/time_gemm.py): 20 = generic 128x128 tiles only, 21 = 128x320
 * tile, 22 = tile DMA in one burst, 23 = 4-wave 128x320 tile, 24..27 = forced split-K factor, 28 = tap-major K order for 3x3.  Results agree to
 * fp32 summation order.  Process-wide overrides for A/B runs inside a captured graph (read once): SD_GEMM_TUNE / SD_GEMM_TUNE_1X1 / SD_GEMM_TUNE_3X3 =
 * <mask of these bits>, SD_GEMM_FORCE="N,K,mask;...", SD_GEMM_M16 / SD_GEMM_M16_1X1 = <minimum K for the 16x16x32 K loop, 0 = 32x32x16 everywhere>. */
#define SD_EPI_TUNING_MASK 0x1ff00000

/* out[m, n] = sum_k A[m, k] * W[n, k] (+ epilogue) with A gathered from one or two NHWC sources:
 *   m = (b, oy, ox), k = (tap, ci);  taps = 9: 3x3, zero pad 1;  taps = 1: 1x1 / linear;  taps = 4: one sub-pixel phase (see `phase`)
 *   upsample = 1: the 3x3 window slides over the nearest-x2 upsampling of the [in_h, in_w] input
 *   ci runs over the concatenation [a0 (c0 channels) | a1 (c1 channels)]; c0, c1 multiples of 64
 * replaces: torch.nn.Conv2d / nn.Linear / torch.cat / F.interpolate(nearest) inside diffusers'
 *           UNet2DConditionModel and AutoencoderKL (call sites utils/adaptive_mask_inpainting.py:1001, :1086, :680). */
typedef struct sd_conv_gemm_desc {
  const void* a0;       /* fp16 [batch, in_h, in_w, c0] */
  const void* a1;       /* fp16 [batch, in_h, in_w, c1] or NULL */
  int c0, c1;
  int batch, in_h, in_w, out_h, out_w;
  int taps, stride, upsample;
  int pad;              /* zero padding on the low (top/left) side: 1 for a "same" 3x3, 0 for the VAE encoder's
                           asymmetric (0,1,0,1) stride-2 convolution; the high side is bounds-checked */
  int n;                /* output channels (rows of w) */
  const void* w;        /* fp16 [n][taps*(c0+c1)] */
  const void* bias;     /* fp16 [n] or NULL */
  const void* bias_bn;  /* fp16 [batch][ldbb] or NULL: per-sample bias (time-embedding projection) */
  int ldbb;             /* row stride of bias_bn (0 -> n) */
  const void* res;      /* fp16 [M][ldr] residual added after the activation, or NULL */
  int ldr;
  void* out;            /* fp16 [M][ldo]  (ldo = 0 -> n, or n/2 with GEGLU) */
  int ldo;
  int epi;
  int nbatch_z;         /* >1: independent problems along grid z with the element strides below */
  int64_t stride_a, stride_w, stride_out, stride_res;
  void* workspace;      /* optional fp32 scratch for split-K (small M*N, deep K); NULL disables split-K */
  size_t workspace_bytes;
  float* colstats;      /* optional fp32 [M/32][2][n]: per 32-row block, column sums and sums of squares of the stored output
                           (the GroupNorm statistics of the consumer, sd_groupnorm_colstats_f16 colstats0/1); needs M % 32 == 0 */
  /* Optional second output: columns [n_split, n) of the product leave TRANSPOSED, per sample, in the key order of
   * sd_attention_f16(vt_perm16 = 1): out_t fp16 [M / rows_per_sample][n - n_split][ldo_t], element (b, c, pos) = product row
   * b * rows_per_sample + key(pos), column n_split + c, where every group of 16 positions holds the keys (0-3, 8-11, 4-7, 12-15).
   * Columns [0, n_split) go to `out` as usual (ldo >= n_split).  This is to_q | to_k | to_v of a self-attention in ONE launch: V^T is
   * what the attention kernel's PV product wants as its LDS operand.  Plain linear only (taps = 1, no bias / residual / GEGLU /
   * statistics / batching / split-K); n_split and n - n_split multiples of 640, M and rows_per_sample multiples of 32, ldo_t % 8 == 0. */
  void* out_t;
  int n_split, ldo_t, rows_per_sample;
  /* Sub-pixel phase of `conv3x3(F.interpolate(x, scale_factor=2, mode="nearest"))` (diffusers Upsample2D): output pixel (2y + a, 2x + b) only
   * sees the 2 x 2 source pixels (y - 1 + a .. y + a) x (x - 1 + b .. x + b), each weighted by the SUM of the 3x3 taps that land on it -- 16
   * products per output 2 x 2 block instead of 36, exactly (no transform).  phase = 1 + 2 a + b selects the parity; taps = 4, w fp16
   * [n][4][c0+c1] = the summed weights of that phase (window order (dy, dx) row-major), batch / in_h / in_w = the SOURCE size (out_h = in_h,
   * out_w = in_w, in_w a power of two), `out` = the full [batch, 2 in_h, 2 in_w, ldo] tensor: row m = (b, y, x) of the product is written to
   * pixel (2y + a, 2x + b).  Four launches (phase 1..4) make the convolution; colstats (optional) must have room for 4 M / 32 slots.
   * Plain epilogue only (bias).  0 = off (every other launch; a zero-initialised descriptor). */
  int phase;
} sd_conv_gemm_desc;

int sd_conv_gemm_f16(const sd_conv_gemm_desc* desc, void* stream);
size_t sd_conv_gemm_workspace_bytes(void); /* recommended workspace size */
/* Tuning aid: with SD_GEMM_DBG set in the environment, the first 4096 workgroups of every sd_conv_gemm_f16 launch record the
 * shader clock at {entry, first K tile landed, end of the K loop, end of the epilogue}; this copies [n_blocks][4] u64 stamps of the
 * most recent launch to the host (synchronises the device). */
int sd_debug_timestamps(unsigned long long* host_dst, int n_blocks);

/* GroupNorm (+ optional SiLU) over NHWC fp16, reading the channel concatenation of two sources and writing one
 * tensor [batch, hw, c0+c1].  replaces: nn.GroupNorm(groups, C, eps) + nn.SiLU in diffusers ResnetBlock2D /
 * Transformer2DModel / VAE (and the torch.cat feeding the UNet up blocks).
 * stats: fp32 scratch of batch*(c0+c1)*2 + batch*ceil(hw/64)*groups*2 floats (affine table + partial sums). */
int sd_groupnorm_f16(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                     const void* gamma, const void* beta, int silu, void* out, float* stats, void* stream);
/* Same, but the statistics come from the column sums the producing GEMMs left behind (sd_conv_gemm_desc.colstats of x0
 * and, with two sources, of x1): no statistics pass over the tensor.  hw % 32 == 0. */
int sd_groupnorm_colstats_f16(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                              const void* gamma, const void* beta, int silu, void* out, float* stats, const float* colstats0,
                              const float* colstats1, void* stream);

/* LayerNorm over the last dim of fp16 [rows, c].  replaces: nn.LayerNorm(C) in BasicTransformerBlock. */
int sd_layernorm_f16(const void* x, int64_t rows, int c, float eps, const void* gamma, const void* beta, void* out,
                     void* stream);

/* Fused multi-head attention (online softmax), fp16 in/out, fp32 accumulate.
 *   q   : [batch, lq, ldq]   head h at columns [h*d, (h+1)*d)          (ldq >= heads*d)
 *   k   : [batch, lk, ldk]   same column convention
 *   vt  : [batch, heads*d, ldv]  V TRANSPOSED: row h*d+i holds component i of every key (ldv >= lk rounded up to 8, a
 *         multiple of 8); the pad columns lk..ldv-1 must hold finite values (they meet zero softmax weights).
 *         vt_perm16 = 1: the keys of every group of 16 are stored in the order (0-3, 8-11, 4-7, 12-15) (SD_EPI_PERM16_N of the
 *         producing GEMM), ldv a multiple of 16 -- the kernel then fetches each V^T MFMA operand with ONE conflict-free 16-byte
 *         LDS read instead of two 8-byte ones that collide two-way.  vt_perm16 is a bit field: bit 0 = the layout flag above;
 *         bit 1 = use the software-pipelined kernel wherever it is legal (d = 40, lk a multiple of 64 and >= 128, bit 0 set),
 *         bit 2 = never use it.  By default the library picks it when the launch fills the chip (the 64 x 64 level of the UNet).
 *         Same results up to fp16 rounding of the weights: that kernel pre-multiplies Q by scale*log2(e) in fp16 and carries
 *         the running maximum on fp16 values.
 *   out : [batch, lq, ldo]
 * out = softmax(q k^T * scale) v per (batch, head).  d: any multiple of 8 up to 160.  The K and V^T slices of one
 * (batch, head) must stay below 2 GiB (32-bit LDS-DMA offsets); violations return an error, nothing is launched.
 * replaces: diffusers Attention (xformers / AttnProcessor2_0 scaled_dot_product_attention) in the UNet. */
int sd_attention_f16(const void* q, const void* k, const void* vt, void* out, int batch, int heads, int lq, int lk,
                     int d, int ldq, int ldk, int ldv, int ldo, float scale, int vt_perm16, void* stream);

/* Fused attention for WIDE heads (32 <= d <= 512, d a multiple of 32; the VAE mid-block: one head of 512 over 4096 tokens,
 * diffusers AttentionBlock reached through self.vae.decode / self.vae.encode, utils/adaptive_mask_inpainting.py:1112, :677-680).
 * Same tensors as sd_attention_f16, except: lk must be a multiple of 64 and V^T must be in the SD_EPI_PERM32_N key order
 * (ldv >= lk).  A wave owns 16 queries on v_mfma_f32_16x16x32_f16 (O^T of 512 x 16 is 128 accumulator registers), K and V^T
 * tiles of 64 keys alternate through two single-buffered 64 KB LDS regions.  out = softmax(q k^T * scale) v. */
int sd_attention_wide_f16(const void* q, const void* k, const void* vt, void* out, int batch, int heads, int lq, int lk, int d,
                          int ldq, int ldk, int ldv, int ldo, float scale, void* stream);

/* Only the per-(sample, channel) affine table of a GroupNorm: fp32 [batch][c0][2] = (scale, shift) at the start of `stats` (same
 * scratch size as sd_groupnorm_f16), from the producer's column sums (colstats0 != NULL: fp32 [batch * hw / rows_per_slot][2][c0], rows_per_slot
 * = 32 for sd_conv_gemm_desc.colstats -- 0 means 32 -- and 256 for sd_conv3x3_halo_f16) or from a statistics pass over x0; nothing is
 * applied.  For consumers that apply the affine themselves (sd_xfront_f16, sd_conv3x3_halo_f16, sd_conv3x3_small_n_f16). */
int sd_groupnorm_table_f16(const void* x0, int c0, int batch, int hw, int groups, float eps, const void* gamma, const void* beta,
                           float* stats, const float* colstats0, int rows_per_slot, void* stream);
/* The same table for a GroupNorm over the channel concatenation of two tensors, from both producers' column sums only (hw % 32 == 0). */
int sd_groupnorm_table_cat_f16(int c0, int c1, int batch, int hw, int groups, float eps, const void* gamma, const void* beta, float* stats,
                               const float* colstats0, const float* colstats1, void* stream);

/* The row-local FRONT of a transformer block at C = 320 in ONE launch (five launches of the unfused graph):
 *   n = x * scale + shift (gn_affine [samples][320][2], Transformer2DModel.norm without SiLU);  h = n Wpi^T + bpi (proj_in);
 *   n1 = LayerNorm(h; gamma1, beta1);  qk[:, 0:320] = n1 Wq^T, qk[:, 320:640] = n1 Wk^T (wqk = [Wq ; Wk], fp16 [640, 320]);
 *   vt[sample] = Wv n1^T, fp16 [320, ldv] with the tokens of every 16 in the SD_EPI_PERM16_N order (what sd_attention_f16 reads).
 * x fp16 [rows, 320]; outputs h fp16 [rows, 320], qk fp16 [rows, 640], vt fp16 [samples, 320, ldv]; rows_per_sample a multiple of 64.
 * replaces: Transformer2DModel.norm / proj_in and BasicTransformerBlock.norm1 / attn1.to_q / to_k / to_v inside self.unet(...),
 *           utils/adaptive_mask_inpainting.py:1001-1007. */
int sd_xfront_f16(const void* x, const float* gn_affine, const void* wpi, const void* bpi, const void* gamma1, const void* beta1,
                  const void* wqk, const void* wv, void* h, void* qk, void* vt, int64_t rows, int rows_per_sample, int ldv, float eps,
                  void* stream);

/* The row-local TAIL of a transformer block at C = 320 in ONE launch (three launches and a 4x-wide hidden tensor in the unfused graph):
 *   f = GEGLU(n3 W1^T + b1) (W1 fp16 [2560, 320], b1 [2560]: value / gate rows interleaved per 32 as for SD_EPI_GEGLU);
 *   h3 = f W2^T + b2 + h2 (W2 fp16 [320, 1280]);  out = h3 Wpo^T + bpo + x (Wpo fp16 [320, 320]).
 * n3, h2, x, out: fp16 [rows, 320], rows a multiple of 128.  colstats (optional): fp32 [rows / 32][2][320], per 32-row block the
 * column sums and sums of squares of the stored `out` (the layout of sd_conv_gemm_desc.colstats: the next GroupNorm's statistics).
 * replaces: BasicTransformerBlock.ff (GEGLU, Linear) + residual and Transformer2DModel.proj_out + residual inside self.unet(...),
 *           utils/adaptive_mask_inpainting.py:1001-1007. */
int sd_xtail_f16(const void* n3, const void* h2, const void* x, const void* w1, const void* b1, const void* w2, const void* b2,
                 const void* wpo, const void* bpo, void* out, float* colstats, int64_t rows, void* stream);

/* The row-local middle of a BasicTransformerBlock at C = 320 (8 heads of 40) in ONE launch:
 *   h1 = attn1_out Wo1^T + bo1 + h;  n2 = LayerNorm(h1; gamma2, beta2);  q2 = n2 Wq2^T;
 *   a2 = softmax(q2 K2^T / sqrt(40)) V2 per head over the lk <= 96 text tokens;  h2 = a2 Wo2^T + bo2 + h1;  n3 = LayerNorm(h2; gamma3, beta3)
 * attn1_out, h: fp16 [rows, 320];  weights fp16 [320, 320] (nn.Linear layout);  k2 fp16 [rows / rows_per_sample, lk, 320];
 * vt2 fp16 [samples, 320, ldv2] = V2 transposed, keys in the SD_EPI_PERM16_N order (ldv2 >= 80, pad columns finite);
 * outputs h2, n3: fp16 [rows, 320] (h2 is also used as scratch for h1).  rows_per_sample a multiple of 64.
 * debug_out / debug_stage (tests): when debug_out != NULL the kernel stops after stage 1 (h1), 2 (n2), 3 (q2) or 4 (a2) and
 * writes that [rows, 320] tensor there.
 * replaces: attn1.to_out.0 + residual, norm2, attn2 (to_q, attention, to_out.0 + residual), norm3 of diffusers'
 *           BasicTransformerBlock inside self.unet(...), utils/adaptive_mask_inpainting.py:1001-1007 -- six launches of the
 *           unfused graph (sd_conv_gemm_f16 x3, sd_layernorm_f16 x2, sd_attention_f16). */
int sd_xattn_chain_f16(const void* attn1_out, const void* h, const void* wo1, const void* bo1, const void* gamma2, const void* beta2,
                       const void* wq2, const void* k2, const void* vt2, const void* wo2, const void* bo2, const void* gamma3,
                       const void* beta3, void* h2, void* n3, int64_t rows, int rows_per_sample, int lk, int ldv2, float eps,
                       void* debug_out, int debug_stage, void* stream);

/* Winograd F(2x2,3x3) around a plane-batched GEMM: a 3x3 / stride 1 / pad 1 convolution as 16 independent [T, C_in] x [C_in, C_out]
 * products (T = batch * h/2 * w/2 output tiles) run by sd_conv_gemm_f16 with nbatch_z = 16 -- 2.25 x fewer MFMA flops.
 *   sd_winograd_input_f16   v fp16 [16][T][c0+c1] = B^T d B of every 4x4 input patch (two concatenated NHWC sources, zero pad);
 *                           upsample = 1: [h, w] is the nearest-x2 upsampling of the [h/2, w/2] sources (diffusers Upsample2D + conv);
 *                           gn_affine != NULL: fp32 [batch][c0+c1][2] = (scale, shift) of a GroupNorm applied to the source on the fly
 *                           (+ SiLU if silu), rounded to fp16 as the GroupNorm kernel would have stored it -- the normalised tensor is never written
 *   sd_winograd_weight_f16  u fp16 [16][n][c] = G g G^T of w fp16 [n][9][c] (computed in fp32, rounded once; at prep time)
 *   sd_winograd_output_f16  out fp16 [batch*h*w, ldo] = A^T m A of m fp16 [16][T][ldm] + bias + per-sample bias, SiLU, + residual;
 *                           colstats != NULL (w = 32, n % 128 == 0): also fp32 [batch*h*w/32][2][n], the column sums / sums of squares of
 *                           `out` per 32 rows (sd_conv_gemm_desc.colstats layout: the consumer's GroupNorm statistics)
 * h, w even; channel counts multiples of 8.  sd_winograd_input_f16 / _output_f16 are recordable (sd_winograd_weight_f16 runs at prep time).
 * fp16 range: V and the plane products are STORED as fp16 (the direct path keeps that sum in fp32 registers), so the transforms take
 * power-of-two scales -- v = vscale * B^T d B, u = uscale * G g G^T, out = mscale * A^T m A + bias ... with mscale = 1 / (uscale * vscale):
 * exact in the normal range; the product uses uscale = 1/4 (and vscale = 1/4 where the input is an un-normalised residual stream, V alone
 * reaching 4 max|d|), which keeps the stored planes 4 x / 16 x further from 65504 (tests/test_sd_ops_gpu.py: outputs peaking at 4e3 ... 3e4).
 * replaces: diffusers Conv2d(3x3) inside self.unet(...) / self.vae.decode, utils/adaptive_mask_inpainting.py:1001-1007, :1086, :1112. */
int sd_winograd_input_f16(const void* x0, const void* x1, int c0, int c1, int batch, int h, int w, int upsample, const float* gn_affine, int silu,
                          float vscale, void* v, void* stream);
int sd_winograd_weight_f16(const void* w, int n, int c, float uscale, void* u, void* stream);
int sd_winograd_output_f16(const void* m, int ldm, int batch, int h, int w, int n, const void* bias, const void* bias_bn, int ldbb,
                           const void* res, int ldr, void* out, int ldo, int silu, float mscale, float* colstats, void* stream);
/* GroupNorm (+ SiLU) of a SMALL feature map fused with the Winograd input transform, one workgroup per (sample, group), the group's
 * slice in LDS (h * w * C / groups <= 20480 elements: the 16 x 16 / 8 x 8 levels of the UNet):
 *   source = the channel concatenation [x0 | x1] (NHWC fp16; m == NULL), or
 *   source = mscale * A^T m A + bias + per-sample bias of the 16 plane products m fp16 [16][batch*h/2*w/2][ldm] of the PREVIOUS Winograd
 *            convolution (x0 == x1 == NULL, c0 = its output channels, c1 = 0), rounded to fp16 and never written;
 *   v fp16 [16][batch*h/2*w/2][c0+c1] = B^T act(GroupNorm(source)) B.
 * Replaces sd_winograd_output_f16 -> sd_groupnorm_f16 -> sd_winograd_input_f16 (conv1 -> norm2 -> SiLU -> conv2 of a ResnetBlock2D)
 * or sd_groupnorm_f16 -> sd_winograd_input_f16 (norm1 -> SiLU -> conv1) with one launch.  Recordable. */
int sd_gn_winograd_input_f16(const void* x0, const void* x1, int c0, int c1, const void* m, int ldm, const void* bias, const void* bias_bn,
                             int ldbb, int batch, int h, int w, int groups, float eps, const void* gamma, const void* beta, int silu, float mscale,
                             void* v, void* stream);

/* GroupNorm affine + SiLU folded into a 3x3 / stride 1 / pad 1 convolution with n = 128 ... 512 OUTPUT channels, as a halo-patch ("direct")
 * convolution (coma_amd/csrc/sd_haloconv.hip): the ResNet convolutions of the VAE.  At n = 128 (512 x 512) the implicit GEMM re-stages every
 * activation nine times for only 128 columns; at every width it sits behind a GroupNorm apply pass over the whole tensor.  A 16 x 16 pixel
 * tile is computed by n / 128 workgroups, 128 output channels each.
 *   x fp16 NHWC [batch][h][w][c], c a multiple of 64 up to 512, h and w multiples of 16;
 *   gn_affine fp32 [batch][c][2] = (scale, shift) of the GroupNorm (sd_groupnorm_table_f16), or NULL for a plain convolution;
 *   out[m, 0:n] = conv3x3(act(x * scale + shift)) + bias (+ res[m, 0:n]), act = SiLU if silu, zero padding of the ACTIVATED tensor;
 *   colstats != NULL: fp32 [batch*h*w/256][2][n], sums / sums of squares of the stored output per 16 x 16 pixel tile (one slot per
 *   workgroup: sd_groupnorm_table_f16 with rows_per_slot = 256; 8 x fewer slots than the GEMM epilogue's 32-row slots -- at 512 x 512 the
 *   consumer's table launch read 67 MB of them, 84-120 us).  Recordable.
 * replaces: GroupNorm -> SiLU -> Conv2d(3x3) of ResnetBlock2D inside self.vae.decode / self.vae.encode,
 * utils/adaptive_mask_inpainting.py:1086, :1112, :677-680. */
int sd_conv3x3_halo_f16(const void* x, int c, const float* gn_affine, int silu, const void* w, const void* bias, const void* res, int ldr,
                        int batch, int h, int w_, int n, void* out, int ldo, float* colstats, void* stream);

/* GroupNorm affine + SiLU folded into a 3x3 / stride 1 / pad 1 convolution with FEW (n <= 4) output channels:
 *   out[m, 0:n] = conv3x3(act(x * scale + shift))[m, 0:n] + bias,   act = SiLU if silu else identity
 * x fp16 NHWC [batch, h, w, c] (c = 128 or 320); gn_affine fp32 [batch][c][2] = (scale, shift) per sample and channel (the table of
 * sd_groupnorm_table_f16) or NULL for a plain convolution; w fp16 [n][9][c]; bias fp16 [n] or NULL; out fp16 [batch*h*w, ldo]
 * (ldo >= 8, multiple of 8): channels 0..7 of every pixel are written (n results + zeros), channels >= 8 are left alone.
 * The normalised tensor is never written: the halo patch of a 16 x 16 pixel tile is activated on its way into LDS and rounded to
 * fp16 there, i.e. the result equals GroupNorm kernel -> convolution up to fp32 summation order.  Recordable.
 * replaces: decoder.conv_norm_out + SiLU + decoder.conv_out of AutoencoderKL (self.vae.decode, utils/adaptive_mask_inpainting.py:1086, :1112)
 *           and conv_norm_out + SiLU + conv_out of UNet2DConditionModel (self.unet(...), :1001-1007). */
int sd_conv3x3_small_n_f16(const void* x, const float* gn_affine, int silu, const void* w, const void* bias, int batch, int h, int w_,
                           int c, int n, void* out, int ldo, void* stream);

/* 3x3 / stride 1 / pad 1 neighbourhoods of a 3-channel image as rows of 32 halfs: out[m][3 * tap + ch] = x[pixel m shifted by tap][ch]
 * (tap = 3 * ky + kx, zero padding, columns 27..31 zero), so that a convolution with 3 input channels is a plain K = 32 product
 * (sd_conv_gemm_f16 with weights [n][32] = [n][ky][kx][c] padded).  x fp16 NHWC [batch, h, w, ldx] (channels 0..2 read, ldx % 4 == 0),
 * out fp16 [batch*h*w][32].  Recordable.
 * replaces: the im2col half of encoder.conv_in of AutoencoderKL (self.vae.encode, utils/adaptive_mask_inpainting.py:677-680). */
int sd_im2col3x3_c3_f16(const void* x, int ldx, int batch, int h, int w_, void* out, void* stream);

/* The whole 3-input-channel convolution in one launch (r5): out[m][n] = bias[n] + sum_k w32[n][k] * x[pixel m shifted by tap][ch], k = 3 * tap + ch, for
 * n = 128 output channels, h and w multiples of 16.  x fp16 NHWC [batch, h, w, ldx] (channels 0..2 read, ldx % 4 == 0); w32 fp16 [128][32] =
 * [n][ky][kx][c] padded with 5 zeros (the layout the K = 32 product above takes); bias fp16 [128] or NULL; out fp16 [batch*h*w][ldo], ldo % 8 == 0.
 * colstats (or NULL): fp32 [batch*h*w/256][2][128], the column sums / sums of squares of the stored output per 16 x 16 pixel tile -- the slot layout
 * of sd_conv3x3_halo_f16, read by sd_groupnorm_table_f16 with rows_per_slot = 256.  Same products as im2col + K = 32 GEMM (fp32 accumulation in the
 * MFMA, another summation order); the packed [batch*h*w][32] matrix is never written.  Recordable.
 * replaces: encoder.conv_in of AutoencoderKL (self.vae.encode, utils/adaptive_mask_inpainting.py:677-680) + the statistics pass of the first
 *           ResNet block's GroupNorm. */
int sd_conv3x3_c3_f16(const void* x, int ldx, const void* w32, const void* bias, int batch, int h, int w_, int n, void* out, int ldo,
                      float* colstats, void* stream);

/* Row softmax in place over fp16 [rows, n] with scale (VAE mid-block attention, un-fused). */
int sd_softmax_f16(void* x, int64_t rows, int n, int ld, float scale, void* stream);

/* Classifier-free guidance + DDIM step (eta = 0) + assembly of the next UNet input, one elementwise pass.
 *   eps = eps_u + s (eps_c - eps_u);  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);  x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
 * eps_uc: fp16 NHWC [2B, hw, cpad] UNet output (first B = unconditional, last B = conditional, 4 valid channels)
 * latents: fp32 [B, hw, 4] in/out;  x0_out: fp32 [B, hw, 4] or NULL (pred_original_sample)
 * mask: fp16 [B, hw] ; masked_latents fp16 [B, hw, 4]
 * unet_in: fp16 NHWC [2B, hw, 64]: channels [latents(4) | mask(1) | masked_latents(4) | zero pad] for both halves.
 * replaces: utils/adaptive_mask_inpainting.py:990-996 (input assembly), :1010-1012 (CFG), :1015-1017 (DDIM step). */
int sd_cfg_ddim_step(const void* eps_uc, int eps_ld, float* latents, float* x0_out, const void* mask,
                     const void* masked_latents, void* unet_in, int batch, int hw, float guidance, float alpha_t,
                     float alpha_prev, int write_latents, void* stream);

/* Sinusoidal timestep embedding (flip_sin_to_cos, shift 0) -> fp16 [batch, dim]. */
int sd_timestep_embedding_f16(const float* timesteps, int batch, int dim, void* out, void* stream);

/* NCHW fp32 <-> NHWC fp16 with channel padding / cropping (boundary conversions of the module API). */
int sd_nchw_to_nhwc_f16(const float* x, int batch, int c, int hw, int cpad, void* out, void* stream);
int sd_nhwc_to_nchw_f32(const void* x, int batch, int c, int hw, int ld, float* out, void* stream);

/* VAE decode epilogue: NHWC fp16 [B, hw, ld] (3 valid channels, range ~[-1,1]) -> uint8 HWC image
 * (x/2+0.5).clamp(0,1)*255, truncated (round_mode 0: the `.astype(np.uint8)` of utils/adaptive_mask_inpainting.py:1114)
 * or rounded half-to-even (round_mode 1: diffusers' numpy_to_pil used for the final image, :1097). */
int sd_image_to_u8(const void* x, int batch, int hw, int ld, int round_mode, uint8_t* out, void* stream);

/* VAE posterior sample: moments NHWC fp16 [npix, ld] = [mean(4) | logvar(4) | pad] ->
 * (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale as fp32 [npix,4] and/or fp16 [npix,4] (noise NULL = mode).
 * replaces: `self.vae.encode(img).latent_dist.sample(generator)` * scaling_factor, utils/adaptive_mask_inpainting.py:675-684. */
int sd_vae_sample(const void* moments, int ld, const float* noise, float scale, int64_t npix, float* latents_f32,
                  void* latents_f16, void* stream);

/* out = sqrt(alpha) x0 + sqrt(1-alpha) noise  (scheduler.add_noise, utils/adaptive_mask_inpainting.py:658). */
int sd_add_noise(const float* x0, const float* noise, float alpha, int64_t n, float* out, void* stream);

/* Mask adaptation glue of the adaptive loop, on device:
 *   mask = use_default ? default : AND(dilate_{3x3 ones, dilate_iters}(seg), default)     (u8 [H,W], 0/1)
 *   mask_latent  fp16 [H/8, W/8]        nearest down-sample (source pixel (8y, 8x))
 *   masked_image fp16 NHWC [H*W, cpad]  image * (mask < 0.5), image = fp32 NCHW [3,H,W] in [-1,1]
 * replaces: utils/adaptive_mask_inpainting.py:1130-1141 (cv2.dilate / logical_and / prepare_mask_and_masked_image)
 *           and the mask interpolation of :686-694. */
int sd_mask_adapt(const uint8_t* seg, const uint8_t* default_mask, int H, int W, int dilate_iters, int use_default,
                  const float* image_nchw, int cpad, uint8_t* mask_full, void* mask_latent, void* masked_image, void* stream);

/* The same for `batch` images in one call, with the reference's "mask too small -> default mask" test done on the device
 * (no host round trip per re-estimation):  area[b] = sum(seg[b]) (sum of VALUES, as `mask.sum()`),
 *   mask[b] = (force_default || (double)area[b] < area_thres) ? default[b] : AND(dilate(seg[b]), default[b]).
 * seg / default_mask / mask_full: u8 [batch,H,W]; image_nchw fp32 [batch,3,H,W]; mask_latent fp16 [batch, H/8*W/8];
 * masked_image fp16 NHWC [batch*H*W, cpad], cpad % 8 == 0: channels 0..7 are always written (3 data + zeros), channels >= 8
 * only when write_pad != 0 (a caller that zero-initialised the buffer once passes 0).  area: int32 [batch] (output);
 * scratch: u8 [batch,H,W] workspace.  seg / scratch may be NULL when force_default.
 * replaces: utils/adaptive_mask_inpainting.py:1123-1141 (adapt_mask up to prepare_mask_latents), :686-694. */
int sd_mask_adapt_batched(const uint8_t* seg, const uint8_t* default_mask, int batch, int H, int W, int dilate_iters,
                          int force_default, double area_thres, const float* image_nchw, int cpad, int write_pad,
                          uint8_t* mask_full, void* mask_latent, void* masked_image, int32_t* area, uint8_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Models: the launch list and the hipGraph of a network live in the library (coma_amd/csrc/sd_plan.hip).
 * A model = a registry of device buffers + named bindings (inputs / outputs) + named plans (recorded launch lists).
 * Recording: between sd_model_record_begin and sd_model_record_end every sd_* LAUNCH entry point called on the thread
 * (sd_conv_gemm_f16, sd_groupnorm*_f16, sd_layernorm_f16, sd_attention_f16, sd_softmax_f16,
 * sd_timestep_embedding_f16, sd_copy_d2d) appends its arguments to the plan instead of launching; arguments are validated when
 * the plan first runs.  sd_model_run launches the list eagerly on `stream`; sd_model_replay captures it once into a hipGraph (on a
 * private stream, nothing executes during capture) and launches the graph on `stream`.
 * sd_model_save / sd_model_load: ONE file with the registry, the bindings, the plans and the contents of the SD_BUF_PERSISTENT
 * buffers (weights, constants); a loaded model owns its device memory (freed by sd_model_destroy), a recorded one borrows the
 * caller's buffers, which must outlive it.  Every pointer a plan uses must lie inside a registered buffer for the model to be
 * saved.  Not thread-safe per model; distinct models are independent. */
#define SD_BUF_PERSISTENT 1   /* contents are part of the model (saved / restored) */
#define SD_BUF_ZEROED 2       /* must start as zeros (pad channels that kernels never write); loaded buffers always start zeroed */
#define SD_BUF_IF_NEW 4       /* register as SD_BUF_PERSISTENT unless an entry already covers the range, whose flags then stay as they are
                                 (the recording hook: scratch stays scratch, a constant seen for the first time is saved) */
int sd_model_create(void** model);
int sd_model_destroy(void* model);
int sd_model_register_buffer(void* model, void* ptr, size_t bytes, int flags);
int sd_model_bind(void* model, const char* name, void* ptr, size_t bytes);            /* name: at most 31 characters */
int sd_model_binding(const void* model, const char* name, void** ptr, size_t* bytes);
int sd_model_record_begin(void* model, const char* plan_name);                        /* an existing plan of that name is replaced */
int sd_model_record_end(void* model);
int sd_model_num_launches(const void* model, const char* plan_name);                  /* -1: no such plan */
int sd_model_run(void* model, const char* plan_name, void* stream);
int sd_model_replay(void* model, const char* plan_name, void* stream);
int sd_model_prepare(void* model, const char* plan_name);   /* capture + instantiate the plan's hipGraph now (nothing executes); sd_model_replay does it on first use otherwise */
int sd_model_save(const void* model, const char* path);
int sd_model_load(const char* path, void** model);
/* dst[0:bytes) = src[0:bytes), device to device, on `stream` (recordable: the duplicated CFG halves of the UNet). */
int sd_copy_d2d(void* dst, const void* src, size_t bytes, void* stream);

/* The three networks of the inpainting loop as entry points over a model (recorded by coma_amd/sd/{unet,vae}.py or loaded from a
 * file).  Inputs are copied device-to-device into the model's bound buffers, the plan's hipGraph is launched, the result is copied
 * out; a NULL input / output pointer means "already in place / leave it in the bound buffer" (sd_model_binding gives the address).
 *   sd_unet_set_context  plan "context": ctx fp16 [2B,77,768] -> cross-attention K / V^T of every block   (once per prompt)
 *   sd_unet_forward      plan "step":    x_in fp16 [2B, h*w, 64] (9 valid channels: latents | mask | masked-image latents),
 *                                        timesteps fp32 [2B] -> eps fp16 [2B*h*w, 64] (4 valid channels)
 *                        replaces: self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, ...)[0],
 *                                  utils/adaptive_mask_inpainting.py:1001-1007
 *   sd_vae_decode        plan "decode":  z fp16 [B, h*w, 64] (4 valid channels, already divided by scaling_factor)
 *                                        -> image fp16 [B*8h*8w, 64] (3 valid channels)     replaces: self.vae.decode, :1086, :1112
 *   sd_vae_encode        plan "encode":  x fp16 [B, H*W, 64] (3 valid channels in [-1,1]) -> moments fp16 [B*h*w, 64] (mean 4 | logvar 4)
 *                                        replaces: self.vae.encode(image).latent_dist (before .sample), :677-680 */
int sd_unet_set_context(void* model, const void* ctx, void* stream);
int sd_unet_forward(void* model, const void* x_in, const float* timesteps, void* eps_out, void* stream);
int sd_vae_decode(void* model, const void* z, void* image_out, void* stream);
int sd_vae_encode(void* model, const void* image, void* moments_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SD_HIP_H */
