// sd_xchain.hip -- the row-local middle of a BasicTransformerBlock at C = 320 as ONE kernel (gfx950):
//
//     h1 = attn1_out @ Wo1^T + bo1 + h          (attn1.to_out.0 + residual)
//     n2 = LayerNorm2(h1)
//     q2 = n2 @ Wq2^T                           (attn2.to_q)
//     a2 = softmax(q2 K2^T / sqrt(d)) V2        (cross attention over the 77 text tokens, 8 heads of 40; K2 / V2^T precomputed per prompt)
//     h2 = a2 @ Wo2^T + bo2 + h1                (attn2.to_out.0 + residual)
//     n3 = LayerNorm3(h2)                       (what ff.net.0 consumes)
//
// replaces: six launches of the UNet graph at the 64 x 64 level (linear, LayerNorm, linear, attention, linear, LayerNorm: 185 us,
// each of them bound by reading and writing [65536, 320] fp16 tensors -- the K = 320 linear is its residual read + output write,
// profiles/r03_notes.md section 4) that diffusers runs as BasicTransformerBlock.attn1.to_out / norm2 / attn2 / norm3, reached from the
// reference through self.unet(...) (utils/adaptive_mask_inpainting.py:1001-1007).  Every step is local to a token row once K2 / V2^T
// exist, so a workgroup keeps a tile of 128 rows on the chip from the first product to the last LayerNorm: HBM sees attn1_out and h
// once, h2 and n3 once (h1 makes one round trip through L2 in the h2 buffer: keeping it in registers across the two products in
// between was tried and spills -- 256 registers per wave with two workgroups per CU).
//
// Workgroup = 4 waves, 64 token rows of one sample; 80 KB of LDS, so TWO workgroups share a CU and one's HBM phases (tile loads,
// residual / LayerNorm stores) run under the other's products (with 128-row tiles and one workgroup per CU the phases of all
// workgroups move in lock step: 146 us, of which ~55 us is pure memory time).
//   T   : LDS tile [64][320] fp16 (rows of 640 B, 16-byte chunks XOR-swizzled with the row) -- in turn attn1_out, h / h1, n2, q2
//         (overwritten head by head with a2), h1 / h2.  It is the activation operand of the three products.
//   WB  : two LDS stages of a weight K-slice [320][32] (LDS-DMA, one barrier per slice); 10 slices per product.
//   MFMA: v_mfma_f32_32x32x16_f16 as D = W_frag . A_frag^T: a lane owns one output row; wave (wr, wc), wr, wc in {0, 1}, owns rows 32 wr .. +31 and
//         columns 160 wc .. +159 (5 tiles: 80 accumulator registers).
//   LayerNorm / stores: a ROW pass over T -- 4 lanes per row, 10 chunks each: coalesced 16-byte global stores of the residual stream,
//         mean / variance with two lane exchanges, the normalised row written back in place.
//   cross attention: wave (wr, wc) handles its 32 rows for heads 4 wc .. 4 wc + 3.  77 keys = 3 key tiles: K and V^T fragments come
//         straight from global memory into registers (98 KB per sample, L2-resident; no LDS, no barrier), the next head's fragments
//         are in flight while this head is multiplied; softmax in one pass (all keys at once), denominator from a synthetic ones row.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

namespace xc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 320, TM = 64, NW = 4, BK = 32, HEADS = 8, D = 40;   // 64-row tiles: 80 KB of LDS, TWO workgroups per CU
constexpr int T_BYTES = TM * C * 2;                   // 40 KB
constexpr int WB_STAGE = C * BK;                      // halves per weight slice
constexpr int LDS_BYTES = T_BYTES + 2 * WB_STAGE * 2; // 40 + 40 KB
constexpr unsigned OOB = 0x80000000u;

struct Args {
  const _Float16 *a, *h, *wo1, *bo1, *g2, *b2, *wq, *k2, *vt2, *wo2, *bo2, *g3, *b3;
  _Float16 *h2, *n3, *dbg;
  int M, rows_per_sample, lk, ldv2, stage;
  float scale_log2, eps;
};

// 16-byte chunk slot of logical chunk c (0..39) in row `row` of T: the 8 chunks of a 128-byte group are permuted with the row,
// rows alternate between the two halves of the 256-byte bank window (640 = 512 + 128) -> conflict-free fragment reads
__device__ __forceinline__ int tswz(int row, int c) { return (c & ~7) | ((c ^ (row >> 1)) & 7); }
// weight slice rows are 64 bytes (4 chunks)
__device__ __forceinline__ int wswz(int row, int c) { return c ^ ((row >> 2) & 3); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__global__ __launch_bounds__(NW * 64, 2) void xchain_kernel(Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char xsmem[];
  _Float16* const T = reinterpret_cast<_Float16*>(xsmem);
  _Float16* const WB = reinterpret_cast<_Float16*>(xsmem + T_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * TM;
  const int l31 = lane & 31, hh = lane >> 5;
  const int my_row = wr * 32 + l31;                   // the tile row this lane owns in every product

  // ---- tile DMA: 128 rows x 40 chunks = 80 pieces of 1 KiB, 10 per wave; lane -> (row, slot) of the piece, source chunk un-swizzled
  // (tswz is an involution).  The offsets are recomputed per use (three uses): 10 registers less to carry through the kernel.
  const unsigned tensor_bytes = (unsigned)((long long)g.M * C * 2);
  auto load_tile = [&](const _Float16* src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, tensor_bytes);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int q = (wave * 10 + j) * 64 + lane;
      const int row = q / 40, slot = q - row * 40;
      const unsigned off = (m0 + row) < g.M ? (unsigned)(((long long)(m0 + row) * C + tswz(row, slot) * 8) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(T + (wave * 10 + j) * 512), 16, off, 0, 0, 0);
    }
#endif
  };
  // ---- weight slice DMA: [320 rows][32 k] = 20 pieces, 5 per wave
  static_assert(TM * 40 / 64 / NW == 10 && 20 % NW == 0, "DMA piece counts");
  constexpr int WPW = 20 / NW;
  unsigned w_off[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int p = wave + NW * j;
    const int row = p * 16 + (lane >> 2), slot = lane & 3;
    w_off[j] = (unsigned)((row * C + wswz(row, slot) * 8) * 2);
  }
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int buf, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16* dst = WB + buf * WB_STAGE;
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + (wave + NW * j) * 512), 16, w_off[j] + s * (BK * 2), 0, 0, 0);
#endif
  };

  float16v acc[5];
  // acc = T[128 x 320] . W[320 x 320]^T for this wave's 32 x 160 patch.  Entry: T complete and visible (a barrier has passed).
  auto gemm = [&](const _Float16* w) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(w, (unsigned)(C * C * 2));
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    issue_w(rs, 0, 0);
#pragma unroll 1
    for (int s = 0; s < C / BK; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // slice s landed everywhere; every wave is done with the other stage
      if (s + 1 < C / BK) issue_w(rs, (s + 1) & 1, s + 1);
      const _Float16* Wb = WB + (s & 1) * WB_STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = 2 * s + kk;
        const half8 af = *reinterpret_cast<const half8*>(&T[my_row * C + tswz(my_row, 2 * ks + hh) * 8]);
        half8 wf[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int n = wc * 160 + j * 32 + l31;
          wf[j] = *reinterpret_cast<const half8*>(&Wb[n * BK + wswz(n, 2 * kk + hh) * 8]);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af, acc[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();                    // every wave is done reading T and WB
  };
  // register quad (j, rg) of a lane = 4 consecutive columns starting at col(j, rg) of row my_row
  auto quad_col = [&](int j, int rg) { return wc * 160 + j * 32 + 8 * rg + 4 * hh; };
  auto quad_ptr = [&](int j, int rg) {
    const int col = quad_col(j, rg);
    return &T[my_row * C + tswz(my_row, col >> 3) * 8 + (col & 7)];
  };
  // T <- fp16(acc (+ bias + T))
  auto write_tile = [&](const _Float16* bias, bool add_tile) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        _Float16* p = quad_ptr(j, rg);
        float v[4] = {acc[j][rg * 4 + 0], acc[j][rg * 4 + 1], acc[j][rg * 4 + 2], acc[j][rg * 4 + 3]};
        if (bias) {
          const half4 bv = *reinterpret_cast<const half4*>(bias + quad_col(j, rg));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
        }
        if (add_tile) {
          const half4 tv = *reinterpret_cast<const half4*>(p);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)tv[e];
        }
        *reinterpret_cast<half4*>(p) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      }
  };
  // row pass: 4 lanes per row, 10 chunks each.  Stores the row (as it stands in T) to `res_out`, LayerNorms it and either writes the
  // normalised row back into T (ln_out == nullptr) or stores it to ln_out.
  auto row_pass = [&](_Float16* res_out, const _Float16* gamma, const _Float16* beta, _Float16* ln_out) {
    const int row = tid >> 2, qtr = tid & 3;
    const bool ok = m0 + row < g.M;
    half8 x[10];
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int c = qtr * 10 + i;
      x[i] = *reinterpret_cast<const half8*>(&T[row * C + tswz(row, c) * 8]);
      if (ok && res_out) *reinterpret_cast<half8*>(res_out + (long long)(m0 + row) * C + c * 8) = x[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)x[i][e]; sum += f; sq += f * f; }
    }
    sum += __shfl_xor(sum, 1); sq += __shfl_xor(sq, 1);
    sum += __shfl_xor(sum, 2); sq += __shfl_xor(sq, 2);
    const float mean = sum * (1.0f / C);
    const float var = fmaxf(sq * (1.0f / C) - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + g.eps);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int c = qtr * 10 + i;
      const half8 gm = *reinterpret_cast<const half8*>(gamma + c * 8), bt = *reinterpret_cast<const half8*>(beta + c * 8);
      half8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (_Float16)(((float)x[i][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      if (ln_out) { if (ok) *reinterpret_cast<half8*>(ln_out + (long long)(m0 + row) * C + c * 8) = y; }
      else *reinterpret_cast<half8*>(&T[row * C + tswz(row, c) * 8]) = y;
    }
  };
  auto dump = [&]() {                                 // debug: T in logical order -> g.dbg
    __syncthreads();
    const int row = tid >> 2, qtr = tid & 3;
    if (g.dbg && m0 + row < g.M)
      for (int i = 0; i < 10; ++i) {
        const int c = qtr * 10 + i;
        *reinterpret_cast<half8*>(g.dbg + (long long)(m0 + row) * C + c * 8) = *reinterpret_cast<const half8*>(&T[row * C + tswz(row, c) * 8]);
      }
  };

  // ================================================================================================ phase 1: h1, n2
  load_tile(g.a);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  gemm(g.wo1);
  load_tile(g.h);                                     // T is free: the residual tile comes in coalesced
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  write_tile(g.bo1, true);                            // T = h1 (each quad is read and written by the lane that owns it)
  __syncthreads();
  if (g.stage == 1) { dump(); return; }
  row_pass(g.h2, g.g2, g.b2, nullptr);                // h1 -> the h2 buffer (comes back in phase 4, from L2); T = n2
  __syncthreads();
  if (g.stage == 2) { dump(); return; }
  // ================================================================================================ phase 2: q2
  gemm(g.wq);
  write_tile(nullptr, false);                         // T = q2
  __syncthreads();
  if (g.stage == 3) { dump(); return; }
  // ================================================================================================ phase 3: cross attention
  {
    const int b = m0 / g.rows_per_sample;
    const __amdgpu_buffer_rsrc_t k_rs = make_rsrc(g.k2 + (long long)b * g.lk * C, (unsigned)(g.lk * C * 2));
    const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(g.vt2 + (long long)b * C * g.ldv2, (unsigned)(C * g.ldv2 * 2));
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> half8 {
      const uint4v u = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      return __builtin_bit_cast(half8, u);
    };
    half8 kf[9], vf[10];
    // fragments of head hd: K[key = 32 t + l31][dd = 16 ks + 8 hh ..] (dd >= 40: zero), V^T[dd = 32 t + l31][keys of chunk 2 st + hh].
    // One register set each: the NEXT head's K is requested as soon as this head's QK^T has consumed the registers, the next V^T as
    // soon as P.V has -- both are in flight under the rest of the head (register budget: 256 with two waves per SIMD).
    auto fetch_k = [&](int hd) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int key = t * 32 + l31, dd = ks * 16 + hh * 8;
          const unsigned off = (key < g.lk && dd < D) ? (unsigned)((key * C + hd * D + dd) * 2) : OOB;
          kf[t * 3 + ks] = ld16(k_rs, off);
        }
    };
    auto fetch_v = [&](int hd) {
#pragma unroll
      for (int st = 0; st < 5; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int dd = t * 32 + l31;
          const unsigned off = dd < D ? (unsigned)(((hd * D + dd) * g.ldv2 + (2 * st + hh) * 8) * 2) : OOB;
          vf[st * 2 + t] = ld16(v_rs, off);
        }
    };
    const int hbase = wc * 4;
    fetch_k(hbase);
    fetch_v(hbase);
#pragma unroll
    for (int hi = 0; hi < 4; ++hi) {
      const int hd = hbase + hi;
      __builtin_amdgcn_sched_barrier(0);
      // Q fragments of this head from T (pre-multiplied by scale*log2e: the scores come out as exp2 arguments)
      half8 qf[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int dd = ks * 16 + hh * 8;
        half8 q;
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (_Float16)0.0f;
        if (dd < D) {
          const int col = hd * D + dd;
          q = *reinterpret_cast<const half8*>(&T[my_row * C + tswz(my_row, col >> 3) * 8]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (_Float16)((float)q[e] * g.scale_log2);
      }
      float16v s[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t * 3 + ks], qf[ks], s[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (hi + 1 < 4) fetch_k(hd + 1);
      // keys >= lk masked; all keys are here at once: plain softmax
      float mx = -__builtin_inff();
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          s[t][r] = key < g.lk ? s[t][r] : -__builtin_inff();
          mx = fmaxf(mx, s[t][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      half8 pf[5];
#pragma unroll
      for (int st = 0; st < 5; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[st][e] = (_Float16)__builtin_amdgcn_exp2f(s[st >> 1][(st & 1) * 8 + e] - mx);
      float16v o[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          half8 v = vf[st * 2 + t];
          if (t == 1 && l31 == D - 32) {              // the ones row (dd = 40): the softmax denominator comes out of the same MFMAs
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)1.0f;
          }
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, pf[st], o[t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (hi + 1 < 4) fetch_v(hd + 1);
      const float inv = 1.0f / __shfl(o[1][4], l31);      // row 40 = register 4 of the second 32-row tile, hh = 0 lane
      // a2 of this head over q2 of this head (only this wave reads or writes these columns of its rows)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int dd = t * 32 + 8 * rg + 4 * hh;
          if (dd < D) {
            const int col = hd * D + dd;
            *reinterpret_cast<half4*>(&T[my_row * C + tswz(my_row, col >> 3) * 8 + (col & 7)]) =
                half4{(_Float16)(o[t][rg * 4 + 0] * inv), (_Float16)(o[t][rg * 4 + 1] * inv), (_Float16)(o[t][rg * 4 + 2] * inv),
                      (_Float16)(o[t][rg * 4 + 3] * inv)};
          }
        }
    }
  }
  __syncthreads();
  if (g.stage == 4) { dump(); return; }
  // ================================================================================================ phase 4: h2, n3
  gemm(g.wo2);
  load_tile(g.h2);                                    // h1, written in phase 1 by this workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  write_tile(g.bo2, true);                            // T = h2
  __syncthreads();
  row_pass(g.h2, g.g3, g.b3, g.n3);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// xfront_kernel -- the row-local FRONT of a C = 320 transformer block in one launch:
//     n  = GroupNorm(x) applied as the per-(sample, channel) affine table (no SiLU: Transformer2DModel.norm)
//     h  = n @ Wpi^T + bpi                      (proj_in, a 1x1 convolution)
//     n1 = LayerNorm1(h)
//     q | k = n1 @ [Wq ; Wk]^T                  (attn1.to_q, to_k -> one [M, 640] tensor, what the attention kernel reads)
//     V^T   = Wv @ n1^T per sample              (attn1.to_v, produced TRANSPOSED with the keys of every 16 in the SD_EPI_PERM16_N order)
// replaces five launches of the r2 graph (GroupNorm apply, linear, LayerNorm, linear N = 640, batched V^T GEMM: 165 us at the 64 x 64
// level).  Same tile machinery as xchain_kernel: T = LDS tile [64][320] (x -> n -> h -> n1), weight slices through two LDS stages;
// each product's result is staged in the (then idle) weight stages and leaves as coalesced 16-byte stores.  V^T comes out of the MFMA
// with the operand roles swapped (a lane owns a head-dim row and 4 consecutive tokens), staged as [320][64 tokens].
struct FrontArgs {
  const _Float16 *x, *wpi, *bpi, *g1, *b1, *wqk, *wv;
  const float* gn_affine;          // [samples][320][2]
  _Float16 *h, *qk, *vt;
  int M, rows_per_sample, ldv;
  float eps;
};

__global__ __launch_bounds__(NW * 64, 2) void xfront_kernel(FrontArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char xsmem[];
  _Float16* const T = reinterpret_cast<_Float16*>(xsmem);
  _Float16* const WB = reinterpret_cast<_Float16*>(xsmem + T_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * TM;
  const int l31 = lane & 31, hh = lane >> 5;
  const int my_row = wr * 32 + l31;
  const int b = m0 / g.rows_per_sample, tok0 = m0 - b * g.rows_per_sample;

  const unsigned tensor_bytes = (unsigned)((long long)g.M * C * 2);
  auto load_tile = [&](const _Float16* src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, tensor_bytes);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int q = (wave * 10 + j) * 64 + lane;
      const int row = q / 40, slot = q - row * 40;
      const unsigned off = (m0 + row) < g.M ? (unsigned)(((long long)(m0 + row) * C + tswz(row, slot) * 8) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(T + (wave * 10 + j) * 512), 16, off, 0, 0, 0);
    }
#endif
  };
  constexpr int WPW = 20 / NW;
  unsigned w_off[WPW];
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int p = wave + NW * j;
    const int row = p * 16 + (lane >> 2), slot = lane & 3;
    w_off[j] = (unsigned)((row * C + wswz(row, slot) * 8) * 2);
  }
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int buf, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16* dst = WB + buf * WB_STAGE;
#pragma unroll
    for (int j = 0; j < WPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + (wave + NW * j) * 512), 16, w_off[j] + s * (BK * 2), 0, 0, 0);
#endif
  };
  float16v acc[5];
  // swapped = false: acc[j][.] = rows my_row x columns (wc, j) of T W^T (a lane owns a token row);
  // swapped = true : the operand roles exchanged -- a lane owns weight row wc * 160 + 32 j + l31 and 4 consecutive tokens of tile wr
  auto gemm = [&](const _Float16* w, auto swapc) {
    constexpr bool SWAPPED = decltype(swapc)::value;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(w, (unsigned)(C * C * 2));
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    issue_w(rs, 0, 0);
#pragma unroll 1
    for (int s = 0; s < C / BK; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 1 < C / BK) issue_w(rs, (s + 1) & 1, s + 1);
      const _Float16* Wb = WB + (s & 1) * WB_STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = 2 * s + kk;
        const half8 af = *reinterpret_cast<const half8*>(&T[my_row * C + tswz(my_row, 2 * ks + hh) * 8]);
        half8 wf[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int n = wc * 160 + j * 32 + l31;
          wf[j] = *reinterpret_cast<const half8*>(&Wb[n * BK + wswz(n, 2 * kk + hh) * 8]);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
          acc[j] = SWAPPED ? __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf[j], acc[j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af, acc[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();
  };
  // token-major result -> fp16 into a swizzled [64][320] tile at `dst` (T itself or the idle weight stages)
  auto write_rows = [&](_Float16* dst, const _Float16* bias) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int col = wc * 160 + j * 32 + 8 * rg + 4 * hh;
        float v[4] = {acc[j][rg * 4 + 0], acc[j][rg * 4 + 1], acc[j][rg * 4 + 2], acc[j][rg * 4 + 3]};
        if (bias) {
          const half4 bv = *reinterpret_cast<const half4*>(bias + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
        }
        *reinterpret_cast<half4*>(&dst[my_row * C + tswz(my_row, col >> 3) * 8 + (col & 7)]) =
            half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      }
  };
  // rows of a swizzled [64][320] tile -> global [M, ld] at column offset col0: 4 lanes per row, coalesced 16-byte stores
  auto store_rows = [&](const _Float16* src, _Float16* out, int ld, int col0) {
    const int row = tid >> 2, qtr = tid & 3;
    if (m0 + row < g.M)
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int c = qtr * 10 + i;
        *reinterpret_cast<half8*>(out + (long long)(m0 + row) * ld + col0 + c * 8) = *reinterpret_cast<const half8*>(&src[row * C + tswz(row, c) * 8]);
      }
  };

  // ---- x -> T, GroupNorm affine in place
  load_tile(g.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int row = tid >> 2, qtr = tid & 3;
    const float* aff = g.gn_affine + (long long)b * C * 2;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int c = qtr * 10 + i;
      half8* p = reinterpret_cast<half8*>(&T[row * C + tswz(row, c) * 8]);
      const half8 xv = *p;
      half8 y;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const float4 sc = *reinterpret_cast<const float4*>(aff + (c * 8 + e) * 2);      // (scale, shift) of two channels
        y[e] = (_Float16)((float)xv[e] * sc.x + sc.y);
        y[e + 1] = (_Float16)((float)xv[e + 1] * sc.z + sc.w);
      }
      *p = y;
    }
  }
  __syncthreads();
  // ---- h = n Wpi^T + bpi -> T; h to memory (the residual of the block), n1 = LayerNorm1(h) -> T
  gemm(g.wpi, std::false_type{});
  write_rows(T, g.bpi);
  __syncthreads();
  {
    const int row = tid >> 2, qtr = tid & 3;
    const bool ok = m0 + row < g.M;
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int c = qtr * 10 + i;
      const half8 xv = *reinterpret_cast<const half8*>(&T[row * C + tswz(row, c) * 8]);
      if (ok) *reinterpret_cast<half8*>(g.h + (long long)(m0 + row) * C + c * 8) = xv;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)xv[e]; sum += f; sq += f * f; }
    }
    sum += __shfl_xor(sum, 1); sq += __shfl_xor(sq, 1);
    sum += __shfl_xor(sum, 2); sq += __shfl_xor(sq, 2);
    const float mean = sum * (1.0f / C);
    const float rstd = rsqrtf(fmaxf(sq * (1.0f / C) - mean * mean, 0.0f) + g.eps);
    __builtin_amdgcn_sched_barrier(0);               // second sweep re-reads the row from LDS instead of holding it in 40 registers
#pragma unroll 2
    for (int i = 0; i < 10; ++i) {
      const int c = qtr * 10 + i;
      half8* p = reinterpret_cast<half8*>(&T[row * C + tswz(row, c) * 8]);
      const half8 xv = *p;
      const half8 gm = *reinterpret_cast<const half8*>(g.g1 + c * 8), bt = *reinterpret_cast<const half8*>(g.b1 + c * 8);
      half8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (_Float16)(((float)xv[e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      *p = y;
    }
  }
  __syncthreads();
  // ---- q and k: each product staged in the weight stages, stored, then the stages are handed back
#pragma unroll 1
  for (int part = 0; part < 2; ++part) {
    gemm(g.wqk + (long long)part * C * C, std::false_type{});
    write_rows(WB, nullptr);
    __syncthreads();
    store_rows(WB, g.qk, 2 * C, part * C);
    __syncthreads();
  }
  // ---- V^T: lane owns head-dim row dd = 160 wc + 32 j + l31 and tokens 32 wr + 8 rg + 4 hh .. + 3; staged as [320][64 tokens] (rows of
  // 128 B, swz64-style chunks) with every 16 tokens in the order (0-3, 8-11, 4-7, 12-15)
  gemm(g.wv, std::true_type{});
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int dd = wc * 160 + j * 32 + l31;
      const int t = wr * 32 + 8 * rg + 4 * hh;                        // first of 4 consecutive tokens, t % 4 == 0
      const int tq = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);     // its position in the permuted order
      *reinterpret_cast<half4*>(&WB[dd * 64 + (((tq >> 3) ^ (dd >> 1)) & 7) * 8 + (tq & 7)]) =
          half4{(_Float16)acc[j][rg * 4 + 0], (_Float16)acc[j][rg * 4 + 1], (_Float16)acc[j][rg * 4 + 2], (_Float16)acc[j][rg * 4 + 3]};
    }
  __syncthreads();
  {
    _Float16* vout = g.vt + (long long)b * C * g.ldv + tok0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = i * (NW * 64) + tid;                               // 320 rows x 8 chunks
      const int dd = q >> 3, c = q & 7;
      *reinterpret_cast<half8*>(vout + (long long)dd * g.ldv + c * 8) = *reinterpret_cast<const half8*>(&WB[dd * 64 + ((c ^ (dd >> 1)) & 7) * 8]);
    }
  }
}

}  // namespace xc
}  // namespace sd

extern "C" int sd_xattn_chain_f16(const void* attn1_out, const void* h, const void* wo1, const void* bo1, const void* gamma2,
                                  const void* beta2, const void* wq2, const void* k2, const void* vt2, const void* wo2, const void* bo2,
                                  const void* gamma3, const void* beta3, void* h2, void* n3, int64_t rows, int rows_per_sample, int lk,
                                  int ldv2, float eps, void* debug_out, int debug_stage, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_XCHAIN;
    const void* ps[15] = {attn1_out, h, wo1, bo1, gamma2, beta2, wq2, k2, vt2, wo2, bo2, gamma3, beta3, h2, n3};
    for (int k = 0; k < 15; ++k) r.p[k] = const_cast<void*>(ps[k]);
    r.i[0] = rows; r.i[1] = rows_per_sample; r.i[2] = lk; r.i[3] = ldv2;
    r.f[0] = eps;
    return plan_record(r);
  }
  if (!attn1_out || !h || !wo1 || !bo1 || !gamma2 || !beta2 || !wq2 || !k2 || !vt2 || !wo2 || !bo2 || !gamma3 || !beta3 || !h2 || !n3)
    return fail(COMA_E_INVALID, "sd_xattn_chain_f16: null pointer");
  if (rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample || rows_per_sample % xc::TM || lk <= 0 || lk > 96 || ldv2 < 80 || ldv2 % 8 ||
      rows * xc::C * 2 >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_xattn_chain_f16: bad sizes (C = 320, rows per sample a multiple of 64, at most 96 keys, ldv2 >= 80)");
  xc::Args g;
  g.a = (const _Float16*)attn1_out; g.h = (const _Float16*)h; g.wo1 = (const _Float16*)wo1; g.bo1 = (const _Float16*)bo1;
  g.g2 = (const _Float16*)gamma2; g.b2 = (const _Float16*)beta2; g.wq = (const _Float16*)wq2; g.k2 = (const _Float16*)k2;
  g.vt2 = (const _Float16*)vt2; g.wo2 = (const _Float16*)wo2; g.bo2 = (const _Float16*)bo2; g.g3 = (const _Float16*)gamma3;
  g.b3 = (const _Float16*)beta3; g.h2 = (_Float16*)h2; g.n3 = (_Float16*)n3; g.dbg = (_Float16*)debug_out;
  g.M = (int)rows; g.rows_per_sample = rows_per_sample; g.lk = lk; g.ldv2 = ldv2; g.stage = debug_out ? debug_stage : 0;
  g.scale_log2 = 0.15811388300841897f * 1.4426950408889634f;      // 40^-0.5 * log2(e)
  g.eps = eps;
  static coma::LdsOptIn lds_opt;
  if (int rc = coma::opt_in_lds(lds_opt, reinterpret_cast<const void*>(xc::xchain_kernel), xc::LDS_BYTES, "sd_xattn_chain_f16")) return rc;
  hipLaunchKernelGGL(xc::xchain_kernel, dim3((unsigned)(rows / xc::TM)), dim3(xc::NW * 64), xc::LDS_BYTES, (hipStream_t)stream, g);
  return check_launch("xchain_kernel");
}

extern "C" int sd_xfront_f16(const void* x, const float* gn_affine, const void* wpi, const void* bpi, const void* gamma1, const void* beta1,
                             const void* wqk, const void* wv, void* h, void* qk, void* vt, int64_t rows, int rows_per_sample, int ldv, float eps,
                             void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_XFRONT;
    const void* ps[11] = {x, gn_affine, wpi, bpi, gamma1, beta1, wqk, wv, h, qk, vt};
    for (int k = 0; k < 11; ++k) r.p[k] = const_cast<void*>(ps[k]);
    r.i[0] = rows; r.i[1] = rows_per_sample; r.i[2] = ldv; r.f[0] = eps;
    return plan_record(r);
  }
  if (!x || !gn_affine || !wpi || !bpi || !gamma1 || !beta1 || !wqk || !wv || !h || !qk || !vt) return fail(COMA_E_INVALID, "sd_xfront_f16: null pointer");
  if (rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample || rows_per_sample % xc::TM || ldv < rows_per_sample || ldv % 8 ||
      rows * xc::C * 4 >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_xfront_f16: bad sizes (C = 320, rows per sample a multiple of 64, ldv >= rows per sample)");
  xc::FrontArgs g;
  g.x = (const _Float16*)x; g.gn_affine = gn_affine; g.wpi = (const _Float16*)wpi; g.bpi = (const _Float16*)bpi; g.g1 = (const _Float16*)gamma1;
  g.b1 = (const _Float16*)beta1; g.wqk = (const _Float16*)wqk; g.wv = (const _Float16*)wv; g.h = (_Float16*)h; g.qk = (_Float16*)qk;
  g.vt = (_Float16*)vt; g.M = (int)rows; g.rows_per_sample = rows_per_sample; g.ldv = ldv; g.eps = eps;
  static coma::LdsOptIn lds_opt;
  if (int rc = coma::opt_in_lds(lds_opt, reinterpret_cast<const void*>(xc::xfront_kernel), xc::LDS_BYTES, "sd_xfront_f16")) return rc;
  hipLaunchKernelGGL(xc::xfront_kernel, dim3((unsigned)(rows / xc::TM)), dim3(xc::NW * 64), xc::LDS_BYTES, (hipStream_t)stream, g);
  return check_launch("xfront_kernel");
}
