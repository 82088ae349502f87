// sd_xtail.hip -- the row-local TAIL of a transformer block at C = 320 as one kernel (gfx950):
//
//     f   = GEGLU(n3 @ W1^T + b1)               (ff.net.0: 320 -> 2 x 1280, value * gelu(gate))
//     h3  = f @ W2^T + b2 + h2                  (ff.net.2 + residual)
//     out = h3 @ Wpo^T + bpo + x                (Transformer2DModel.proj_out + the block's input as residual)
//     + the per-32-row column sums / sums of squares of `out` (the GroupNorm statistics of the next ResNet block)
//
// replaces three launches of the UNet graph at the 64 x 64 level (GEGLU GEMM 178 us at 600 TF/s -- its erf epilogue is as long as
// its K = 320 main loop --, K = 1280 GEMM 76 us, K = 320 GEMM 37 us) and the 168 MB hidden tensor between the first two; reached from
// the reference through self.unet(...) (utils/adaptive_mask_inpainting.py:1001-1007).
//
// Workgroup = 8 waves, 128 token rows; the hidden dimension is walked in 10 chunks of 128:
//   T  : LDS tile [128][320] fp16 -- n3 (the A operand of every chunk's first product), then h2 / h3, then x / out
//   H  : LDS tile [128][128] fp16 -- the GEGLU output of the current chunk, A operand of the second product
//   WB : 48 KB of weight K-slices by LDS-DMA, one barrier per slice: three stages of [256][32] (W1: a slice is 8 MFMAs per wave, so it
//        is requested two slices ahead) or two of [320][32] (W2 / Wpo); the first slices of every product are requested under the
//        previous phase (W2 under the GEGLU arithmetic, the next chunk's W1 under the last W2 slice)
//   acc1 (4 tiles: two row tiles x (value, gate) of 32 hidden columns -- the first product runs on a 2 x 4 wave grid, wave tile
//   64 x 64, one fragment read per MFMA) lives per chunk; acc2 (5 tiles: this wave's 32 x 160 patch of the ff.net.2 output, 4 x 2 wave
//   grid) accumulates over the 10 chunks: 144 accumulator registers.
// Where a workgroup's ~ 243 k cycles go (shader-clock stamps, profiles/r03_notes.md 10): per chunk first product 10.0 k (160 MFMAs per SIMD =
// 5.1 k: the 32 / 64-row wave tiles read one LDS fragment per MFMA, the LDS port is as busy as the matrix pipe and the two do not overlap
// across the per-slice barrier), GEGLU 3.9 k (VALU), 1.6 k wait, second product 4.6 k; third product 16.6 k, final residual + store 16 k.
// One workgroup per CU (160 KB of LDS, 8 waves); a 4-wave form with 64 x 128 / 64 x 160 wave tiles (0.7 reads per MFMA, accumulators in
// AGPRs) measured 25 % slower.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"
#include "sd_gelu.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

namespace xt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 320, HID = 1280, TM = 128, NW = 8, BK = 32, CH = 128;     // CH hidden columns per chunk
constexpr int RT = TM / 32 / (NW / 2);                 // 32-row MFMA tiles per wave (waves: NW / 2 row groups x 2 column groups)
constexpr int W1_DMA = 16 / NW, W2_DMA = 24 / NW;      // DMA instructions per wave for a W1 / a (padded) W2 or Wpo slice
constexpr int T_BYTES = TM * C * 2;                    // 80 KB
constexpr int W1_STAGE = 256 * BK;                     // halves: three 16 KB stages for the [256][32] slices of W1
constexpr int W2_STAGE = 12288;                        // halves: two stages (at 0 and 24 KB) for the [320][32] slices of W2 / Wpo
constexpr int WB_BYTES = 3 * W1_STAGE * 2;             // 48 KB
constexpr int H_BYTES = TM * CH * 2;                   // 32 KB
constexpr int LDS_BYTES = T_BYTES + WB_BYTES + H_BYTES;
constexpr unsigned OOB = 0x80000000u;

struct Args {
  const _Float16 *n3, *h2, *x, *w1, *b1, *w2, *b2, *wpo, *bpo;
  _Float16* out;
  float* colstats;       // [M/32][2][320] or nullptr
  int M;
};

__device__ __forceinline__ int tswz(int row, int c) { return (c & ~7) | ((c ^ (row >> 1)) & 7); }   // 640-byte rows
__device__ __forceinline__ int hswz(int row, int c) { return c ^ (row & 15); }                      // 256-byte rows (16 chunks)
__device__ __forceinline__ int wswz(int row, int c) { return c ^ ((row >> 2) & 3); }                // 64-byte rows

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__global__ __launch_bounds__(NW * 64, NW / 4) void xtail_kernel(Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char xsmem[];
  _Float16* const T = reinterpret_cast<_Float16*>(xsmem);
  _Float16* const WB = reinterpret_cast<_Float16*>(xsmem + T_BYTES);
  _Float16* const H = reinterpret_cast<_Float16*>(xsmem + T_BYTES + WB_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * TM;
  const int l31 = lane & 31, hh = lane >> 5;
  int my_row[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) my_row[i] = (wr * RT + i) * 32 + l31;
  // the first product uses another wave grid: (NW / 4) row groups x 4 column groups, wave tile (RT1 * 32) rows x 64 W1 rows = (value, gate)
  // of 32 hidden columns -- RT1 + 2 fragment reads per 2 RT1 MFMAs instead of 1 + 4 per 4
  constexpr int RT1 = TM / 32 / (NW / 4);
  const int wr1 = wave >> 2, wc1 = wave & 3;
  int row1[RT1];
#pragma unroll
  for (int i = 0; i < RT1; ++i) row1[i] = (wr1 * RT1 + i) * 32 + l31;

  const unsigned tensor_bytes = (unsigned)((long long)g.M * C * 2);
  // [128 rows][40 chunks] = 80 pieces of 1 KiB, 80 / NW per wave
  auto load_tile = [&](const _Float16* src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, tensor_bytes);
#pragma unroll
    for (int j = 0; j < 80 / NW; ++j) {
      const int q = (wave * (80 / NW) + j) * 64 + lane;
      const int row = q / 40, slot = q - row * 40;
      const unsigned off = (m0 + row) < g.M ? (unsigned)(((long long)(m0 + row) * C + tswz(row, slot) * 8) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(T + (wave * (80 / NW) + j) * 512), 16, off, 0, 0, 0);
    }
#endif
  };
  // weight slice [rows][32 k] of a row-major [n][ld] matrix starting at row r0, column k0: `pieces` pieces of 16 rows; wave w takes
  // pieces w, w + NW, ...
  const int p_row = lane >> 2, p_slot = lane & 3;
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, _Float16* dst, int r0, int ld, int k0, int pieces, int nrows) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < 24 / NW; ++j) {
      const int p = wave + NW * j;
      if (p < pieces) {
        const int row = p * 16 + p_row;
        const unsigned off = row < nrows ? (unsigned)((((long long)(r0 + row)) * ld + k0 + wswz(row, p_slot) * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + p * 512), 16, off, 0, 0, 0);
      }
    }
#endif
  };

  float16v acc1[RT1][2], acc2[RT][5];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t w1_rs = make_rsrc(g.w1, (unsigned)(2 * HID * C * 2));
  const __amdgpu_buffer_rsrc_t w2_rs = make_rsrc(g.w2, (unsigned)(C * HID * 2));
  const __amdgpu_buffer_rsrc_t wpo_rs = make_rsrc(g.wpo, (unsigned)(C * C * 2));
  // The weight region (48 KB) is THREE stages of [256][32] for the first product -- a slice is only 8 MFMAs per wave, so a slice
  // must be requested two slices ahead to cover the L2 latency -- and TWO stages of [320][32] (at 0 and 24 KB) for the second and
  // third products.  The stages overlap; the schedule below only ever requests a slice into bytes no wave can still be reading:
  //   chunk parity 0: W1 slice s -> stage s % 3,       W2 slice k -> stage (k + 1) & 1, next chunk's W1 slice 0 -> stage 2
  //   chunk parity 1: W1 slice s -> stage (s + 2) % 3, W2 slice k -> stage k & 1,       next chunk's W1 slice 0 -> stage 0
  auto issue_w1 = [&](int c, int sl, int stage) { issue_w(w1_rs, WB + stage * W1_STAGE, 256 * c, C, sl * BK, 16, 256); };     // 16 / NW DMAs per wave
  // [320][32] slices are padded to 24 pieces (rows >= 320 out of bounds: zeros into the 4 KB behind the slice) so that every wave
  // issues the same number of DMAs (W2_DMA) and one s_waitcnt immediate serves all of them
  auto issue_w2 = [&](const __amdgpu_buffer_rsrc_t& rs, int stage, int ld, int k0) { issue_w(rs, WB + stage * W2_STAGE, 0, ld, k0, 24, C); };

  load_tile(g.n3);
  issue_w1(0, 0, 0);
  issue_w1(0, 1, 1);

  auto chunk = [&](int c, auto phi_c, bool last) {
    constexpr int PHI = decltype(phi_c)::value;
    // ---- first product of the chunk: acc1 = n3 (T) . W1[256 c .. 256 c + 255]^T, K = 320 in 10 slices, two slices in flight
#pragma unroll
    for (int i = 0; i < RT1; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.0f;
    int st = PHI ? 2 : 0;
#pragma unroll 1
    for (int s = 0; s < C / BK; ++s) {
      if (s + 1 < C / BK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_DMA) : "memory");     // slice s landed (slice s + 1 may still be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int st2 = st == 0 ? 2 : st - 1;            // (st + 2) % 3: the stage slice s - 1 has just left
      if (s + 2 < C / BK) issue_w1(c, s + 2, st2);
      else if (s + 1 == C / BK) issue_w2(w2_rs, PHI ? 0 : 1, HID, c * CH);      // first slice of the second product
      const _Float16* Wb = WB + st * W1_STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = 2 * s + kk;
        half8 af[RT1], wf[2];
#pragma unroll
        for (int i = 0; i < RT1; ++i) af[i] = *reinterpret_cast<const half8*>(&T[row1[i] * C + tswz(row1[i], 2 * ks + hh) * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = wc1 * 64 + j * 32 + l31;
          wf[j] = *reinterpret_cast<const half8*>(&Wb[n * BK + wswz(n, 2 * kk + hh) * 8]);
        }
#pragma unroll
        for (int i = 0; i < RT1; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[i], acc1[i][j], 0, 0, 0);
      }
      st = st == 2 ? 0 : st + 1;
    }
    __builtin_amdgcn_s_barrier();                    // every wave is done with the W1 stages, with T (last chunk) and with H of the previous chunk
    if (last) load_tile(g.h2);                       // n3 is no longer needed: the residual of the second product lands under the GEGLU
    issue_w2(w2_rs, PHI ? 1 : 0, HID, c * CH + BK);  // second slice; both land while the GEGLU arithmetic runs
    // ---- GEGLU: tiles (i, 0) and (i, 1) of this wave are (value, gate) of the same 32 hidden columns -> H
#pragma unroll
    for (int i = 0; i < RT1; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int col = wc1 * 32 + 8 * rg + 4 * hh;                              // hidden column within the chunk
        const int wrow = 256 * c + wc1 * 64 + 8 * rg + 4 * hh;                   // row of the interleaved W1 / b1 of the VALUE
        const half4 bv = *reinterpret_cast<const half4*>(g.b1 + wrow), bg = *reinterpret_cast<const half4*>(g.b1 + wrow + 32);
        half4 o4;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 av = {acc1[i][0][rg * 4 + e] + (float)bv[e], acc1[i][0][rg * 4 + e + 1] + (float)bv[e + 1]};
          const f32x2 ag = {acc1[i][1][rg * 4 + e] + (float)bg[e], acc1[i][1][rg * 4 + e + 1] + (float)bg[e + 1]};
          const f32x2 r = av * gelu_erf2(ag);
          o4[e] = (_Float16)r.x;
          o4[e + 1] = (_Float16)r.y;
        }
        *reinterpret_cast<half4*>(&H[row1[i] * CH + hswz(row1[i], col >> 3) * 8 + (col & 7)]) = o4;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // H complete and visible, W2 slices 0 and 1 (and h2) landed
    // ---- second product: acc2 += H . W2[:, 128 c .. 128 c + 127]^T, K = 128 in 4 slices
#pragma unroll 1
    for (int s = 0; s < CH / BK; ++s) {
      if (s >= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // slice s landed, every wave is done with slice s - 1
        if (s + 1 < CH / BK) issue_w2(w2_rs, PHI ? ((s + 1) & 1) : (s & 1), HID, c * CH + (s + 1) * BK);
        else if (!last) issue_w1(c + 1, 0, PHI ? 0 : 2);           // first W1 slice of the next chunk
        else issue_w2(wpo_rs, 0, C, 0);                            // (last chunk has parity 1: stage 0 is free) first slice of proj_out
      }
      const _Float16* Wb = WB + (PHI ? (s & 1) : ((s + 1) & 1)) * W2_STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = 2 * s + kk;
        half8 af[RT], wf[5];
#pragma unroll
        for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(&H[my_row[i] * CH + hswz(my_row[i], 2 * ks + hh) * 8]);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int n = wc * 160 + j * 32 + l31;
          wf[j] = *reinterpret_cast<const half8*>(&Wb[n * BK + wswz(n, 2 * kk + hh) * 8]);
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[i], acc2[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();                    // weight stages and H free again
    if (!last) issue_w1(c + 1, 1, PHI ? 1 : 0);      // second W1 slice of the next chunk
    else issue_w2(wpo_rs, 1, C, BK);
  };
#pragma unroll 1
  for (int c2 = 0; c2 < HID / CH / 2; ++c2) {
    chunk(2 * c2, std::integral_constant<int, 0>{}, false);
    chunk(2 * c2 + 1, std::integral_constant<int, 1>{}, c2 + 1 == HID / CH / 2);
  }

  // quad (j, rg) of acc2 = 4 consecutive columns col(j, rg) of row my_row
  auto quad_col = [&](int j, int rg) { return wc * 160 + j * 32 + 8 * rg + 4 * hh; };
  auto quad_ptr = [&](int i, int j, int rg) {
    const int col = quad_col(j, rg);
    return &T[my_row[i] * C + tswz(my_row[i], col >> 3) * 8 + (col & 7)];
  };
  auto write_tile = [&](const _Float16* bias) {       // T <- fp16(acc2 + bias + T)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          _Float16* p = quad_ptr(i, j, rg);
          const half4 bv = *reinterpret_cast<const half4*>(bias + quad_col(j, rg));
          const half4 tv = *reinterpret_cast<const half4*>(p);
          half4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (_Float16)(acc2[i][j][rg * 4 + e] + (float)bv[e] + (float)tv[e]);
          *reinterpret_cast<half4*>(p) = o4;
        }
  };
  // ---- h3 = acc2 + b2 + h2 -> T (h2 arrived under the last chunk)
  write_tile(g.b2);
  __syncthreads();
  // ---- third product: acc2 = h3 (T) . Wpo^T, three stages (0, 24 KB, 48 KB = the idle H tile); slices 0 and 1 were requested during
  // the last chunk
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
  int st3 = 0;
#pragma unroll 1
  for (int s = 0; s < C / BK; ++s) {
    if (s + 1 < C / BK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < C / BK) issue_w2(wpo_rs, st3 == 0 ? 2 : st3 - 1, C, (s + 2) * BK);
    const _Float16* Wb = WB + st3 * W2_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = 2 * s + kk;
      half8 af[RT], wf[5];
#pragma unroll
      for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(&T[my_row[i] * C + tswz(my_row[i], 2 * ks + hh) * 8]);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int n = wc * 160 + j * 32 + l31;
        wf[j] = *reinterpret_cast<const half8*>(&Wb[n * BK + wswz(n, 2 * kk + hh) * 8]);
      }
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[i], acc2[i][j], 0, 0, 0);
    }
    st3 = st3 == 2 ? 0 : st3 + 1;
  }
  __builtin_amdgcn_s_barrier();
  // ---- out = acc2 + bpo + x -> T, then coalesced stores and the GroupNorm column statistics of the next block
  load_tile(g.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  write_tile(g.bpo);
  __syncthreads();
  {
    const int qtr = tid & 3;
#pragma unroll
    for (int row = tid >> 2; row < TM; row += NW * 16)
      if (m0 + row < g.M)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const int c = qtr * 10 + i;
          *reinterpret_cast<half8*>(g.out + (long long)(m0 + row) * C + c * 8) = *reinterpret_cast<const half8*>(&T[row * C + tswz(row, c) * 8]);
        }
    if (g.colstats && tid < 160) {                    // (32-row block rb, 8-column chunk c): sums of the STORED fp16 values, rows in order
      const int rb = tid / 40, c = tid - rb * 40;
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.0f; s2[e] = 0.0f; }
      for (int r = 0; r < 32; ++r) {
        const int row2 = rb * 32 + r;
        const half8 v = *reinterpret_cast<const half8*>(&T[row2 * C + tswz(row2, c) * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1[e] += f; s2[e] += f * f; }
      }
      float* cs = g.colstats + ((long long)(m0 / 32 + rb) * 2) * C + c * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[e] = s1[e]; cs[C + e] = s2[e]; }
    }
  }
}

}  // namespace xt
}  // namespace sd

extern "C" int sd_xtail_f16(const void* n3, const void* h2, const void* x, const void* w1, const void* b1, const void* w2, const void* b2,
                            const void* wpo, const void* bpo, void* out, float* colstats, int64_t rows, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_XTAIL;
    const void* ps[11] = {n3, h2, x, w1, b1, w2, b2, wpo, bpo, out, colstats};
    for (int k = 0; k < 11; ++k) r.p[k] = const_cast<void*>(ps[k]);
    r.i[0] = rows;
    return plan_record(r);
  }
  if (!n3 || !h2 || !x || !w1 || !b1 || !w2 || !b2 || !wpo || !bpo || !out) return fail(COMA_E_INVALID, "sd_xtail_f16: null pointer");
  if (rows <= 0 || rows % xt::TM || rows * xt::C * 2 >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_xtail_f16: bad sizes (C = 320, rows a multiple of 128, tensors below 2 GiB)");
  xt::Args g;
  g.n3 = (const _Float16*)n3; g.h2 = (const _Float16*)h2; g.x = (const _Float16*)x; g.w1 = (const _Float16*)w1; g.b1 = (const _Float16*)b1;
  g.w2 = (const _Float16*)w2; g.b2 = (const _Float16*)b2; g.wpo = (const _Float16*)wpo; g.bpo = (const _Float16*)bpo; g.out = (_Float16*)out;
  g.colstats = colstats; g.M = (int)rows;
  static coma::LdsOptIn lds_opt;
  if (int rc = coma::opt_in_lds(lds_opt, reinterpret_cast<const void*>(xt::xtail_kernel), xt::LDS_BYTES, "sd_xtail_f16")) return rc;
  hipLaunchKernelGGL(xt::xtail_kernel, dim3((unsigned)(rows / xt::TM)), dim3(xt::NW * 64), xt::LDS_BYTES, (hipStream_t)stream, g);
  return check_launch("xtail_kernel");
}
