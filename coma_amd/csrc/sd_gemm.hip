// sd_gemm.hip -- fp16 implicit-GEMM for the diffusion UNet / VAE on gfx950 (MI355X).
//
// One kernel family covers every matmul-shaped layer of SD-1.5 (the graph the reference drives through
// diffusers, utils/adaptive_mask_inpainting.py:1001-1007 UNet, :1086/:1112 VAE decode, :680 VAE encode):
//   * 3x3 convolution (stride 1 or 2, zero pad 1, optionally reading a nearest-x2-upsampled input)
//   * 1x1 convolution / nn.Linear (taps = 1)
//   * all of the above over a channel-concatenation of TWO sources (UNet skip connections) without
//     materialising the concat
//   * batched A.B^T with a per-batch B operand (attention V^T projection, VAE attention)
// as   out[m, n] = sum_k A[m, k] * W[n, k]   with A gathered on the fly from NHWC fp16 activations:
//   m = (batch, oy, ox),  k = (tap, ci),  A[m,k] = X[batch, oy*stride + ky - pad, ox*stride + kx - pad, ci].
// Both operands are K-contiguous, so A and W fragments are the same 16-byte LDS reads.
//
// Tiling (wave64, MFMA v_mfma_f32_32x32x16_f16): block = WM x WN waves, a wave owns 64 x (TN * 32) outputs; tiles from
// 128 x 64 to 256 x 320 / 512 x 128 (table at the kernel).  K tiles of 32 or 64 reach LDS by LDS-DMA
// (`buffer_load_dwordx4 ... offen lds`) into unpadded XOR-swizzled rows, 2-4 stages with counted vmcnt waits and one raw
// s_barrier per K tile; zero padding is the buffer range check.  The epilogue fuses bias, per-(batch) bias (time
// embedding), residual add, SiLU, GEGLU (value/gate columns interleaved per wave at weight-prep time) and, on request,
// per-32-row column sums / sums of squares of the stored tensor (the consumer's GroupNorm statistics).
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "sd_gelu.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int BKMIN = 32;   // source channel counts must be multiples of this (and of 64 for the deep-K variant)

// Tuning aid (SD_GEMM_DBG=1): blocks 0..DBG_BLOCKS-1 record s_memtime at kernel entry, first tile landed, end of the K loop and
// end of the epilogue; read back with sd_debug_timestamps().
constexpr int DBG_BLOCKS = 4096;
__device__ unsigned long long g_dbg_stamps[DBG_BLOCKS * 4];

struct GemmArgs {
  const _Float16* a0;
  const _Float16* a1;
  int c0, c1;                 // channels of the two A sources (c1 = 0 -> single source)
  int in_h, in_w, out_h, out_w;
  int taps, stride, upsample, pad;
  int M, N, K;
  int n_valid;                // rows of W that exist (N is rounded up to 16 with SD_EPI_PERM16_N)
  int rows_per_batch;
  const _Float16* w;
  const _Float16* bias;       // [N] (or [M] with EPI_BIAS_ROWS)
  const _Float16* bias_bn;    // [B][ldbb]
  int ldbb;
  const _Float16* res;
  int ldr;
  _Float16* out;
  int ldo;
  int epi;
  float* colstats;            // optional [M/32][2][N] fp32: per 32-row block column sums / sums of squares of the OUTPUT
  int ksplit;                 // >1: blockIdx.z selects a K range and fp32 partials go to `partial`
  float* partial;             // [ksplit][M][N] fp32
  long long sa, sw, so, sr;   // per-blockIdx.z strides in elements (batched mode, ksplit == 1)
  unsigned long long* dbg;    // nullptr unless SD_GEMM_DBG is set
  _Float16* out_t;            // optional: columns >= n_split leave transposed per sample, keys in the PERM16 order (sd_conv_gemm_desc.out_t)
  int n_split, ldo_t, rps;
  // Sub-pixel phase of `conv3x3(nearest-upsample-x2(x))` (sd_conv_gemm_desc.phase): taps = 4 is a 2 x 2 window over the SOURCE whose origin
  // is (y - pad, x - pad_x), and row m = (b, y, x) of the product is output pixel (2 y + a, 2 x + b): row 2 m + 2 W floor(m / W) + a 2 W + b
  // of `out` (W = in_w, a power of two).  kw = taps per window row (3, or 2 in phase mode).
  int kw, pad_x;
  int phase;                  // 0: off; 1 + 2 a + b otherwise
  int ph_wshift, ph_rowoff;   // log2(in_w); a * 2 W + b
};

// output row of product row `m` (identity unless this launch is a sub-pixel phase)
__device__ __forceinline__ long long out_row(const GemmArgs& g, int m) {
  return g.phase ? 2LL * m + ((long long)(m >> g.ph_wshift) << (g.ph_wshift + 1)) + g.ph_rowoff : (long long)m;
}

__device__ __forceinline__ void dbg_stamp(const GemmArgs& g, int slot) {
  if (g.dbg && blockIdx.y == 0 && blockIdx.x < DBG_BLOCKS && threadIdx.x == 0) g.dbg[blockIdx.x * 4 + slot] = __builtin_readcyclecounter();
}

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
// gelu_erf2: see sd_gelu.h
__device__ __forceinline__ int kappa16(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }
// SD_EPI_PERM32_N: position p = 8g + e of every group of 32 columns holds key 16 (e >> 2) + 4g + (e & 3)
__device__ __forceinline__ int kappa32(int p) { return (p & ~28) | (((p >> 3) & 3) << 2) | (((p >> 2) & 1) << 4); }

// Shared epilogue math: v = acc (+bias)(+per-batch bias) -> SiLU -> (+residual)
__device__ __forceinline__ float epilogue_value(const GemmArgs& g, float v, int row, int col, const _Float16* resp) {
  if (g.bias) v += (float)g.bias[(g.epi & SD_EPI_BIAS_ROWS) ? row : col];
  if (g.bias_bn) v += (float)g.bias_bn[(long long)(row / g.rows_per_batch) * g.ldbb + col];
  if (g.epi & SD_EPI_SILU) v = silu(v);
  if (resp) v += (float)resp[(long long)row * g.ldr + col];
  return v;
}

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS tile image: [rows][BK halves], unpadded rows filled by LDS-DMA (global_load_lds, 16 B per lane), so one wave
// instruction writes 1 KiB lane-linearly (8 rows of 128 B at BK = 64, 16 rows of 64 B at BK = 32).  Bank conflicts of
// the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE side: the 16-byte slot p of row r
// holds K-chunk p ^ f(r).  A 256-byte bank row spans 2 (BK=64) or 4 (BK=32) tile rows, so with
//   BK = 64: f(r) = (r >> 1) & 7        BK = 32: f(r) = (r >> 2) & 3
// (r mod rows-per-bank-row, slot) enumerates the 16 slots of a bank row and the 16 rows read by one lane group of a
// ds_read_b128 land on 16 distinct slots -> conflict-free, for both the A and the W fragments.
template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
  return BK == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Epilogue shared by every kernel of the family: acc[2][TN] (MFMA C layout, see below) -> bias / per-sample bias / SiLU /
// residual / GEGLU / GroupNorm column statistics -> fp16 output (or fp32 split-K slab), staged through LDS per wave.
// accq(i, j, q) = the q-th register quad of output tile (i, j) of this wave: four consecutive columns starting at column qcol(q) of row
// qrow(q) of the tile.  32x32x16 MFMAs (M16 = false): row = lane & 31, column 8 q + 4 (lane >> 5).  16x16x32 MFMAs (M16 = true): quad
// q = 2 a + b is sub-tile (a, b): row 16 a + (lane & 15), column 16 b + 4 (lane >> 4).  Everything after the staging write works on the
// row-major read-back and does not care.
template <int WM, int WN, int TN, int TM, bool M16, class AccQ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, AccQ accq, _Float16* lds, const int m0, const int n0,
                                              const int wave, const int lane, const long long z, const bool split) {
  constexpr int EP_STRIDE = 32 + 4;                 // floats per staged row (one 32x32 MFMA tile per wave at a time)
  const int wr = wave / WN, wc = wave % WN;
  // ---------------------------------------------------------------- epilogue
  // The MFMAs were issued as D = W_frag . A_frag^T, so a lane owns ONE output row m = (lane & 31) of each 32-row
  // tile and its 16 registers run along output columns n = 8*(r>>2) + 4*(lane>>5) + (r&3): four consecutive
  // registers are four consecutive columns.  Each wave stages one 32 x 32 fp32 MFMA tile at a time in LDS (16-byte
  // writes) and re-reads it row-major so that bias / residual / output move as coalesced 16-byte vectors.
  wait_vmcnt<0>();
  __syncthreads();                              // every wave is done with the operand tiles (all DMA drained)
  float* stage = reinterpret_cast<float*>(lds) + wave * (32 * EP_STRIDE);
  const int lrow = lane & 31, hh = lane >> 5;
  auto qrow = [&](int q) { return M16 ? 16 * (q >> 1) + (lane & 15) : lrow; };
  auto qcol = [&](int q) { return M16 ? 16 * (q & 1) + 4 * (lane >> 4) : 8 * q + 4 * hh; };
  if (split) {
    float* part = g.partial + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wr * (TM * 32) + i * 32 + qrow(q);
          const int col = n0 + wc * (TN * 32) + j * 32 + qcol(q);
          if (row < g.M && col < g.N)   // N % 8 == 0 on this path
            *reinterpret_cast<float4*>(part + (long long)row * g.N + col) = accq(i, j, q);
        }
    }
    return;
  }
  if (g.out_t && n0 >= g.n_split) {
    // this workgroup's columns are the transposed tail (a whole number of N tiles by contract): tile by tile through the wave's staging
    // area, read back COLUMN-wise -- lane = (channel c, half hf): 16 keys of one channel = one 16-key group, stored as two 16-byte
    // vectors in the order (0-3, 8-11, 4-7, 12-15).  No bias, no residual.
    const int c = lane & 31, hf = lane >> 5;
    const int cv = g.N - g.n_split;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mbase = m0 + wr * (TM * 32) + i * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(stage + qrow(q) * EP_STRIDE + qcol(q)) = accq(i, j, q);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mbase < g.M) {
          float v[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = stage[(hf * 16 + t) * EP_STRIDE + c];
          half8 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lo[e] = (_Float16)v[e];
            lo[4 + e] = (_Float16)v[8 + e];
            hi[e] = (_Float16)v[4 + e];
            hi[4 + e] = (_Float16)v[12 + e];
          }
          const int b = mbase / g.rps;
          const int ch = n0 - g.n_split + wc * (TN * 32) + j * 32 + c;
          _Float16* dst = g.out_t + ((long long)b * cv + ch) * g.ldo_t + (mbase - b * g.rps) + hf * 16;
          *reinterpret_cast<half8*>(dst) = lo;
          *reinterpret_cast<half8*>(dst + 8) = hi;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    return;
  }
  _Float16* outp = g.out + z * g.so;
  const _Float16* resp = g.res ? g.res + z * g.sr : nullptr;
  const bool geglu = (g.epi & SD_EPI_GEGLU) != 0;
  // (a per-row bias takes the generic path)
  const bool vec_ok = (g.ldo % 8 == 0) && (g.N % 8 == 0) && (!resp || g.ldr % 8 == 0) && !(g.epi & SD_EPI_BIAS_ROWS) &&
                      (!g.bias_bn || (g.rows_per_batch % 32 == 0 && g.ldbb % 8 == 0));
  // read-back role: 32 rows x 4 chunks of 8 columns = 128 items, two per lane; the column chunk is fixed per lane
  const int cl = (lane & 3) * 8;
  if (geglu) {
    if constexpr (TN % 2 == 0) {
      // tile pairs (2p, 2p+1) = (32 value columns, their 32 gate columns): weight rows interleaved at prep time.  The wave's whole
      // output (TM*32 rows x P*32 columns) is staged ONCE, as fp16, and read back as 16-byte chunks of full 64 / 128-byte row
      // segments: two LDS synchronisation points per wave instead of two per 32 x 32 tile (the per-tile form spent 13 k of a
      // 32 k-cycle workgroup in this epilogue, nearly all of it waiting on LDS round trips -- profiles/r02_notes.md section 11)
      constexpr int P = TN / 2, GST = P * 32 + 8;          // staged row: P*32 halves + 16 bytes of padding
      _Float16* const gs = lds + wave * (TM * 32 * GST);
      const int ncolw = n0 + wc * (TN * 32);
      half4 hbv[P][4], hbg[P][4];
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = ncolw + p * 64 + qcol(q);
          hbv[p][q] = half4{0, 0, 0, 0};
          hbg[p][q] = half4{0, 0, 0, 0};
          if (g.bias) { hbv[p][q] = *reinterpret_cast<const half4*>(g.bias + c); hbg[p][q] = *reinterpret_cast<const half4*>(g.bias + c + 32); }
        }
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            half4 o;
            const float4 qv = accq(i, 2 * p, q), qg = accq(i, 2 * p + 1, q);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              f32x2 av = e == 0 ? (f32x2){qv.x, qv.y} : (f32x2){qv.z, qv.w};
              f32x2 ag = e == 0 ? (f32x2){qg.x, qg.y} : (f32x2){qg.z, qg.w};
              const f32x2 r = (av + (f32x2){(float)hbv[p][q][e], (float)hbv[p][q][e + 1]}) *
                              gelu_erf2(ag + (f32x2){(float)hbg[p][q][e], (float)hbg[p][q][e + 1]});
              o[e] = (_Float16)r.x;
              o[e + 1] = (_Float16)r.y;
            }
            *reinterpret_cast<half4*>(gs + (i * 32 + qrow(q)) * GST + p * 32 + qcol(q)) = o;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int CH = P * 4;                              // 16-byte chunks per staged row
      const int ocolw = ncolw >> 1;
#pragma unroll
      for (int k = 0; k < TM * 32 * CH / 64; ++k) {
        const int it = lane + 64 * k;
        const int rl = it / CH, c = it % CH;
        const int row = m0 + wr * (TM * 32) + rl;
        const half8 o = *reinterpret_cast<const half8*>(gs + rl * GST + c * 8);
        if (row < g.M) *reinterpret_cast<half8*>(outp + out_row(g, row) * g.ldo + ocolw + c * 8) = o;
      }
    }
    return;
  }
  // Residual / per-sample-bias tiles are fetched DEPTH tiles ahead of their use: with one tile in flight per wave the
  // epilogue was latency-bound (16 loads in flight per CU; measured 34 k cycles for a 256 x 320 tile -- longer than a
  // K = 320 main loop); DEPTH tiles ahead it moves at the HBM rate.  All of it goes through buffer descriptors: one
  // 32-bit row offset per lane for the whole epilogue, the tile position in the scalar offset, rows >= M dropped by the
  // range check -- no 64-bit address arithmetic and few live registers next to the 32 * TN accumulators.
  constexpr int NT = TM * TN;                           // 32 x 32 tiles of this wave, j-major (column tile), i-minor
  constexpr int DEPTH = NT <= 4 ? NT : (TN >= 5 ? 2 : 4);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto rsrc_of = [](const void* p, long long bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
  };
  const bool fast = vec_ok && (long long)(g.M + 512) * g.ldo * 2 * (g.phase ? 4 : 1) < 0x7fffffffLL &&
                    (!resp || (long long)(g.M + 512) * g.ldr * 2 < 0x7fffffffLL) && !(g.bias_bn && resp);
  if (fast) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t out_rsrc = rsrc_of(outp, (long long)g.M * g.ldo * 2 * (g.phase ? 4 : 1));
    const __amdgpu_buffer_rsrc_t pre_rsrc = resp ? rsrc_of(resp, (long long)g.M * g.ldr * 2)
                                                 : rsrc_of(g.bias_bn ? g.bias_bn : g.out, g.bias_bn ? 0x7fffffffLL : 0);
    const int colbase = n0 + wc * (TN * 32) + cl;
    const int rowbase = m0 + wr * (TM * 32) + (lane >> 2);              // + 16 k + 32 i
    const int ld_pre = resp ? g.ldr : 0;
    int voff_out[2], voff_pre[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      voff_out[k] = ((rowbase + 16 * k) * g.ldo + colbase) * 2;
      voff_pre[k] = ((rowbase + 16 * k) * ld_pre + colbase) * 2;
    }
    int voff_ph[TM][2];                                          // sub-pixel phase: the rows of a tile are not equally spaced in `out`
    const bool phase_on = g.phase != 0;
    if (phase_on) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = rowbase + 16 * k + 32 * i;
          voff_ph[i][k] = row < g.M ? (int)((out_row(g, row) * g.ldo + colbase) * 2) : (int)0x80000000;
        }
    }
    u32x4 pf[NT][2];                                             // residual rows (k = 0, 1) or, in [0], the per-sample bias
    auto prefetch = [&](const int t) {
      const int j = t / TM, i = t % TM;
      if (resp) {
        const int soff = __builtin_amdgcn_readfirstlane((i * 32 * g.ldr + j * 32) * 2);
#pragma unroll
        for (int k = 0; k < 2; ++k) pf[t][k] = __builtin_amdgcn_raw_buffer_load_b128(pre_rsrc, voff_pre[k], soff, 0);
      } else if (g.bias_bn) {
        const int mb = min(m0 + wr * (TM * 32) + i * 32, g.M - 1);
        const int soff = __builtin_amdgcn_readfirstlane(((mb / g.rows_per_batch) * g.ldbb + j * 32) * 2);
        pf[t][0] = __builtin_amdgcn_raw_buffer_load_b128(pre_rsrc, voff_pre[0], soff, 0);
      }
    };
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) prefetch(t);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      half8 bcol = {0, 0, 0, 0, 0, 0, 0, 0};                        // kept packed: registers are scarce next to 32 * TN accumulators
      const bool oob = colbase + j * 32 + 8 > g.N;                // column chunk of this lane beyond N: loads give 0, stores are dropped
      if (g.bias && !oob) bcol = *reinterpret_cast<const half8*>(g.bias + colbase + j * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int t = TM * j + i;
        if (t + DEPTH < NT) prefetch(t + DEPTH);
        float cs[8], cq[8];        // GroupNorm statistics of the consumer: column sums over the 32 rows of this tile
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(stage + qrow(q) * EP_STRIDE + qcol(q)) = accq(i, j, q);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int soff_out = __builtin_amdgcn_readfirstlane((i * 32 * g.ldo + j * 32) * 2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int rl = (lane + 64 * k) >> 2;
          const float4 x = *reinterpret_cast<const float4*>(stage + rl * EP_STRIDE + cl);
          const float4 y = *reinterpret_cast<const float4*>(stage + rl * EP_STRIDE + cl + 4);
          float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)bcol[e];
          if (g.bias_bn) {
            const half8 tb = __builtin_bit_cast(half8, pf[t][0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)tb[e];
          }
          if (g.epi & SD_EPI_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
          }
          if (resp) {
            const half8 r8 = __builtin_bit_cast(half8, pf[t][k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
          }
          half8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
          if (phase_on) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, oob ? (int)0x80000000 : voff_ph[i][k], j * 64, 0);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rsrc, oob ? (int)0x80000000 : voff_out[k], soff_out, 0);
          if (g.colstats && !oob && m0 + wr * (TM * 32) + i * 32 + rl < g.M) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float f = (float)o[e];     // statistics of the stored (fp16-rounded) tensor, as a GroupNorm pass would see it
              cs[e] += f;
              cq[e] += f * f;
            }
          }
        }
        if (g.colstats) {
          // fold the 16 lanes that share a column chunk (lane & 3 fixed), fixed order -> reproducible: lanes +4, +8, +12 of
          // the 16-lane row by two DPP row rotations (VALU speed), then the four rows by two cross-lane exchanges
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            cs[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs[e]), 0x128, 0xf, 0xf, false));
            cq[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cq[e]), 0x128, 0xf, 0xf, false));
            cs[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs[e]), 0x124, 0xf, 0xf, false));
            cq[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cq[e]), 0x124, 0xf, 0xf, false));
          }
#pragma unroll
          for (int mask = 16; mask < 64; mask <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              cs[e] += __shfl_xor(cs[e], mask);
              cq[e] += __shfl_xor(cq[e], mask);
            }
          if (lane < 4 && !oob) {
            int slot = (m0 + wr * (TM * 32) + i * 32) >> 5;
            if (g.phase) {               // the consumer's GroupNorm sums the slots of a sample in any order: phase p of sample b owns
              const int per = g.rows_per_batch >> 5, b = slot / per;     // slots [4 b per + p per, 4 b per + (p + 1) per)
              slot = (4 * b + g.phase - 1) * per + (slot - b * per);
            }
            float* dst = g.colstats + (long long)slot * 2 * g.N + colbase + j * 32;
            *reinterpret_cast<float4*>(dst) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
            *reinterpret_cast<float4*>(dst + g.N) = make_float4(cq[0], cq[1], cq[2], cq[3]);
            *reinterpret_cast<float4*>(dst + g.N + 4) = make_float4(cq[4], cq[5], cq[6], cq[7]);
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
#endif
    return;
  }
  // generic path (ragged N, unaligned leading dimensions, row bias, both a residual and a per-sample bias): one tile at a time
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wc * (TN * 32) + j * 32 + cl;
    const bool col_ok = col + 8 <= g.N;
    float bcol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bcol[e] = 0.0f;
    if (vec_ok && col_ok && g.bias) {
      const half8 bvv = *reinterpret_cast<const half8*>(g.bias + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) bcol[e] = (float)bvv[e];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = m0 + wr * (TM * 32) + i * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(stage + qrow(q) * EP_STRIDE + qcol(q)) = accq(i, j, q);
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rl = (lane + 64 * k) >> 2;
        const int row = mbase + rl;
        const float4 x = *reinterpret_cast<const float4*>(stage + rl * EP_STRIDE + cl);
        const float4 y = *reinterpret_cast<const float4*>(stage + rl * EP_STRIDE + cl + 4);
        float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        if (row >= g.M) continue;
        if (vec_ok && col_ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bcol[e];
          if (g.bias_bn) {
            const half8 tb = *reinterpret_cast<const half8*>(g.bias_bn + (long long)(mbase / g.rows_per_batch) * g.ldbb + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)tb[e];
          }
          if (g.epi & SD_EPI_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
          }
          if (resp) {
            const half8 r8 = *reinterpret_cast<const half8*>(resp + (long long)row * g.ldr + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
          }
          half8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
          *reinterpret_cast<half8*>(outp + out_row(g, row) * g.ldo + col) = o;
        } else {
          for (int e = 0; e < 8; ++e)
            if (col + e < g.N)
              outp[out_row(g, row) * g.ldo + col + e] = (_Float16)epilogue_value(g, v[e], row, col + e, resp);
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// STAGES-deep LDS-DMA pipeline: tiles kt+1 .. kt+STAGES-1 are in flight while tile kt is multiplied.
// Block = WM x WN waves; a wave owns 64 rows x (TN * 32) columns = 2 x TN MFMA tiles.  Instantiations:
//   <WM=2, WN=2, TN=2, BK=64, ST=2>  128 x 128, 64 KiB, 2 blocks/CU : generic deep-K (and GEGLU, which needs TN even)
//   <WM=2, WN=2, TN=2, BK=32, ST=4>  128 x 128, 64 KiB, 2 blocks/CU : generic short-K (3 tiles in flight per block)
//   <WM=4, WN=2, TN=5, BK=64, ST=2>  256 x 320, 144 KiB, 1 block/CU (8 waves): every N of this UNet is a multiple of
//        320, so activations are read ONCE per 320 output columns and the LDS-DMA issue cost -- measured at 66-114
//        cycles per wave instruction and NOT overlapped with MFMA issue, the limiter of the 128 x 128 tile (0.5 DMA per
//        MFMA) -- drops to 0.225 DMA per MFMA.
//   <WM=2, WN=1/2, ...> 64-wide fallbacks for small N.
template <int WM, int WN, int TN, int BK, int STAGES, bool SPREAD = false, int TM = 2, bool M16 = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_gemm_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;                       // waves per block
  constexpr int BM_ = WM * TM * 32, BN = WN * TN * 32;
  constexpr int CPR = BK / 8;                       // 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR;                     // tile rows written by one wave-wide DMA instruction
  static_assert((BM_ / RPI) % NW == 0 && (BN / RPI) % NW == 0, "DMA instructions must divide evenly over the waves");
  constexpr int A_LD = BM_ / RPI / NW;              // DMA instructions per wave for the A tile
  constexpr int B_LD = BN / RPI / NW;
  constexpr int LPT = A_LD + B_LD;                  // DMA instructions per wave per K tile
  constexpr int EP_STRIDE = 32 + 4;                 // floats per staged row (one 32x32 MFMA tile per wave at a time)
  constexpr int EPI_BYTES = NW * 32 * EP_STRIDE * 4;
  constexpr int TILE_BYTES = STAGES * (BM_ + BN) * BK * 2;
  __shared__ __attribute__((aligned(1024))) _Float16 lds[(TILE_BYTES > EPI_BYTES ? TILE_BYTES : EPI_BYTES) / 2];
  _Float16* const As0 = lds;
  _Float16* const Bs0 = lds + STAGES * BM_ * BK;
  static_assert(TN % 2 != 0 || NW * TM * 32 * (TN / 2 * 32 + 8) * 2 <= (int)sizeof(lds), "GEGLU staging must fit the tile buffers");

  dbg_stamp(g, 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with a private
  // L2.  With enough M tiles every XCD gets a contiguous band of them and walks the N tiles of one M tile back to back,
  // so the A rows are fetched once per XCD and re-read from its L2; small grids keep the plain order.
  const int n_tiles = (g.N + BN - 1) / BN;
  const int m_tiles = (g.M + BM_ - 1) / BM_;
  int m_tile, n_tile;
  if (m_tiles >= 16) {
    const int mq = (m_tiles + 7) >> 3;               // M tiles per XCD band (grid is padded to 8 * mq * n_tiles)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    m_tile = xcd * mq + seq / n_tiles;
    n_tile = seq % n_tiles;
  } else {
    m_tile = blockIdx.x % m_tiles;
    n_tile = blockIdx.x / m_tiles;
  }
  if (m_tile >= m_tiles || n_tile >= n_tiles) return;   // padding block (exits before any barrier)
  // integer divisions run on the VALU; readfirstlane moves their (wave-uniform) results back to SGPRs so that the
  // loop control, the tile bookkeeping and the DMA descriptors below stay scalar
  const int m0 = __builtin_amdgcn_readfirstlane(m_tile * BM_);
  const int n0 = __builtin_amdgcn_readfirstlane(n_tile * BN);
  const bool split = g.ksplit > 1;
  const long long z = split ? 0 : blockIdx.y;
  const _Float16* a0 = g.a0 + z * g.sa;
  const _Float16* wp = g.w + z * g.sw;
  const int ctot = g.c0 + g.c1;

  // K range of this block (split-K) in units of BK tiles
  const int nk_all = g.K / BK;
  int kt0 = 0, nk = nk_all;
  if (split) {
    const int per = (nk_all + g.ksplit - 1) / g.ksplit;
    kt0 = blockIdx.y * per;
    nk = min(per, nk_all - kt0);
    if (nk < 0) nk = 0;
  }
  kt0 = __builtin_amdgcn_readfirstlane(kt0);
  nk = __builtin_amdgcn_readfirstlane(nk);

  // DMA role of this lane: instruction j of this wave covers tile rows (wave*A_LD + j)*RPI .. +RPI-1; lane -> row
  // +lane/CPR, LDS slot lane%CPR, which must receive K-chunk slot ^ swizzle(row).  The transfers are
  // `buffer_load_dwordx4 ... offen lds`: per row ONE 32-bit byte offset in a VGPR (rebuilt only when the tap or the
  // source tensor changes), the K position of the tile in the scalar offset, so a tile's DMA issue costs no VALU work.
  // Rows that are zero padding (or beyond M) carry the offset 0x80000000, which fails the buffer range check: the
  // hardware then writes zeros into LDS.
  constexpr unsigned OOB = 0x80000000u;
  const int l_row = lane / CPR, l_slot = lane % CPR;
  int a_n[A_LD], a_yx[A_LD];          // sample index; (y, x) of tap (0,0) packed as two int16 (y = -16384 -> row >= M)
  unsigned a_koff[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int r = (wave * A_LD + j) * RPI + l_row;
    a_koff[j] = swz<BK>(r, l_slot) * 16;
    const int m = m0 + r;
    const bool ok = m < g.M;
    const int mm = ok ? m : 0;
    a_n[j] = mm / g.rows_per_batch;
    const int rem = mm - a_n[j] * g.rows_per_batch;
    const int oy = rem / g.out_w;
    const int y = ok ? oy * g.stride - g.pad : -16384;
    const int x = (rem - oy * g.out_w) * g.stride - g.pad_x;
    a_yx[j] = (y << 16) | (x & 0xffff);
  }
  unsigned b_off[B_LD];               // byte offset of (weight row, swizzled chunk); rows >= N re-read row N-1 (never stored)
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int r = (wave * B_LD + j) * RPI + l_row;
    int nn = n0 + r;
    if (g.epi & SD_EPI_PERM16_N) nn = kappa16(nn);   // output column j <- W row with bits 2, 3 of j swapped
    if (g.epi & SD_EPI_PERM32_N) nn = kappa32(nn);   // ... or with bits 2, 3, 4 rotated (sd_attention_wide_f16's V^T operand)
    const int n = min(nn, g.n_valid - 1);
    b_off[j] = (unsigned)(n * g.K) * 2u + swz<BK>(r, l_slot) * 16;
  }
  // buffer resources and scalar offsets must live in SGPRs: pin them with readfirstlane (every input is wave-uniform,
  // but the compiler's divergence analysis would otherwise wrap each DMA in a waterfall loop)
  auto make_rsrc = [](const void* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wp);
  const int lim_h = g.upsample ? 2 * g.in_h : g.in_h;
  const int lim_w = g.upsample ? 2 * g.in_w : g.in_w;

  // running position of the NEXT tile to load: tap index and channel offset inside the concatenation.
  // 3x3 convolutions walk K TAP-MINOR (for every 64-channel chunk the nine taps back to back): the nine tiles then touch
  // the same (pixel, channel-chunk) lines shifted by one pixel -- ~50 KB per workgroup that stays in L2 -- where the
  // tap-major order re-fetched the whole 164 KB activation tile of the workgroup per tap (working set of an XCD 5 MB > L2:
  // rocprofv3 showed 9x the activation bytes on the fabric).  The sum over K is the same set of products in another order.
  const bool tapminor = g.taps == 9 && !g.upsample && !(g.epi & (1 << 28));
  int ld_tap, ld_ci;
  if (tapminor) {
    ld_tap = __builtin_amdgcn_readfirstlane(kt0 % 9);
    ld_ci = __builtin_amdgcn_readfirstlane((kt0 / 9) * BK);
  } else {
    ld_tap = __builtin_amdgcn_readfirstlane((kt0 * BK) / ctot);
    ld_ci = kt0 * BK - ld_tap * ctot;
  }
  int ld_src = -1;                    // 0: a0, 1: a1
  unsigned a_off[A_LD];
  unsigned a_base[A_LD];              // tap-minor: byte offset of tap (0,0) in the current source (may wrap below zero)
  auto repoint = [&]() {
    const int ky = g.taps == 9 ? ld_tap / 3 : (g.taps == 4 ? ld_tap >> 1 : 0);      // (constant divisors: no runtime division in the K loop)
    const int kx = g.taps == 9 ? ld_tap - ky * 3 : (g.taps == 4 ? ld_tap & 1 : 0);
    ld_src = __builtin_amdgcn_readfirstlane(ld_ci >= g.c0 ? 1 : 0);
    const int csrc2 = (ld_src ? g.c1 : g.c0) * 2;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int iy = (a_yx[j] >> 16) + ky, ix = (int)(short)(a_yx[j] & 0xffff) + kx;
      const bool ok = iy >= 0 && iy < lim_h && ix >= 0 && ix < lim_w;
      if (g.upsample) { iy >>= 1; ix >>= 1; }
      a_off[j] = ok ? (unsigned)(((a_n[j] & 0xffff) * g.in_h + iy) * g.in_w + ix) * (unsigned)csrc2 + a_koff[j] : OOB;
    }
  };
  // tap-minor: per row the nine validity bits ride in the upper half of a_n; a tile's offsets are base + one scalar delta
  auto retap = [&]() {
    const int ky = ld_tap / 3, kx = ld_tap - ky * 3;
    const int csrc2 = (ld_src ? g.c1 : g.c0) * 2;
    const unsigned sdelta = (unsigned)__builtin_amdgcn_readfirstlane((ky * g.in_w + kx) * csrc2);
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_off[j] = ((a_n[j] >> (16 + ld_tap)) & 1) ? a_base[j] + sdelta : OOB;
  };
  auto rebase = [&]() {
    ld_src = __builtin_amdgcn_readfirstlane(ld_ci >= g.c0 ? 1 : 0);
    const int csrc2 = (ld_src ? g.c1 : g.c0) * 2;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int y0 = a_yx[j] >> 16, x0 = (int)(short)(a_yx[j] & 0xffff);
      a_base[j] = (unsigned)((((a_n[j] & 0xffff) * g.in_h + y0) * g.in_w + x0) * csrc2) + a_koff[j];
    }
    retap();
  };
  if (tapminor) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int y0 = a_yx[j] >> 16, x0 = (int)(short)(a_yx[j] & 0xffff);
      unsigned mask = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y0 + t / 3, ix = x0 + t % 3;
        mask |= (iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w) ? (1u << t) : 0u;      // rows >= M carry y0 = -16384: all clear
      }
      a_n[j] |= (int)(mask << 16);
    }
    rebase();
  } else {
    repoint();
  }

  // LDS-DMA of the tile at (ld_tap, ld_ci) into stage `buf`: instruction `idx` of this wave's LPT (A first, then W)
  auto issue_one = [&](int buf, auto idx_c) {
    constexpr int idx = decltype(idx_c)::value;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (and cannot type the LDS-DMA builtin)
    if constexpr (idx < A_LD) {
      const int ci = ld_src ? ld_ci - g.c0 : ld_ci;
      _Float16* ad = As0 + buf * (BM_ * BK) + wave * A_LD * RPI * BK;
      const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(ld_src ? g.a1 : a0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(ad + idx * RPI * BK), 16, a_off[idx], __builtin_amdgcn_readfirstlane(ci * 2), 0, 0);
    } else {
      constexpr int j = idx - A_LD;
      const int k0 = ld_tap * ctot + ld_ci;
      _Float16* bd = Bs0 + buf * (BN * BK) + wave * B_LD * RPI * BK;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(bd + j * RPI * BK), 16, b_off[j], __builtin_amdgcn_readfirstlane(k0 * 2), 0, 0);
    }
#endif
  };
  auto issue_range = [&](int buf, auto lo_c, auto hi_c) {
    constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
    if constexpr (lo < hi) {
      issue_one(buf, std::integral_constant<int, lo>{});
      if constexpr (lo + 1 < hi) issue_one(buf, std::integral_constant<int, lo + 1>{});
      if constexpr (lo + 2 < hi) issue_one(buf, std::integral_constant<int, lo + 2>{});
      if constexpr (lo + 3 < hi) issue_one(buf, std::integral_constant<int, lo + 3>{});
      static_assert(hi - lo <= 4, "at most four DMA instructions per K step");
    }
  };
  auto advance_tile = [&]() {
    if (tapminor) {
      ld_tap = __builtin_amdgcn_readfirstlane(ld_tap + 1);
      if (ld_tap == 9) {
        ld_tap = 0;
        ld_ci = __builtin_amdgcn_readfirstlane(ld_ci + BK);
        if ((ld_ci >= g.c0) != (ld_src == 1)) { rebase(); return; }
      }
      retap();
      return;
    }
    ld_ci = __builtin_amdgcn_readfirstlane(ld_ci + BK);
    if (ld_ci >= ctot) { ld_ci = 0; ld_tap = __builtin_amdgcn_readfirstlane(ld_tap + 1); repoint(); }
    else if ((ld_ci >= g.c0) != (ld_src == 1)) repoint();
  };
  auto issue_tile = [&](int buf) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (issue_one(buf, std::integral_constant<int, I>{}), ...); }
    (std::make_integer_sequence<int, LPT>{});
    advance_tile();
  };

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // M16: the K loop runs on v_mfma_f32_16x16x32_f16 (a 32 x 32 output tile = 2 x 2 of them).  Same flops per clock as 32x32x16, but
  // less power per flop: under the 1400 W package cap a register-resident loop of them sustains 1.95 PF against 1.69 PF
  // (scripts/probes/mfma_shape.hip), and the vendor GEMM uses this shape.  acc16[2i+a][2j+b] holds rows 16a + (lane & 15), columns
  // 16b + 4 (lane >> 4) + e of tile (i, j); the epilogue takes either layout (gemm_epilogue's accq / qrow / qcol).
  float4v acc16[M16 ? 2 * TM : 1][M16 ? 2 * TN : 1];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
  }

  // fragment addressing: row = base + (lane & 31), K-chunk = 2*ks + (lane >> 5), slot = chunk ^ swizzle(row)
  // (M16: row = base + (lane & 15), K-chunk = 4*ks + (lane >> 4))
  const int frow = M16 ? (lane & 15) : (lane & 31), fhalf = M16 ? (lane >> 4) : (lane >> 5);
  constexpr int FR = M16 ? 16 : 32;                  // rows per fragment
  int a_fr[TM * 32 / FR], b_fr[TN * 32 / FR];
#pragma unroll
  for (int i = 0; i < TM * 32 / FR; ++i) a_fr[i] = wr * (TM * 32) + i * FR + frow;
#pragma unroll
  for (int j = 0; j < TN * 32 / FR; ++j) b_fr[j] = wc * (TN * 32) + j * FR + frow;

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue_tile(s);
  constexpr int KS = BK / 16;
  auto k_tile = [&](int kt, auto next_c) {
    constexpr bool NEXT = decltype(next_c)::value;     // a tile kt+STAGES-1 exists and is issued during this one
    // wait for tile kt only: the (up to STAGES-2) younger tiles stay in flight across the barrier.  A raw s_barrier is
    // used on purpose -- __syncthreads() would drain the DMA queue (its release carries vmcnt(0)).
    const int younger = min(nk - 1, kt + STAGES - 2) - kt;
    if constexpr (STAGES == 2) {
      wait_vmcnt<0>();
    } else if constexpr (STAGES == 3) {
      if (younger >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    } else {
      if (younger >= 2) wait_vmcnt<2 * LPT>(); else if (younger == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt == 0) dbg_stamp(g, 1);
    // every wave has finished reading the stage that tile kt+STAGES-1 overwrites (it held tile kt-1)
    const int nbuf = (kt + STAGES - 1) % STAGES;
    if constexpr (NEXT && !SPREAD) issue_tile(nbuf);
    const _Float16* Ab = As0 + (kt % STAGES) * (BM_ * BK);
    const _Float16* Bb = Bs0 + (kt % STAGES) * (BN * BK);
    if constexpr (!M16) {
    [&]<int... KSI>(std::integer_sequence<int, KSI...>) {
      ([&] {
        constexpr int ks = KSI;
        half8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const half8*>(Ab + a_fr[i] * BK + swz<BK>(a_fr[i], 2 * ks + fhalf) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = *reinterpret_cast<const half8*>(Bb + b_fr[j] * BK + swz<BK>(b_fr[j], 2 * ks + fhalf) * 8);
        if constexpr (NEXT && SPREAD)      // this K step's share of the next tile's DMA, between its LDS reads and MFMAs
          issue_range(nbuf, std::integral_constant<int, ks * LPT / KS>{}, std::integral_constant<int, (ks + 1) * LPT / KS>{});
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
      }(), ...);
    }(std::make_integer_sequence<int, KS>{});
    } else {
    constexpr int KS2 = BK / 32;
    [&]<int... KSI>(std::integer_sequence<int, KSI...>) {
      ([&] {
        constexpr int ks = KSI;
        half8 af[2 * TM], bf[TN];
        // one lane offset per operand and K step: the swizzle term only looks at row bits 1-3 (BK = 64) / 2-3 (BK = 32), which the
        // 16-row block index does not touch, so block i is a constant byte offset (an immediate of the ds_read)
        const int offa = a_fr[0] * BK + swz<BK>(a_fr[0], 4 * ks + fhalf) * 8;
        const int offb = b_fr[0] * BK + swz<BK>(b_fr[0], 4 * ks + fhalf) * 8;
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) af[i] = *reinterpret_cast<const half8*>(Ab + offa + i * 16 * BK);
        // the W fragments come in two halves (TN at a time) to keep 20 fewer registers live; the next tile's DMA goes out in
        // 2 * KS2 shares, one before each half of this K step's MFMAs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bf[j] = *reinterpret_cast<const half8*>(Bb + offb + (h * TN + j) * 16 * BK);
          if constexpr (NEXT && SPREAD) {
            if (h == 0) issue_range(nbuf, std::integral_constant<int, (2 * ks) * LPT / (2 * KS2)>{}, std::integral_constant<int, (2 * ks + 1) * LPT / (2 * KS2)>{});
            else issue_range(nbuf, std::integral_constant<int, (2 * ks + 1) * LPT / (2 * KS2)>{}, std::integral_constant<int, (2 * ks + 2) * LPT / (2 * KS2)>{});
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i)
              acc16[i][h * TN + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc16[i][h * TN + j], 0, 0, 0);
        }
      }(), ...);
    }(std::make_integer_sequence<int, KS2>{});
    }
    if constexpr (NEXT && SPREAD) advance_tile();
  };
  {
    int kt = 0;
    for (; kt + STAGES - 1 < nk; ++kt) k_tile(kt, std::true_type{});
    for (; kt < nk; ++kt) k_tile(kt, std::false_type{});
  }
  dbg_stamp(g, 2);

  if constexpr (M16) {
    gemm_epilogue<WM, WN, TN, TM, true>(g, [&](int i, int j, int q) {
      const float4v v = acc16[2 * i + (q >> 1)][2 * j + (q & 1)];
      return make_float4(v[0], v[1], v[2], v[3]);
    }, lds, m0, n0, wave, lane, z, split);
  } else {
    gemm_epilogue<WM, WN, TN, TM, false>(g, [&](int i, int j, int q) {
      return make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }, lds, m0, n0, wave, lane, z, split);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dbg_stamp(g, 3);
}

// sum the split-K slabs in a fixed order and apply the epilogue (8 columns per thread)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n8 = g.N / 8;
  if (i >= (long long)g.M * n8) return;
  const int row = (int)(i / n8), col0 = (int)(i - (long long)row * n8) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.0f;
  for (int s = 0; s < g.ksplit; ++s) {
    const float4* p = reinterpret_cast<const float4*>(g.partial + ((long long)s * g.M + row) * g.N + col0);
    float4 x = p[0], y = p[1];
    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
    v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
  }
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)epilogue_value(g, v[j], row, col0 + j, g.res);
  *reinterpret_cast<half8*>(g.out + (long long)row * g.ldo + col0) = o;
}

}  // namespace sd

using namespace sd;

extern "C" size_t sd_conv_gemm_workspace_bytes(void) { return (size_t)64 << 20; }

extern "C" int sd_debug_timestamps(unsigned long long* host_dst, int n_blocks) {
  if (!host_dst || n_blocks <= 0 || n_blocks > DBG_BLOCKS) return fail(COMA_E_INVALID, "sd_debug_timestamps: bad arguments");
  if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dbg_stamps), (size_t)n_blocks * 4 * sizeof(unsigned long long)) != hipSuccess)
    return fail(COMA_E_LAUNCH, "sd_debug_timestamps: copy failed");
  return COMA_OK;
}

extern "C" int sd_conv_gemm_f16(const sd_conv_gemm_desc* d_in, void* stream) {
  if (sd::plan_recording()) {
    if (!d_in) return sd::fail(COMA_E_INVALID, "sd_conv_gemm_f16: null descriptor");
    const sd_conv_gemm_desc& d = *d_in;
    sd::PlanRec r{};
    r.kind = sd::PK_CONV;
    void* ps[10] = {(void*)d.a0, (void*)d.a1, (void*)d.w, (void*)d.bias, (void*)d.bias_bn, (void*)d.res, d.out, d.workspace, d.colstats, d.out_t};
    for (int k = 0; k < 10; ++k) r.p[k] = ps[k];
    const int64_t is[22] = {d.c0, d.c1, d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.taps, d.stride, d.upsample, d.pad, d.n, d.ldbb, d.ldr, d.ldo,
                            d.epi, d.nbatch_z, d.stride_a, d.stride_w, d.stride_out, d.stride_res, (int64_t)d.workspace_bytes};
    for (int k = 0; k < 22; ++k) r.i[k] = is[k];
    if (d.out_t && (d.n_split < 0 || d.n_split >= (1 << 20) || d.ldo_t < 0 || d.ldo_t >= (1 << 20) || d.rows_per_sample < 0 || d.rows_per_sample >= (1 << 20)))
      return sd::fail(COMA_E_INVALID, "sd_conv_gemm_f16: out_t sizes out of range");
    if (d.phase < 0 || d.phase > 4) return sd::fail(COMA_E_INVALID, "sd_conv_gemm_f16: phase must be 0..4");
    r.i[22] = (int64_t)d.n_split | ((int64_t)d.ldo_t << 20) | ((int64_t)d.rows_per_sample << 40) | ((int64_t)d.phase << 60);    // three 20-bit fields + the phase
    return sd::plan_record(r);
  }
  if (!d_in) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: null descriptor");
  // SD_GEMM_TUNE=<mask> (or _1X1 / _3X3 for those launches only): OR tuning-knob bits (SD_EPI_TUNING_MASK) into every launch -- lets a dispatch rule be A/B-tested inside the
  // captured UNet (scripts/time_unet.py), where cache state differs from a layer timed alone (profiles/r02_notes.md section 14)
  static const int tune_env = getenv("SD_GEMM_TUNE") ? (int)strtol(getenv("SD_GEMM_TUNE"), nullptr, 0) & SD_EPI_TUNING_MASK : 0;
  sd_conv_gemm_desc d_copy = *d_in;
  static const int tune1_env = getenv("SD_GEMM_TUNE_1X1") ? (int)strtol(getenv("SD_GEMM_TUNE_1X1"), nullptr, 0) & SD_EPI_TUNING_MASK : 0;
  static const int tune9_env = getenv("SD_GEMM_TUNE_3X3") ? (int)strtol(getenv("SD_GEMM_TUNE_3X3"), nullptr, 0) & SD_EPI_TUNING_MASK : 0;
  d_copy.epi |= tune_env | (d_in->taps == 1 ? tune1_env : tune9_env);
  {   // SD_GEMM_FORCE="N,K,mask;N,K,mask;...": knob bits for the launches with that N and K only (K = taps * channels)
    struct Force { int n, k, mask; };
    static const std::vector<Force> forced = [] {
      std::vector<Force> v;
      const char* e = getenv("SD_GEMM_FORCE");
      while (e && *e) {
        Force f{0, 0, 0};
        char* end = nullptr;
        f.n = (int)strtol(e, &end, 0); if (*end != ',') break;
        f.k = (int)strtol(end + 1, &end, 0); if (*end != ',') break;
        f.mask = (int)strtol(end + 1, &end, 0) & SD_EPI_TUNING_MASK;
        v.push_back(f);
        e = *end == ';' ? end + 1 : end;
        if (*end != ';') break;
      }
      return v;
    }();
    for (const Force& f : forced)
      if (f.n == d_in->n && f.k == d_in->taps * (d_in->c0 + d_in->c1)) d_copy.epi |= f.mask;
  }
  // a plain product with more than 65536 rows given as batch x 1 x 1: the kernel keeps the sample index in 16 bits, so fold a power of
  // two of the rows into the "image" (same row order; not with a per-sample bias, whose index is the sample)
  if (d_copy.taps == 1 && d_copy.in_h == 1 && d_copy.in_w == 1 && d_copy.out_h == 1 && d_copy.out_w == 1 && d_copy.batch > 65536 && !d_copy.bias_bn &&
      !d_copy.out_t) {
    int f = 1;
    while (d_copy.batch / f > 65536 && (d_copy.batch % (2 * f)) == 0 && f < 32768) f *= 2;
    d_copy.batch /= f; d_copy.in_w = d_copy.out_w = f;
  }
  const sd_conv_gemm_desc* d = &d_copy;
  if (!d->a0 || !d->w || !d->out) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: null pointer");
  const int phase = d->phase;
  if (phase < 0 || phase > 4) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: phase must be 0..4");
  if (phase) {
    // sub-pixel phase of conv3x3(nearest-upsample-x2(x)): a 2 x 2 window over the source, rows scattered to the output pixels of that parity
    if (d->taps != 4 || d->stride != 1 || d->upsample || d->out_h != d->in_h || d->out_w != d->in_w || (d->in_w & (d->in_w - 1)) ||
        (d->nbatch_z > 1) || d->res || d->bias_bn || d->out_t || (d->epi & ~SD_EPI_TUNING_MASK) || d->n % 8 ||
        (d->ldo > 0 && d->ldo % 8))
      return fail(COMA_E_INVALID, "sd_conv_gemm_f16: a phase launch needs taps = 4, stride 1, out = in size, in_w a power of two, n %% 8 == 0 and "
                                  "a plain epilogue (bias, optional colstats)");
    const long long ldo_eff = d->ldo > 0 ? d->ldo : d->n;
    if ((4LL * d->batch * d->in_h * d->in_w + 2048) * ldo_eff * 2 >= 0x7fffffffLL)
      return fail(COMA_E_INVALID, "sd_conv_gemm_f16: the output of a phase launch must stay below 2 GiB");
  } else if (d->taps != 1 && d->taps != 9) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: taps must be 1 or 9 (4 only with a phase)");
  if (d->stride != 1 && d->stride != 2) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: stride must be 1 or 2");
  if (d->c0 <= 0 || d->c0 % BKMIN || d->c1 < 0 || d->c1 % BKMIN || (d->c1 > 0 && !d->a1))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: source channels must be multiples of %d (c0=%d c1=%d)", BKMIN, d->c0, d->c1);
  if (d->batch <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->n <= 0)
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: bad sizes");
  if (d->pad < 0 || d->pad > 1) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: pad must be 0 or 1");
  if (d->upsample && (d->stride != 1 || d->taps != 9))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: upsample only with 3x3 stride 1");
  const int nz = d->nbatch_z > 0 ? d->nbatch_z : 1;
  GemmArgs g;
  g.a0 = (const _Float16*)d->a0; g.a1 = (const _Float16*)d->a1; g.c0 = d->c0; g.c1 = d->c1;
  g.in_h = d->in_h; g.in_w = d->in_w; g.out_h = d->out_h; g.out_w = d->out_w;
  g.taps = d->taps; g.stride = d->stride; g.upsample = d->upsample; g.pad = d->taps == 9 ? d->pad : 0;
  g.kw = 3; g.pad_x = g.pad; g.phase = phase; g.ph_wshift = 0; g.ph_rowoff = 0;
  if (phase) {
    const int pa = (phase - 1) >> 1, pb = (phase - 1) & 1;
    g.kw = 2; g.pad = 1 - pa; g.pad_x = 1 - pb;                  // window rows y - 1 + a, y + a; columns x - 1 + b, x + b
    while ((1 << g.ph_wshift) < d->in_w) ++g.ph_wshift;
    g.ph_rowoff = pa * 2 * d->in_w + pb;
  }
  g.ldbb = d->ldbb > 0 ? d->ldbb : d->n;
  g.rows_per_batch = d->out_h * d->out_w;
  long long M = (long long)d->batch * g.rows_per_batch;
  if (M > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: M too large");
  if (d->batch > 65536) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: at most 65536 samples per launch (the kernel keeps the sample index in 16 bits): "
                                    "pass a large row count as batch x in_h x in_w");
  if (d->taps != 1 && d->batch > 65535) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: a 3x3 convolution takes at most 65535 samples per launch");
  // the LDS-DMA addresses rows with 32-bit byte offsets from the tensor base (bit 31 marks zero padding)
  const long long a_bytes = (long long)d->batch * d->in_h * d->in_w * (d->c0 > d->c1 ? d->c0 : d->c1) * 2;
  const long long w_bytes = (long long)d->n * d->taps * (d->c0 + d->c1) * 2;
  if (a_bytes >= 0x80000000LL || w_bytes >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: a source tensor (%lld B) or the weights (%lld B) exceed 2 GiB per launch; split the batch",
                a_bytes, w_bytes);
  g.M = (int)M; g.N = d->n; g.K = d->taps * (d->c0 + d->c1);
  g.n_valid = d->n;
  if (d->epi & SD_EPI_PERM16_N) {
    g.N = (d->n + 15) & ~15;            // whole 16-column groups: positions past n hold clamped (finite) rows the consumer masks
    if ((d->ldo > 0 ? d->ldo : d->n) < g.N) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: SD_EPI_PERM16_N needs ldo >= n rounded up to 16");
    if ((d->epi & SD_EPI_GEGLU) || d->res || d->bias_bn || d->colstats) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: SD_EPI_PERM16_N takes no GEGLU / residual / per-sample bias / colstats");
  }
  if (d->epi & SD_EPI_PERM32_N) {
    if ((d->epi & (SD_EPI_PERM16_N | SD_EPI_GEGLU)) || d->n % 32 || d->res || d->bias_bn || d->colstats)
      return fail(COMA_E_INVALID, "sd_conv_gemm_f16: SD_EPI_PERM32_N needs n %% 32 == 0 and takes no other column-dependent epilogue");
  }
  g.w = (const _Float16*)d->w; g.bias = (const _Float16*)d->bias; g.bias_bn = (const _Float16*)d->bias_bn;
  g.res = (const _Float16*)d->res; g.ldr = d->ldr > 0 ? d->ldr : d->n;
  g.out = (_Float16*)d->out; g.epi = d->epi;
  const bool geglu = (d->epi & SD_EPI_GEGLU) != 0;
  g.ldo = d->ldo > 0 ? d->ldo : (geglu ? d->n / 2 : d->n);
  g.sa = d->stride_a; g.sw = d->stride_w; g.so = d->stride_out; g.sr = d->stride_res;
  if (geglu && (d->n % 128 != 0 || d->bias_bn || d->res))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: GEGLU needs N %% 128 == 0 and no residual / batch bias");
  // ---- tile configuration
  const bool k64 = d->c0 % 64 == 0 && d->c1 % 64 == 0;
  // BK = 64 with two stages from K = 1024 (was 2048: the K = 1280 linears of the 16 x 16 level, -0.15 ms per UNet forward measured
  // inside the captured graph)
  const bool deep = g.K >= 1024 && k64;
  // 256 x 320 tile: N a multiple of 320 (every layer of the SD UNet), enough rows to fill the chip with 1 block / CU
  // ... or z-batched plain products whose tiles fill it together (the 16 plane products of a Winograd convolution at the 16 x 16 level:
  // 4 x 4 tiles x 16 planes = one block per CU, where the generic 128 x 128 tile needs 2.5 rounds; SD_GEMM_ZBIG=0 switches it off for A/B)
  static const bool zbig_on = !getenv("SD_GEMM_ZBIG") || atoi(getenv("SD_GEMM_ZBIG")) != 0;
  const bool zplain = nz > 1 && zbig_on && d->taps == 1 && !d->colstats && !d->out_t &&
                      !(d->epi & ~SD_EPI_TUNING_MASK);
  const bool big = !geglu && (nz == 1 || zplain) && k64 && g.N % 320 == 0 && (long long)((g.M + 255) / 256) * (g.N / 320) * nz >= 192 &&
                   !(d->epi & ((1 << 20) | (1 << 21)));
  // GEGLU (needs an even number of MFMA column tiles per wave): 256 x 256, 8 waves, wave tile 64 x 128
  const bool big_geglu = geglu && nz == 1 && k64 && g.N % 256 == 0 && g.M >= 256 * 16 && !(d->epi & (1 << 20));
  // VAE widths (128 / 256 / 512 channels at up to 512 x 512 pixels): 256 x 256 and 256 x 128 tiles, 8 waves
  const bool big256 = !geglu && !big && nz == 1 && k64 && g.N % 256 == 0 && g.M >= 256 * 128 && !(d->epi & (1 << 20));
  const bool big128 = !geglu && !big && !big256 && nz == 1 && k64 && g.N % 128 == 0 && g.M >= 256 * 256 && !(d->epi & (1 << 20));
  // 256 x 128 with FOUR waves (wave tile 64 x 128), BK = 32, 3 stages = 72 KiB of LDS, so that two independent blocks share a CU
  // (one block's epilogue and barriers run under the other's K loop): the 128-channel 3x3 convs of the VAE at 512 x 512
  // (K = 1152, where the output pass is a quarter of the launch).  800 TF/s isolated, the same as 512 x 128 with 8 waves (789) and
  // +15 % over 128 x 128; at K >= 2304 and for N >= 256 the 8-wave tiles win by 10-15 % (profiles/r02_notes.md section 10)
  const bool tall128 = big128 && g.K <= 1152 && !(d->epi & (1 << 20));
  // 128 x 320, 4 waves (wave tile 64 x 160): mid-size M where 256-row tiles would leave CUs idle
  // ... and, split in two along K, for the deep 3x3 convs at 16 x 16 (M = 4096, N = 1280, K >= 11520): 32 x 4 tiles x 2 splits = one
  // block per CU, where 128 x 128 tiles leave 320 blocks for 512 slots (+4 % at K = 11520, +15 % at K = 23040, profiles/r02_notes.md 12)
  const bool midsk = !big && !geglu && nz == 1 && k64 && g.N % 320 == 0 && g.N >= 1280 && g.M >= 4096 && g.M < 128 * 64 && g.K >= 8192 &&
                     !d->colstats && d->workspace && d->workspace_bytes >= (size_t)2 * g.M * g.N * sizeof(float) &&
                     (long long)((g.M + 127) / 128) * (g.N / 320) * 2 <= 256 && !(d->epi & ((1 << 20) | (1 << 21)));
  const bool mid = (!big && !big256 && !big128 && !geglu && nz == 1 && g.N % 320 == 0 && (g.M >= 128 * 64 || (d->epi & (1 << 21))) &&
                    (g.N <= 640 || (d->epi & (1 << 21))) && !(d->epi & (1 << 20))) || midsk;
  // 128 x 320 with EIGHT waves (wave tile 32 x 160, two waves per SIMD) instead of four: the partner wave covers each
  // wave's LDS-read / DMA-issue latency, which the 4-wave tile leaves exposed (knob 23 selects the 4-wave form)
  const bool mid8 = mid && k64 && g.K >= 256 && !(d->epi & (1 << 23));
  const bool wide = g.N % 128 == 0 || g.N > 256;
  const int bm = ((big || big_geglu || big256 || big128) ? 256 : 128);
  const int bn = (big || mid) ? 320 : ((big_geglu || big256) ? 256 : ((wide || big128) ? 128 : 64));
  const int bk = ((mid && !mid8) || tall128) ? 32 : ((big || big_geglu || big256 || big128 || deep || mid8) ? 64 : 32);
  const unsigned gx = (unsigned)((g.M + bm - 1) / bm), gy = (unsigned)((g.N + bn - 1) / bn);
  // split-K when the tile grid cannot fill the chip: as many splits as keep every block resident at once (2 per CU,
  // 512 in total -- a partial second round costs more than it buys), at least 384 of K per split
  g.ksplit = 1;
  static const bool dbg_on = getenv("SD_GEMM_DBG") != nullptr;
  g.dbg = nullptr;
  if (dbg_on) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg_stamps)) == hipSuccess) g.dbg = (unsigned long long*)p; }
  g.partial = (float*)d->workspace;
  g.colstats = (float*)d->colstats;
  g.out_t = (_Float16*)d->out_t; g.n_split = d->n_split; g.ldo_t = d->ldo_t; g.rps = d->rows_per_sample;
  if (g.out_t) {
    if (geglu || nz != 1 || d->taps != 1 || d->c1 > 0 || d->bias || d->bias_bn || d->res || d->colstats ||
        (d->epi & ~SD_EPI_TUNING_MASK) || d->n_split <= 0 || d->n_split % 640 || (d->n - d->n_split) <= 0 || (d->n - d->n_split) % 640 ||
        g.M % 32 || d->rows_per_sample <= 0 || d->rows_per_sample % 32 || g.M % d->rows_per_sample || d->ldo_t % 8 || d->ldo_t < d->rows_per_sample ||
        g.ldo < d->n_split || g.ldo % 8)
      return fail(COMA_E_INVALID, "sd_conv_gemm_f16: out_t needs a plain linear (no bias / residual / GEGLU / statistics / batching), n_split and "
                                  "n - n_split multiples of 640, M and rows_per_sample multiples of 32, ldo >= n_split, ldo_t >= rows_per_sample, both % 8 == 0");
    if (d->n_split % bn) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: out_t: the %d-column tile of this shape does not divide n_split = %d", bn, d->n_split);
  }
  if (g.colstats && (geglu || nz != 1 || g.M % 32 || g.N % 8 || (d->epi & SD_EPI_BIAS_ROWS) || g.ldo % 8 || (g.res && g.ldr % 8) ||
                     (g.bias_bn && (g.res || g.rows_per_batch % 32 || g.ldbb % 8)) || (long long)(g.M + 512) * g.ldo * 2 >= 0x7fffffffLL ||
                     (g.res && (long long)(g.M + 512) * g.ldr * 2 >= 0x7fffffffLL)))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: colstats needs M %% 32 == 0, N %% 8 == 0, 16-byte aligned rows, < 2 GiB tensors, no GEGLU / batching");
  // tap-minor K order only where it measured faster: the 320-column tiles with a single N tile (+7 % at 64x64, C = 320; elsewhere the
  // second N tile re-reads A from L2 anyway and the order is neutral to -10 %, profiles/r02_notes.md).  The 8-wave 128 x 320 tile
  // follows the same rule so that a half-batch launch of such a layer (the shared CFG prefix) accumulates in the same order as the
  // full-batch one and stays bit-identical to it.
  if (!((big || mid8) && gy == 1)) g.epi |= (1 << 28);
  const long long blocks = (long long)gx * gy;
  const int nk = g.K / bk;
  const int min_tiles = 384 / bk;
  if (!phase && !g.colstats && !g.out_t && !big && !big256 && !big128 && nz == 1 && !geglu && d->workspace && blocks < 200 && nk >= 2 * min_tiles && g.N % 8 == 0 && g.ldo % 8 == 0) {
    int s = (int)((mid8 ? 256 : 512) / blocks);         // the 8-wave 128 x 320 tile is resident once per CU, the others twice
    if (s > nk / min_tiles) s = nk / min_tiles;
    if (s > 16) s = 16;
    while (s > 1 && (size_t)s * g.M * g.N * sizeof(float) > d->workspace_bytes) --s;
    if (s > 1) g.ksplit = s;
    const int force = (d->epi >> 24) & 15;      // tuning knob: forced split factor
    if (force && (size_t)force * g.M * g.N * sizeof(float) <= d->workspace_bytes) g.ksplit = force > nk ? nk : force;
  }
  const long long lin_blocks = gx >= 16 ? 8LL * ((gx + 7) / 8) * gy : (long long)gx * gy;   // XCD-banded order (see kernel)
  if (lin_blocks > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: grid too large");
  dim3 grid((unsigned)lin_blocks, (unsigned)(g.ksplit > 1 ? g.ksplit : nz));
  hipStream_t st = (hipStream_t)stream;
  // DMA issue spread over the K steps of a tile (3x3: +0.5 % of a UNet forward) or in one burst (1x1: +0.6 %) -- both measured inside the
  // captured forward; knob 22 forces the burst
  const bool spread = !(d->epi & (1 << 22)) && d->taps == 9;
#define GEMM_LAUNCH(WM_, WN_, TN_, BK_, ST_, SP_, TM_, THREADS_)                                                                  \
  do {                                                                                                                           \
    if (m16) hipLaunchKernelGGL((conv_gemm_kernel<WM_, WN_, TN_, BK_, ST_, SP_, TM_, true>), grid, dim3(THREADS_), 0, st, g);    \
    else hipLaunchKernelGGL((conv_gemm_kernel<WM_, WN_, TN_, BK_, ST_, SP_, TM_, false>), grid, dim3(THREADS_), 0, st, g);       \
  } while (0)
  // 16x16x32 MFMAs in the K loop of every launch with K >= 256 (measured inside the captured graphs, A B A B: UNet forward -0.2 ms from the
  // 3x3 convolutions -- the power-limited ones -- and another -0.45 ms from the 1x1 / linear launches, most of it at K = 320; VAE decode
  // -0.65 ms).  SD_GEMM_M16 / SD_GEMM_M16_1X1 = <minimum K, 0 = off> override the rule for A/B runs.
  static const int m16_env = getenv("SD_GEMM_M16") ? atoi(getenv("SD_GEMM_M16")) : 256;
  static const int m16_1x1 = getenv("SD_GEMM_M16_1X1") ? atoi(getenv("SD_GEMM_M16_1X1")) : 256;
  const bool m16 = (d->taps == 9 ? (m16_env && g.K >= m16_env) : (m16_1x1 && g.K >= m16_1x1));
  if (big && spread) GEMM_LAUNCH(4, 2, 5, 64, 2, true, 2, 512);
  else if (wide && deep && spread && !big_geglu && !big256 && !big128 && !mid) GEMM_LAUNCH(2, 2, 2, 64, 2, true, 2, 256);
  else if (big) GEMM_LAUNCH(4, 2, 5, 64, 2, false, 2, 512);
  else if ((big_geglu || big256) && spread) GEMM_LAUNCH(4, 2, 4, 64, 2, true, 2, 512);
  else if (big_geglu || big256) GEMM_LAUNCH(4, 2, 4, 64, 2, false, 2, 512);
  else if (tall128) GEMM_LAUNCH(4, 1, 4, 32, 3, false, 2, 256);
  else if (big128) GEMM_LAUNCH(4, 2, 2, 64, 2, false, 2, 512);
  else if (mid8) GEMM_LAUNCH(4, 2, 5, 64, 2, true, 1, 512);
  else if (mid) GEMM_LAUNCH(2, 2, 5, 32, 3, false, 2, 256);
  else if (wide && deep) GEMM_LAUNCH(2, 2, 2, 64, 2, false, 2, 256);
  else if (wide) GEMM_LAUNCH(2, 2, 2, 32, 4, false, 2, 256);
  else if (deep) GEMM_LAUNCH(2, 2, 1, 64, 2, false, 2, 256);
  else GEMM_LAUNCH(2, 2, 1, 32, 4, false, 2, 256);
#undef GEMM_LAUNCH
  if (g.ksplit > 1) {
    const long long n8 = (long long)g.M * (g.N / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, g);
  }
  return check_launch("conv_gemm_kernel");
}
