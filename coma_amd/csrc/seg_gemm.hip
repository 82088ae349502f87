// seg_gemm.hip -- fp32 implicit-GEMM convolution / linear layer on the fp32 MFMA of gfx950 (v_mfma_f32_32x32x2_f32): every
// convolution and fully connected layer of the person-segmentation network behind PointRendPredictor.__call__
// (utils/adaptive_mask_inpainting.py:1225-1236 -> detectron2 DefaultPredictor; src/generation/segment_human.py:24-169), which the
// reference runs in fp32 -- so this path computes in fp32 too: exact-f32 MFMA (a k-ordered fmaf chain, one rounding
// per product), roofline = the 157.3 TFLOP/s fp32 matrix peak, not the fp16 one.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] (+ res) ),  m = (b, oy, ox),  k = (ky, kx, c)
//
// * A is gathered on the fly from an NHWC fp32 tensor (kh x kw window, stride, zero padding); a linear layer is the 1 x 1 case.
// * FrozenBatchNorm is folded into W / bias by the host (coma_amd/seg/weights.py), ReLU is an epilogue flag, the bottleneck shortcut
//   is the residual operand, and FPN's top-down pathway (`lateral + F.interpolate(coarser, 2, "nearest")`) is the residual read at
//   (oy >> 1, ox >> 1) -- no upsampled tensor is ever written.
// * rows can be limited by counts that live on the device: the M rows form units of `unit_rows` rows (one per image), and of unit u only
//   the first m_dev[u] * rows_per_item rows are computed -- the mask heads run on however many detections each image produced without the
//   host ever learning the number; workgroups with no valid row exit, invalid rows are neither gathered nor stored.
// Workgroup = 4 waves, tile 128 x 128 (2 x 2 waves, 2 x 2 MFMA tiles each; two workgroups per CU), 128 x 64 or 128 x 32 (4 x 1 waves; three per
// CU); the grid is 1-D in an XCD-banded tile order; K in chunks of 32 through two LDS stages (unpadded rows, XOR-swizzled 16-byte slots, one
// barrier per chunk) and two register sets: a chunk's global loads are issued two chunks ahead and stored to LDS in the
// middle of the chunk before it is used; operand addresses of the 1 x 1 / 3 x 3 layers advance incrementally on the scalar unit (a VALU
// instruction does not overlap with the issuing wave's own MFMAs, so address arithmetic is paid in matrix-pipe time).  Per 8 k values a wave reads
// ONE ds_read_b128 per operand tile: lanes 0-31 take k = 8g .. 8g+3, lanes 32-63 take k = 8g+4 .. 8g+7, and MFMA step e consumes element e
// of both (the k order inside a sum is free as long as A and W agree) -- 16 MFMAs (1024 cycles) per 4 LDS reads.
// Epilogue: the accumulators go through LDS (the two operand stages are free by then) and leave as whole rows -- float4 stores of 128 / 64 / 32
// consecutive columns, the residual and the bias read the same way.  (Storing straight from the MFMA layout is one 4-byte column per lane,
// two 128-byte row pieces per instruction: the layers with K <= 128 were 4 x off the HBM floor that way, profiles/r06_notes.md 3.)
// Split-K (deterministic): launches of fewer than 1024 tiles with a long K cut K into S slices (blockIdx.z) when the per-CU cost model in
// launch_gemm says so; every slice stores its raw partial tile into a slab of the caller's workspace and seg_splitk_reduce_kernel sums the slabs
// in slice order and applies bias / residual / ReLU -- the same result on every run (no atomics).
// Algorithmic bytes per launch: A read once (x taps when the window overlaps is NOT counted: the re-reads hit L2), W once, out once.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "sd_plan.h"
#include "../../include/seg_hip.h"

namespace seg {

using coma::check_launch;
using coma::fail;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // operand registers: a first-class vector value (HIP's float4 is a struct)

__device__ float kZerosDev[4] = {0.f, 0.f, 0.f, 0.f};      // what an out-of-range operand load reads; reaches the kernel as an ordinary global pointer
                                                       // (GemmArgs::zeros): selecting between a kernel-argument pointer and the symbol itself made every
                                                       // load a flat_load and sent the register sets to scratch

constexpr int kBK = 32, kLd = kBK;          // LDS row stride in floats: 128 B, no padding -- the eight 16-byte slots of a row are XOR-swizzled with
                                            // the row index (slot ^ (row & 7)), which spreads eight consecutive rows over all banks like the 144-byte
                                            // stride did, and lets the 128 x 64 tile fit three times per CU (3 x 48 KB; 55 KB padded: two)

struct GemmArgs {
  const float* x; const float* w; const float* bias; const float* res; float* out; const int* m_dev;
  int B, H, W, C, ldx, N, Kpad, kh, kw, stride, pad, OH, OW, ldr, res_mode, ldo, relu, rows_per_item, unit_rows;
  long long M;
  const float* zeros;                        // 16 bytes of zeros in global memory
  float* ws; int splits, nk_per, ws_ld;      // split-K: slab z of the workspace is [gx * BM][ws_ld] raw partial sums
  int gx, gy, band;                                                  // row / column tiles (the grid is 1-D: see the tile order in the kernel)
};

template <int WM, int WN, int TM, int TN, bool UNI, bool PRE>
__global__ __launch_bounds__(256, (WM * TM * WN * TN <= 8 ? 3 : 2)) void conv_gemm_f32_kernel(const GemmArgs a) {      // 128 x 64 / 128 x 32 tiles: three per CU
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NA = BM / 32, NB = BN / 32;
  static_assert(BN % 32 == 0 && BM % 32 == 0, "whole 32-row load steps");       // float4 loads per thread and chunk (A rows / W rows in steps of 32)
  extern __shared__ float lds[];
  float* As = lds;                                       // [2][BM][kLd]
  float* Bs = lds + 2 * BM * kLd;                        // [2][BN][kLd]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (one L2 each).  Workgroup L of the
  // 1-D grid runs on XCD L % 8 as that XCD's (L / 8)-th workgroup; XCD i is given the CONTIGUOUS band of tiles [start_i, start_i + count_i)
  // of the (row tile major, column tile minor) order, so the column tiles that share an A row band -- and the neighbouring row bands whose
  // 3 x 3 windows overlap -- are resident on ONE XCD at about the same time and the operand is pulled into one L2 instead of gy of them.
  int tile_id;
  {
    const int T = a.gx * a.gy, L = (int)blockIdx.x;
    const int xcd = L & 7, slot = L >> 3, q = T >> 3, r = T & 7;
    tile_id = a.band ? xcd * q + (xcd < r ? xcd : r) + slot : L;
  }
  int tmx, tnx;
  if (a.band) { tmx = tile_id / a.gy; tnx = tile_id - tmx * a.gy; }       // banded: column tiles of one row band adjacent
  else { tnx = tile_id / a.gx; tmx = tile_id - tnx * a.gx; }             // plain: row tiles fastest (the order of a 2-D grid)
  const int m0 = tmx * BM;                               // M < 2^31 - BM (checked by the launcher): row arithmetic in 32 bits
  const int n0 = tnx * BN;
  const int Mv = (int)a.M;
  if (m0 >= Mv) return;
  auto row_ok = [&](int m) __attribute__((always_inline)) {
    if (m >= Mv) return false;
    if (!a.m_dev) return true;
    const unsigned u = (unsigned)m / (unsigned)a.unit_rows;
    return (long long)(m - (int)u * a.unit_rows) < (long long)a.m_dev[u] * a.rows_per_item;
  };
  if (a.m_dev) {                                         // valid rows are a prefix of every unit: does any unit this tile touches have one here?
    const int last = (m0 + BM - 1 < Mv ? m0 + BM - 1 : Mv - 1);
    bool any = false;
    for (int u = m0 / a.unit_rows; u <= last / a.unit_rows; ++u) {
      const int lo = u * a.unit_rows > m0 ? u * a.unit_rows : m0;
      any = any || row_ok(lo);
    }
    if (!any) return;
  }

  // ---- per-thread gather geometry: rows r = tid / 8 + 32 j, k offset (tid % 8) * 4 inside a chunk
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  long long rbase[NA];                                   // element offset of (b, iy0, ix0) -- may point outside; bounds are checked per tap
  int riy[NA], rix[NA];
  const int ohw = a.OH * a.OW;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + lr + 32 * j;
    if (row_ok(m)) {
      const int b = (int)((unsigned)m / (unsigned)ohw), rem = m - b * ohw;
      const int oy = rem / a.OW, ox = rem - oy * a.OW;
      riy[j] = oy * a.stride - a.pad;
      rix[j] = ox * a.stride - a.pad;
      rbase[j] = (long long)b * a.H * a.W;
    } else {
      riy[j] = -(1 << 28); rix[j] = 0; rbase[j] = 0;     // never in bounds
    }
  }
  const int ntaps = a.kh * a.kw;
  const int nk_all = a.Kpad / kBK;
  const int kc0 = a.splits > 1 ? (int)blockIdx.z * a.nk_per : 0;
  const int nk = a.splits > 1 ? (kc0 + a.nk_per < nk_all ? a.nk_per : nk_all - kc0) : nk_all;      // >= 1 by construction of the launch

  f32x4 ra[NA], rb[NB];
  // Uniform-tap path (UNI: C % 32 == 0, every layer but the stem): a chunk of 32 k values lies inside ONE tap, so (ky, kx, c0) are wave-
  // uniform and advance incrementally -- no division in the loop, one 64-bit add per operand row (VALU instructions do not overlap with the
  // issuing wave's own MFMAs, profiles/r01_probe_mfma_valu.txt: the per-chunk address arithmetic is paid in matrix-pipe time).
  const float* xrow[NA];                                 // (b, oy * stride - pad, ox * stride - pad, channel lk): only dereferenced for taps in bounds
  const float* wrow[NB];
  int u_ky = 0, u_kx = 0, u_c0 = 0;
  long long u_koff = 0;
  if constexpr (UNI) {
#pragma unroll
    for (int j = 0; j < NA; ++j) xrow[j] = a.x + (rbase[j] + (long long)riy[j] * a.W + rix[j]) * a.ldx + lk;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = n0 + lr + 32 * j;
      wrow[j] = n < a.N ? a.w + (long long)n * a.Kpad + lk : nullptr;
    }
    const int tap0 = (kc0 * kBK) / a.C;
    u_c0 = kc0 * kBK - tap0 * a.C;
    u_ky = tap0 / a.kw;
    u_kx = tap0 - u_ky * a.kw;
    u_koff = (long long)kc0 * kBK;
  }
  auto gload = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], int kc) __attribute__((always_inline)) {
    // branch-free and wait-free: an invalid tap / row / column reads 16 bytes of zeros (GemmArgs::zeros) instead -- a conditional load costs
    // a divergent branch per load, and a select on the loaded value would make the wave wait for the load right here instead of at the LDS
    // store one chunk later
    if constexpr (UNI) {                                 // called once per chunk in ascending order: the running state IS chunk kc
      const long long tapoff = ((long long)u_ky * a.W + u_kx) * a.ldx + u_c0;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int iy = riy[j] + u_ky, ix = rix[j] + u_kx;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        ra[j] = *reinterpret_cast<const f32x4*>(ok ? xrow[j] + tapoff : a.zeros);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wrow[j] ? wrow[j] + u_koff : a.zeros);
      u_koff += kBK;
      u_c0 += kBK;
      if (u_c0 >= a.C) {
        u_c0 = 0;
        if (++u_kx == a.kw) { u_kx = 0; ++u_ky; }
      }
    } else {
      const int k = kc * kBK + lk;
      const int tap = k / a.C, c = k - tap * a.C;
      const int ky = tap / a.kw, kx = tap - ky * a.kw;
      const bool tv = tap < ntaps;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int iy = riy[j] + ky, ix = rix[j] + kx;
        const bool ok = tv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const float* src = ok ? a.x + (rbase[j] + (long long)iy * a.W + ix) * a.ldx + c : a.zeros;
        ra[j] = *reinterpret_cast<const f32x4*>(src);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int n = n0 + lr + 32 * j;
        const bool ok = n < a.N;                          // rows lr + 32 j >= BN do not exist: NB * 32 == BN for every instantiated tile
        const float* src = ok ? a.w + (long long)n * a.Kpad + k : a.zeros;
        rb[j] = *reinterpret_cast<const f32x4*>(src);
      }
    }
  };
  const int lk_sw = (((tid & 7) ^ (lr & 7))) * 4;         // swizzled slot of this thread's 16 bytes (stage bases and the 32-row steps are multiples of 8 rows)
  auto lstore = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(As + (buf * BM + lr + 32 * j) * kLd + lk_sw) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(Bs + (buf * BN + lr + 32 * j) * kLd + lk_sw) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  auto mma = [&](int buf, int g) __attribute__((always_inline)) {                       // 8 k values: one ds_read_b128 per operand tile, 4 MFMA steps on each tile pair
    const int sw = ((lh + 2 * g) ^ (li & 7)) * 4;         // slot lh + 2 g of row li, swizzled (every row offset below is a multiple of 8 rows)
    const float* Ab = As + (buf * BM + wm * TM * 32 + li) * kLd + sw;
    const float* Bb = Bs + (buf * BN + wn * TN * 32 + li) * kLd + sw;
    float4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * kLd);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * kLd);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
      }
  };
  // ---- PRE (launches with a residual operand and no split-K): the residual rows and the bias of this thread's output pieces are requested at
  // the top of the LAST K chunk (whose load slot is free), so their memory latency hides under that chunk's MFMAs and the LDS staging
  // instead of being paid per group of passes after it -- the K <= 512 1 x 1 layers with a shortcut operand were 2 x off both of their
  // floors (326 -> 220 us at K = 64, 200 -> 140 us at K = 128, profiles/r06_notes.md 5).  Rows that are not computed read the zero buffer.
  // A separate instantiation: the 64 registers of the residual pieces put the kernel at 254 VGPRs, and the long-K layers WITHOUT a residual
  // lost 20 % to the schedule the compiler found under that pressure (fpn_output3 761 -> 905 us) when both shared one kernel.
  constexpr int CLD = BN + 8;
  constexpr int LPR = BN / 4, RPP = 256 / LPR, NP = BM / RPP;      // lanes per output row, rows per pass, passes
  const int ecol = n0 + (tid % LPR) * 4, err = tid / LPR;
  const bool evec = ecol + 3 < a.N && !(a.ldo & 3) && (a.res_mode == 0 || !(a.ldr & 3));
  const bool pre_res = PRE && evec;                                 // PRE launches have res_mode != 0 and splits == 1 (the launcher's rule)
  f32x4 rres[PRE ? NP : 1];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  auto epi_prefetch = [&]() __attribute__((always_inline)) {
    if (a.bias && ecol < a.N) {
      if (ecol + 3 < a.N && !((uintptr_t)a.bias & 15)) bias4 = *reinterpret_cast<const f32x4*>(a.bias + ecol);
      else
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[e] = ecol + e < a.N ? a.bias[ecol + e] : 0.f;
    }
    if (pre_res) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int mi = m0 + p * RPP + err;
        const float* rp = a.zeros;
        if (row_ok(mi)) {
          if (a.res_mode == 1) {
            rp = a.res + (long long)mi * a.ldr + ecol;
          } else {                                       // nearest x2 up-sampling of a [B, OH / 2, OW / 2] tensor
            const int b = (int)((unsigned)mi / (unsigned)ohw), rem = mi - b * ohw;
            const int oy = (int)((unsigned)rem / (unsigned)a.OW), ox = rem - oy * a.OW;
            rp = a.res + (((long long)b * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1)) * a.ldr + ecol;
          }
        }
        rres[PRE ? p : 0] = *reinterpret_cast<const f32x4*>(rp);
      }
    }
  };
  {
    // two register sets: chunk kc + 2 is requested from memory at the top of chunk kc (two chunks of MFMA time to arrive) and chunk kc + 1,
    // requested one chunk earlier, goes into the free LDS stage after the first quarter of this chunk's MFMAs -- the barrier at the end of
    // the chunk then only synchronises; nothing waits on memory or on LDS writes there
    f32x4 sa[NA], sb[NB];
    gload(ra, rb, kc0);
    lstore(ra, rb, 0);
    if (nk > 1) gload(ra, rb, kc0 + 1);                  // (requesting chunk 1 before the LDS store of chunk 0 measured neutral: not kept)
    __syncthreads();
    auto chunk = [&](int kc, f32x4 (&xa)[NA], f32x4 (&xb)[NB], f32x4 (&ya)[NA], f32x4 (&yb)[NB]) __attribute__((always_inline)) {   // x: holds chunk kc + 1; y: free
      const int buf = kc & 1;
      if (kc + 2 < nk) gload(ya, yb, kc0 + kc + 2);
      mma(buf, 0);
      if (kc + 1 < nk) lstore(xa, xb, buf ^ 1);
#pragma unroll
      for (int g = 1; g < kBK / 8; ++g) mma(buf, g);
      __syncthreads();
    };
    int kc = 0;
    if constexpr (PRE) {
      for (; kc + 2 < nk; kc += 2) {
        chunk(kc, ra, rb, sa, sb);
        chunk(kc + 1, sa, sb, ra, rb);
      }
      // one or two chunks left; the last one is peeled out of the loop so that the epilogue operands requested in front of it are not
      // loop-carried registers
      if (kc + 2 == nk) {
        chunk(kc, ra, rb, sa, sb);
        epi_prefetch();
        chunk(kc + 1, sa, sb, ra, rb);
      } else {
        epi_prefetch();
        chunk(kc, ra, rb, sa, sb);
      }
    } else {
      for (; kc + 1 < nk; kc += 2) {
        chunk(kc, ra, rb, sa, sb);
        chunk(kc + 1, sa, sb, ra, rb);
      }
      if (kc < nk) chunk(kc, ra, rb, sa, sb);
    }
  }

  // ---- epilogue through LDS: D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] -> Cs[BM][BN + 8] (the row stride puts the two
  // lane halves, 4 rows apart, 32 banks apart: conflict-free 4-byte writes), then whole rows leave as float4
  float* Cs = lds;                                       // the loop's last __syncthreads() released both operand stages
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[((wm * TM + i) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * CLD + (wn * TN + j) * 32 + li] = acc[i][j][r];
  __syncthreads();
  const int col = ecol, rr = err;
  if (col >= a.N) return;
  if constexpr (!PRE) {
    if (a.splits <= 1) epi_prefetch();                   // the bias only
  }
  if (a.splits > 1) {                                    // raw partial sums; bias / residual / ReLU belong to the reduce pass
    float* slab = a.ws + (long long)blockIdx.z * ((long long)a.gx * BM) * a.ws_ld;
#pragma unroll 4
    for (int p = 0; p < NP; ++p) {
      const int row = p * RPP + rr;
      if (!row_ok(m0 + row)) continue;
      *reinterpret_cast<float4*>(slab + (long long)(m0 + row) * a.ws_ld + col) = *reinterpret_cast<const float4*>(Cs + row * CLD + (tid % LPR) * 4);
    }
    return;
  }
  if (evec) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = p * RPP + rr;
      const int mi = m0 + row;
      if (!row_ok(mi)) continue;
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(Cs + row * CLD + (tid % LPR) * 4);
      f32x4 v = c4 + bias4;                              // (acc + bias) + residual: the order of the reduce pass and of the scalar tail below
      if (pre_res) v += rres[PRE ? p : 0];
      else if (a.res_mode == 1) v += *reinterpret_cast<const f32x4*>(a.res + (long long)mi * a.ldr + col);
      else if (a.res_mode == 2) {
        const int b = (int)((unsigned)mi / (unsigned)ohw), rem = mi - b * ohw;
        const int oy = (int)((unsigned)rem / (unsigned)a.OW), ox = rem - oy * a.OW;
        v += *reinterpret_cast<const f32x4*>(a.res + (((long long)b * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1)) * a.ldr + col);
      }
      if (a.relu)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(a.out + (long long)mi * a.ldo + col) = v;
    }
    return;
  }
  // ragged tail (N or a leading dimension not a multiple of 4): scalar stores, the residual read here
#pragma unroll 2
  for (int p = 0; p < NP; ++p) {
    const int row = p * RPP + rr;
    const int mi = m0 + row;
    if (!row_ok(mi)) continue;
    const long long m = mi;
    const float4 c4 = *reinterpret_cast<const float4*>(Cs + row * CLD + (tid % LPR) * 4);
    const float v[4] = {c4.x + bias4[0], c4.y + bias4[1], c4.z + bias4[2], c4.w + bias4[3]};
    const float* rp = nullptr;
    if (a.res_mode == 1) {
      rp = a.res + m * a.ldr + col;
    } else if (a.res_mode == 2) {
      const int b = (int)((unsigned)mi / (unsigned)ohw), rem = mi - b * ohw;
      const int oy = (int)((unsigned)rem / (unsigned)a.OW), ox = rem - oy * a.OW;
      rp = a.res + (((long long)b * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1)) * a.ldr + col;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e >= a.N) break;
      float x = v[e] + (rp ? rp[e] : 0.f);
      if (a.relu) x = x > 0.f ? x : 0.f;
      a.out[m * a.ldo + col + e] = x;
    }
  }
}

// out = act(sum_z slab_z + bias (+ res)) over the valid rows; one float4 of columns per thread
__global__ __launch_bounds__(256) void seg_splitk_reduce_kernel(const GemmArgs a, long long slab_elems) {
  const int n4 = (a.N + 3) / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.M * n4) return;
  const long long m = t / n4;
  const int col = (int)(t - m * n4) * 4;
  if (a.m_dev) {
    const long long u = m / a.unit_rows;
    if (m - u * a.unit_rows >= (long long)a.m_dev[u] * a.rows_per_item) return;
  }
  const float* src = a.ws + m * a.ws_ld + col;
  float4 acc = *reinterpret_cast<const float4*>(src);
  for (int z = 1; z < a.splits; ++z) {
    const float4 p = *reinterpret_cast<const float4*>(src + z * slab_elems);
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  float v[4] = {acc.x, acc.y, acc.z, acc.w};
  const float* rp = nullptr;
  if (a.res_mode == 1) {
    rp = a.res + m * a.ldr + col;
  } else if (a.res_mode == 2) {
    const int ohw = a.OH * a.OW;
    const int b = (int)(m / ohw), rem = (int)(m - (long long)b * ohw);
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    rp = a.res + (((long long)b * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1)) * a.ldr + col;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (col + e >= a.N) break;
    float x = v[e] + (a.bias ? a.bias[col + e] : 0.f) + (rp ? rp[e] : 0.f);
    if (a.relu) x = x > 0.f ? x : 0.f;
    a.out[m * a.ldo + col + e] = x;
  }
}


template <int WM, int WN, int TM, int TN, bool UNI>
static int launch_gemm(GemmArgs& a, hipStream_t st, int force_split, size_t ws_bytes) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds_ops = (size_t)2 * (BM + BN) * kLd * sizeof(float), lds_epi = (size_t)BM * (BN + 8) * sizeof(float);
  constexpr size_t lds = lds_ops > lds_epi ? lds_ops : lds_epi;      // the epilogue stages the output tile over the operand stages (128 x 128: 68 KB)
  static coma::LdsOptIn opt;
  static coma::LdsOptIn opt_pre;
  if (lds > 65536) {
    if (int rc = coma::opt_in_lds(opt, (const void*)conv_gemm_f32_kernel<WM, WN, TM, TN, UNI, false>, lds, "seg_conv_gemm_f32")) return rc;
    if (int rc = coma::opt_in_lds(opt_pre, (const void*)conv_gemm_f32_kernel<WM, WN, TM, TN, UNI, true>, lds, "seg_conv_gemm_f32")) return rc;
  }
  const long long gx = (a.M + BM - 1) / BM;
  const int gy = (a.N + BN - 1) / BN;
  if (a.M >= 0x7fffffffLL - BM || gx * gy >= 0x7fffffffLL) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: M=%lld rows / %d column tiles exceed the kernel's 32-bit row arithmetic", a.M, gy);
  // split-K: only where the tiles leave more than half of the chip idle and every slice keeps >= 4 chunks (128 k values)
  const int nk = a.Kpad / kBK;
  const long long tiles = gx * gy;
  int S = 1;
  constexpr int slots = 512;
  if (force_split > 0) S = force_split;
  else if (force_split == 0 && a.ws && tiles < 2 * slots && nk >= 8) {
    // per-CU cost model, unit = one K chunk of a workgroup that shares its CU's matrix pipe with another one (~3.5 us on a 128 x 128 tile):
    // a launch of W = tiles * S workgroups of nk / S chunks (+ 3 for prologue and epilogue) gives every CU n = ceil(W / 256) of them, and from two
    // workgroups on the pipe is the bottleneck -- the CU needs n x (nk / S + 3) / 2 units however many of them are resident at a time; a CU with
    // ONE workgroup cannot fill the pipe: 0.6 per chunk.  A split launch also pays the reduce pass: a second launch (~3 us behind the GEMM
    // inside a graph) that reads S slabs at ~4 TB/s.  Measured against forced factors on the plan's shapes (profiles/r06_notes.md 5):
    // res5.x.conv2 picks 3 (220 us; 2: 315, 4: 250), the M = 20 000 3 x 3 layers on 128 x 64 tiles pick 2 (230 us; unsplit 244-252), the point
    // head (392 tiles x 11 chunks) stays unsplit (56 us; split in two 78).  Every slice >= 4 chunks.
    auto cu_cost = [](double w, double chunks) {
      const double n = (double)(long long)((w + 255.0) / 256.0);
      return n * (chunks + 3.0) * (n <= 1.0 ? 0.6 : 0.5);
    };
    double best = cu_cost((double)tiles, (double)nk);
    const double unit_us = 3.5 * ((double)BM * BN / (128.0 * 128.0));
    const double slab_units = 4.0 * (double)a.M * a.N / 4.0e6 / unit_us;      // one slab read back (the GEMM's slab writes overlap with its own work)
    for (int c = 2; c <= 16 && nk / c >= 4; ++c) {
      const double cost = cu_cost((double)tiles * c, (double)nk / c) + 3.0 / unit_us + c * slab_units;
      if (cost < best * 0.95) { best = cost; S = c; }
    }
  }
  a.ws_ld = gy * BN;
  const long long slab = gx * BM * (long long)a.ws_ld;
  if (S > 1 && a.ws) S = (int)std::min<long long>(S, (long long)(ws_bytes / sizeof(float)) / slab);
  if (S > nk) S = nk;
  if (S <= 1 || !a.ws) {
    a.splits = 1; a.nk_per = nk;
    if (force_split > 1) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: split_k=%d needs a workspace of %lld bytes", force_split, slab * 4 * force_split);
  } else {
    a.nk_per = (nk + S - 1) / S;
    a.splits = (nk + a.nk_per - 1) / a.nk_per;            // every slice non-empty
  }
  a.gx = (int)gx; a.gy = gy;
  static const int band_env = getenv("SEG_XCD_BAND") ? atoi(getenv("SEG_XCD_BAND")) : -1;      // A/B aid: 0 = plain order, 1 = banded everywhere
  a.band = band_env >= 0 ? band_env : 1;
  static const int pre_env = getenv("SEG_EPI_PREFETCH") ? atoi(getenv("SEG_EPI_PREFETCH")) : 1;  // A/B aid: 0 = residual read inside the epilogue
  // (only for K <= 128: from K = 256 on the K loop hides the epilogue of the co-resident workgroup anyway, and the 254-register kernel
  // is the slower one -- res5.x.conv3, K = 512: 124 us without, 146 us with)
  if (a.res_mode != 0 && a.splits == 1 && nk <= 4 && pre_env)
    hipLaunchKernelGGL((conv_gemm_f32_kernel<WM, WN, TM, TN, UNI, true>), dim3((unsigned)(gx * gy), 1, 1), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((conv_gemm_f32_kernel<WM, WN, TM, TN, UNI, false>), dim3((unsigned)(gx * gy), 1, (unsigned)a.splits), dim3(256), lds, st, a);
  if (int rc = check_launch("seg::conv_gemm_f32_kernel")) return rc;
  if (a.splits > 1) {
    const long long n = a.M * ((a.N + 3) / 4);
    hipLaunchKernelGGL(seg_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, slab);
    return check_launch("seg::seg_splitk_reduce_kernel");
  }
  return COMA_OK;
}

}  // namespace seg

extern "C" int seg_conv_gemm_f32(const seg_conv_desc* d, void* stream) {
  using namespace seg;
  if (!d) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: null descriptor");
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_SEG;
    r.i[0] = SEG_OP_CONV;
    r.p[0] = (void*)d->x; r.p[1] = (void*)d->w; r.p[2] = (void*)d->bias; r.p[3] = (void*)d->res; r.p[4] = d->out; r.p[5] = (void*)d->m_dev;
    r.i[1] = d->batch; r.i[2] = d->in_h; r.i[3] = d->in_w; r.i[4] = d->c; r.i[5] = d->ldx; r.i[6] = d->n; r.i[7] = d->kpad; r.i[8] = d->kh;
    r.i[9] = d->kw; r.i[10] = d->stride; r.i[11] = d->pad; r.i[12] = d->out_h; r.i[13] = d->out_w; r.i[14] = d->ldr; r.i[15] = d->res_mode;
    r.i[16] = d->ldo; r.i[17] = d->relu; r.i[18] = d->rows_per_item; r.i[19] = d->tile; r.i[20] = d->unit_rows;
    r.p[6] = d->workspace; r.i[21] = d->split_k; r.i[22] = (int64_t)d->workspace_bytes;
    return sd::plan_record(r);
  }
  if (!d->x || !d->w || !d->out) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: null pointer");
  if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->n <= 0)
    return fail(COMA_E_INVALID, "seg_conv_gemm_f32: batch=%d in=%dx%d out=%dx%d n=%d", d->batch, d->in_h, d->in_w, d->out_h, d->out_w, d->n);
  if (d->c <= 0 || d->c % 4 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0)
    return fail(COMA_E_INVALID, "seg_conv_gemm_f32: c=%d (a multiple of 4) kh=%d kw=%d stride=%d pad=%d", d->c, d->kh, d->kw, d->stride, d->pad);
  const int ldx = d->ldx ? d->ldx : d->c;
  if (ldx < d->c || ldx % 4) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: ldx=%d", ldx);
  if (d->kpad % 32 || d->kpad < d->kh * d->kw * d->c)
    return fail(COMA_E_INVALID, "seg_conv_gemm_f32: kpad=%d must be a multiple of 32 covering kh*kw*c=%d", d->kpad, d->kh * d->kw * d->c);
  if (d->res_mode < 0 || d->res_mode > 2) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: res_mode=%d", d->res_mode);
  if (d->res_mode && !d->res) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: res_mode=%d without a residual", d->res_mode);
  if (d->res_mode == 2 && ((d->out_h | d->out_w) & 1)) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: up-sampled residual needs even out_h, out_w");
  if (d->m_dev && (d->rows_per_item <= 0 || d->unit_rows <= 0))
    return fail(COMA_E_INVALID, "seg_conv_gemm_f32: rows_per_item=%d unit_rows=%d", d->rows_per_item, d->unit_rows);
  GemmArgs a;
  a.x = (const float*)d->x; a.w = (const float*)d->w; a.bias = (const float*)d->bias; a.res = (const float*)d->res; a.out = (float*)d->out;
  a.m_dev = (const int*)d->m_dev;
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.C = d->c; a.ldx = ldx; a.N = d->n; a.Kpad = d->kpad; a.kh = d->kh; a.kw = d->kw;
  a.stride = d->stride; a.pad = d->pad; a.OH = d->out_h; a.OW = d->out_w; a.ldr = d->ldr ? d->ldr : d->n; a.res_mode = d->res_mode;
  a.ldo = d->ldo ? d->ldo : d->n; a.relu = d->relu; a.rows_per_item = d->rows_per_item; a.unit_rows = d->unit_rows;
  a.M = (long long)d->batch * d->out_h * d->out_w;
  if (a.ldo < d->n) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: ldo=%d < n=%d", a.ldo, d->n);
  if (d->split_k < -1 || (d->workspace && d->workspace_bytes < 16)) return fail(COMA_E_INVALID, "seg_conv_gemm_f32: split_k=%d workspace_bytes=%zu", d->split_k, d->workspace_bytes);
  a.ws = (float*)d->workspace; a.splits = 1; a.nk_per = 0; a.ws_ld = 0;
  static const float* zeros = [] { void* p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(kZerosDev)) == hipSuccess ? (const float*)p : nullptr; }();
  if (!zeros) return fail(COMA_E_DEVICE, "seg_conv_gemm_f32: hipGetSymbolAddress(kZerosDev) failed");
  a.zeros = zeros;
  const int fs = d->split_k;                     // 0 = by the rule above, -1 = never, S > 1 = exactly S slices (tests)
  const size_t wb = d->workspace_bytes;
  hipStream_t st = (hipStream_t)stream;
  // tile: 0 = by width (n <= 32: 128 x 32, n <= 64: 128 x 64, else 128 x 128); 1 / 2 / 3 force 128 x 128 / 128 x 64 / 128 x 32 (tests)
  int tile = d->tile;
  if (tile == 0) {
    tile = d->n <= 32 ? 3 : (d->n <= 64 ? 2 : 1);
    if (tile == 1) {
      // mid-size launches: 128 x 64 tiles when the 128 x 128 grid leaves the last CU round mostly empty.  Per-CU model: a CU works through
      // ceil(tiles / 256) tiles; a 128 x 64 tile costs 0.45 (short K: prologue / epilogue bound, three of them per CU) ... 0.58 (K >= 2048: it
      // re-reads its A rows from LDS for half the columns) of a 128 x 128 one.  Measured on the plan's shapes (profiles/r06_notes.md 5):
      // res3.x.conv2 241 -> 204 us, res3.x.conv1 116 -> 97, res4.x.conv1 141 -> 111, res5.x.conv3 124 -> 102, res2.x.conv3 201 -> 188,
      // res4.x.conv3 119 -> 109; the long-K large layers (fpn_output2 / 3, rpn_conv, box_fc1) stay on 128 x 128.
      const long long t1 = ((a.M + 127) / 128) * ((d->n + 127) / 128), t2 = ((a.M + 127) / 128) * ((d->n + 63) / 64);
      const double nk = d->kpad / 32.0;
      const double r = 0.45 + 0.13 * (nk < 64.0 ? nk / 64.0 : 1.0);      // (three narrow-tile workgroups per CU since the unpadded LDS rows)
      if ((double)((t2 + 255) / 256) * r < 0.95 * (double)((t1 + 255) / 256)) tile = 2;
    }
  }
  // uniform-tap addressing wherever a 32-wide K chunk lies inside one tap (every layer but the stem, C = 4, and the point head, C = 336)
  if (d->c % kBK == 0) {
    if (tile == 1) return launch_gemm<2, 2, 2, 2, true>(a, st, fs, wb);
    if (tile == 2) return launch_gemm<4, 1, 1, 2, true>(a, st, fs, wb);
    if (tile == 3) return launch_gemm<4, 1, 1, 1, true>(a, st, fs, wb);
  } else {
    if (tile == 1) return launch_gemm<2, 2, 2, 2, false>(a, st, fs, wb);
    if (tile == 2) return launch_gemm<4, 1, 1, 2, false>(a, st, fs, wb);
    if (tile == 3) return launch_gemm<4, 1, 1, 1, false>(a, st, fs, wb);
  }
  return fail(COMA_E_INVALID, "seg_conv_gemm_f32: tile=%d", d->tile);
}
