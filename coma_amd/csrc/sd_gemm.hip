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
// Tiling (wave64, MFMA v_mfma_f32_32x32x16_f16): block = 4 waves (2x2) computing 128 x BN (BN = 128 or 64),
// each wave 64 x BN/2 as 2 x (BN/64) MFMA tiles; BK = 32 (two MFMA k-steps); register-staged global->LDS
// double buffer, one barrier per K tile; LDS rows padded to 80 B so the ds_read_b128 fragment reads of a
// 16-lane group hit 16 distinct 16-byte slots.  Epilogue fuses bias, per-(batch) bias (time embedding),
// residual add, SiLU, and GEGLU (value/gate columns interleaved per wave at weight-prep time).
#include <hip/hip_fp16.h>

#include "common.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 8;   // halves; 80-byte rows

struct GemmArgs {
  const _Float16* a0;
  const _Float16* a1;
  int c0, c1;                 // channels of the two A sources (c1 = 0 -> single source)
  int in_h, in_w, out_h, out_w;
  int taps, stride, upsample, pad;
  int M, N, K;
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
  long long sa, sw, so, sr;   // per-blockIdx.z strides in elements
};

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmArgs g) {
  constexpr int WN = BN / 64;          // MFMA n-tiles per wave
  constexpr int B_ITEMS = BN * 4 / 256;  // 16-byte chunks of the W tile per thread
  __shared__ __attribute__((aligned(16))) _Float16 As[2][BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const long long z = blockIdx.z;
  const _Float16* a0 = g.a0 + z * g.sa;
  const _Float16* a1 = g.a1;
  const _Float16* wp = g.w + z * g.sw;
  const int ctot = g.c0 + g.c1;
  const int pad = g.pad;
  const int lim_h = g.upsample ? 2 * g.in_h : g.in_h;
  const int lim_w = g.upsample ? 2 * g.in_w : g.in_w;

  // A-tile staging role: two (row, 16-byte chunk) items per thread
  int a_row[2], a_chunk[2], a_n[2], a_y[2], a_x[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int item = tid + i * 256;
    a_row[i] = item >> 2;
    a_chunk[i] = item & 3;
    int m = m0 + a_row[i];
    a_ok[i] = m < g.M;
    int mm = a_ok[i] ? m : 0;
    a_n[i] = mm / g.rows_per_batch;
    int rem = mm - a_n[i] * g.rows_per_batch;
    a_y[i] = rem / g.out_w;
    a_x[i] = rem - a_y[i] * g.out_w;
  }
  int b_row[B_ITEMS], b_chunk[B_ITEMS];
#pragma unroll
  for (int i = 0; i < B_ITEMS; ++i) {
    int item = tid + i * 256;
    b_row[i] = item >> 2;
    b_chunk[i] = item & 3;
  }

  uint4 ra[2], rb[B_ITEMS];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / ctot;
    int ci = k0 - tap * ctot;
    const _Float16* src = a0;
    int csrc = g.c0;
    if (ci >= g.c0) { src = a1; csrc = g.c1; ci -= g.c0; }
    const int ky = g.taps == 9 ? tap / 3 : 0;
    const int kx = g.taps == 9 ? tap - ky * 3 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int iy = a_y[i] * g.stride + ky - pad;
      int ix = a_x[i] * g.stride + kx - pad;
      bool ok = a_ok[i] && iy >= 0 && iy < lim_h && ix >= 0 && ix < lim_w;
      if (g.upsample) { iy >>= 1; ix >>= 1; }
      if (ok) {
        const _Float16* p = src + (((long long)a_n[i] * g.in_h + iy) * g.in_w + ix) * csrc + ci + a_chunk[i] * 8;
        ra[i] = *reinterpret_cast<const uint4*>(p);
      } else {
        ra[i] = make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_ITEMS; ++i) {
      int n = n0 + b_row[i];
      if (n < g.N) {
        rb[i] = *reinterpret_cast<const uint4*>(wp + (long long)n * g.K + k0 + b_chunk[i] * 8);
      } else {
        rb[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<uint4*>(&As[buf][a_row[i] * LDS_STRIDE + a_chunk[i] * 8]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ITEMS; ++i)
      *reinterpret_cast<uint4*>(&Bs[buf][b_row[i] * LDS_STRIDE + b_chunk[i] * 8]) = rb[i];
  };

  float16v acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = g.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 af[2], bf[WN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const half8*>(&As[cur][(wr * 64 + i * 32 + frow) * LDS_STRIDE + ks * 16 + fk]);
#pragma unroll
      for (int j = 0; j < WN; ++j)
        bf[j] = *reinterpret_cast<const half8*>(&Bs[cur][(wc * (BN / 2) + j * 32 + frow) * LDS_STRIDE + ks * 16 + fk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  _Float16* outp = g.out + z * g.so;
  const _Float16* resp = g.res ? g.res + z * g.sr : nullptr;
  const bool geglu = (g.epi & SD_EPI_GEGLU) != 0;
  const bool act_silu = (g.epi & SD_EPI_SILU) != 0;
  const bool bias_rows = (g.epi & SD_EPI_BIAS_ROWS) != 0;
  const int col_in_wave = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= g.M) continue;
      if (geglu) {
        // wave tile = [32 value cols | 32 gate cols]; output column = (n0 + wc*64)/2 + lane
        if constexpr (WN == 2) {
          const int ncol = n0 + wc * 64 + col_in_wave;          // permuted value column
          const int ocol = (n0 >> 1) + wc * 32 + col_in_wave;
          float v = acc[i][0][r], gt = acc[i][1][r];
          if (g.bias) { v += (float)g.bias[ncol]; gt += (float)g.bias[ncol + 32]; }
          outp[(long long)row * g.ldo + ocol] = (_Float16)(v * gelu_erf(gt));
        }
      } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int col = n0 + wc * (BN / 2) + j * 32 + col_in_wave;
          if (col >= g.N) continue;
          float v = acc[i][j][r];
          if (g.bias) v += (float)g.bias[bias_rows ? row : col];
          if (g.bias_bn) v += (float)g.bias_bn[(long long)(row / g.rows_per_batch) * g.ldbb + col];
          if (act_silu) v = silu(v);
          if (resp) v += (float)resp[(long long)row * g.ldr + col];
          outp[(long long)row * g.ldo + col] = (_Float16)v;
        }
      }
    }
  }
}

}  // namespace sd

using namespace sd;

extern "C" int sd_conv_gemm_f16(const sd_conv_gemm_desc* d, void* stream) {
  if (!d) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: null descriptor");
  if (!d->a0 || !d->w || !d->out) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: null pointer");
  if (d->taps != 1 && d->taps != 9) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: taps must be 1 or 9");
  if (d->stride != 1 && d->stride != 2) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: stride must be 1 or 2");
  if (d->c0 <= 0 || d->c0 % BK || d->c1 < 0 || d->c1 % BK || (d->c1 > 0 && !d->a1))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: source channels must be multiples of %d (c0=%d c1=%d)", BK, d->c0, d->c1);
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
  g.ldbb = d->ldbb > 0 ? d->ldbb : d->n;
  g.rows_per_batch = d->out_h * d->out_w;
  long long M = (long long)d->batch * g.rows_per_batch;
  if (M > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv_gemm_f16: M too large");
  g.M = (int)M; g.N = d->n; g.K = d->taps * (d->c0 + d->c1);
  g.w = (const _Float16*)d->w; g.bias = (const _Float16*)d->bias; g.bias_bn = (const _Float16*)d->bias_bn;
  g.res = (const _Float16*)d->res; g.ldr = d->ldr > 0 ? d->ldr : d->n;
  g.out = (_Float16*)d->out; g.epi = d->epi;
  const bool geglu = (d->epi & SD_EPI_GEGLU) != 0;
  g.ldo = d->ldo > 0 ? d->ldo : (geglu ? d->n / 2 : d->n);
  g.sa = d->stride_a; g.sw = d->stride_w; g.so = d->stride_out; g.sr = d->stride_res;
  if (geglu && (d->n % 128 != 0 || d->bias_bn || d->res))
    return fail(COMA_E_INVALID, "sd_conv_gemm_f16: GEGLU needs N %% 128 == 0 and no residual / batch bias");
  const unsigned gx = (unsigned)((g.M + BM - 1) / BM);
  if (d->n % 128 == 0 || d->n > 256) {
    dim3 grid(gx, (unsigned)((d->n + 127) / 128), (unsigned)nz);
    hipLaunchKernelGGL(conv_gemm_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, g);
  } else {
    dim3 grid(gx, (unsigned)((d->n + 63) / 64), (unsigned)nz);
    hipLaunchKernelGGL(conv_gemm_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, g);
  }
  return check_launch("conv_gemm_kernel");
}
