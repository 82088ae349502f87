// K4: reducers over the learned histograms (gfx950).  All of them are single-pass, HBM-bound
// row kernels: one wave per (h,o) row of N bins, lanes strided over bins (coalesced 256-B segments),
// wave-level butterfly reductions, no LDS.
//
// replaces: utils/coma.py:328-330 (normalize), :342-356 (contact map), :376-377 (significant pairs),
//           :402-427 (masked max), :455-475 (entropy score).
#include <cstdint>

#include "common.h"

namespace coma {

constexpr int kRowWaves = 4;   // rows (waves) per block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// prob[m,:] /= (sum + eps);  contact[m] = (sum_k prob*(1 - p.n_k)/2) * nom/den
__global__ __launch_bounds__(kRowWaves* kWave) void contact_map_kernel(
    float* __restrict__ prob, const float* __restrict__ grid, float px, float py, float pz,
    const float* __restrict__ nom, const float* __restrict__ den, int64_t M, int N, float eps,
    float* __restrict__ contact) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t m = (int64_t)blockIdx.x * kRowWaves + threadIdx.x / kWave;
  if (m >= M) return;
  float* row = prob + m * N;
  float s = 0.0f;
  for (int k = lane; k < N; k += kWave) s += row[k];
  s = wave_sum(s) + eps;
  float c = 0.0f;
  for (int k = lane; k < N; k += kWave) {
    float v = row[k] / s;
    row[k] = v;
    float dot = (px * grid[3 * k] + py * grid[3 * k + 1]) + pz * grid[3 * k + 2];
    c += v * ((1.0f - dot) / 2.0f);
  }
  if (contact) {
    c = wave_sum(c);
    if (lane == 0) contact[m] = c * (nom[m] / den[m]);
  }
}

// prob normalised in place, then 1 + sum_k plogp(round(p*n_bin)/n_bin) / ln(n_bin)
__global__ __launch_bounds__(kRowWaves* kWave) void entropy_kernel(float* __restrict__ prob, int64_t M,
                                                                    int N, float eps, float n_bin,
                                                                    float log_n_bin,
                                                                    float* __restrict__ score) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t m = (int64_t)blockIdx.x * kRowWaves + threadIdx.x / kWave;
  if (m >= M) return;
  float* row = prob + m * N;
  float s = 0.0f;
  for (int k = lane; k < N; k += kWave) s += row[k];
  s = wave_sum(s) + eps;
  float e = 0.0f;
  for (int k = lane; k < N; k += kWave) {
    float v = row[k] / s;
    row[k] = v;
    float q = rintf(v * n_bin) / n_bin;   // torch.round = half-to-even
    e += (q == 0.0f) ? 0.0f : q * logf(q);
  }
  e = wave_sum(e);
  if (lane == 0) score[m] = e / log_n_bin + 1.0f;
}

__global__ void pairs_kernel(const float* __restrict__ cnt, float thr, int64_t M, uint8_t* __restrict__ pairs) {
  int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) pairs[m] = cnt[m] >= thr ? 1 : 0;
}

// blockIdx.y == 0: col_any[o] = any_h pairs[h,o]   (threads over o, coalesced)
// blockIdx.y == 1: row_any[h] = any_o pairs[h,o]   (one wave per row)
__global__ void any_kernel(const uint8_t* __restrict__ pairs, int H, int O, uint8_t* __restrict__ col_any,
                           uint8_t* __restrict__ row_any) {
  if (blockIdx.y == 0) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    uint8_t a = 0;
    for (int h = 0; h < H; ++h) a |= pairs[(int64_t)h * O + o];
    col_any[o] = a ? 1 : 0;
  } else {
    int h = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (h >= H) return;
    int lane = threadIdx.x & (kWave - 1);
    int a = 0;
    for (int o = lane; o < O; o += kWave) a |= pairs[(int64_t)h * O + o];
    a = __any(a);
    if (lane == 0) row_any[h] = a ? 1 : 0;
  }
}

__device__ __forceinline__ float nanmax(float m, float v) { return (v > m || v != v) ? v : m; }

// which == 0: out[h] = max_{o: col_any[o]} C[h,o]   (one wave per h)
// which == 1: out[o] = max_{h: row_any[h]} C[h,o]   (one thread per o)
__global__ void masked_max_kernel(const float* __restrict__ C, const uint8_t* __restrict__ col_any,
                                  const uint8_t* __restrict__ row_any, int H, int O, int which,
                                  float* __restrict__ out) {
  const float ninf = -__builtin_inff();
  if (which == 0) {
    int h = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (h >= H) return;
    int lane = threadIdx.x & (kWave - 1);
    float m = ninf;
    int any = 0;
    for (int o = lane; o < O; o += kWave)
      if (col_any[o]) {
        m = nanmax(m, C[(int64_t)h * O + o]);
        any = 1;
      }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = nanmax(m, __shfl_xor(m, d));
    any = __any(any);
    if (lane == 0) out[h] = any ? m : 0.0f;
  } else {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float m = ninf;
    int any = 0;
    for (int h = 0; h < H; ++h)
      if (row_any[h]) {
        m = nanmax(m, C[(int64_t)h * O + o]);
        any = 1;
      }
    out[o] = any ? m : 0.0f;
  }
}

// ---- consumer of the accumulator state (src/application/optimize.py:190-196)
// NumPy ordering: argmax returns the FIRST maximum and treats NaN as the maximum (first NaN wins); max propagates NaN.
__device__ __forceinline__ bool np_better(float v, int64_t i, float bv, int64_t bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;                 // a NaN beats any number
  if (vn) return i < bi;                   // both NaN: first one
  return v > bv || (v == bv && i < bi);
}

// idx[m] = argmax_k x[m*row_stride + col_offset + k], k < n; optional val[m] = np.max of the same row
__global__ __launch_bounds__(kRowWaves* kWave) void row_argmax_kernel(const float* __restrict__ x, int64_t M, int n,
                                                                       int64_t row_stride, int64_t col_offset,
                                                                       int64_t* __restrict__ idx, float* __restrict__ val) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t m = (int64_t)blockIdx.x * kRowWaves + threadIdx.x / kWave;
  if (m >= M) return;
  const float* row = x + m * row_stride + col_offset;
  float bv = -__builtin_inff();
  int64_t bi = INT64_MAX;
  for (int k = lane; k < n; k += kWave) {
    const float v = row[k];
    if (np_better(v, k, bv, bi)) { bv = v; bi = k; }
  }
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
    const float ov = __shfl_xor(bv, msk);
    const int64_t oi = __shfl_xor(bi, msk);
    if (np_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) {
    if (idx) idx[m] = bi;
    if (val) val[m] = bv;
  }
}

// sel[h] = (max_o nom[h,o] / den[h,o]) > thr   (NaN anywhere in the row -> max is NaN -> false, as np.max / `>` give)
__global__ __launch_bounds__(kRowWaves* kWave) void contact_select_kernel(const float* __restrict__ nom, const float* __restrict__ den,
                                                                           int H, int O, float thr, uint8_t* __restrict__ sel) {
  const int lane = threadIdx.x & (kWave - 1);
  const int h = blockIdx.x * kRowWaves + threadIdx.x / kWave;
  if (h >= H) return;
  float mx = -__builtin_inff();
  bool nan = false;
  for (int o = lane; o < O; o += kWave) {
    const float v = nom[(int64_t)h * O + o] / den[(int64_t)h * O + o];
    nan |= v != v;
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk));
  nan = __any(nan);
  if (lane == 0) sel[h] = (!nan && mx > thr) ? 1 : 0;
}

}  // namespace coma

using namespace coma;

extern "C" int coma_contact_map_f32(float* prob, const float* sphere_grid, const float* principle_vec,
                                    const float* nom, const float* den, int64_t M, int N, float eps,
                                    float* contact, void* stream) {
  if (!prob || !sphere_grid || !principle_vec) return fail(COMA_E_INVALID, "coma_contact_map_f32: null pointer");
  if (contact && (!nom || !den)) return fail(COMA_E_INVALID, "coma_contact_map_f32: nom/den required");
  if (M <= 0 || N <= 0) return fail(COMA_E_INVALID, "coma_contact_map_f32: bad sizes");
  int64_t blocks = (M + kRowWaves - 1) / kRowWaves;
  if (blocks > 0x7fffffffLL) return fail(COMA_E_INVALID, "coma_contact_map_f32: M too large");
  hipLaunchKernelGGL(contact_map_kernel, dim3((unsigned)blocks), dim3(kRowWaves * kWave), 0,
                     (hipStream_t)stream, prob, sphere_grid, principle_vec[0], principle_vec[1],
                     principle_vec[2], nom, den, M, N, eps, contact);
  return check_launch("contact_map_kernel");
}

extern "C" int coma_entropy_f32(float* prob, int64_t M, int N, float eps, float n_bin, float* score,
                                void* stream) {
  if (!prob || !score) return fail(COMA_E_INVALID, "coma_entropy_f32: null pointer");
  if (M <= 0 || N <= 0 || !(n_bin > 1.0f)) return fail(COMA_E_INVALID, "coma_entropy_f32: bad sizes");
  int64_t blocks = (M + kRowWaves - 1) / kRowWaves;
  if (blocks > 0x7fffffffLL) return fail(COMA_E_INVALID, "coma_entropy_f32: M too large");
  // math.log(n_bin) is a Python double, cast to f32 by the in-place division (utils/coma.py:462)
  float log_n_bin = (float)log((double)n_bin);
  hipLaunchKernelGGL(entropy_kernel, dim3((unsigned)blocks), dim3(kRowWaves * kWave), 0,
                     (hipStream_t)stream, prob, M, N, eps, n_bin, log_n_bin, score);
  return check_launch("entropy_kernel");
}

extern "C" int coma_significant_pairs_u8(const float* cnt, float threshold, int H, int O, uint8_t* pairs,
                                         uint8_t* col_any, uint8_t* row_any, void* stream) {
  if (!cnt || !pairs || !col_any || !row_any) return fail(COMA_E_INVALID, "coma_significant_pairs_u8: null pointer");
  if (H <= 0 || O <= 0) return fail(COMA_E_INVALID, "coma_significant_pairs_u8: bad sizes");
  int64_t M = (int64_t)H * O;
  hipLaunchKernelGGL(pairs_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     cnt, threshold, M, pairs);
  int bx = (int)((O + 255) / 256);
  int by = (H + 3) / 4;
  hipLaunchKernelGGL(any_kernel, dim3((unsigned)(bx > by ? bx : by), 2), dim3(256), 0, (hipStream_t)stream,
                     pairs, H, O, col_any, row_any);
  return check_launch("pairs/any kernels");
}

extern "C" int coma_masked_max_f32(const float* contact, const uint8_t* col_any, const uint8_t* row_any,
                                   int H, int O, int which, float* out, void* stream) {
  if (!contact || !col_any || !row_any || !out) return fail(COMA_E_INVALID, "coma_masked_max_f32: null pointer");
  if (H <= 0 || O <= 0 || (which != 0 && which != 1)) return fail(COMA_E_INVALID, "coma_masked_max_f32: bad args");
  unsigned blocks = which == 0 ? (unsigned)((H + 3) / 4) : (unsigned)((O + 255) / 256);
  hipLaunchKernelGGL(masked_max_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, contact, col_any,
                     row_any, H, O, which, out);
  return check_launch("masked_max_kernel");
}

extern "C" int coma_row_argmax_i64(const float* x, int64_t rows, int n, int64_t row_stride, int64_t col_offset, int64_t* idx,
                                   float* val, void* stream) {
  if (!x || (!idx && !val)) return fail(COMA_E_INVALID, "coma_row_argmax_i64: null pointer");
  if (rows <= 0 || n <= 0 || row_stride < n || col_offset < 0) return fail(COMA_E_INVALID, "coma_row_argmax_i64: bad sizes");
  const int64_t blocks = (rows + kRowWaves - 1) / kRowWaves;
  hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)blocks), dim3(kRowWaves * kWave), 0, (hipStream_t)stream, x, rows, n,
                     row_stride, col_offset, idx, val);
  return check_launch("row_argmax_kernel");
}

extern "C" int coma_contact_select_u8(const float* nom, const float* den, int H, int O, float threshold, uint8_t* selected,
                                      void* stream) {
  if (!nom || !den || !selected) return fail(COMA_E_INVALID, "coma_contact_select_u8: null pointer");
  if (H <= 0 || O <= 0) return fail(COMA_E_INVALID, "coma_contact_select_u8: bad sizes");
  hipLaunchKernelGGL(contact_select_kernel, dim3((unsigned)((H + kRowWaves - 1) / kRowWaves)), dim3(kRowWaves * kWave), 0,
                     (hipStream_t)stream, nom, den, H, O, threshold, selected);
  return check_launch("contact_select_kernel");
}
