// Two-view DLT triangulation, candidate scoring and the RANSAC reprojection matrix (gfx950), all in f64.
// replaces: src/generation/optimize_depth.py:202-237 (solve_DLT: per joint a 4x3 pseudo-inverse), :291-295 (the two
//           reprojection MSEs of a candidate) and :329-350 (candidates^2 reprojection MSEs of the RANSAC search).
// Every camera enters as a 28-double record built on the host with the reference's own NumPy expressions:
//   rot[9], trans[3]   projection of get_projection_matrix (:164-183)
//   mr[9],  tmr[3]     R @ C and t @ (R @ C) of get_view2joints_render (:185-200)
//   scale, maxres, half_x, half_y
// A (4x3) depends only on the view pair, so the pseudo-inverse is (A^T A)^-1 A^T from a 3x3 cofactor inverse, applied to
// every joint of the pair (full column rank: two distinct views); one workgroup per candidate, one thread per joint.
#include "common.h"

namespace coma {

constexpr int VIEW_DOUBLES = 28;

struct View {
  double rot[9], trans[3], mr[9], tmr[3], scale, maxres, hx, hy;
};

__device__ __forceinline__ View load_view(const double* __restrict__ v) {
  View w;
#pragma unroll
  for (int i = 0; i < 9; ++i) w.rot[i] = v[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) w.trans[i] = v[9 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) w.mr[i] = v[12 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) w.tmr[i] = v[21 + i];
  w.scale = v[24]; w.maxres = v[25]; w.hx = v[26]; w.hy = v[27];
  return w;
}

// get_view2joints_render: X @ (R C) - t (R C), then / scale * max(res) + res/2 (same operation order as the reference)
__device__ __forceinline__ void render(const View& w, double x, double y, double z, double& px, double& py) {
  const double cx = (x * w.mr[0] + y * w.mr[3] + z * w.mr[6]) - w.tmr[0];
  const double cy = (x * w.mr[1] + y * w.mr[4] + z * w.mr[7]) - w.tmr[1];
  px = cx / w.scale * w.maxres + w.hx;
  py = cy / w.scale * w.maxres + w.hy;
}

__device__ __forceinline__ double block_sum_128(double v, double* sh) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = sh[0] + sh[1];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(128) void dlt_score_kernel(const double* __restrict__ views, int ref_view, const double* __restrict__ ref_xy,
                                                        const int* __restrict__ cand_view, const double* __restrict__ cand_xy, int J,
                                                        double* __restrict__ tri, double* __restrict__ ref_mse, double* __restrict__ other_mse) {
  __shared__ double sh[2];
  const int p = blockIdx.x;
  const View r = load_view(views + (long long)ref_view * VIEW_DOUBLES);
  const View o = load_view(views + (long long)cand_view[p] * VIEW_DOUBLES);
  // A = [r.rot row 0; r.rot row 1; o.rot row 0; o.rot row 1];  N = A^T A (symmetric 3x3), inverse by cofactors
  const double* a[4] = {r.rot, r.rot + 3, o.rot, o.rot + 3};
  double n[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) n[i][k] = ((a[0][i] * a[0][k] + a[1][i] * a[1][k]) + a[2][i] * a[2][k]) + a[3][i] * a[3][k];
  const double c00 = n[1][1] * n[2][2] - n[1][2] * n[2][1], c01 = n[1][2] * n[2][0] - n[1][0] * n[2][2], c02 = n[1][0] * n[2][1] - n[1][1] * n[2][0];
  const double det = n[0][0] * c00 + n[0][1] * c01 + n[0][2] * c02;
  const double inv[3][3] = {{c00 / det, (n[0][2] * n[2][1] - n[0][1] * n[2][2]) / det, (n[0][1] * n[1][2] - n[0][2] * n[1][1]) / det},
                            {c01 / det, (n[0][0] * n[2][2] - n[0][2] * n[2][0]) / det, (n[0][2] * n[1][0] - n[0][0] * n[1][2]) / det},
                            {c02 / det, (n[0][1] * n[2][0] - n[0][0] * n[2][1]) / det, (n[0][0] * n[1][1] - n[0][1] * n[1][0]) / det}};
  double er = 0.0, eo = 0.0;
  for (int j = threadIdx.x; j < J; j += 128) {
    const double rx = ref_xy[2 * j], ry = ref_xy[2 * j + 1];
    const double ox = cand_xy[((long long)p * J + j) * 2], oy = cand_xy[((long long)p * J + j) * 2 + 1];
    const double b[4] = {(rx - r.hx) - r.trans[0], (ry - r.hy) - r.trans[1], (ox - o.hx) - o.trans[0], (oy - o.hy) - o.trans[1]};
    double atb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) atb[i] = ((a[0][i] * b[0] + a[1][i] * b[1]) + a[2][i] * b[2]) + a[3][i] * b[3];
    double x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = (inv[i][0] * atb[0] + inv[i][1] * atb[1]) + inv[i][2] * atb[2];
    double* t = tri + ((long long)p * J + j) * 3;
    t[0] = x[0]; t[1] = x[1]; t[2] = x[2];
    double px, py;
    render(r, x[0], x[1], x[2], px, py);
    er += (px - rx) * (px - rx) + (py - ry) * (py - ry);
    render(o, x[0], x[1], x[2], px, py);
    eo += (px - ox) * (px - ox) + (py - oy) * (py - oy);
  }
  er = block_sum_128(er, sh);
  eo = block_sum_128(eo, sh);
  if (threadIdx.x == 0) { ref_mse[p] = er / J; other_mse[p] = eo / J; }
}

// mse[a][b] = mean_j |xy_b[j] - render_{view(b)}(tri_a[j])|^2 over the selected candidates (thread per (a, b))
__global__ __launch_bounds__(128) void ransac_mse_kernel(const double* __restrict__ views, const double* __restrict__ tri,
                                                         const int* __restrict__ cand_view, const double* __restrict__ cand_xy,
                                                         const int* __restrict__ sel, int C, int J, double* __restrict__ mse) {
  const int a = blockIdx.y, b = blockIdx.x * 128 + threadIdx.x;
  if (b >= C) return;
  const int ia = sel[a], ib = sel[b];
  const View w = load_view(views + (long long)cand_view[ib] * VIEW_DOUBLES);
  const double* t = tri + (long long)ia * J * 3;
  const double* xy = cand_xy + (long long)ib * J * 2;
  double e = 0.0;
  for (int j = 0; j < J; ++j) {
    double px, py;
    render(w, t[3 * j], t[3 * j + 1], t[3 * j + 2], px, py);
    const double dx = xy[2 * j] - px, dy = xy[2 * j + 1] - py;
    e += dx * dx + dy * dy;
  }
  mse[(long long)a * C + b] = e / J;
}

__global__ __launch_bounds__(128) void ransac_count_kernel(const double* __restrict__ mse, int C, double threshold, int* __restrict__ counts) {
  __shared__ int sh[2];
  int n = 0;
  for (int b = threadIdx.x; b < C; b += 128) n += mse[(long long)blockIdx.x * C + b] < threshold ? 1 : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1];
}

}  // namespace coma

using namespace coma;

extern "C" int coma_dlt_score_f64(const double* views, int n_views, int ref_view, const double* ref_xy, const int* cand_view,
                                  const double* cand_xy, int P, int J, double* tri, double* ref_mse, double* other_mse, void* stream) {
  if (!views || !ref_xy || !cand_view || !cand_xy || !tri || !ref_mse || !other_mse) return fail(COMA_E_INVALID, "coma_dlt_score_f64: null pointer");
  if (P < 0 || J <= 0 || n_views <= 0 || ref_view < 0 || ref_view >= n_views)
    return fail(COMA_E_INVALID, "coma_dlt_score_f64: bad sizes P=%d J=%d views=%d ref=%d", P, J, n_views, ref_view);
  if (P == 0) return COMA_OK;
  hipLaunchKernelGGL(dlt_score_kernel, dim3((unsigned)P), dim3(128), 0, (hipStream_t)stream, views, ref_view, ref_xy, cand_view, cand_xy, J,
                     tri, ref_mse, other_mse);
  return check_launch("dlt_score_kernel");
}

extern "C" int coma_ransac_mse_f64(const double* views, const double* tri, const int* cand_view, const double* cand_xy, const int* sel,
                                   int C, int J, double threshold, double* mse, int* counts, void* stream) {
  if (!views || !tri || !cand_view || !cand_xy || !sel || !mse || !counts) return fail(COMA_E_INVALID, "coma_ransac_mse_f64: null pointer");
  if (C < 0 || J <= 0) return fail(COMA_E_INVALID, "coma_ransac_mse_f64: bad sizes C=%d J=%d", C, J);
  if (C == 0) return COMA_OK;
  hipLaunchKernelGGL(ransac_mse_kernel, dim3((unsigned)((C + 127) / 128), (unsigned)C), dim3(128), 0, (hipStream_t)stream, views, tri,
                     cand_view, cand_xy, sel, C, J, mse);
  hipLaunchKernelGGL(ransac_count_kernel, dim3((unsigned)C), dim3(128), 0, (hipStream_t)stream, mse, C, threshold, counts);
  return check_launch("ransac kernels");
}
