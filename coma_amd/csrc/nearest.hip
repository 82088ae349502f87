// K7: nearest-vertex index map (gfx950).
// replaces: utils/coma.py:87-91 -- argmin_v sum((p_i - m_v)^2) over an f64 [V,P] table, first minimum
// wins ties (np.argmin).  One workgroup per query point, threads strided over the mesh vertices
// (coalesced 24-B records), f64 arithmetic in the reference's order ((dx^2+dy^2)+dz^2, no FMA), then a
// (distance, index) lexicographic min-reduction so the tie-break is identical.
#include "common.h"

namespace coma {

__global__ __launch_bounds__(256) void nearest_vertex_kernel(const double* __restrict__ pts,
                                                             const double* __restrict__ verts, int V,
                                                             int64_t* __restrict__ idx) {
  __shared__ double sd[4];
  __shared__ int si[4];
  const double px = pts[3 * (int64_t)blockIdx.x + 0], py = pts[3 * (int64_t)blockIdx.x + 1],
               pz = pts[3 * (int64_t)blockIdx.x + 2];
  double best = __builtin_inf();
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    double dx = px - verts[3 * (int64_t)v + 0], dy = py - verts[3 * (int64_t)v + 1],
           dz = pz - verts[3 * (int64_t)v + 2];
    double d = (dx * dx + dy * dy) + dz * dz;
    if (d < best) { best = d; bi = v; }   // ascending v per thread: strict < keeps the first minimum
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    double od = __shfl_xor(best, m);
    int oi = __shfl_xor(bi, m);
    if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sd[w] < best || (sd[w] == best && si[w] < bi)) { best = sd[w]; bi = si[w]; }
    // a NaN row never satisfies d < best; np.argmin would return the first NaN -- report 0 like an
    // all-NaN argmin of an empty comparison chain only if nothing was ever selected
    idx[blockIdx.x] = (bi == 0x7fffffff) ? 0 : (int64_t)bi;
  }
}

}  // namespace coma

using namespace coma;

extern "C" int coma_nearest_vertex_i64(const double* points, const double* verts, int P, int V,
                                       int64_t* idx, void* stream) {
  if (!points || !verts || !idx) return fail(COMA_E_INVALID, "coma_nearest_vertex_i64: null pointer");
  if (P < 0 || V <= 0) return fail(COMA_E_INVALID, "coma_nearest_vertex_i64: bad sizes P=%d V=%d", P, V);
  if (P == 0) return COMA_OK;
  hipLaunchKernelGGL(nearest_vertex_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, points,
                     verts, V, idx);
  return check_launch("nearest_vertex_kernel");
}
