// Sample ingestion: area-weighted vertex normals of S posed meshes sharing one topology.
// replaces: the per-sample open3d `TriangleMesh.compute_vertex_normals()` + normalize_vectors_np of
// prepare_affordance_extraction_inputs (utils/coma.py:672-686).  open3d (third party, absent here -> parity
// unpinned) sums the UN-normalised triangle cross products per vertex, walking the faces in order, then normalises;
// the same order is kept here through a vertex->face CSR list (ascending face index), all in f64, no atomics
// -> deterministic.  A zero normal becomes (0,0,1) as in open3d's NormalizeNormals.
#include "common.h"

namespace coma {

__global__ void vertex_normals_kernel(const double* __restrict__ verts, const int32_t* __restrict__ faces,
                                      const int32_t* __restrict__ off, const int32_t* __restrict__ vf, int V, double eps,
                                      double* __restrict__ normals) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (v >= V) return;
  const double* P = verts + (int64_t)s * V * 3;
  double nx = 0.0, ny = 0.0, nz = 0.0;
  for (int e = off[v]; e < off[v + 1]; ++e) {
    const int f = vf[e];
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const double ax = P[3 * i1] - P[3 * i0], ay = P[3 * i1 + 1] - P[3 * i0 + 1], az = P[3 * i1 + 2] - P[3 * i0 + 2];
    const double bx = P[3 * i2] - P[3 * i0], by = P[3 * i2 + 1] - P[3 * i0 + 1], bz = P[3 * i2 + 2] - P[3 * i0 + 2];
    nx += ay * bz - az * by;
    ny += az * bx - ax * bz;
    nz += ax * by - ay * bx;
  }
  double n = sqrt((nx * nx + ny * ny) + nz * nz);
  if (n > 0.0) { nx /= n; ny /= n; nz /= n; } else { nx = 0.0; ny = 0.0; nz = 1.0; }
  if (eps >= 0.0) {   // normalize_vectors_np(., eps): v / (|v| + eps)
    n = sqrt((nx * nx + ny * ny) + nz * nz) + eps;
    nx /= n; ny /= n; nz /= n;
  }
  double* o = normals + ((int64_t)s * V + v) * 3;
  o[0] = nx; o[1] = ny; o[2] = nz;
}

}  // namespace coma

using namespace coma;

extern "C" int coma_vertex_normals_f64(const double* verts, const int32_t* faces, const int32_t* vf_offsets,
                                       const int32_t* vf_faces, int S, int V, int F, double eps, double* normals, void* stream) {
  if (!verts || !faces || !vf_offsets || !vf_faces || !normals) return fail(COMA_E_INVALID, "coma_vertex_normals_f64: null pointer");
  if (S <= 0 || V <= 0 || F <= 0) return fail(COMA_E_INVALID, "coma_vertex_normals_f64: bad sizes");
  hipLaunchKernelGGL(vertex_normals_kernel, dim3((V + 127) / 128, S), dim3(128), 0, (hipStream_t)stream, verts, faces, vf_offsets,
                     vf_faces, V, eps, normals);
  return check_launch("vertex_normals_kernel");
}
