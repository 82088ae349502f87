// sd_winograd.hip -- Winograd F(2x2, 3x3) transforms around a plane-batched GEMM (gfx950).
//
// A 3x3 / stride-1 / pad-1 convolution (diffusers Conv2d inside self.unet(...) / self.vae.decode,
// utils/adaptive_mask_inpainting.py:1001-1007, :1086, :1112) as
//     Y = A^T [ (G g G^T) o (B^T d B) ] A
// per 2x2 output tile: 16 element-wise positions ("planes"), each of them a plain [T, C_in] x [C_in, C_out] product
// (T = batch * H/2 * W/2 tiles) that runs through sd_conv_gemm_f16 with nbatch_z = 16.  4 * M * C_in * C_out multiplies instead
// of 9 * M * C_in * C_out: 2.25 x fewer MFMA flops, paid for with a 4 x larger (transformed) activation tensor on the way in and
// a 4 x larger product tensor on the way out.  fp16 storage / fp32 arithmetic like every other sd_* operator.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include <hip/hip_fp16.h>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// thread = (tile, chunk of 8 channels); V[p][t][c], p = 4 i + j
__global__ void __launch_bounds__(256) winograd_input_kernel(const _Float16* __restrict__ x0, const _Float16* __restrict__ x1, int c0,
                                                            int c1, int batch, int h, int w, _Float16* __restrict__ v) {
  const int c = c0 + c1, cch = c >> 3;
  const int th = h >> 1, tw = w >> 1;
  const long long ntile = (long long)batch * th * tw;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ntile * cch) return;
  const int ch = (int)(gid % cch) * 8;
  const long long t = gid / cch;
  const int tx = (int)(t % tw);
  const int ty = (int)((t / tw) % th);
  const int b = (int)(t / ((long long)tw * th));
  const _Float16* src;
  int ld, cc;
  if (ch < c0) { src = x0; ld = c0; cc = ch; } else { src = x1; ld = c1; cc = ch - c0; }
  float d[4][4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 2 * tx - 1 + j;
      half8 q = {0, 0, 0, 0, 0, 0, 0, 0};
      if (y >= 0 && y < h && x >= 0 && x < w) q = *reinterpret_cast<const half8*>(src + (((long long)b * h + y) * w + x) * ld + cc);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[i][j][e] = (float)q[e];
    }
  }
  // t = B^T d (columns), then V = t B (rows)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d0 = d[0][j][e], d1 = d[1][j][e], d2 = d[2][j][e], d3 = d[3][j][e];
      d[0][j][e] = d0 - d2; d[1][j][e] = d1 + d2; d[2][j][e] = d2 - d1; d[3][j][e] = d1 - d3;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    half8 o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t0 = d[i][0][e], t1 = d[i][1][e], t2 = d[i][2][e], t3 = d[i][3][e];
      o[0][e] = (_Float16)(t0 - t2); o[1][e] = (_Float16)(t1 + t2); o[2][e] = (_Float16)(t2 - t1); o[3][e] = (_Float16)(t1 - t3);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<half8*>(v + ((long long)(4 * i + j) * ntile + t) * c + ch) = o[j];
  }
}

// U[p][n][c] = (G g G^T)[p], g = w[n][ky*3+kx][c]; thread = (n, c)
__global__ void winograd_weight_kernel(const _Float16* __restrict__ wsrc, int n, int c, _Float16* __restrict__ u) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n * c) return;
  const int ci = (int)(gid % c);
  const int ni = (int)(gid / c);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = (float)wsrc[((long long)ni * 9 + a * 3 + b) * c + ci];
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
    u[((long long)(4 * a + 0) * n + ni) * c + ci] = (_Float16)u0;
    u[((long long)(4 * a + 1) * n + ni) * c + ci] = (_Float16)u1;
    u[((long long)(4 * a + 2) * n + ni) * c + ci] = (_Float16)u2;
    u[((long long)(4 * a + 3) * n + ni) * c + ci] = (_Float16)u3;
  }
}

// thread = (tile, chunk of 8 output channels): Y = A^T m A (+ bias, per-sample bias, SiLU, residual) -> 4 output pixels
__global__ void __launch_bounds__(256) winograd_output_kernel(const _Float16* __restrict__ m, int ldm, int batch, int h, int w, int n,
                                                             const _Float16* __restrict__ bias, const _Float16* __restrict__ bias_bn,
                                                             int ldbb, const _Float16* __restrict__ res, int ldr,
                                                             _Float16* __restrict__ out, int ldo, int silu) {
  const int nch = n >> 3;
  const int th = h >> 1, tw = w >> 1;
  const long long ntile = (long long)batch * th * tw;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ntile * nch) return;
  const int ch = (int)(gid % nch) * 8;
  const long long t = gid / nch;
  const int tx = (int)(t % tw);
  const int ty = (int)((t / tw) % th);
  const int b = (int)(t / ((long long)tw * th));
  float s[2][4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float mm[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const half8 q = *reinterpret_cast<const half8*>(m + ((long long)(4 * i + j) * ntile + t) * ldm + ch);
#pragma unroll
      for (int e = 0; e < 8; ++e) mm[i][e] = (float)q[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[0][j][e] = mm[0][e] + mm[1][e] + mm[2][e];
      s[1][j][e] = mm[1][e] - mm[2][e] - mm[3][e];
    }
  }
  float add[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) add[e] = 0.0f;
  if (bias) {
    const half8 q = *reinterpret_cast<const half8*>(bias + ch);
#pragma unroll
    for (int e = 0; e < 8; ++e) add[e] += (float)q[e];
  }
  if (bias_bn) {
    const half8 q = *reinterpret_cast<const half8*>(bias_bn + (long long)b * ldbb + ch);
#pragma unroll
    for (int e = 0; e < 8; ++e) add[e] += (float)q[e];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const long long row = ((long long)b * h + 2 * ty + a) * w + 2 * tx + bb;
      half8 r = {0, 0, 0, 0, 0, 0, 0, 0};
      if (res) r = *reinterpret_cast<const half8*>(res + row * ldr + ch);
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = (bb == 0 ? s[a][0][e] + s[a][1][e] + s[a][2][e] : s[a][1][e] - s[a][2][e] - s[a][3][e]) + add[e];
        if (silu) y = y / (1.0f + __expf(-y));
        o[e] = (_Float16)(y + (float)r[e]);
      }
      *reinterpret_cast<half8*>(out + row * ldo + ch) = o;
    }
}

}  // namespace sd

extern "C" {

int sd_winograd_input_f16(const void* x0, const void* x1, int c0, int c1, int batch, int h, int w, void* v, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_WINO_IN;
    r.p[0] = (void*)x0; r.p[1] = (void*)x1; r.p[2] = v; r.i[0] = c0; r.i[1] = c1; r.i[2] = batch; r.i[3] = h; r.i[4] = w;
    return plan_record(r);
  }
  if (!x0 || !v) return fail(COMA_E_INVALID, "sd_winograd_input_f16: null pointer");
  if (c0 <= 0 || c0 % 8 || c1 < 0 || c1 % 8 || (c1 > 0 && !x1)) return fail(COMA_E_INVALID, "sd_winograd_input_f16: channel counts must be multiples of 8");
  if (batch <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return fail(COMA_E_INVALID, "sd_winograd_input_f16: even h, w required");
  const long long total = (long long)batch * (h / 2) * (w / 2) * ((c0 + c1) / 8);
  hipLaunchKernelGGL(winograd_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x0, (const _Float16*)x1, c0, c1, batch, h, w, (_Float16*)v);
  return check_launch("sd_winograd_input_f16");
}

int sd_winograd_weight_f16(const void* w, int n, int c, void* u, void* stream) {
  using namespace sd;
  if (!w || !u || n <= 0 || c <= 0) return fail(COMA_E_INVALID, "sd_winograd_weight_f16: bad arguments");
  const long long total = (long long)n * c;
  hipLaunchKernelGGL(winograd_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)w, n, c, (_Float16*)u);
  return check_launch("sd_winograd_weight_f16");
}

int sd_winograd_output_f16(const void* m, int ldm, int batch, int h, int w, int n, const void* bias, const void* bias_bn, int ldbb,
                           const void* res, int ldr, void* out, int ldo, int silu, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_WINO_OUT;
    r.p[0] = (void*)m; r.p[1] = (void*)bias; r.p[2] = (void*)bias_bn; r.p[3] = (void*)res; r.p[4] = out;
    r.i[0] = ldm; r.i[1] = batch; r.i[2] = h; r.i[3] = w; r.i[4] = n; r.i[5] = ldbb; r.i[6] = ldr; r.i[7] = ldo; r.i[8] = silu;
    return plan_record(r);
  }
  if (!m || !out) return fail(COMA_E_INVALID, "sd_winograd_output_f16: null pointer");
  if (n <= 0 || n % 8 || ldm % 8 || batch <= 0 || (h & 1) || (w & 1)) return fail(COMA_E_INVALID, "sd_winograd_output_f16: bad shape");
  if (ldo == 0) ldo = n;
  if (ldr == 0) ldr = n;
  if (ldbb == 0) ldbb = n;
  const long long total = (long long)batch * (h / 2) * (w / 2) * (n / 8);
  hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)m, ldm, batch, h, w, n, (const _Float16*)bias, (const _Float16*)bias_bn, ldbb, (const _Float16*)res,
                     ldr, (_Float16*)out, ldo, silu);
  return check_launch("sd_winograd_output_f16");
}

}  // extern "C"
