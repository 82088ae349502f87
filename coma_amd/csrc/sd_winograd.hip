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
// fp16 RANGE: V and the plane products are stored as fp16 where the direct path keeps its sum in fp32 registers.
// V = B^T d B reaches 4 max|d| and a plane product can exceed the convolution output it cancels into, so both transforms take a
// power-of-two scale (uscale on U = G g G^T, vscale on V) that the output transform undoes in fp32 (mscale = 1 / (uscale vscale)):
// exact in the normal range, and the stored planes sit 4 x (ResNet convolutions, whose inputs are GroupNorm-bounded) or 16 x
// (Upsample2D convolutions, whose input is the raw residual stream: V itself needs the headroom) further from 65504.
#include <hip/hip_fp16.h>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// thread = (tile, chunk of 8 channels); V[p][t][c], p = 4 i + j
// affine != nullptr: the source is first normalised per (sample, channel) -- y = x * scale + shift, fp32 [batch][c0+c1][2], the table of a
// GroupNorm -- optionally passed through SiLU, and rounded to fp16 (what the GroupNorm kernel would have stored); pad pixels stay zero.
__global__ void __launch_bounds__(256) winograd_input_kernel(const _Float16* __restrict__ x0, const _Float16* __restrict__ x1, int c0,
                                                            int c1, int batch, int h, int w, int up, const float* __restrict__ affine,
                                                            int silu, float vscale, _Float16* __restrict__ v) {
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
  float sc[8], sh[8];
  if (affine) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 a = reinterpret_cast<const float2*>(affine)[(long long)b * c + ch + e];
      sc[e] = a.x; sh[e] = a.y;
    }
  }
  float d[4][4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 2 * tx - 1 + j;
      half8 q = {0, 0, 0, 0, 0, 0, 0, 0};
      // up = 1: the 3x3 window slides over the nearest-x2 upsampling of the [h/2, w/2] source (diffusers Upsample2D)
      if (y >= 0 && y < h && x >= 0 && x < w)
      {
        q = *reinterpret_cast<const half8*>(src + (((long long)b * (h >> up) + (y >> up)) * (w >> up) + (x >> up)) * ld + cc);
        if (affine) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)q[e], sc[e], sh[e]);
            if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));
            q[e] = (_Float16)f;
          }
        }
      }
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
      o[0][e] = (_Float16)(vscale * (t0 - t2)); o[1][e] = (_Float16)(vscale * (t1 + t2)); o[2][e] = (_Float16)(vscale * (t2 - t1));
      o[3][e] = (_Float16)(vscale * (t1 - t3));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<half8*>(v + ((long long)(4 * i + j) * ntile + t) * c + ch) = o[j];
  }
}

// U[p][n][c] = (G g G^T)[p], g = w[n][ky*3+kx][c]; thread = (n, c)
__global__ void winograd_weight_kernel(const _Float16* __restrict__ wsrc, int n, int c, float uscale, _Float16* __restrict__ u) {
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
    const float u0 = uscale * t[a][0], u1 = uscale * (0.5f * (t[a][0] + t[a][1] + t[a][2])), u2 = uscale * (0.5f * (t[a][0] - t[a][1] + t[a][2])),
                u3 = uscale * t[a][2];
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
                                                             _Float16* __restrict__ out, int ldo, int silu, float mscale) {
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
        float y = mscale * (bb == 0 ? s[a][0][e] + s[a][1][e] + s[a][2][e] : s[a][1][e] - s[a][2][e] - s[a][3][e]) + add[e];
        if (silu) y = y / (1.0f + __expf(-y));
        o[e] = (_Float16)(y + (float)r[e]);
      }
      *reinterpret_cast<half8*>(out + row * ldo + ch) = o;
    }
}

// The same output transform with the consumer's GroupNorm statistics: one block = one ROW of 2x2 tiles (w / 2 = 16 tiles) x 128 output
// channels, thread = (tile, chunk of 8 channels).  The two image rows the block writes are one 32-row slot each of the column-sum buffer
// cs fp32 [batch * h * w / 32][2][n] (sd_conv_gemm_desc.colstats layout): sums and sums of squares of the STORED fp16 values, folded over
// the 16 tiles through LDS in a fixed order (reproducible).  w == 32 only (the 32 x 32 level of the UNet).
__global__ void __launch_bounds__(256) winograd_output_cs_kernel(const _Float16* __restrict__ m, int ldm, int batch, int h, int n,
                                                                const _Float16* __restrict__ bias, const _Float16* __restrict__ bias_bn,
                                                                int ldbb, const _Float16* __restrict__ res, int ldr,
                                                                _Float16* __restrict__ out, int ldo, int silu, float mscale, float* __restrict__ cs) {
  constexpr int w = 32, tw = 16;
  __shared__ float red[2][2][16][128];                      // [sum | sumsq][image row a][tile][channel of the block]
  const int th = h >> 1;
  const long long ntile = (long long)batch * th * tw;
  const int tx = threadIdx.x >> 4, chunk = threadIdx.x & 15;
  const int R = blockIdx.x, b = R / th, ty = R - b * th;
  const int ch = blockIdx.y * 128 + chunk * 8;
  const long long t = (long long)R * tw + tx;
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
  for (int a = 0; a < 2; ++a) {
    float ps[8], pq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ps[e] = pq[e] = 0.0f;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const long long row = ((long long)b * h + 2 * ty + a) * w + 2 * tx + bb;
      half8 r = {0, 0, 0, 0, 0, 0, 0, 0};
      if (res) r = *reinterpret_cast<const half8*>(res + row * ldr + ch);
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = mscale * (bb == 0 ? s[a][0][e] + s[a][1][e] + s[a][2][e] : s[a][1][e] - s[a][2][e] - s[a][3][e]) + add[e];
        if (silu) y = y / (1.0f + __expf(-y));
        o[e] = (_Float16)(y + (float)r[e]);
        const float f = (float)o[e];
        ps[e] += f; pq[e] += f * f;
      }
      *reinterpret_cast<half8*>(out + row * ldo + ch) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][a][tx][chunk * 8 + e] = ps[e]; red[1][a][tx][chunk * 8 + e] = pq[e]; }
  }
  __syncthreads();
  // 2 (sum | sumsq) x 2 rows x 128 channels = 512 results, two per thread, each the fixed-order sum over the 16 tiles
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = threadIdx.x + 256 * k, which = idx >> 8, a = (idx >> 7) & 1, c = idx & 127;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += red[which][a][i][c];
    const long long slot = (long long)b * h + 2 * ty + a;              // w == 32: one image row = one 32-row slot
    cs[(slot * 2 + which) * n + blockIdx.y * 128 + c] = acc;
  }
}


// ---- GroupNorm (+ SiLU) of a SMALL feature map fused with the Winograd input transform: one block per (sample, group), the group's
// hw x (C / groups) slice held in LDS.  The slice comes either from NHWC tensors (mode 0: norm1 of a ResNet block, two concatenated
// sources) or straight from the 16 plane products of the PREVIOUS Winograd convolution (mode 1: conv1 -> norm2 -> conv2 of a ResNet
// block; h = A^T m A + bias + per-sample bias is formed here, rounded to fp16 as the stored tensor would be, and never written).
// Three launches of the unfused chain (output transform, GroupNorm, input transform) become one; the arithmetic of every stage is the
// unfused kernels' (same statistics formula, same affine + SiLU expression, transforms in fp32 rounded once).
constexpr int kGnWinoMaxSlice = 20480;          // halfs: 16 x 16 pixels x 80 channels (C = 2560, 32 groups)
typedef _Float16 half4w __attribute__((ext_vector_type(4)));

struct GnWinoArgs {
  const _Float16 *x0, *x1;
  int c0, c1;
  const _Float16* m;          // != nullptr: mode 1, fp16 [16][batch * T][ldm]
  int ldm;
  const _Float16 *bias, *bias_bn;
  int ldbb;
  int h, w, groups;
  float eps;
  const _Float16 *gamma, *beta;
  int silu;
  float mscale;               // mode 1: h = mscale * A^T m A + bias (the planes were stored scaled by 1 / mscale)
  _Float16* v;                // fp16 [16][batch * T][C]
};

__device__ __forceinline__ float wave_sum64(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// VEC = channels per lane in the plane-product load, the normalisation and the transform (8 when the group width allows 16-byte
// accesses, else 4).  The NHWC load keeps 4-channel chunks in sd_groupnorm_f16's own thread order, so that the statistics -- and with
// them every output bit -- equal the unfused GroupNorm kernel -> input transform chain.
template <int VEC>
__global__ __launch_bounds__(256) void gn_winograd_input_kernel(GnWinoArgs a) {
  typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
  __shared__ __attribute__((aligned(16))) _Float16 slice[kGnWinoMaxSlice];
  __shared__ float rs[4], rq[4];
  const int C = a.c0 + a.c1, cg = C / a.groups, q4 = cg >> 2, qv = cg / VEC;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int hw = a.h * a.w, th = a.h >> 1, tw = a.w >> 1, T = th * tw;
  const long long ntile = (long long)gridDim.y * T;
  float s = 0.0f, q = 0.0f;
  if (!a.m) {
    for (int it = tid; it < hw * q4; it += 256) {
      const int p = it / q4, cc = (it - p * q4) * 4, c = g * cg + cc;
      const long long pix = (long long)b * hw + p;
      const half4w v4 = *reinterpret_cast<const half4w*>(c < a.c0 ? a.x0 + pix * a.c0 + c : a.x1 + pix * a.c1 + (c - a.c0));
      *reinterpret_cast<half4w*>(slice + p * cg + cc) = v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float f = (float)v4[j]; s += f; q += f * f; }
    }
  } else {
    for (int it = tid; it < T * qv; it += 256) {
      const int t = it / qv, cc = (it - t * qv) * VEC, c = g * cg + cc;
      const int ty = t / tw, tx = t - ty * tw;
      const long long tg = (long long)b * T + t;
      float sm[2][4][VEC];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float mm[4][VEC];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const hv v4 = *reinterpret_cast<const hv*>(a.m + ((long long)(4 * i + j) * ntile + tg) * a.ldm + c);
#pragma unroll
          for (int e = 0; e < VEC; ++e) mm[i][e] = (float)v4[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          sm[0][j][e] = mm[0][e] + mm[1][e] + mm[2][e];
          sm[1][j][e] = mm[1][e] - mm[2][e] - mm[3][e];
        }
      }
      float add[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) add[e] = 0.0f;
      if (a.bias) { const hv v4 = *reinterpret_cast<const hv*>(a.bias + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) add[e] += (float)v4[e]; }
      if (a.bias_bn) { const hv v4 = *reinterpret_cast<const hv*>(a.bias_bn + (long long)b * a.ldbb + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) add[e] += (float)v4[e]; }
#pragma unroll
      for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          hv o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float y = a.mscale * (xb == 0 ? sm[ya][0][e] + sm[ya][1][e] + sm[ya][2][e] : sm[ya][1][e] - sm[ya][2][e] - sm[ya][3][e]) + add[e];
            o[e] = (_Float16)y;
            const float f = (float)o[e];            // the statistics see the fp16 tensor the unfused chain would have stored
            s += f; q += f * f;
          }
          *reinterpret_cast<hv*>(slice + ((2 * ty + ya) * a.w + 2 * tx + xb) * cg + cc) = o;
        }
    }
  }
  s = wave_sum64(s);
  q = wave_sum64(q);
  if ((tid & 63) == 0) { rs[tid >> 6] = s; rq[tid >> 6] = q; }
  __syncthreads();
  s = (rs[0] + rs[1]) + (rs[2] + rs[3]);
  q = (rq[0] + rq[1]) + (rq[2] + rq[3]);
  const float count = (float)hw * (float)cg;
  const float mean = s / count;
  const float var = fmaxf(q / count - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + a.eps);
  for (int it = tid; it < hw * qv; it += 256) {
    const int p = it / qv, cc = (it - p * qv) * VEC, c = g * cg + cc;
    const hv ga = *reinterpret_cast<const hv*>(a.gamma + c), be = *reinterpret_cast<const hv*>(a.beta + c);
    hv v4 = *reinterpret_cast<hv*>(slice + p * cg + cc);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float sc = rstd * (float)ga[j];
      float y = fmaf((float)v4[j], sc, (float)be[j] - mean * sc);
      if (a.silu) y = y / (1.0f + __expf(-y));
      v4[j] = (_Float16)y;
    }
    *reinterpret_cast<hv*>(slice + p * cg + cc) = v4;
  }
  __syncthreads();
  for (int it = tid; it < T * qv; it += 256) {
    const int t = it / qv, cc = (it - t * qv) * VEC, c = g * cg + cc;
    const int ty = t / tw, tx = t - ty * tw;
    float d[4][4][VEC];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 2 * ty - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = 2 * tx - 1 + j;
        hv v4;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v4[e] = (_Float16)0.0f;
        if (y >= 0 && y < a.h && x >= 0 && x < a.w) v4 = *reinterpret_cast<const hv*>(slice + (y * a.w + x) * cg + cc);
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[i][j][e] = (float)v4[e];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d0 = d[0][j][e], d1 = d[1][j][e], d2 = d[2][j][e], d3 = d[3][j][e];
        d[0][j][e] = d0 - d2; d[1][j][e] = d1 + d2; d[2][j][e] = d2 - d1; d[3][j][e] = d1 - d3;
      }
    const long long tg = (long long)b * T + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hv o[4];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float t0 = d[i][0][e], t1 = d[i][1][e], t2 = d[i][2][e], t3 = d[i][3][e];
        o[0][e] = (_Float16)(t0 - t2); o[1][e] = (_Float16)(t1 + t2); o[2][e] = (_Float16)(t2 - t1); o[3][e] = (_Float16)(t1 - t3);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<hv*>(a.v + ((long long)(4 * i + j) * ntile + tg) * C + c) = o[j];
    }
  }
}

}  // namespace sd

extern "C" {

int sd_winograd_input_f16(const void* x0, const void* x1, int c0, int c1, int batch, int h, int w, int upsample, const float* gn_affine,
                          int silu, float vscale, void* v, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_WINO_IN;
    r.p[0] = (void*)x0; r.p[1] = (void*)x1; r.p[2] = v; r.p[3] = (void*)gn_affine;
    r.i[0] = c0; r.i[1] = c1; r.i[2] = batch; r.i[3] = h; r.i[4] = w; r.i[5] = upsample; r.i[6] = silu; r.f[0] = vscale;
    return plan_record(r);
  }
  if (upsample != 0 && upsample != 1) return fail(COMA_E_INVALID, "sd_winograd_input_f16: upsample must be 0 or 1");
  if (!x0 || !v) return fail(COMA_E_INVALID, "sd_winograd_input_f16: null pointer");
  if (!(vscale > 0.0f)) return fail(COMA_E_INVALID, "sd_winograd_input_f16: vscale must be positive");
  if (c0 <= 0 || c0 % 8 || c1 < 0 || c1 % 8 || (c1 > 0 && !x1)) return fail(COMA_E_INVALID, "sd_winograd_input_f16: channel counts must be multiples of 8");
  if (batch <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return fail(COMA_E_INVALID, "sd_winograd_input_f16: even h, w required");
  const long long total = (long long)batch * (h / 2) * (w / 2) * ((c0 + c1) / 8);
  hipLaunchKernelGGL(winograd_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x0, (const _Float16*)x1, c0, c1, batch, h, w, upsample, gn_affine, silu, vscale, (_Float16*)v);
  return check_launch("sd_winograd_input_f16");
}

int sd_winograd_weight_f16(const void* w, int n, int c, float uscale, void* u, void* stream) {
  using namespace sd;
  if (!w || !u || n <= 0 || c <= 0 || !(uscale > 0.0f)) return fail(COMA_E_INVALID, "sd_winograd_weight_f16: bad arguments");
  const long long total = (long long)n * c;
  hipLaunchKernelGGL(winograd_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)w, n, c, uscale, (_Float16*)u);
  return check_launch("sd_winograd_weight_f16");
}

int sd_winograd_output_f16(const void* m, int ldm, int batch, int h, int w, int n, const void* bias, const void* bias_bn, int ldbb,
                           const void* res, int ldr, void* out, int ldo, int silu, float mscale, float* colstats, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_WINO_OUT;
    r.p[0] = (void*)m; r.p[1] = (void*)bias; r.p[2] = (void*)bias_bn; r.p[3] = (void*)res; r.p[4] = out; r.p[5] = colstats;
    r.i[0] = ldm; r.i[1] = batch; r.i[2] = h; r.i[3] = w; r.i[4] = n; r.i[5] = ldbb; r.i[6] = ldr; r.i[7] = ldo; r.i[8] = silu; r.f[0] = mscale;
    return plan_record(r);
  }
  if (!m || !out) return fail(COMA_E_INVALID, "sd_winograd_output_f16: null pointer");
  if (!(mscale > 0.0f)) return fail(COMA_E_INVALID, "sd_winograd_output_f16: mscale must be positive");
  if (n <= 0 || n % 8 || ldm % 8 || batch <= 0 || (h & 1) || (w & 1)) return fail(COMA_E_INVALID, "sd_winograd_output_f16: bad shape");
  if (ldo == 0) ldo = n;
  if (ldr == 0) ldr = n;
  if (ldbb == 0) ldbb = n;
  if (colstats) {
    if (w != 32 || n % 128) return fail(COMA_E_INVALID, "sd_winograd_output_f16: column sums need w = 32 and n %% 128 == 0 (w=%d n=%d)", w, n);
    hipLaunchKernelGGL(winograd_output_cs_kernel, dim3((unsigned)(batch * (h / 2)), (unsigned)(n / 128)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)m, ldm, batch, h, n, (const _Float16*)bias, (const _Float16*)bias_bn, ldbb, (const _Float16*)res, ldr,
                       (_Float16*)out, ldo, silu, mscale, colstats);
    return check_launch("sd_winograd_output_f16");
  }
  const long long total = (long long)batch * (h / 2) * (w / 2) * (n / 8);
  hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)m, ldm, batch, h, w, n, (const _Float16*)bias, (const _Float16*)bias_bn, ldbb, (const _Float16*)res,
                     ldr, (_Float16*)out, ldo, silu, mscale);
  return check_launch("sd_winograd_output_f16");
}

int sd_gn_winograd_input_f16(const void* x0, const void* x1, int c0, int c1, const void* m, int ldm, const void* bias, const void* bias_bn,
                             int ldbb, int batch, int h, int w, int groups, float eps, const void* gamma, const void* beta, int silu, float mscale,
                             void* v, void* stream) {
  using namespace sd;
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_GN_WINO_IN;
    r.p[0] = (void*)x0; r.p[1] = (void*)x1; r.p[2] = (void*)m; r.p[3] = (void*)bias; r.p[4] = (void*)bias_bn; r.p[5] = (void*)gamma;
    r.p[6] = (void*)beta; r.p[7] = v;
    r.i[0] = c0; r.i[1] = c1; r.i[2] = ldm; r.i[3] = ldbb; r.i[4] = batch; r.i[5] = h; r.i[6] = w; r.i[7] = groups; r.i[8] = silu; r.f[0] = eps; r.f[1] = mscale;
    return plan_record(r);
  }
  if ((!x0 && !m) || !gamma || !beta || !v) return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: null pointer");
  if (m && !(mscale > 0.0f)) return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: mscale must be positive");
  if (m && (x0 || x1 || c1)) return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: either NHWC sources or plane products, not both");
  const int C = c0 + c1;
  if (c0 <= 0 || c0 % 4 || c1 < 0 || c1 % 4 || (c1 > 0 && !x1) || groups <= 0 || C % groups || (C / groups) % 4)
    return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: channels per group must be a multiple of 4 (c0=%d c1=%d groups=%d)", c0, c1, groups);
  if (batch <= 0 || batch > 65535 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: even h, w required");
  if ((long long)h * w * (C / groups) > kGnWinoMaxSlice)
    return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: a group's slice (%d x %d pixels x %d channels) exceeds %d elements", h, w, C / groups, kGnWinoMaxSlice);
  if (m && (ldm % 4 || ldm < C)) return fail(COMA_E_INVALID, "sd_gn_winograd_input_f16: ldm = %d", ldm);
  GnWinoArgs a;
  a.x0 = (const _Float16*)x0; a.x1 = (const _Float16*)x1; a.c0 = c0; a.c1 = c1; a.m = (const _Float16*)m; a.ldm = ldm;
  a.bias = (const _Float16*)bias; a.bias_bn = (const _Float16*)bias_bn; a.ldbb = ldbb > 0 ? ldbb : C; a.h = h; a.w = w; a.groups = groups;
  a.eps = eps; a.gamma = (const _Float16*)gamma; a.beta = (const _Float16*)beta; a.silu = silu; a.mscale = mscale; a.v = (_Float16*)v;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((C / groups) % 8 == 0 && (!m || ldm % 8 == 0) && a.ldbb % 8 == 0 && al16(m) && al16(bias) && al16(bias_bn) && al16(gamma) && al16(beta) && al16(v))
    hipLaunchKernelGGL(gn_winograd_input_kernel<8>, dim3((unsigned)groups, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(gn_winograd_input_kernel<4>, dim3((unsigned)groups, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("sd_gn_winograd_input_f16");
}

}  // extern "C"
