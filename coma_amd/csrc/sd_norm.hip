// sd_norm.hip -- GroupNorm(+SiLU), LayerNorm and row softmax over NHWC fp16 (gfx950).  HBM-bound kernels:
// every global access is a 16-byte vector of 8 consecutive channels, statistics in fp32.
//
// GroupNorm runs as (1) per-(pixel-chunk) partial sums written to a scratch slab, (2) a tiny finalize that folds
// the partials in a fixed order (deterministic, no atomics, no memset) into mean / rstd, (3) a fully coalesced
// normalise + affine + SiLU pass.  It can read the channel concatenation of TWO tensors and writes one, which is
// how the UNet's skip-connection torch.cat is materialised for free.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;
using coma::kWave;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int GN_PIX = 64;       // pixels per partial-sum block
constexpr int GN_MAX_GROUPS = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ half8 load8(const _Float16* x0, const _Float16* x1, int c0, int c1, long long pix, int c) {
  const _Float16* p = c < c0 ? x0 + pix * c0 + c : x1 + pix * c1 + (c - c0);
  return *reinterpret_cast<const half8*>(p);
}

// partial[b][chunk][g][2] = (sum, sumsq) over GN_PIX pixels x (C/G) channels.
// Thread mapping: tx = 8-channel chunk, ty = pixel lane; all 256 threads stay busy for every C (a C=320 tensor has only
// 40 channel chunks, so 6 pixel lanes share them) and each thread walks GN_PIX/ny pixels.  The per-(pixel lane, channel)
// sums are folded in LDS in a fixed order -> bitwise reproducible statistics, no atomics.
constexpr int GN_MAX_C = 2560;
__global__ __launch_bounds__(256) void gn_partial_kernel(const _Float16* __restrict__ x0, const _Float16* __restrict__ x1,
                                                         int c0, int c1, int hw, int groups, float* __restrict__ partial) {
  __shared__ float chs[GN_MAX_C], chq[GN_MAX_C];   // per-channel sums of this pixel chunk
  __shared__ float red[16 * 256];                  // [8 sums | 8 sumsq][thread]: consecutive threads -> consecutive banks
  const int C = c0 + c1, cg = C / groups, c8 = C / 8;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int p0 = chunk * GN_PIX, p1 = min(hw, p0 + GN_PIX);
  const int nx = min(c8, 256), ny = 256 / nx;      // nx channel chunks per pass, ny pixel lanes
  const int tx = threadIdx.x % nx, ty = threadIdx.x / nx;
  for (int cb = 0; cb < c8; cb += nx) {            // passes over channel chunks (1 unless C > 2048)
    const int ch = cb + tx;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0f;
    if (ty < ny && ch < c8) {
      for (int p = p0 + ty; p < p1; p += ny) {
        half8 v = load8(x0, x1, c0, c1, (long long)b * hw + p, ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = (float)v[j];
          s[j] += f;
          q[j] += f * f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[j * 256 + threadIdx.x] = s[j];
      red[(8 + j) * 256 + threadIdx.x] = q[j];
    }
    __syncthreads();
    if (ty == 0 && ch < c8) {                      // fold the pixel lanes in a fixed order
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float ss = 0.0f, qq = 0.0f;
        for (int y = 0; y < ny; ++y) {
          ss += red[j * 256 + y * nx + tx];
          qq += red[(8 + j) * 256 + y * nx + tx];
        }
        chs[ch * 8 + j] = ss;
        chq[ch * 8 + j] = qq;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < groups) {   // fixed summation order -> bitwise reproducible statistics
    float s = 0.0f, q = 0.0f;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) {
      s += chs[c];
      q += chq[c];
    }
    float* p = partial + ((long long)b * nchunk + chunk) * groups * 2 + threadIdx.x * 2;
    p[0] = s;
    p[1] = q;
  }
}

// One wave per (b, g): fold the per-chunk partial sums (fixed order), then write the affine table the apply pass uses:
//   y = x * scale[b][c] + shift[b][c],  scale = rstd * gamma[c],  shift = beta[c] - mean * rstd * gamma[c]
__device__ __forceinline__ void gn_write_affine(float s, float q, float count, float eps, int b, int g, int cg, int C,
                                                const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                                float* __restrict__ affine, int lane) {
  const float mean = s / count;
  const float var = fmaxf(q / count - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + eps);
  for (int c = g * cg + lane; c < (g + 1) * cg; c += 64) {
    const float sc = rstd * (float)gamma[c];
    affine[((long long)b * C + c) * 2] = sc;
    affine[((long long)b * C + c) * 2 + 1] = (float)beta[c] - mean * sc;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int groups, int C,
                                                          float count, float eps, const _Float16* __restrict__ gamma,
                                                          const _Float16* __restrict__ beta, float* __restrict__ affine, int total) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, g)
  const int lane = threadIdx.x & 63;
  if (i >= total) return;
  const int b = i / groups, g = i - b * groups;
  float s = 0.0f, q = 0.0f;
  for (int c = lane; c < nchunk; c += 64) {
    const float* p = partial + ((long long)b * nchunk + c) * groups * 2 + g * 2;
    s += p[0];
    q += p[1];
  }
  s = wave_sum(s);
  q = wave_sum(q);
  gn_write_affine(s, q, count, eps, b, g, C / groups, C, gamma, beta, affine, lane);
}

// the same from per-32-row-block column sums [rb][2][C] of one or two producers: one 256-thread block per (b, g),
// thread t owns channel (t % cg) of the group and every (256 / cg)-th row block; LDS fold in a fixed order
__global__ __launch_bounds__(256) void gn_finalize_colstats_kernel(const float* __restrict__ cs0, const float* __restrict__ cs1,
                                                                  int c0, int c1, int hw, int rbs, int groups, float eps,
                                                                  const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                                                  float* __restrict__ affine) {
  // rbs = slots per sample (hw / 32 for the GEMM epilogues' 32-row slots; hw / 256 for sd_conv3x3_halo_f16's per-tile slots)
  __shared__ float rs[256], rq[256];
  const int C = c0 + c1, cg = C / groups;
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int per = 256 / cg;                       // row-block lanes (cg <= 80 in every SD layer)
  const int tc = threadIdx.x % cg, tr = threadIdx.x / cg;
  float s = 0.0f, q = 0.0f;
  if (tr < per) {
    const int c = g * cg + tc;
    const float* base = c < c0 ? cs0 + c : cs1 + (c - c0);
    const int cw = c < c0 ? c0 : c1;
    // eight row blocks per trip, all sixteen loads independent and the tail predicated (index clamped, weight 0) instead of a serial
    // remainder loop: at the UNet's shapes (128 row blocks, 25 row lanes) the whole sum is ONE round of memory latency where the
    // four-per-trip loop + remainder took three (11.5 us per launch, 31 launches per forward); sums still in row-block order
    const long long st = (long long)per * 2 * cw;
    for (int rb = tr; rb < rbs; rb += 8 * per) {
      const float* p = base + ((long long)(b * rbs + rb) * 2) * cw;
      float av[8], bv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = rb + k * per < rbs;
        const float* pk = ok ? p + k * st : p;
        av[k] = pk[0];
        bv[k] = pk[cw];
        if (!ok) av[k] = bv[k] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s += av[k];
        q += bv[k];
      }
    }
  }
  rs[threadIdx.x] = s;
  rq[threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < 64) {
    float ss = 0.0f, qq = 0.0f;
    for (int i = threadIdx.x; i < per * cg; i += 64) {
      ss += rs[i];
      qq += rq[i];
    }
    ss = wave_sum(ss);
    qq = wave_sum(qq);
    gn_write_affine(ss, qq, (float)hw * (float)cg, eps, b, g, cg, C, gamma, beta, affine, threadIdx.x);
  }
}

// Small feature maps (16 x 16 and 8 x 8 levels of the UNet): one launch, one block per (sample, group).  The group's
// hw x (C/G) slice (<= GN_SMALL_ITEMS * 256 four-channel chunks) is read ONCE into registers, reduced through LDS in a
// fixed order, normalised and written -- three launches and two extra passes over L2 become one.
constexpr int GN_SMALL_ITEMS = 20;
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gn_small_kernel(const _Float16* __restrict__ x0, const _Float16* __restrict__ x1, int c0,
                                                       int c1, int hw, int groups, float eps, const _Float16* __restrict__ gamma,
                                                       const _Float16* __restrict__ beta, int silu, _Float16* __restrict__ out) {
  __shared__ float rs[4], rq[4];
  const int C = c0 + c1, cg = C / groups, q4 = cg / 4;
  const int g = blockIdx.x, b = blockIdx.y;
  const int items = hw * q4;
  half4v v[GN_SMALL_ITEMS];
  float s = 0.0f, q = 0.0f;
#pragma unroll
  for (int k = 0; k < GN_SMALL_ITEMS; ++k) {
    const int it = threadIdx.x + k * 256;
    if (it < items) {
      const int p = it / q4, c = g * cg + (it - p * q4) * 4;
      const long long pix = (long long)b * hw + p;
      v[k] = *reinterpret_cast<const half4v*>(c < c0 ? x0 + pix * c0 + c : x1 + pix * c1 + (c - c0));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = (float)v[k][j];
        s += f;
        q += f * f;
      }
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rq[threadIdx.x >> 6] = q; }
  __syncthreads();
  s = (rs[0] + rs[1]) + (rs[2] + rs[3]);
  q = (rq[0] + rq[1]) + (rq[2] + rq[3]);
  const float count = (float)hw * (float)cg;
  const float mean = s / count;
  const float var = fmaxf(q / count - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < GN_SMALL_ITEMS; ++k) {
    const int it = threadIdx.x + k * 256;
    if (it < items) {
      const int p = it / q4, c = g * cg + (it - p * q4) * 4;
      const half4v ga = *reinterpret_cast<const half4v*>(gamma + c), be = *reinterpret_cast<const half4v*>(beta + c);
      half4v o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = rstd * (float)ga[j];
        float y = fmaf((float)v[k][j], sc, (float)be[j] - mean * sc);
        if (silu) y = y / (1.0f + __expf(-y));
        o[j] = (_Float16)y;
      }
      *reinterpret_cast<half4v*>(out + ((long long)b * hw + p) * C + c) = o;
    }
  }
}

// y = silu(x * scale + shift): blockIdx.y = sample, tx = 8-channel chunk, ty = pixel lane -> no integer division
constexpr int GN_APPLY_PIX = 64;     // pixels per block
static int gn_pix() { return GN_APPLY_PIX; }
__global__ __launch_bounds__(256) void gn_apply_kernel(const _Float16* __restrict__ x0, const _Float16* __restrict__ x1,
                                                       int c0, int c1, int hw, const float* __restrict__ affine, int silu,
                                                       _Float16* __restrict__ out, int pix) {
  const int C = c0 + c1, c8 = C / 8;
  const int b = blockIdx.y;
  const int nx = min(c8, 256), ny = 256 / nx;
  const int tx = threadIdx.x % nx, ty = threadIdx.x / nx;
  if (ty >= ny) return;
  const int p0 = blockIdx.x * pix, p1 = min(hw, p0 + pix);
  for (int ch = tx; ch < c8; ch += nx) {
    const float4* ap = reinterpret_cast<const float4*>(affine + ((long long)b * C + ch * 8) * 2);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = ap[k];
      sc[2 * k] = v.x; sh[2 * k] = v.y; sc[2 * k + 1] = v.z; sh[2 * k + 1] = v.w;
    }
    // four pixels per trip, and the NEXT trip's loads are issued before this trip's arithmetic (exp + rcp per element: 8.5 us of
    // transcendental issue per 65536 x 320 tensor next to the 12.5 us a plain copy of it takes): one trip in flight per thread left
    // the kernel at 3.2-3.5 TB/s with memory idle during the SiLU arithmetic
    const int step = 4 * ny;
    half8 cur[4], nxt[4];
    auto load4 = [&](half8 (&v)[4], int p) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + u * ny;
        v[u] = load8(x0, x1, c0, c1, (long long)b * hw + (pp < p1 ? pp : p), ch * 8);
      }
    };
    int p = p0 + ty;
    if (p < p1) load4(cur, p);
    for (; p < p1; p += step) {
      const bool more = p + step < p1;
      if (more) load4(nxt, p + step);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + u * ny;
        if (pp >= p1) break;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = fmaf((float)cur[u][j], sc[j], sh[j]);
          if (silu) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
          o[j] = (_Float16)y;
        }
        *reinterpret_cast<half8*>(out + ((long long)b * hw + pp) * C + ch * 8) = o;
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
      }
    }
  }
}

// R rows per wave (R = 4 for C <= 512, 2 for C <= 1024, else 1): all loads of the R rows are issued before the first is
// used, so a wave keeps R x 16 B per lane in flight instead of one (C = 320 fills only 40 of the 64 lanes with one chunk each)
template <int R, int CH>
__global__ __launch_bounds__(256) void layernorm_kernel(const _Float16* __restrict__ x, long long rows, int C, float eps,
                                                        const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                                        _Float16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int c8 = C / 8;
  half8 v[R][CH];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long row = row0 + r < rows ? row0 + r : row0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int ch = lane + i * 64;
      if (ch < c8) v[r][i] = *reinterpret_cast<const half8*>(x + row * C + ch * 8);
    }
  }
  half8 ga[CH], be[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int ch = lane + i * 64;
    if (ch < c8) {
      ga[i] = *reinterpret_cast<const half8*>(gamma + ch * 8);
      be[i] = *reinterpret_cast<const half8*>(beta + ch * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (lane + i * 64 < c8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[r][i][j];
      }
    const float mean = wave_sum(s) / C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (lane + i * 64 < c8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)v[r][i][j] - mean;
          q += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int ch = lane + i * 64;
      if (ch < c8) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)(((float)v[r][i][j] - mean) * rstd * (float)ga[i][j] + (float)be[i][j]);
        *reinterpret_cast<half8*>(out + (row0 + r) * C + ch * 8) = o;
      }
    }
  }
}

// The same arithmetic with a row owned by a GROUP of LPR lanes (LPR = C / 40: 8 / 16 / 32 for the UNet's C = 320 / 640 / 1280), five
// 16-byte chunks per lane: every lane of the wave carries data (the wave-per-row form above fills 40 of 64 lanes at C = 320),
// the two reductions take log2(LPR) exchange steps instead of six, and a wave keeps U x 5 loads per lane in flight.
// 21.3 -> ~13 us at 65536 x 320, where a plain copy of the tensor takes 12.5 us (profiles/r02_notes.md section 13).
template <int LPR, int U>
__global__ __launch_bounds__(256) void layernorm_group_kernel(const _Float16* __restrict__ x, long long rows, int C, float eps,
                                                              const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                                              _Float16* __restrict__ out) {
  constexpr int CPL = 5, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (U * RPW) + grp;
  half8 v[U][CPL];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long row = row0 + u * RPW < rows ? row0 + u * RPW : rows - 1;
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[u][j] = *reinterpret_cast<const half8*>(x + row * C + (j * LPR + sub) * 8);
  }
  half8 ga[CPL], be[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    ga[j] = *reinterpret_cast<const half8*>(gamma + (j * LPR + sub) * 8);
    be[j] = *reinterpret_cast<const half8*>(beta + (j * LPR + sub) * 8);
  }
  const float inv_c = 1.0f / (float)C;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[u][j][e];
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[u][j][e] - mean;
        q += d * d;
      }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    const float rstd = rsqrtf(q * inv_c + eps);
    const long long row = row0 + u * RPW;
    if (row < rows) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)v[u][j][e] - mean) * rstd * (float)ga[j][e] + (float)be[j][e]);
        *reinterpret_cast<half8*>(out + row * C + (j * LPR + sub) * 8) = o;
      }
    }
  }
}

// in-place softmax(scale * x) over each row of fp16 [rows, n] (block per row)
__global__ __launch_bounds__(256) void softmax_kernel(_Float16* __restrict__ x, int n, int ld, float scale) {
  __shared__ float red[4];
  _Float16* row = x + (long long)blockIdx.x * ld;
  float m = -__builtin_inff();
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, (float)row[i] * scale);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += __expf((float)row[i] * scale - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = threadIdx.x; i < n; i += 256) row[i] = (_Float16)(__expf((float)row[i] * scale - m) * inv);
}

}  // namespace sd

using namespace sd;

// scratch layout (floats): [batch*C*2] affine table, then [batch*nchunk*groups*2] partial sums
extern "C" int sd_groupnorm_f16(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                                const void* gamma, const void* beta, int silu, void* out, float* stats, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_GN;
    r.p[0] = (void*)x0; r.p[1] = (void*)x1; r.p[2] = (void*)gamma; r.p[3] = (void*)beta; r.p[4] = out; r.p[5] = stats;
    r.i[0] = c0; r.i[1] = c1; r.i[2] = batch; r.i[3] = hw; r.i[4] = groups; r.i[5] = silu; r.f[0] = eps;
    return sd::plan_record(r);
  }
  if (!x0 || !gamma || !beta || !out || !stats) return fail(COMA_E_INVALID, "sd_groupnorm_f16: null pointer");
  if (c1 > 0 && !x1) return fail(COMA_E_INVALID, "sd_groupnorm_f16: x1 missing");
  const int C = c0 + c1;
  if (batch <= 0 || hw <= 0 || groups <= 0 || groups > GN_MAX_GROUPS || C % groups || c0 % 8 || c1 % 8 || C > GN_MAX_C)
    return fail(COMA_E_INVALID, "sd_groupnorm_f16: bad shape C=%d groups=%d", C, groups);
  hipStream_t s = (hipStream_t)stream;
  const int cg = C / groups;
  if (cg % 4 == 0 && c0 % 4 == 0 && (long long)hw * (cg / 4) <= GN_SMALL_ITEMS * 256) {
    hipLaunchKernelGGL(gn_small_kernel, dim3(groups, batch), dim3(256), 0, s, (const _Float16*)x0, (const _Float16*)x1, c0, c1, hw,
                       groups, eps, (const _Float16*)gamma, (const _Float16*)beta, silu, (_Float16*)out);
    return check_launch("gn_small_kernel");
  }
  const int nchunk = (hw + GN_PIX - 1) / GN_PIX;
  float* partial = stats + (size_t)batch * C * 2;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, batch), dim3(256), 0, s, (const _Float16*)x0, (const _Float16*)x1, c0, c1,
                     hw, groups, partial);
  const int total = batch * groups;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, s, partial, nchunk, groups, C,
                     (float)hw * (float)(C / groups), eps, (const _Float16*)gamma, (const _Float16*)beta, stats, total);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((hw + gn_pix() - 1) / gn_pix(), batch), dim3(256), 0, s, (const _Float16*)x0,
                     (const _Float16*)x1, c0, c1, hw, stats, silu, (_Float16*)out, gn_pix());
  return check_launch("groupnorm kernels");
}

extern "C" int sd_layernorm_f16(const void* x, int64_t rows, int c, float eps, const void* gamma, const void* beta,
                                void* out, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_LN;
    r.p[0] = (void*)x; r.p[1] = (void*)gamma; r.p[2] = (void*)beta; r.p[3] = out; r.i[0] = rows; r.i[1] = c; r.f[0] = eps;
    return sd::plan_record(r);
  }
  if (!x || !gamma || !beta || !out) return fail(COMA_E_INVALID, "sd_layernorm_f16: null pointer");
  if (rows <= 0 || c <= 0 || c % 8 || c > 2048) return fail(COMA_E_INVALID, "sd_layernorm_f16: bad shape c=%d", c);
#define SD_LN_LAUNCH(R_, CH_)                                                                                                  \
  hipLaunchKernelGGL((layernorm_kernel<R_, CH_>), dim3((unsigned)((rows + 4 * R_ - 1) / (4 * R_))), dim3(256), 0, (hipStream_t)stream, \
                     (const _Float16*)x, (long long)rows, c, eps, (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)out)
#define SD_LN_GROUP(LPR_, U_)                                                                                                   \
  hipLaunchKernelGGL((layernorm_group_kernel<LPR_, U_>), dim3((unsigned)((rows + 4 * U_ * (64 / LPR_) - 1) / (4 * U_ * (64 / LPR_)))),  \
                     dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (long long)rows, c, eps, (const _Float16*)gamma,    \
                     (const _Float16*)beta, (_Float16*)out)
  // (measured: two row groups per wave pay from 32768 rows at C = 320; at C = 640 one group is as fast, and the 4096 x 1280 tensors
  // of the 16 x 16 level are launch / latency bound either way: 8.4 us for the wave-per-row form, 8.6 for this one)
  if (c == 320) { if (rows >= 32768) SD_LN_GROUP(8, 2); else SD_LN_GROUP(8, 1); }
  else if (c == 640) { if (rows >= 65536) SD_LN_GROUP(16, 2); else SD_LN_GROUP(16, 1); }
  else if (c == 1280 && rows >= 16384) { if (rows >= 32768) SD_LN_GROUP(32, 2); else SD_LN_GROUP(32, 1); }
  else if (c <= 512) SD_LN_LAUNCH(4, 1);
  else if (c <= 1024) SD_LN_LAUNCH(2, 2);
  else SD_LN_LAUNCH(1, 4);
#undef SD_LN_LAUNCH
#undef SD_LN_GROUP
  return check_launch("layernorm_kernel");
}

extern "C" int sd_softmax_f16(void* x, int64_t rows, int n, int ld, float scale, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_SOFTMAX;
    r.p[0] = x; r.i[0] = rows; r.i[1] = n; r.i[2] = ld; r.f[0] = scale;
    return sd::plan_record(r);
  }
  if (!x) return fail(COMA_E_INVALID, "sd_softmax_f16: null pointer");
  if (rows <= 0 || n <= 0 || ld < n) return fail(COMA_E_INVALID, "sd_softmax_f16: bad shape");
  hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (_Float16*)x, n, ld, scale);
  return check_launch("softmax_kernel");
}

extern "C" int sd_groupnorm_colstats_f16(const void* x0, const void* x1, int c0, int c1, int batch, int hw, int groups, float eps,
                                         const void* gamma, const void* beta, int silu, void* out, float* stats,
                                         const float* colstats0, const float* colstats1, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_GN_COLSTATS;
    r.p[0] = (void*)x0; r.p[1] = (void*)x1; r.p[2] = (void*)gamma; r.p[3] = (void*)beta; r.p[4] = out; r.p[5] = stats;
    r.p[6] = (void*)colstats0; r.p[7] = (void*)colstats1;
    r.i[0] = c0; r.i[1] = c1; r.i[2] = batch; r.i[3] = hw; r.i[4] = groups; r.i[5] = silu; r.f[0] = eps;
    return sd::plan_record(r);
  }
  if (!x0 || !gamma || !beta || !out || !stats || !colstats0) return fail(COMA_E_INVALID, "sd_groupnorm_colstats_f16: null pointer");
  if (c1 > 0 && (!x1 || !colstats1)) return fail(COMA_E_INVALID, "sd_groupnorm_colstats_f16: second source incomplete");
  const int C = c0 + c1;
  if (batch <= 0 || hw <= 0 || hw % 32 || groups <= 0 || groups > GN_MAX_GROUPS || C % groups || c0 % 8 || c1 % 8)
    return fail(COMA_E_INVALID, "sd_groupnorm_colstats_f16: bad shape C=%d groups=%d hw=%d", C, groups, hw);
  hipStream_t s = (hipStream_t)stream;
  const int total = batch * groups;
  if (C / groups > 256) return fail(COMA_E_INVALID, "sd_groupnorm_colstats_f16: more than 256 channels per group");
  hipLaunchKernelGGL(gn_finalize_colstats_kernel, dim3(total), dim3(256), 0, s, colstats0, colstats1, c0, c1, hw, hw / 32, groups, eps,
                     (const _Float16*)gamma, (const _Float16*)beta, stats);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((hw + gn_pix() - 1) / gn_pix(), batch), dim3(256), 0, s, (const _Float16*)x0,
                     (const _Float16*)x1, c0, c1, hw, stats, silu, (_Float16*)out, gn_pix());
  return check_launch("groupnorm (colstats) kernels");
}

// Only the per-(sample, channel) affine table (scale, shift) of a GroupNorm -- fp32 [batch][C][2] at the start of `stats` -- from the
// producer's column sums (colstats0 != NULL) or from a statistics pass over the tensor; no apply pass.  For consumers that apply the
// affine themselves (sd_xfront_f16).
extern "C" int sd_groupnorm_table_f16(const void* x0, int c0, int batch, int hw, int groups, float eps, const void* gamma, const void* beta,
                                      float* stats, const float* colstats0, int rows_per_slot, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_GN_TABLE;
    r.p[0] = (void*)x0; r.p[1] = (void*)gamma; r.p[2] = (void*)beta; r.p[3] = stats; r.p[4] = (void*)colstats0;
    r.i[0] = c0; r.i[1] = batch; r.i[2] = hw; r.i[3] = groups; r.i[4] = rows_per_slot; r.f[0] = eps;
    return sd::plan_record(r);
  }
  if (!x0 || !gamma || !beta || !stats) return fail(COMA_E_INVALID, "sd_groupnorm_table_f16: null pointer");
  if (rows_per_slot == 0) rows_per_slot = 32;
  const int C = c0;
  if (batch <= 0 || hw <= 0 || groups <= 0 || groups > GN_MAX_GROUPS || C % groups || c0 % 8 || C > GN_MAX_C || C / groups > 256)
    return fail(COMA_E_INVALID, "sd_groupnorm_table_f16: bad shape C=%d groups=%d", C, groups);
  hipStream_t s = (hipStream_t)stream;
  const int total = batch * groups;
  if (colstats0) {
    if (rows_per_slot < 32 || hw % rows_per_slot) return fail(COMA_E_INVALID, "sd_groupnorm_table_f16: colstats need hw %% rows_per_slot == 0 (hw=%d, rows_per_slot=%d)", hw, rows_per_slot);
    hipLaunchKernelGGL(gn_finalize_colstats_kernel, dim3(total), dim3(256), 0, s, colstats0, (const float*)nullptr, c0, 0, hw, hw / rows_per_slot, groups, eps,
                       (const _Float16*)gamma, (const _Float16*)beta, stats);
  } else {
    const int nchunk = (hw + GN_PIX - 1) / GN_PIX;
    float* partial = stats + (size_t)batch * C * 2;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, batch), dim3(256), 0, s, (const _Float16*)x0, (const _Float16*)nullptr, c0, 0, hw, groups,
                       partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, s, partial, nchunk, groups, C, (float)hw * (float)(C / groups), eps,
                       (const _Float16*)gamma, (const _Float16*)beta, stats, total);
  }
  return check_launch("groupnorm table kernels");
}

// The affine table of a GroupNorm over the channel concatenation of TWO tensors, from the column sums their producers left (no pass over
// the tensors): fp32 [batch][c0 + c1][2] at the start of `stats`.  For consumers that apply the affine themselves
// (sd_winograd_input_f16 with gn_affine: norm1 of the UNet's up-block ResNets, whose input is [hidden | skip]).
extern "C" int sd_groupnorm_table_cat_f16(int c0, int c1, int batch, int hw, int groups, float eps, const void* gamma, const void* beta, float* stats,
                                          const float* colstats0, const float* colstats1, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_GN_TABLE_CAT;
    r.p[0] = (void*)gamma; r.p[1] = (void*)beta; r.p[2] = stats; r.p[3] = (void*)colstats0; r.p[4] = (void*)colstats1;
    r.i[0] = c0; r.i[1] = c1; r.i[2] = batch; r.i[3] = hw; r.i[4] = groups; r.f[0] = eps;
    return sd::plan_record(r);
  }
  if (!gamma || !beta || !stats || !colstats0 || (c1 > 0 && !colstats1)) return fail(COMA_E_INVALID, "sd_groupnorm_table_cat_f16: null pointer");
  const int C = c0 + c1;
  if (batch <= 0 || hw <= 0 || hw % 32 || groups <= 0 || groups > GN_MAX_GROUPS || c0 <= 0 || c1 < 0 || C % groups || c0 % 8 || c1 % 8 || C > GN_MAX_C ||
      C / groups > 256)
    return fail(COMA_E_INVALID, "sd_groupnorm_table_cat_f16: bad shape c0=%d c1=%d groups=%d hw=%d", c0, c1, groups, hw);
  hipLaunchKernelGGL(gn_finalize_colstats_kernel, dim3(batch * groups), dim3(256), 0, (hipStream_t)stream, colstats0, colstats1, c0, c1, hw, hw / 32, groups, eps,
                     (const _Float16*)gamma, (const _Float16*)beta, stats);
  return check_launch("gn_finalize_colstats_kernel");
}
