// K1+K2+K3: fused contact / relative-orientation accumulator for gfx950 (MI355X).
//
// Reference op chain being replaced (per sample): utils/coma.py:284-291 (distance, count, proximity),
// :295-309 (two canonicalisations, utils/coma.py:123-172) and :312-323 (two geodesic-Gaussian soft
// histograms, utils/coma.py:102-112).  The reference streams ~10 passes over [H,O,N] f64 temporaries
// per sample; here a wave OWNS a tile of pairs, keeps both histograms of those pairs in registers
// across ALL samples of the call, and touches HBM once at the end (one coalesced read-modify-write of
// 2*N floats per pair).  The kernel is therefore FP32-VALU / transcendental bound, not HBM bound.
//
// Mapping (wave64):
//   * lanes <-> orientation bins: lane l holds bins {l, l+64, l+128, l+192} of a 256-bin chunk
//     (blockIdx.y walks chunks when N > 256); the bin vectors sit in 12 VGPRs for the whole kernel.
//   * a wave owns PT = 8 consecutive pairs (h,o) of the flattened [H*O] index -> 8 pairs x 2 grids x
//     4 bins = 64 accumulator VGPRs, output rows contiguous in memory.
//   * per chunk of SC = 8 samples the 64 lanes compute the per-(pair,sample) scalars (distance,
//     proximity, count, both canonical normals) one per lane, then the bin loop broadcasts them with
//     v_readlane into SGPRs.  Scalar work is thus ~1 % of the bin work instead of 47 %.
//   * weights are accumulated as 2^64 * w (the exponent bias rides in the FMA that forms the exp2
//     argument) so that contributions down to 2^-213 survive v_exp_f32's flush-to-zero and the sum
//     is rescaled once at the end: bins the reference keeps as f32 denormals are kept here too.
#include "common.h"

namespace coma {

constexpr int PT = 8;    // pairs per wave
constexpr int SC = 8;    // samples per scalar chunk (PT*SC == 64 lanes)
constexpr int NB = 4;    // bins per lane
constexpr int kBinsPerChunk = NB * kWave;
constexpr int kWavesPerBlock = 4;
constexpr float kPi = 3.14159265358979323846f;

struct ContactArgs {
  const float* hv;
  const float* hn;
  const float* ov;
  const float* on;
  int64_t obj_stride;   // floats between consecutive samples' object arrays (0 = shared)
  const float* grid;    // [N,3]
  int S, H, O, N;
  int64_t M;            // H*O
  float p[3], sp[3];    // normalised principle / sub-principle vectors
  float size, thres, eps;
  float cexp;           // -log2(e)/sigma^2
  float* P1;
  float* P2;
  float* nom;
  float* den;
  float* cnt;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }

// Two bin weights at once (same bin, the two histograms), written on float2 so that the Horner chain,
// the dot product and the exponent argument issue as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: a plain
// v_fma_f32 costs ~3 cycles per wave64 on gfx950, a packed one 4 cycles for two results.
//
//   w = 2^64 * exp(-acos(clip(g.c))^2 / sigma^2)                       reference: utils/coma.py:108-110
//   acos(|x|) = sqrt(1-|x|) * P(|x|), P = degree-7 interpolant of acos(x)/sqrt(1-x) at Chebyshev nodes of
//   [0,1] (|error| <= 2.9e-8 exact, 2.5e-7 evaluated in f32);  acos(x) = pi/2 + sign(x) * (acos(|x|) - pi/2).
//   sqrt(|1-|x||) stands in for the clip: |x| exceeds 1 by at most 2 ulp, which moves theta by < 5e-4 near
//   theta = 0 or pi where d w / d theta vanishes (relative effect on w < 1e-5).
__device__ __forceinline__ f32x2 bin_weight2(float gx, float gy, float gz, f32x2 cx, f32x2 cy, f32x2 cz,
                                             f32x2 cexp) {
  f32x2 x = pk_fma(splat(gz), cz, pk_fma(splat(gy), cy, splat(gx) * cx));
  f32x2 ax = {fabsf(x.x), fabsf(x.y)};
  f32x2 t = splat(1.0f) - ax;
  f32x2 r = splat(-0.001211737748235464f);
  r = pk_fma(r, ax, splat(0.006491521373391151f));
  r = pk_fma(r, ax, splat(-0.01684105210006237f));
  r = pk_fma(r, ax, splat(0.03072212263941765f));
  r = pk_fma(r, ax, splat(-0.05011430382728577f));
  r = pk_fma(r, ax, splat(0.08896885067224503f));
  r = pk_fma(r, ax, splat(-0.214598149061203f));
  r = pk_fma(r, ax, splat(1.570796251296997f));
  f32x2 s = {__builtin_amdgcn_sqrtf(fabsf(t.x)), __builtin_amdgcn_sqrtf(fabsf(t.y))};
  f32x2 u = pk_fma(s, r, splat(-0.5f * kPi));                       // acos(|x|) - pi/2
  f32x2 sg = {__builtin_copysignf(1.0f, x.x), __builtin_copysignf(1.0f, x.y)};
  f32x2 th = pk_fma(sg, u, splat(0.5f * kPi));
  f32x2 e = pk_fma(th * th, cexp, splat(64.0f));
  return f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
}

struct V3 {
  float x, y, z;
};

__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

__device__ __forceinline__ V3 unit(V3 v, float eps) {   // utils/transformations.py:14-17
  float n = sqrtf(dot3(v, v)) + eps;
  return {v.x / n, v.y / n, v.z / n};
}

// canonicalize_a_wrt_b_to_p for one (a,b) pair, f32, literal formula incl. the reference's incomplete
// skew matrix (utils/coma.py:149-155).  a,b must already be normalised.
__device__ __forceinline__ V3 canon(V3 a, V3 b, V3 p, V3 sp, float eps) {
  float c = dot3(b, p);
  float ab = dot3(a, b);
  float ap = dot3(a, p);
  float as = dot3(a, sp);
  V3 v = {(b.x * p.x + (-b.z) * p.y) + b.y * p.z, (b.z * p.x + 0.0f * p.y) + (-b.x) * p.z,
          ((-b.y) * p.x + 0.0f * p.y) + 0.0f * p.z};
  float av = dot3(a, v);
  V3 f;
  float opc = 1.0f + c;
  if (opc < eps) {
    f = {2.0f * as * sp.x - a.x, 2.0f * as * sp.y - a.y, 2.0f * as * sp.z - a.z};
  } else {
    f.x = ((v.x * av) / opc + c * a.x) + ab * p.x - ap * b.x;
    f.y = ((v.y * av) / opc + c * a.y) + ab * p.y - ap * b.y;
    f.z = ((v.z * av) / opc + c * a.z) + ab * p.z - ap * b.z;
  }
  float n = sqrtf(dot3(f, f));
  return {f.x / n, f.y / n, f.z / n};
}

__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(kWavesPerBlock* kWave) void contact_accumulate_kernel(ContactArgs A) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t pair0 = tile * PT;
  if (pair0 >= A.M) return;   // wave-uniform
  const int kbase = blockIdx.y * kBinsPerChunk;

  // this lane's bins
  float gx[NB], gy[NB], gz[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int k = kbase + b * kWave + lane;
    bool ok = k < A.N;
    gx[b] = ok ? A.grid[3 * k + 0] : 0.0f;
    gy[b] = ok ? A.grid[3 * k + 1] : 0.0f;
    gz[b] = ok ? A.grid[3 * k + 2] : 0.0f;
  }

  f32x2 acc[PT][NB];   // .x: human-wrt-object histogram, .y: object-wrt-human histogram
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[p][b] = splat(0.0f);
  const f32x2 cexp2 = splat(A.cexp);

  // scalar role of this lane: pair (lane & 7), sample slot (lane >> 3)
  const int64_t my_pair = pair0 + (lane & (PT - 1));
  const bool pair_ok = my_pair < A.M;
  const int h = pair_ok ? (int)(my_pair / A.O) : 0;
  const int o = pair_ok ? (int)(my_pair % A.O) : 0;
  const V3 p = {A.p[0], A.p[1], A.p[2]}, sp = {A.sp[0], A.sp[1], A.sp[2]};
  float nom_part = 0.0f, cnt_part = 0.0f;

  for (int s0 = 0; s0 < A.S; s0 += SC) {
    const int si = s0 + (lane >> 3);
    const bool ok = pair_ok && si < A.S;
    V3 c1 = {0.f, 0.f, 1.f}, c2 = {0.f, 0.f, 1.f};
    if (ok) {
      const float* hvp = A.hv + ((int64_t)si * A.H + h) * 3;
      const float* hnp = A.hn + ((int64_t)si * A.H + h) * 3;
      const float* ovp = A.ov + (int64_t)si * A.obj_stride + (int64_t)o * 3;
      const float* onp = A.on + (int64_t)si * A.obj_stride + (int64_t)o * 3;
      // K1 -- exact f32 sequence of utils/coma.py:284 (sub, square, (x+y)+z, correctly rounded sqrt)
      float dx = hvp[0] - ovp[0], dy = hvp[1] - ovp[1], dz = hvp[2] - ovp[2];
      float d = sqrtf((dx * dx + dy * dy) + dz * dz);
      cnt_part += (d < A.thres) ? 1.0f : 0.0f;
      nom_part += expf(-d / A.size);
      // K2
      V3 a = unit({hnp[0], hnp[1], hnp[2]}, A.eps);
      V3 b = unit({onp[0], onp[1], onp[2]}, A.eps);
      c1 = canon(a, b, p, sp, A.eps);   // human normal, object normal taken to p
      c2 = canon(b, a, p, sp, A.eps);   // object normal, human normal taken to p
    }
    const int ns = min(SC, A.S - s0);
    // K3
    for (int s = 0; s < ns; ++s) {
#pragma unroll
      for (int q = 0; q < PT; ++q) {
        const int j = s * PT + q;
        const f32x2 cx = {bcast(c1.x, j), bcast(c2.x, j)};
        const f32x2 cy = {bcast(c1.y, j), bcast(c2.y, j)};
        const f32x2 cz = {bcast(c1.z, j), bcast(c2.z, j)};
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[q][b] += bin_weight2(gx[b], gy[b], gz[b], cx, cy, cz, cexp2);
      }
    }
  }

  // histograms: one coalesced read-modify-write per (pair, grid, bin)
  constexpr float kUnscale = 5.421010862427522e-20f;   // 2^-64
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int64_t pair = pair0 + q;
    if (pair < A.M) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int k = kbase + b * kWave + lane;
        if (k < A.N) {
          const int64_t idx = pair * A.N + k;
          A.P1[idx] += acc[q][b].x * kUnscale;
          A.P2[idx] += acc[q][b].y * kUnscale;
        }
      }
    }
  }

  // per-pair scalars: fold the 8 sample slots (lane bits 3..5); bin chunk 0 owns the update
  if (blockIdx.y == 0) {
#pragma unroll
    for (int m = PT; m < kWave; m <<= 1) {
      nom_part += __shfl_xor(nom_part, m);
      cnt_part += __shfl_xor(cnt_part, m);
    }
    if (lane < PT && pair_ok) {
      A.nom[my_pair] += nom_part;
      A.cnt[my_pair] += cnt_part;
      A.den[my_pair] += (float)A.S;   // S additions of 1.0f, exact
    }
  }
}

}  // namespace coma

using namespace coma;

static void unit_host(const float* v, float eps, float* out) {
  // same f32 sequence as utils/transformations.py:14-17
  float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + eps;
  for (int i = 0; i < 3; ++i) out[i] = v[i] / n;
}

extern "C" int coma_contact_accumulate_f32(const float* human_verts, const float* human_normals,
                                           const float* obj_verts, const float* obj_normals,
                                           int64_t obj_sample_stride, const float* sphere_grid, int S,
                                           int H, int O, int N, const float* principle_vec,
                                           const float* sub_principle_vec, float spatial_grid_size,
                                           float spatial_grid_thres, float normal_gaussian_sigma,
                                           float eps, float* prob_h_wrt_o, float* prob_o_wrt_h,
                                           float* nom, float* den, float* cnt, void* stream) {
  if (!human_verts || !human_normals || !obj_verts || !obj_normals || !sphere_grid || !principle_vec ||
      !sub_principle_vec || !prob_h_wrt_o || !prob_o_wrt_h || !nom || !den || !cnt)
    return fail(COMA_E_INVALID, "coma_contact_accumulate_f32: null pointer");
  if (S < 0 || H <= 0 || O <= 0 || N <= 0)
    return fail(COMA_E_INVALID, "coma_contact_accumulate_f32: bad sizes S=%d H=%d O=%d N=%d", S, H, O, N);
  if (obj_sample_stride != 0 && obj_sample_stride != (int64_t)O * 3)
    return fail(COMA_E_INVALID, "coma_contact_accumulate_f32: obj_sample_stride must be 0 or 3*O");
  if (!(normal_gaussian_sigma > 0.f) || !(spatial_grid_size > 0.f))
    return fail(COMA_E_INVALID, "coma_contact_accumulate_f32: sigma and spatial_grid_size must be > 0");
  if (S == 0) return COMA_OK;

  ContactArgs A;
  A.hv = human_verts; A.hn = human_normals; A.ov = obj_verts; A.on = obj_normals;
  A.obj_stride = obj_sample_stride; A.grid = sphere_grid;
  A.S = S; A.H = H; A.O = O; A.N = N; A.M = (int64_t)H * O;
  unit_host(principle_vec, eps, A.p);
  unit_host(sub_principle_vec, eps, A.sp);
  A.size = spatial_grid_size; A.thres = spatial_grid_thres; A.eps = eps;
  A.cexp = (float)(-1.4426950408889634 / ((double)normal_gaussian_sigma * (double)normal_gaussian_sigma));
  A.P1 = prob_h_wrt_o; A.P2 = prob_o_wrt_h; A.nom = nom; A.den = den; A.cnt = cnt;

  const int64_t waves = (A.M + PT - 1) / PT;
  const int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 0x7fffffffLL) return fail(COMA_E_INVALID, "coma_contact_accumulate_f32: H*O too large");
  dim3 grid((unsigned)blocks, (unsigned)((N + kBinsPerChunk - 1) / kBinsPerChunk));
  hipLaunchKernelGGL(contact_accumulate_kernel, grid, dim3(kWavesPerBlock * kWave), 0,
                     (hipStream_t)stream, A);
  return check_launch("contact_accumulate_kernel");
}
