// sd_smallconv.hip -- 3x3 convolutions with a handful of channels on one side (gfx950): the two ends of the VAE.
//
// sd_conv3x3_small_n_f16: GroupNorm affine + SiLU folded into a 3x3 / stride 1 / pad 1 convolution with n <= 4 OUTPUT channels --
//   the VAE decoder's conv_norm_out -> SiLU -> conv_out (128 -> 3 at 512 x 512; self.vae.decode, utils/adaptive_mask_inpainting.py:1086,
//   :1112) and the UNet's (320 -> 4 at 64 x 64; self.unet(...), :1001-1007).  Through the implicit GEMM this layer computed a 64-column tile for 3 channels (0.45 ms per 8 images) behind a separate
//   GroupNorm pass that read and wrote the 0.5 GB tensor once more.  Here a workgroup owns a 16 x 16 pixel tile: the (18 x 18) x 64
//   channel halo patch is normalised, activated and rounded to fp16 on its way into LDS (what the GroupNorm kernel would have
//   stored), the weights of the chunk live in registers as 18 MFMA operands (n padded to 16 rows, 9 taps x 64 channels), and every
//   `v_mfma_f32_16x16x32_f16` takes its pixel operand with one conflict-free 16-byte LDS read.  Bound: HBM -- the input read once.
#include <hip/hip_fp16.h>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {
namespace sc {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kTile = 16, kHalo = 18, kChunk = 64;       // pixels per tile edge, with halo, channels staged per pass

// slot of 16-byte channel octet v (0..7) of halo pixel p: 16 consecutive pixels x one octet hit 16 distinct bank groups
__device__ __forceinline__ int slot(int p, int v) { return p * 8 + (v ^ ((p >> 1) & 7)); }

template <int C>
__global__ __launch_bounds__(256, 2) void conv3x3_small_n_kernel(const _Float16* __restrict__ x, const float* __restrict__ affine, int silu,
                                                              const _Float16* __restrict__ w, const _Float16* __restrict__ bias, int n_out,
                                                              int H, int W, _Float16* __restrict__ out, int ldo) {
  static_assert(C % kChunk == 0, "channels in chunks of 64");
  __shared__ half8 tile[kHalo * kHalo * 8];               // 41 472 bytes
  __shared__ float2 aff[C];
  const int b = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lo = lane >> 4;

  if (affine)
    for (int i = tid; i < C; i += 256) aff[i] = reinterpret_cast<const float2*>(affine)[(size_t)b * C + i];

  float4v acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = float4v{0.f, 0.f, 0.f, 0.f};

  // Staging is split in two so that no thread ever waits on one load at a time: `fetch` issues ALL of a chunk's global loads of this
  // thread into registers (11 independent 16-byte loads), `commit` activates them and writes LDS.  The second chunk is fetched before
  // the first chunk's MFMAs, so its HBM latency runs under them.
  constexpr int kVec = kHalo * kHalo * 8, kPer = (kVec + 255) / 256;      // 2592 vectors, 11 per thread
  half8 q[kPer];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int idx = tid + it * 256;
      const int p = idx >> 3, v = idx & 7;
      const int r = p / kHalo, c = p - r * kHalo;
      const int y = ty0 + r - 1, xx = tx0 + c - 1;
      half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (idx < kVec && y >= 0 && y < H && xx >= 0 && xx < W)
        z = *reinterpret_cast<const half8*>(x + (((size_t)b * H + y) * W + xx) * C + chunk * kChunk + v * 8);
      q[it] = z;
    }
  };
  auto commit = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int idx = tid + it * 256;
      if (idx >= kVec) break;
      const int p = idx >> 3, v = idx & 7;
      const int r = p / kHalo, c = p - r * kHalo;
      const int y = ty0 + r - 1, xx = tx0 + c - 1;
      half8 z = q[it];
      if (affine && y >= 0 && y < H && xx >= 0 && xx < W) {       // zero padding applies to the ACTIVATED tensor: pad pixels stay 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float2 a = aff[chunk * kChunk + v * 8 + e];
          float f = fmaf((float)z[e], a.x, a.y);
          if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));
          z[e] = (_Float16)f;
        }
      }
      tile[slot(p, v)] = z;
    }
  };

  // one chunk: commit the fetched patch, bring this chunk's weights into registers, (optionally) put the next chunk's loads in
  // flight, then 9 taps x 2 K-halves x 4 pixel rows of MFMAs
  auto chunk_body = [&](int chunk, bool prefetch_next) {
    commit(chunk);
    // this chunk's weights -> registers (18 MFMA operands): operand row = output channel (zero rows above n_out), K octet = lane >> 4
    half8 wf[9][kChunk / 32];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kh = 0; kh < kChunk / 32; ++kh) {
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lp < n_out) z = *reinterpret_cast<const half8*>(w + ((size_t)lp * 9 + tap) * C + chunk * kChunk + kh * 32 + lo * 8);
        wf[tap][kh] = z;
      }
    if (prefetch_next) fetch(chunk + 1);                   // in flight under this chunk's MFMAs
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int kh = 0; kh < kChunk / 32; ++kh) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int p = (4 * wave + t + dy) * kHalo + lp + dx;
          const half8 px = tile[slot(p, kh * 4 + lo)];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tap][kh], px, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                       // every wave is done reading the tile before the next commit overwrites it
  };
  if constexpr (C / kChunk <= 2) {
    // two chunks (the VAE's 128 channels): unrolled, the second chunk's HBM latency runs under the first chunk's MFMAs
    fetch(0);
    __syncthreads();                                       // `aff` is visible
#pragma unroll
    for (int chunk = 0; chunk < C / kChunk; ++chunk) chunk_body(chunk, chunk + 1 < C / kChunk);
  } else {
    // more chunks (the UNet's 320 channels): a rolled loop; holding the prefetched patch across the back edge next to the weights
    // spills, so the fetch sits in front of its commit and the second resident workgroup of the CU covers its latency
    __syncthreads();
#pragma unroll 1
    for (int chunk = 0; chunk < C / kChunk; ++chunk) {
      fetch(chunk);
      chunk_body(chunk, false);
    }
  }
  // D[n][pixel]: lane = pixel + 16 * (n / 4), register = n % 4 -> lanes 0..15 hold the n_out <= 4 real channels of their pixel
  if (lane < 16) {
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = (bias && j < n_out) ? (float)bias[j] : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int y = ty0 + 4 * wave + t, xx = tx0 + lp;
      if (y < H && xx < W) {
        half8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < n_out) o[j] = (_Float16)(acc[t][j] + bv[j]);
        *reinterpret_cast<half8*>(out + (((size_t)b * H + y) * W + xx) * ldo) = o;
      }
    }
  }
}


// ---- the other end: a 3x3 convolution with 3 INPUT channels (the VAE encoder's conv_in, 3 -> 128 at 512 x 512; self.vae.encode,
// utils/adaptive_mask_inpainting.py:677-680).  Through the implicit GEMM it multiplied a 64-channel padded input (K = 576 for 27 real
// products per output).  Here one elementwise pass packs every pixel's 3 x 3 x 3 neighbourhood into 32 halfs (k = 3 * tap + channel, 5
// zeros) and the convolution becomes a plain K = 32 product of the existing GEMM.
typedef _Float16 half4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void im2col3x3_c3_kernel(const _Float16* __restrict__ x, int ldx, int batch, int H, int W,
                                                          _Float16* __restrict__ out) {
  const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
  if (m >= (long long)batch * H * W) return;
  const int xx = (int)(m % W), y = (int)((m / W) % H);
  const long long b = m / ((long long)W * H);
  _Float16 v[32];
#pragma unroll
  for (int k = 27; k < 32; ++k) v[k] = (_Float16)0.0f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xc = xx + tap % 3 - 1;
    half4s q = {0, 0, 0, 0};
    if (yy >= 0 && yy < H && xc >= 0 && xc < W) q = *reinterpret_cast<const half4s*>(x + ((b * H + yy) * W + xc) * ldx);
    v[3 * tap + 0] = q[0]; v[3 * tap + 1] = q[1]; v[3 * tap + 2] = q[2];
  }
  half8* o = reinterpret_cast<half8*>(out + m * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    half8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[8 * j + e];
    o[j] = t;
  }
}


// ---- the same convolution (3 input channels -> 128, conv_in of the VAE encoder at 512 x 512) WITHOUT the packed copy: a workgroup owns a 16 x 16
// pixel tile, keeps the (18 x 18) x 3 halo patch in LDS as [row][3 col + channel] (a tap row's 9 values are then contiguous), and every wave builds the
// K = 32 operand of 16 pixels (k = 9 ky + 3 kx + channel = 3 tap + channel, 5 zeros) with eight 2-byte LDS reads per lane -- the im2col matrix
// (134 MB per 8 images, written and read back) never exists.  D = W . X^T on `v_mfma_f32_16x16x32_f16`: the weights are eight register operands, a lane
// owns pixel (lane & 15) and channels 16 nb + 4 (lane >> 4) + i.  The output leaves through a per-wave LDS staging tile as full 256-byte rows, and the
// tile's column sums / sums of squares of the STORED fp16 values go to one statistics slot per tile (the layout sd_conv3x3_halo_f16 writes and
// sd_groupnorm_table_f16(rows_per_slot = 256) reads): the first ResNet's GroupNorm no longer re-reads the 0.5 GB tensor for its statistics.
// Bound: HBM -- the output written once (537 MB per 8 images).
constexpr int kC3Row = 56;                                 // halves per patch row (18 x 3 = 54, padded)
constexpr int kC3Zero = kHalo * kC3Row;                    // one zero half behind the patch: the source of k = 27 .. 31
constexpr int kC3Stage = 136;                              // halves per staged pixel row (128 + 8: the 8-byte writes of 16 pixels x 4 groups spread over all banks)

// sum over the 16 lanes of a DPP row, in every lane (row rotations by 8, 4, 2, 1: a fixed order)
template <int CTRL>
__device__ __forceinline__ float dpp_ror_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) { return dpp_ror_add<0x121>(dpp_ror_add<0x122>(dpp_ror_add<0x124>(dpp_ror_add<0x128>(v)))); }

__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const _Float16* __restrict__ x, int ldx, const _Float16* __restrict__ w32,
                                                        const _Float16* __restrict__ bias, int H, int W, _Float16* __restrict__ out, int ldo,
                                                        float* __restrict__ colstats) {
  __shared__ _Float16 patch[kHalo * kC3Row + 8];
  __shared__ __attribute__((aligned(16))) _Float16 stage[4][16 * kC3Stage];
  __shared__ float wsum[4][2][128];
  const int b = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;

  // weights: operand nb of lane (n = lane & 15, g) = w32[16 nb + n][8 g .. 8 g + 7]; bias of this lane's output channels 16 nb + 4 g + i
  half8 wf[8];
  float bv[8][4];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    wf[nb] = *reinterpret_cast<const half8*>(w32 + (nb * 16 + px) * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[nb][i] = bias ? (float)bias[nb * 16 + 4 * g + i] : 0.0f;
  }
  // halo patch: 324 pixels, 3 halves each (zero outside the image)
  for (int p = tid; p < kHalo * kHalo; p += 256) {
    const int r = p / kHalo, c = p - r * kHalo;
    const int yy = ty0 + r - 1, xx = tx0 + c - 1;
    half4s q = {0, 0, 0, 0};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) q = *reinterpret_cast<const half4s*>(x + (((long long)b * H + yy) * W + xx) * ldx);
    _Float16* d = patch + r * kC3Row + c * 3;
    d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
  }
  if (tid < 8) patch[kC3Zero + tid] = (_Float16)0.0f;
  // where element e of this lane's operand (k = 8 g + e) sits relative to the patch row of the output pixel's row
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e, ky = k / 9, j = k - 9 * ky;
    koff[e] = k < 27 ? ky * kC3Row + px * 3 + j : -1;
  }
  __syncthreads();

  float cs[8][4], cq[8][4];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[nb][i] = cq[nb][i] = 0.0f;
  _Float16* const stg = stage[wave];
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int y = wave * 4 + r;                            // tile row = the 16 pixels of this MFMA
    half8 xf;
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[e] = patch[koff[e] >= 0 ? koff[e] + y * kC3Row : kC3Zero];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const float4v acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nb], xf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      typedef _Float16 half4v __attribute__((ext_vector_type(4)));
      half4v o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (_Float16)(acc[i] + bv[nb][i]);
        const float f = (float)o[i];                       // statistics of the stored (fp16-rounded) tensor
        cs[nb][i] += f;
        cq[nb][i] += f * f;
      }
      *reinterpret_cast<half4v*>(stg + px * kC3Stage + nb * 16 + 4 * g) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // 16 pixels x 256 bytes: lane -> (pixel 4 it + (lane >> 4), 16-byte chunk lane & 15): every store instruction covers four full rows
    _Float16* const orow = out + (((long long)b * H + ty0 + y) * W + tx0) * ldo;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pr = it * 4 + g;
      *reinterpret_cast<half8*>(orow + (long long)pr * ldo + px * 8) = *reinterpret_cast<const half8*>(stg + pr * kC3Stage + px * 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (colstats) {
    // fold the 16 pixel lanes of a row group (DPP row rotations inside 16 lanes, fixed order), then the four waves through LDS
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs[nb][i] = row_sum16(cs[nb][i]);
        cq[nb][i] = row_sum16(cq[nb][i]);
        if (px == 0) {
          wsum[wave][0][nb * 16 + 4 * g + i] = cs[nb][i];
          wsum[wave][1][nb * 16 + 4 * g + i] = cq[nb][i];
        }
      }
    __syncthreads();
    const int which = tid >> 7, c = tid & 127;             // 256 threads = [sum | sumsq][128 channels], waves added in wave order
    const float rsum = ((wsum[0][which][c] + wsum[1][which][c]) + wsum[2][which][c]) + wsum[3][which][c];
    const long long slot_id = ((long long)b * (H / kTile) + blockIdx.y) * (W / kTile) + blockIdx.x;
    colstats[(slot_id * 2 + which) * 128 + c] = rsum;
  }
}

}  // namespace sc
}  // namespace sd

extern "C" int sd_conv3x3_small_n_f16(const void* x, const float* gn_affine, int silu, const void* w, const void* bias, int batch, int h,
                                      int w_, int c, int n, void* out, int ldo, void* stream) {
  using namespace sd::sc;
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_CONV_SMALL_N;
    r.p[0] = (void*)x; r.p[1] = (void*)gn_affine; r.p[2] = (void*)w; r.p[3] = (void*)bias; r.p[4] = out;
    r.i[0] = silu; r.i[1] = batch; r.i[2] = h; r.i[3] = w_; r.i[4] = c; r.i[5] = n; r.i[6] = ldo;
    return sd::plan_record(r);
  }
  if (!x || !w || !out) return fail(COMA_E_INVALID, "sd_conv3x3_small_n_f16: null pointer");
  if (c != 128 && c != 320) return fail(COMA_E_INVALID, "sd_conv3x3_small_n_f16: c = %d (built for 128 and 320 input channels)", c);
  if (n < 1 || n > 4 || batch <= 0 || h <= 0 || w_ <= 0 || ldo < 8 || ldo % 8)
    return fail(COMA_E_INVALID, "sd_conv3x3_small_n_f16: bad shape n=%d batch=%d h=%d w=%d ldo=%d", n, batch, h, w_, ldo);
  if ((size_t)batch * h * w_ * c >= (1ull << 40)) return fail(COMA_E_INVALID, "sd_conv3x3_small_n_f16: tensor too large");
  const dim3 grid((unsigned)((w_ + kTile - 1) / kTile), (unsigned)((h + kTile - 1) / kTile), (unsigned)batch);
  if (c == 128)
    hipLaunchKernelGGL(conv3x3_small_n_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, gn_affine, silu,
                       (const _Float16*)w, (const _Float16*)bias, n, h, w_, (_Float16*)out, ldo);
  else
    hipLaunchKernelGGL(conv3x3_small_n_kernel<320>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, gn_affine, silu,
                       (const _Float16*)w, (const _Float16*)bias, n, h, w_, (_Float16*)out, ldo);
  return check_launch("conv3x3_small_n_kernel");
}

extern "C" int sd_im2col3x3_c3_f16(const void* x, int ldx, int batch, int h, int w_, void* out, void* stream) {
  using namespace sd::sc;
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_IM2COL_C3;
    r.p[0] = (void*)x; r.p[1] = out; r.i[0] = ldx; r.i[1] = batch; r.i[2] = h; r.i[3] = w_;
    return sd::plan_record(r);
  }
  if (!x || !out) return fail(COMA_E_INVALID, "sd_im2col3x3_c3_f16: null pointer");
  if (ldx < 4 || ldx % 4 || batch <= 0 || h <= 0 || w_ <= 0) return fail(COMA_E_INVALID, "sd_im2col3x3_c3_f16: bad shape ldx=%d batch=%d h=%d w=%d", ldx, batch, h, w_);
  const long long M = (long long)batch * h * w_;
  if (M > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_im2col3x3_c3_f16: too many pixels");
  hipLaunchKernelGGL(im2col3x3_c3_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, ldx, batch, h, w_,
                     (_Float16*)out);
  return check_launch("im2col3x3_c3_kernel");
}

extern "C" int sd_conv3x3_c3_f16(const void* x, int ldx, const void* w32, const void* bias, int batch, int h, int w_, int n, void* out, int ldo,
                                 float* colstats, void* stream) {
  using namespace sd::sc;
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_CONV_C3;
    r.p[0] = (void*)x; r.p[1] = (void*)w32; r.p[2] = (void*)bias; r.p[3] = out; r.p[4] = colstats;
    r.i[0] = ldx; r.i[1] = batch; r.i[2] = h; r.i[3] = w_; r.i[4] = n; r.i[5] = ldo;
    return sd::plan_record(r);
  }
  if (!x || !w32 || !out) return fail(COMA_E_INVALID, "sd_conv3x3_c3_f16: null pointer");
  if (n != 128) return fail(COMA_E_INVALID, "sd_conv3x3_c3_f16: n = %d (built for 128 output channels)", n);
  if (ldx < 4 || ldx % 4 || batch <= 0 || batch > 65535 || h <= 0 || w_ <= 0 || h % kTile || w_ % kTile || ldo < n || ldo % 8)
    return fail(COMA_E_INVALID, "sd_conv3x3_c3_f16: bad shape ldx=%d batch=%d h=%d w=%d ldo=%d (h, w multiples of 16; ldx %% 4 == 0; ldo %% 8 == 0)", ldx, batch, h, w_, ldo);
  dim3 grid((unsigned)(w_ / kTile), (unsigned)(h / kTile), (unsigned)batch);
  hipLaunchKernelGGL(conv3x3_c3_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, ldx, (const _Float16*)w32, (const _Float16*)bias, h, w_,
                     (_Float16*)out, ldo, colstats);
  return check_launch("conv3x3_c3_kernel");
}
