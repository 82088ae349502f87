// sd_elementwise.hip -- the small elementwise pieces of the inpainting loop (gfx950): CFG + DDIM step + next-input
// assembly in one pass, sinusoidal timestep embedding, boundary layout conversions, image -> uint8.
// replaces: utils/adaptive_mask_inpainting.py:990-996, :1010-1017, :1111-1115 and diffusers' DDIMScheduler.step /
// get_timestep_embedding / VaeImageProcessor.postprocess (third party, diffusers==0.20.2).
#include <hip/hip_fp16.h>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

// one thread per (b, pixel)
__global__ void cfg_ddim_kernel(const _Float16* __restrict__ eps_uc, int eps_ld, float* __restrict__ latents,
                                float* __restrict__ x0_out, const _Float16* __restrict__ mask,
                                const _Float16* __restrict__ masked, _Float16* __restrict__ unet_in, int batch, int hw,
                                float guidance, float sa_t, float s1ma_t, float sa_p, float s1ma_p, int write_latents) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)batch * hw) return;
  float x[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float xt = latents[i * 4 + c];
    if (eps_uc) {
      float eu = (float)eps_uc[i * eps_ld + c];
      float ec = (float)eps_uc[(i + (long long)batch * hw) * eps_ld + c];
      float e = eu + guidance * (ec - eu);
      float x0 = (xt - s1ma_t * e) / sa_t;
      if (x0_out) x0_out[i * 4 + c] = x0;
      xt = sa_p * x0 + s1ma_p * e;
      if (write_latents) latents[i * 4 + c] = xt;
    }
    x[c] = xt;
  }
  if (unet_in) {
    // channels: latents(4) | mask(1) | masked_image_latents(4) | zero pad to 64; identical for the two CFG halves
    _Float16 v[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) v[c] = (_Float16)0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (_Float16)x[c];
    v[4] = mask[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[5 + c] = masked[i * 4 + c];
    uint4* o0 = reinterpret_cast<uint4*>(unet_in + i * 64);
    uint4* o1 = reinterpret_cast<uint4*>(unet_in + (i + (long long)batch * hw) * 64);
    const uint4* src = reinterpret_cast<const uint4*>(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      o0[q] = src[q];
      o1[q] = src[q];
    }
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int batch, int dim, _Float16* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * dim) return;
  int b = i / dim, j = i - b * dim, half = dim / 2;
  int k = j < half ? j : j - half;
  float freq = __expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
  float a = t[b] * freq;
  out[i] = (_Float16)(j < half ? cosf(a) : sinf(a));                    // flip_sin_to_cos: [cos | sin]
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int batch, int c, int hw, int cpad,
                                    _Float16* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)batch * hw * cpad) return;
  int ch = (int)(i % cpad);
  long long bp = i / cpad;
  int p = (int)(bp % hw);
  int b = (int)(bp / hw);
  out[i] = ch < c ? (_Float16)x[((long long)b * c + ch) * hw + p] : (_Float16)0.0f;
}

__global__ void nhwc_to_nchw_kernel(const _Float16* __restrict__ x, int batch, int c, int hw, int ld,
                                    float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)batch * c * hw) return;
  int p = (int)(i % hw);
  long long bc = i / hw;
  int ch = (int)(bc % c);
  int b = (int)(bc / c);
  out[i] = (float)x[((long long)b * hw + p) * ld + ch];
}

__global__ void image_to_u8_kernel(const _Float16* __restrict__ x, long long npix, int ld, int round_mode,
                                   uint8_t* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * 3) return;
  long long p = i / 3;
  int c = (int)(i - p * 3);
  float v = (float)x[p * ld + c] * 0.5f + 0.5f;
  v = fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f;
  out[i] = (uint8_t)(round_mode ? rintf(v) : v);   // astype(uint8) truncates; PIL path rounds
}

}  // namespace sd

using namespace sd;

extern "C" int sd_cfg_ddim_step(const void* eps_uc, int eps_ld, float* latents, float* x0_out, const void* mask,
                                const void* masked_latents, void* unet_in, int batch, int hw, float guidance, float alpha_t,
                                float alpha_prev, int write_latents, void* stream) {
  if (!latents) return fail(COMA_E_INVALID, "sd_cfg_ddim_step: null latents");
  if (unet_in && (!mask || !masked_latents)) return fail(COMA_E_INVALID, "sd_cfg_ddim_step: mask inputs missing");
  if (batch <= 0 || hw <= 0) return fail(COMA_E_INVALID, "sd_cfg_ddim_step: bad sizes");
  if (eps_uc && (!(alpha_t > 0.f) || alpha_t > 1.f || alpha_prev < 0.f || alpha_prev > 1.f))
    return fail(COMA_E_INVALID, "sd_cfg_ddim_step: alphas out of range");
  long long n = (long long)batch * hw;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)eps_uc, eps_ld, latents, x0_out, (const _Float16*)mask, (const _Float16*)masked_latents,
                     (_Float16*)unet_in, batch, hw, guidance, sqrtf(alpha_t), sqrtf(1.0f - alpha_t), sqrtf(alpha_prev),
                     sqrtf(1.0f - alpha_prev), write_latents);
  return check_launch("cfg_ddim_kernel");
}

extern "C" int sd_timestep_embedding_f16(const float* timesteps, int batch, int dim, void* out, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_TEMB;
    r.p[0] = (void*)timesteps; r.p[1] = out; r.i[0] = batch; r.i[1] = dim;
    return sd::plan_record(r);
  }
  if (!timesteps || !out || batch <= 0 || dim <= 0 || dim % 2) return fail(COMA_E_INVALID, "sd_timestep_embedding_f16: bad args");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((batch * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, timesteps,
                     batch, dim, (_Float16*)out);
  return check_launch("timestep_embedding_kernel");
}

extern "C" int sd_nchw_to_nhwc_f16(const float* x, int batch, int c, int hw, int cpad, void* out, void* stream) {
  if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || cpad < c) return fail(COMA_E_INVALID, "sd_nchw_to_nhwc_f16: bad args");
  long long n = (long long)batch * hw * cpad;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, batch, c, hw,
                     cpad, (_Float16*)out);
  return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int sd_nhwc_to_nchw_f32(const void* x, int batch, int c, int hw, int ld, float* out, void* stream) {
  if (!x || !out || batch <= 0 || c <= 0 || hw <= 0 || ld < c) return fail(COMA_E_INVALID, "sd_nhwc_to_nchw_f32: bad args");
  long long n = (long long)batch * hw * c;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, batch, c, hw, ld, out);
  return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int sd_image_to_u8(const void* x, int batch, int hw, int ld, int round_mode, uint8_t* out, void* stream) {
  if (!x || !out || batch <= 0 || hw <= 0 || ld < 3) return fail(COMA_E_INVALID, "sd_image_to_u8: bad args");
  long long npix = (long long)batch * hw;
  hipLaunchKernelGGL(image_to_u8_kernel, dim3((unsigned)((npix * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, npix, ld, round_mode, out);
  return check_launch("image_to_u8_kernel");
}

// ------------------------------------------------------------------------------------------------
// VAE latent sampling + scaling, and the mask glue of the adaptive loop
// ------------------------------------------------------------------------------------------------
namespace sd {

// moments: NHWC fp16 [B*hw, ld] = [mean(4) | logvar(4) | pad];  latent = (mean + exp(0.5*clamp(logvar,-30,20))*noise)*scale
// replaces: DiagonalGaussianDistribution.sample (diffusers) * vae.config.scaling_factor,
//           utils/adaptive_mask_inpainting.py:675-684.
__global__ void vae_sample_kernel(const _Float16* __restrict__ mom, int ld, const float* __restrict__ noise, float scale,
                                  long long n, float* __restrict__ lat32, _Float16* __restrict__ lat16) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  long long p = i >> 2;
  int c = (int)(i & 3);
  float mean = (float)mom[p * ld + c];
  float logvar = fminf(fmaxf((float)mom[p * ld + 4 + c], -30.0f), 20.0f);
  float v = (mean + __expf(0.5f * logvar) * (noise ? noise[i] : 0.0f)) * scale;
  if (lat32) lat32[i] = v;
  if (lat16) lat16[i] = (_Float16)v;
}

// x = sqrt(a) * x0 + sqrt(1-a) * noise   (DDIMScheduler.add_noise, used when strength < 1)
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, float sa, float s1ma, long long n,
                                 float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sa * x0[i] + s1ma * noise[i];
}

// Mask adaptation on device (K12): given the segmentation mask (u8 [H,W], non-zero = human) and the default box mask,
//   m = AND( dilate_{3x3, iters}(seg), default )          cv2.dilate with a 3x3 ones kernel, `iters` iterations
//                                                          == max over a (2*iters+1)^2 window, zero border
// and the products the pipeline needs from it:
//   mask_full  u8  [H,W]       (0/1)
//   mask_lat   f16 [H/8*W/8]   nearest-neighbour down-sample (F.interpolate, mode="nearest": source pixel (8y, 8x))
//   masked_img f16 NHWC [H*W, cpad]  image * (mask < 0.5), image given as fp32 NCHW in [-1,1]
// replaces: utils/adaptive_mask_inpainting.py:1136-1137 (dilate, logical_and), :131-245 (binarise, masked image),
//           :686-694 (mask interpolate).
__global__ void mask_adapt_kernel(const uint8_t* __restrict__ seg, const uint8_t* __restrict__ dflt, int H, int W, int iters,
                                  int use_default, const float* __restrict__ image, int cpad, uint8_t* __restrict__ mask_full,
                                  _Float16* __restrict__ mask_lat, _Float16* __restrict__ masked_img) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int m;
  if (use_default) {
    m = dflt[y * W + x] != 0;
  } else {
    m = 0;
    const int y0 = max(0, y - iters), y1 = min(H - 1, y + iters), x0 = max(0, x - iters), x1 = min(W - 1, x + iters);
    for (int yy = y0; yy <= y1 && !m; ++yy)
      for (int xx = x0; xx <= x1; ++xx)
        if (seg[yy * W + xx]) { m = 1; break; }
    m = m && (dflt[y * W + x] != 0);
  }
  mask_full[y * W + x] = (uint8_t)m;
  if ((y & 7) == 0 && (x & 7) == 0) mask_lat[(y >> 3) * (W >> 3) + (x >> 3)] = (_Float16)(float)m;
  const long long p = (long long)y * W + x;
  for (int c = 0; c < cpad; ++c)
    masked_img[p * cpad + c] = c < 3 ? (_Float16)(m ? 0.0f : image[(long long)c * H * W + p]) : (_Float16)0.0f;
}


// ---- batched form (B images per call, area test on the device: the adaptive loop never syncs with the host)
// pass 1: horizontal (2k+1) box max of seg into `rowmax`, and area[b] += sum(seg[b]) (the VALUE sum, as `mask.sum()` at :1132)
__global__ void mask_rowdilate_kernel(const uint8_t* __restrict__ seg, int H, int W, int iters, uint8_t* __restrict__ rowmax,
                                      int* __restrict__ area) {
  extern __shared__ uint8_t row[];
  const int y = blockIdx.x, b = blockIdx.y;
  const uint8_t* src = seg + ((long long)b * H + y) * W;
  int part = 0;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    uint8_t v = src[x];
    row[x] = v;
    part += v;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0 && part) atomicAdd(area + b, part);
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const int x0 = max(0, x - iters), x1 = min(W - 1, x + iters);
    int m = 0;
    for (int xx = x0; xx <= x1; ++xx) m |= row[xx];
    rowmax[((long long)b * H + y) * W + x] = (uint8_t)(m != 0);
  }
}

// pass 2: vertical (2k+1) max over `rowmax`, AND with the default mask (or the default mask itself when the whole call is forced
// to it or the image's segmentation is smaller than area_thres), and the three products.  The masked image leaves as ONE 16-byte
// store per pixel (channels 0-7: 3 data + 5 zeros); channels >= 8 are rewritten as zeros only when write_pad is set.
__global__ void mask_finish_kernel(const uint8_t* __restrict__ rowmax, const uint8_t* __restrict__ dflt, int H, int W, int iters,
                                   int force_default, double area_thres, const int* __restrict__ area,
                                   const float* __restrict__ image, int cpad, int write_pad, uint8_t* __restrict__ mask_full,
                                   _Float16* __restrict__ mask_lat, _Float16* __restrict__ masked_img) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const long long p = (long long)y * W + x, img = (long long)b * H * W;
  const bool use_default = force_default || (double)area[b] < area_thres;
  int m = dflt[img + p] != 0;
  if (!use_default && m) {
    const int y0 = max(0, y - iters), y1 = min(H - 1, y + iters);
    int any = 0;
    for (int yy = y0; yy <= y1; ++yy) any |= rowmax[img + (long long)yy * W + x];
    m = any != 0;
  }
  mask_full[img + p] = (uint8_t)m;
  if ((y & 7) == 0 && (x & 7) == 0) mask_lat[(long long)b * (H >> 3) * (W >> 3) + (y >> 3) * (W >> 3) + (x >> 3)] = (_Float16)(float)m;
  union { uint4 q; _Float16 h[8]; } v;
  v.q = make_uint4(0u, 0u, 0u, 0u);
  if (!m) {
    const float* ip = image + img * 3 + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) v.h[c] = (_Float16)ip[(long long)c * H * W];
  }
  uint4* o = reinterpret_cast<uint4*>(masked_img + (img + p) * cpad);
  o[0] = v.q;
  if (write_pad)
    for (int q = 1; q < cpad / 8; ++q) o[q] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace sd

extern "C" int sd_vae_sample(const void* moments, int ld, const float* noise, float scale, int64_t npix, float* latents_f32,
                             void* latents_f16, void* stream) {
  if (!moments || npix <= 0 || ld < 8 || (!latents_f32 && !latents_f16)) return sd::fail(COMA_E_INVALID, "sd_vae_sample: bad args");
  hipLaunchKernelGGL(sd::vae_sample_kernel, dim3((unsigned)((npix * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)moments, ld, noise, scale, (long long)npix, latents_f32, (_Float16*)latents_f16);
  return sd::check_launch("vae_sample_kernel");
}

extern "C" int sd_add_noise(const float* x0, const float* noise, float alpha, int64_t n, float* out, void* stream) {
  if (!x0 || !noise || !out || n <= 0 || !(alpha > 0.f) || alpha > 1.f) return sd::fail(COMA_E_INVALID, "sd_add_noise: bad args");
  hipLaunchKernelGGL(sd::add_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, noise,
                     sqrtf(alpha), sqrtf(1.0f - alpha), (long long)n, out);
  return sd::check_launch("add_noise_kernel");
}

extern "C" int sd_mask_adapt(const uint8_t* seg, const uint8_t* default_mask, int H, int W, int dilate_iters, int use_default,
                             const float* image_nchw, int cpad, uint8_t* mask_full, void* mask_latent, void* masked_image,
                             void* stream) {
  if (!default_mask || !image_nchw || !mask_full || !mask_latent || !masked_image || (!use_default && !seg))
    return sd::fail(COMA_E_INVALID, "sd_mask_adapt: null pointer");
  if (H <= 0 || W <= 0 || (H & 7) || (W & 7) || dilate_iters < 0 || cpad < 3)
    return sd::fail(COMA_E_INVALID, "sd_mask_adapt: bad sizes");
  hipLaunchKernelGGL(sd::mask_adapt_kernel, dim3((W + 127) / 128, H), dim3(128), 0, (hipStream_t)stream, seg, default_mask, H, W,
                     dilate_iters, use_default, image_nchw, cpad, mask_full, (_Float16*)mask_latent, (_Float16*)masked_image);
  return sd::check_launch("mask_adapt_kernel");
}

extern "C" int sd_mask_adapt_batched(const uint8_t* seg, const uint8_t* default_mask, int batch, int H, int W, int dilate_iters,
                                     int force_default, double area_thres, const float* image_nchw, int cpad, int write_pad,
                                     uint8_t* mask_full, void* mask_latent, void* masked_image, int32_t* area, uint8_t* scratch,
                                     void* stream) {
  if (!default_mask || !image_nchw || !mask_full || !mask_latent || !masked_image || !area || (!force_default && (!seg || !scratch)))
    return sd::fail(COMA_E_INVALID, "sd_mask_adapt_batched: null pointer");
  if (batch <= 0 || batch > 65535 || H <= 0 || H > 65535 || W <= 0 || W > 16384 || (H & 7) || (W & 7) || dilate_iters < 0 || cpad < 8 ||
      (cpad & 7))
    return sd::fail(COMA_E_INVALID, "sd_mask_adapt_batched: bad sizes (H, W multiples of 8; cpad a multiple of 8)");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(area, 0, sizeof(int32_t) * (size_t)batch, st) != hipSuccess)
    return sd::fail(COMA_E_LAUNCH, "sd_mask_adapt_batched: memset failed");
  if (!force_default)
    hipLaunchKernelGGL(sd::mask_rowdilate_kernel, dim3(H, batch), dim3(256), (size_t)W, st, seg, H, W, dilate_iters, scratch, area);
  hipLaunchKernelGGL(sd::mask_finish_kernel, dim3((W + 127) / 128, H, batch), dim3(128), 0, st, scratch, default_mask, H, W,
                     dilate_iters, force_default, area_thres, area, image_nchw, cpad, write_pad, mask_full,
                     (_Float16*)mask_latent, (_Float16*)masked_image);
  return sd::check_launch("mask_finish_kernel");
}
