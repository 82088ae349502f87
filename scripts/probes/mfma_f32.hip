// Practical fp32 matrix peak under the package power cap: v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 with register-resident random
// operands, 4 / 8 waves per CU, every CU busy for ~0.3 s each; prints TFLOP/s (nominal peak 157.3).  gfx950 probe for seg_gemm.hip's roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k32(const float* src, float* out, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[threadIdx.x * 8 + i]; b[i] = src[threadIdx.x * 8 + 4 + i]; }
  float16v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k16(const float* src, float* out, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[threadIdx.x * 8 + i]; b[i] = src[threadIdx.x * 8 + 4 + i]; }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* src; float* out;
  (void)hipMalloc(&src, 512 * 8 * 4); (void)hipMalloc(&out, 512 * 512 * 4);
  float h[512 * 8];
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 0.001f; }
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400000;
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
      (void)hipEventRecord(e0, 0);
      double per_wave_iter;
      const char* name;
      int threads = 512;
      if (mode == 0) { hipLaunchKernelGGL(k32<4>, dim3(512), dim3(512), 0, 0, src, out, iters); per_wave_iter = 4 * 4096.0; name = "32x32x2 f32, 4 acc chains, 16 waves/CU"; }
      else if (mode == 1) { threads = 256; hipLaunchKernelGGL(k32<4>, dim3(512), dim3(256), 0, 0, src, out, iters); per_wave_iter = 4 * 4096.0; name = "32x32x2 f32, 4 acc chains, 8 waves/CU"; }
      else if (mode == 2) { threads = 256; hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, src, out, iters); per_wave_iter = 4 * 4096.0; name = "32x32x2 f32, 4 acc chains, 4 waves/CU"; }
      else if (mode == 3) { threads = 256; hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, src, out, iters); per_wave_iter = 1 * 4096.0; name = "32x32x2 f32, ONE dependent chain, 4 waves/CU"; }
      else { hipLaunchKernelGGL(k16, dim3(512), dim3(512), 0, 0, src, out, iters); per_wave_iter = 8 * 2048.0; name = "16x16x4 f32, 8 acc chains, 16 waves/CU"; }
      (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const int grid = (mode == 2 || mode == 3) ? 256 : 512;
      const double flops = (double)grid * (threads / 64) * iters * per_wave_iter;
      printf("%-50s %.1f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    }
  return 0;
}
