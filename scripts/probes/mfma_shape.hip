// Which MFMA shape delivers more under the package power cap?  Register-resident operands (random fp16), 8 waves per CU,
// every CU busy for ~0.5 s; prints TFLOP/s for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16.  gfx950 probe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k32(const _Float16* src, float* out, int iters) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = *(const half8*)(src + (threadIdx.x * 8 + i) * 8); b[i] = *(const half8*)(src + (threadIdx.x * 8 + 4 + i) * 8); }
  float16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k16(const _Float16* src, float* out, int iters) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = *(const half8*)(src + (threadIdx.x * 8 + i) * 8); b[i] = *(const half8*)(src + (threadIdx.x * 8 + 4 + i) * 8); }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  _Float16* src; float* out;
  (void)hipMalloc(&src, 512 * 64 * 2 * 2); (void)hipMalloc(&out, 512 * 512 * 4);
  _Float16 h[512 * 64 * 2];
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 8) % 2001 - 1000) * 0.001f); }
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1000000;
  for (int rep = 0; rep < 6; ++rep)
    for (int mode = rep / 3; mode <= rep / 3; ++mode) {
      (void)hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(k32, dim3(512), dim3(512), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(k16, dim3(512), dim3(512), 0, 0, src, out, iters);
      (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops = mode == 0 ? 512.0 * 8 * iters * 4 * 32768 : 512.0 * 8 * iters * 8 * 16384;
      printf("%s: %.1f ms  %.0f TFLOP/s\n", mode == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9);
    }
  return 0;
}
