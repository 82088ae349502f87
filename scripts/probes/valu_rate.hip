// Throughput probe: cycles per wave-instruction for v_exp_f32, v_pk_fma_f32, v_max3_f32, v_cvt_pkrtz on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed, int iters) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-6f + i; b[i] = a[i] * 0.5f; c[i] = a[i] * 0.25f; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (MODE == 1) a[i] = fmaf(a[i], 1.0001f, 0.5f);
      if (MODE == 2) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), seed);
      if (MODE == 3) { asm volatile("v_exp_f16 %0, %0" : "+v"(a[i])); }
      if (MODE == 4) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); }
      if (MODE == 5) { a[i] = __builtin_amdgcn_exp2f(a[i]); b[i] = fmaf(b[i], 1.0001f, 0.5f); c[i] = fmaf(c[i], 1.0002f, 0.25f); b[i] = fmaf(b[i], 0.9999f, c[i]); }
      if (MODE == 6) { b[i] = fmaf(b[i], 1.0001f, 0.5f); c[i] = fmaf(c[i], 1.0002f, 0.25f); b[i] = fmaf(b[i], 0.9999f, c[i]); }
    }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i] + c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
  float* d; hipMalloc(&d, 4 << 20);
  const char* names[] = {"v_exp_f32", "v_fma_f32", "v_max3_f32", "v_exp_f16", "v_rcp_f32", "exp+3fma", "3fma"};
  for (int waves = 4; waves <= 8; waves *= 2)
  for (int m = 0; m < 7; ++m) {
    const int iters = 4096; float h;
    dim3 g(256 * waves);   // 256 blocks x 4 waves: one wave per SIMD (x waves)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      if (m == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 4) hipLaunchKernelGGL(k<4>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 5) hipLaunchKernelGGL(k<5>, g, dim3(256), 0, 0, d, 0.001f, iters);
      if (m == 6) hipLaunchKernelGGL(k<6>, g, dim3(256), 0, 0, d, 0.001f, iters);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-12s blocks/CU=%d: %.2f clock64 ticks per wave-instruction ; wall: %.2f cycles@2.4GHz per wave-instruction per SIMD\n", names[m], waves, h / (iters * 8.0), ms * 1e-3 * 2.4e9 / ((double)waves * iters * 8.0));
  }
  return 0;
}
