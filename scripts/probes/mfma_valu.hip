// Do MFMA (wave A) and VALU (wave B) streams on the same SIMD overlap?  gfx950 probe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  const int prio = mode >> 8;            // 1: VALU/exp wave runs at priority 3; 2: MFMA wave runs at priority 3
  mode &= 255;
  const bool do_mfma = (mode & 1) && wave < 4, do_valu = (mode & 2) && wave >= 4, do_exp = (mode & 4) && wave >= 4;
  if (prio == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);
  float16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f); b[j] = (_Float16)0.5f; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);    // 32 fma = 128 cycles = 4 MFMAs
    }
  }
  if (do_exp) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);    // 8 exp = 128 cycles
    }
  }
  if (mode == 8 || mode == 9) {     // same wave: 1 MFMA followed by 8 (mode 8) / 4 (mode 9) independent fma, all 8 waves
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (mode == 8 || j < 4) v[j] = fmaf(v[j], 1.0001f, 0.5f);
      }
    }
  }
  if (mode == 10) {   // all 8 waves MFMA only
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  if (mode == 11) {   // all 8 waves fma only (32 per iteration)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 512 * 256 * 4);
  const char* names[] = {"", "mfma only", "fma only", "mfma + fma", "exp only", "mfma + exp", "", "", "1mfma:8fma x8w", "1mfma:4fma x8w", "mfma x8w", "32fma x8w"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {1, 2, 3, 256 + 3, 512 + 3, 4, 5, 256 + 5, 512 + 5, 8, 9, 10, 11}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, 20000);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-16s %s %.3f ms\n", names[mode & 255], (mode >> 8) == 1 ? "[VALU wave prio 3]" : (mode >> 8) == 2 ? "[MFMA wave prio 3]" : "", ms);
  }
  return 0;
}
