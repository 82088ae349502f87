// How many VALU instructions hide in the issue shadow of an MFMA on gfx950 -- measured in SHADER CYCLES (s_memtime), so that the
// package power cap / clock does not enter.  r1's probe (mfma_valu.hip) timed bulk streams with wall-clock events on all 256 CUs
// and saw MFMA and VALU time add up; MI355X_MICROARCH.md says a hand-placed stream hides <= 5 single-issue VALU per 32-cycle gap.
//   mode 0: one wave per SIMD, per iteration 4 x { MFMA 32x32x16 ; k independent v_fma_f32 }          k = 0..10
//   mode 1: the same with v_exp_f32 fillers                                                              k = 0..6
//   mode 2: the same with v_pk_fma_f32 fillers
//   mode 3: two waves per SIMD: waves 0-3 MFMA only, waves 4-7 run k v_fma per "slot" -- cycles of EACH half
//   mode 4: 16x16x32 MFMA with k v_fma fillers
//   mode 5: k v_pk_fma_f16 fillers (r4: what a packed-fp16 exp2 polynomial would be made of)
//   mode 6: softmax slot as shipped, per MFMA 4 score values: 4 v_exp_f32 + 2 v_cvt_pk_f16_f32 (k ignored)
//   mode 7: the same 4 values with HALF of the exponentials as Cody-Waite + degree-3 polynomial in packed fp16:
//           2 v_exp_f32 + 1 cvt  |  1 cvt + 3 v_pk_add_f16 (magic-number round, fraction) + 3 v_pk_fma_f16 + v_pk_lshlrev_b16 + v_pk_add_u16
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_gap mfma_gap.hip ; run: ./mfma_gap [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int K>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  float16v acc[4];
  float4v acc4[4];
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  }
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)((threadIdx.x & 63) * 0.001f + j); b[j] = (_Float16)(0.5f + j * 0.01f); }
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.01f;
  f32x2 pv[6];
  for (int i = 0; i < 6; ++i) pv[i] = f32x2{threadIdx.x * 1e-3f, i * 0.1f};
  unsigned hv[12];
  for (int i = 0; i < 12; ++i) hv[i] = 0x3c003800u + threadIdx.x + i;
  unsigned hc1 = 0x3c013c01u + (threadIdx.x & 1), hc2 = 0x38003800u;
  float c1 = 1.0001f + threadIdx.x * 1e-9f, c2 = 0.5f;
  f32x2 pc1 = {c1, c1}, pc2 = {c2, c2};
  const bool mfma_wave = MODE != 3 || wave < 4;
  const bool valu_wave = MODE != 3 || wave >= 4;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (mfma_wave) {
        if constexpr (MODE == 4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == 6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[0]) : "v"(v[0]), "v"(v[1]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[1]) : "v"(v[2]), "v"(v[3]));
      } else if constexpr (MODE == 7) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[1]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[0]) : "v"(v[0]), "v"(v[1]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[1]) : "v"(v[2]), "v"(v[3]));
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(hv[2]) : "v"(hv[1]), "v"(hc1));          // t = h + 1536
        asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hv[3]) : "v"(hv[2]), "v"(hc1));   // n = t - 1536
        asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hv[4]) : "v"(hv[1]), "v"(hv[3]));  // f = h - n
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(hv[5]) : "v"(hv[4]), "v"(hc2), "v"(hc1));
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(hv[5]) : "v"(hv[5]), "v"(hv[4]), "v"(hc2));
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(hv[5]) : "v"(hv[5]), "v"(hv[4]), "v"(hc1));
        asm volatile("v_pk_lshlrev_b16 %0, 10, %1" : "=v"(hv[6]) : "v"(hv[2]));
        asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(hv[1]) : "v"(hv[5]), "v"(hv[6]));
      } else if (valu_wave) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          // asm volatile: the fillers stay exactly here, between this MFMA and the next
          if constexpr (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
          else if constexpr (MODE == 2) { if (j < 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j]) : "v"(pc1), "v"(pc2)); }
          else if constexpr (MODE == 5) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(hv[j]) : "v"(hc1), "v"(hc2));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
  for (int i = 0; i < 12; ++i) s += v[i];
  for (int i = 0; i < 6; ++i) s += pv[i].x + pv[i].y;
  for (int i = 0; i < 12; ++i) s += (float)hv[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int K>
void run(int blocks, int threads, float* d, long long* c) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(blocks), dim3(threads), 0, 0, d, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long h[8];
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  const double per = 1.0 / (iters * 4.0);
  if (MODE == 3) printf("mode 3 k=%2d blocks=%3d : MFMA wave %.1f cyc/slot, VALU wave %.1f cyc/slot   (%.3f ms)\n", K, blocks, h[0] * per, h[4] * per, ms);
  else printf("mode %d k=%2d blocks=%3d : %.1f cycles per MFMA(+fillers)   (%.3f ms, %.2f GHz)\n", MODE, K, blocks, h[0] * per, ms, h[0] / (ms * 1e6));
}

int main(int argc, char** argv) {
  float* d; long long* c;
  hipMalloc(&d, 512 * 256 * 4); hipMalloc(&c, 256 * 8 * 8);
  for (int blocks : {1, 256}) {
    run<0, 0>(blocks, 256, d, c); run<0, 1>(blocks, 256, d, c); run<0, 2>(blocks, 256, d, c); run<0, 3>(blocks, 256, d, c); run<0, 4>(blocks, 256, d, c);
    run<0, 5>(blocks, 256, d, c); run<0, 6>(blocks, 256, d, c); run<0, 8>(blocks, 256, d, c); run<0, 10>(blocks, 256, d, c); run<0, 12>(blocks, 256, d, c);
    run<1, 1>(blocks, 256, d, c); run<1, 2>(blocks, 256, d, c); run<1, 3>(blocks, 256, d, c); run<1, 4>(blocks, 256, d, c); run<1, 6>(blocks, 256, d, c);
    run<2, 2>(blocks, 256, d, c); run<2, 4>(blocks, 256, d, c); run<2, 6>(blocks, 256, d, c);
    run<3, 0>(blocks, 512, d, c); run<3, 4>(blocks, 512, d, c); run<3, 8>(blocks, 512, d, c); run<3, 12>(blocks, 512, d, c);
    run<5, 2>(blocks, 256, d, c); run<5, 4>(blocks, 256, d, c); run<5, 6>(blocks, 256, d, c); run<5, 8>(blocks, 256, d, c); run<5, 10>(blocks, 256, d, c);
    run<6, 0>(blocks, 256, d, c); run<7, 0>(blocks, 256, d, c);
    run<4, 0>(blocks, 256, d, c); run<4, 2>(blocks, 256, d, c); run<4, 3>(blocks, 256, d, c); run<4, 4>(blocks, 256, d, c);
  }
  return 0;
}
