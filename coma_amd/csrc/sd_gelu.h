// Internal: the GELU of the GEGLU epilogues (sd_gemm.hip, sd_xtail.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sd {

// GELU (exact erf form, diffusers' GEGLU) without transcendentals: x * Phi(x), Phi(x) = 0.5 + xc * R(u) with xc = clamp(x, +-5),
// u = 2 xc^2 / 25 - 1 in [-1, 1] and R a degree-12 Chebyshev fit of (Phi(sqrt t) - 0.5) / sqrt t (coefficients <= 0.15 in
// magnitude: Horner in u is well conditioned in fp32).  |Phi error| <= 4e-7, |gelu error| <= 2e-6 for |x| <= 6 and <= 4e-7 |x|
// beyond (1 - Phi(5) = 2.9e-7): 1/30 of an fp16 ulp at |gelu| = 0.06.  Two values per instruction (v_pk_fma_f32): 17 packed
// operations per pair against 32 scalar ones + 4 quarter-rate transcendentals for the Abramowitz-Stegun erfc form used before
// (|error| 4e-7, relative in the negative tail) -- the VALU time of the GEGLU epilogue is not hidden behind anything, it was
// a third of the kernel (profiles/r02_notes.md section 11).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -5.0f, 5.0f), __builtin_amdgcn_fmed3f(x.y, -5.0f, 5.0f)};
  const f32x2 u = __builtin_elementwise_fma(xc * xc, (f32x2){0.08f, 0.08f}, (f32x2){-1.0f, -1.0f});
  constexpr float c[13] = {1.413638145e-01f, -7.029590756e-02f, 5.151792988e-02f, -4.045128077e-02f, 3.147675842e-02f,
                           -2.321312763e-02f, 1.623608917e-02f, -1.130712498e-02f, 6.766527425e-03f, -2.526916796e-03f,
                           1.374596148e-03f, -1.676730928e-03f, 7.353763795e-04f};
  f32x2 r = {c[12], c[12]};
#pragma unroll
  for (int k = 11; k >= 0; --k) r = __builtin_elementwise_fma(r, u, (f32x2){c[k], c[k]});
  return x * __builtin_elementwise_fma(xc, r, (f32x2){0.5f, 0.5f});
}

}  // namespace sd
