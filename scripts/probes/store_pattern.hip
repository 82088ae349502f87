// How much does the row-segment width of a wave's stores / loads matter?  gfx950 probe.
// A [rows][320] fp16 tensor is written (or read and written) by waves whose every instruction covers
//   mode 0: 32 rows x 64 B   (the GEMM epilogue: one 32-column tile, 4 lanes x 16 B per row)
//   mode 1: 16 rows x 128 B  (two tiles side by side)
//   mode 2:  8 rows x 256 B
//   mode 3: fully linear 1 KiB
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int C = 320;
__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ in, _Float16* __restrict__ out, int rows, int mode, int rd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block = 256 rows x 320 columns (a 256 x 320 GEMM tile); 4 waves x 64 rows each
  const long long r0 = (long long)blockIdx.x * 256 + wave * 64;
  const int lpr = mode == 0 ? 4 : mode == 1 ? 8 : mode == 2 ? 16 : 40;   // lanes per row segment
  if (mode == 3) {
    for (int i = lane; i < 64 * 40; i += 64) {
      const long long off = (r0 * C) + (long long)i * 8;
      half8 v = {1, 2, 3, 4, 5, 6, 7, 8};
      if (rd) v = *reinterpret_cast<const half8*>(in + off);
      *reinterpret_cast<half8*>(out + off) = v;
    }
    return;
  }
  const int rows_per = 64 / lpr, segs = 40 / lpr;   // column segments of lpr*8 columns
  for (int s = 0; s < segs; ++s)
    for (int rr = 0; rr < 64; rr += rows_per) {
      const long long row = r0 + rr + lane / lpr;
      const long long off = row * C + s * lpr * 8 + (lane % lpr) * 8;
      half8 v = {1, 2, 3, 4, 5, 6, 7, 8};
      if (rd) v = *reinterpret_cast<const half8*>(in + off);
      *reinterpret_cast<half8*>(out + off) = v;
    }
}
int main() {
  const int rows = 65536;
  _Float16 *a, *b;
  (void)hipMalloc(&a, (size_t)rows * C * 2); (void)hipMalloc(&b, (size_t)rows * C * 2);
  (void)hipMemset(a, 0, (size_t)rows * C * 2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rd = 0; rd < 2; ++rd)
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(rows / 256), dim3(256), 0, 0, a, b, rows, mode, rd);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5 && ms < best) best = ms;
      }
      const double mb = (double)rows * C * 2 * (1 + rd) / 1e6;
      printf("%s segment %4d B: %7.1f us  %6.2f TB/s\n", rd ? "read+write" : "write     ", mode == 3 ? 1024 : (mode == 0 ? 64 : mode == 1 ? 128 : 256), best * 1e3, mb / best / 1e6 * 1e3);
    }
  return 0;
}
