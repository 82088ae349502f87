// K5/K6: voxel occupancy of human vertices around object point 0 (gfx950).
//
// replaces: utils/coma_occupancy.py:287-295 (splat) and :297-312 (normalise + max over humans).
//
// The reference tests every one of the R^3 voxel centres against every vertex (a dense
// [H,3,R,R,R] f64 broadcast).  Only centres inside the threshold sphere (radius = scale_tolerance
// voxels, ~113 cells at tolerance 3) can pass, so the kernel tests just the bounding box of that
// sphere -- with the reference's exact f64 arithmetic per candidate, so counts are bit-identical:
//     centre_c = centers[c][i_c]                      (table built on the host like load_voxelgrid)
//     d = sqrt(((gx-qx)^2 + (gy-qy)^2) + (gz-qz)^2)   (f64, no FMA contraction: -ffp-contract=off)
//     counts[h,i,j,k] += (d < thres)
// HBM-bound scatter: one workgroup owns one human vertex (one [R,R,R] row of the output) and walks
// that vertex's samples, so all atomics of a workgroup land in one row that stays cache-resident at
// the shipped R=30 (108 KB); f32 atomic adds of 1.0 are exact and order-independent below 2^24.
#include "common.h"

namespace coma {

constexpr int kSplatWaves = 4;

__global__ __launch_bounds__(kSplatWaves* kWave) void occupancy_splat_kernel(
    const float* __restrict__ q, int S, int H, int R, const double* __restrict__ centers, double voxel,
    double thres, float* __restrict__ counts) {
  const int h = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  float* row = counts + (int64_t)h * R * R * R;
  const double inv = 1.0 / voxel;
  for (int s = wave; s < S; s += kSplatWaves) {
    const float* qp = q + ((int64_t)s * H + h) * 3;
    const double qc[3] = {(double)qp[0], (double)qp[1], (double)qp[2]};
    int lo[3], n[3];
    bool empty = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // conservative index range of centres within thres of q along axis c (0.01-voxel margin)
      double c0 = centers[c * R];
      int a = (int)floor((qc[c] - thres - c0) * inv - 0.01);
      int b = (int)ceil((qc[c] + thres - c0) * inv + 0.01);
      a = a < 0 ? 0 : a;
      b = b > R - 1 ? R - 1 : b;
      lo[c] = a;
      n[c] = b - a + 1;
      empty |= n[c] <= 0;
    }
    if (empty) continue;   // wave-uniform: vertex farther than thres from the whole grid
    const int total = n[0] * n[1] * n[2];
    for (int t = lane; t < total; t += kWave) {
      int iz = t % n[2];
      int r = t / n[2];
      int iy = r % n[1];
      int ix = r / n[1];
      ix += lo[0]; iy += lo[1]; iz += lo[2];
      double dx = centers[ix] - qc[0];
      double dy = centers[R + iy] - qc[1];
      double dz = centers[2 * R + iz] - qc[2];
      double d = sqrt((dx * dx + dy * dy) + dz * dz);
      if (d < thres) atomicAdd(row + ((int64_t)ix * R + iy) * R + iz, 1.0f);
    }
  }
}

// rowsum[h] = sum over the R^3 cells (integer-valued -> exact in any order below 2^24)
__global__ __launch_bounds__(256) void occupancy_rowsum_kernel(const float* __restrict__ counts, int64_t R3,
                                                               float* __restrict__ rowsum) {
  __shared__ float part[4];
  const float* row = counts + (int64_t)blockIdx.x * R3;
  float s = 0.0f;
  if ((R3 & 3) == 0) {                       // 16-byte loads, four independent partial sums
    const float4* row4 = reinterpret_cast<const float4*>(row);
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int64_t i = threadIdx.x; i < (R3 >> 2); i += 256) {
      const float4 v = row4[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    s = (a.x + a.y) + (a.z + a.w);
  } else {
    for (int64_t i = threadIdx.x; i < R3; i += 256) s += row[i];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rowsum[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// counts[h,i] /= rowsum[h] (IEEE division, 0/0 = NaN as in the reference); out[i] = max over selected h
__global__ __launch_bounds__(256) void occupancy_norm_max_kernel(float* __restrict__ counts,
                                                                 const uint8_t* __restrict__ select,
                                                                 const float* __restrict__ rowsum, int H,
                                                                 int64_t R3, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R3) return;
  float m = -__builtin_inff();
  for (int h = 0; h < H; ++h) {
    float v = counts[(int64_t)h * R3 + i] / rowsum[h];
    counts[(int64_t)h * R3 + i] = v;
    if (!select || select[h]) m = (v > m || v != v) ? v : m;
  }
  out[i] = m;
}

// the same with four cells per thread (R^3 % 4 == 0): 16-byte loads and stores
__global__ __launch_bounds__(256) void occupancy_norm_max4_kernel(float* __restrict__ counts,
                                                                  const uint8_t* __restrict__ select,
                                                                  const float* __restrict__ rowsum, int H,
                                                                  int64_t R3, float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= R3) return;
  float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int h = 0; h < H; ++h) {
    float4* p = reinterpret_cast<float4*>(counts + (int64_t)h * R3 + i);
    const float4 c = *p;
    const float r = rowsum[h];
    float v[4] = {c.x / r, c.y / r, c.z / r, c.w / r};
    *p = make_float4(v[0], v[1], v[2], v[3]);
    if (!select || select[h]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = (v[j] > m[j] || v[j] != v[j]) ? v[j] : m[j];
    }
  }
  *reinterpret_cast<float4*>(out + i) = make_float4(m[0], m[1], m[2], m[3]);
}


// ---------------------------------------------------------------------------------------------------------------------
// Structure B of SURVEY.md 8d: splat + normalise + max over humans in one pass, the grid written ONCE.
// replaces: utils/coma_occupancy.py:272-312 end to end (aggregate every cached sample into a zero grid, then
//           return_aggregated_spatial_grids): afterwards counts holds the NORMALISED grid exactly as the reference leaves it.
//   pass 1  occupancy_rowprep_kernel   one workgroup per human vertex: the number of hits of every sample (-> row sum, exact integer) and
//                                       the row's (sample, x-plane) incidences bucketed by plane, 16 bytes each
//   pass 2  occupancy_fused_kernel     workgroup = (slab of P x-planes, row group), all resident at once: for every row of the group the
//                                       slab lives in LDS as 16-bit counters (two buffers), the incidences of its bucket are tested a
//                                       y-row of 8 z-cells per lane -- the tests of row h + 1 between the store chunks of row h --, then
//                                       one sweep converts / normalises (quotient table) / stores the slab (1 KB per wave store) and
//                                       folds it into a running maximum each thread keeps in registers for its own cells
//   pass 3  occupancy_groupmax_kernel  NaN-propagating max over the row groups
// The distance test is the reference's: d = sqrt((dx^2 + dy^2) + dz^2) < thres in f64.  sqrt is correctly rounded, hence
// monotone, so "sqrt(x) < thres" is EXACTLY "x < T2" with T2 = the smallest double whose square root is >= thres (found on the
// host by stepping ulps) -- same bits, no f64 sqrt per candidate.  Where f32 can decide "x < T2" it does (error bound at the tests below);
// a candidate too close to T2 for that is evaluated with the f64 expression above.
constexpr int kFusedThreads = 512;
constexpr int kFusedWavesPerSimd = 4;                            // __launch_bounds__ below: <= 128 VGPRs, two workgroups of 8 waves per CU
constexpr int kFusedResident = 256 * (kFusedWavesPerSimd * 4 / (kFusedThreads / 64));   // workgroups the chip holds at once
constexpr int kFusedMaxChunks = 5;                               // 8-cell chunks per thread -> slabs of <= 20480 cells
constexpr int kFusedMaxCells = kFusedThreads * kFusedMaxChunks * 8;
constexpr int kFusedMaxWindow = 16;                              // candidate cells per axis the fused pass accepts (scale_tolerance <= 7)
constexpr size_t occ_align256(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int kFusedListCap = 512;                               // items of one (row, slab) staged per round (8 KB of LDS)

// per (row, sample): the position (f32, as given) and where its candidate window starts along y and z
// a (sample, x-plane) incidence as the fused pass stages it in LDS: the record + the plane
struct OccItem { float x, y, z; unsigned pack; };                // pack: lo_y | lo_z << 8 | plane << 16

__device__ __forceinline__ void axis_range(double qc, double thres, double c0, double inv, int R, int& lo, int& n) {
  int a = (int)floor((qc - thres - c0) * inv - 0.01);           // conservative (0.01-voxel margin), as in the splat kernel
  int b = (int)ceil((qc + thres - c0) * inv + 0.01);
  a = a < 0 ? 0 : a;
  b = b > R - 1 ? R - 1 : b;
  lo = a;
  n = b - a + 1;
}

// pass 1, one workgroup per human vertex (row): (i) the row sum -- every candidate of every sample tested once, an exact
// integer; (ii) the row's (sample, x-plane) incidences BUCKETED BY PLANE: histogram over planes in LDS, exclusive prefix ->
// plane_off[h][0..R], scatter of (sample | plane << 16) words into items[h][...], plus one 16-byte record per (row, sample).
// The fused pass then reads exactly the incidences of its slab (r2 scanned all
// S records of the row once per slab: 128 x 32 KB per row at R = 128, 5.4 GB of L2 reads per call, and compacted them with LDS
// atomics).  Order inside a bucket is whatever the atomics give; the counts do not depend on it.
__global__ __launch_bounds__(256) void occupancy_rowprep_kernel(const float* __restrict__ q, int S, int H, int R, int W,
                                                                const double* __restrict__ centers, double voxel, double thres,
                                                                double t2, OccItem* __restrict__ items,
                                                                unsigned* __restrict__ plane_off, float* __restrict__ rowsum) {
  __shared__ unsigned part[4];
  __shared__ unsigned hist[256], base[257];
  __shared__ double cen[3 * 256];                                 // per-axis centres: LDS, not three dependent global loads per test
  for (int i = threadIdx.x; i < 3 * R; i += 256) cen[i] = centers[i];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const int h = blockIdx.x;
  const double inv = 1.0 / voxel;
  unsigned hits = 0;
  const float t2_f = (float)t2, t2_band = (float)(t2 * 0x1p-18);
  const bool pre_ok = t2 > 1e-30 && t2 < 1e30;        // (the error bound above assumes neither under- nor overflow in f32)
  for (int s = threadIdx.x; s < S; s += 256) {
    const float* qp = q + ((int64_t)s * H + h) * 3;
    const double qc[3] = {(double)qp[0], (double)qp[1], (double)qp[2]};
    int lo[3], n[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) axis_range(qc[c], thres, cen[c * R], inv, R, lo[c], n[c]);
    if (n[0] <= 0 || n[1] <= 0 || n[2] <= 0) continue;
    // exact count of one x-plane: the reference's predicate, (dx*dx + dy*dy) + dz*dz < t2 in f64, every product rounded on its own
    auto plane_exact = [&](int ix) {
      const double dx = cen[ix] - qc[0];
      unsigned c = 0;
      for (int iy = lo[1]; iy < lo[1] + n[1]; ++iy) {
        const double dy = cen[R + iy] - qc[1];
        const double dxy = dx * dx + dy * dy;
        for (int iz = lo[2]; iz < lo[2] + n[2]; ++iz) {
          const double dz = cen[2 * R + iz] - qc[2];
          c += (dxy + dz * dz) < t2 ? 1u : 0u;
        }
      }
      return c;
    };
    if (W <= 8 && pre_ok) {
      // The usual case (a reach of up to 8 cells per axis) is decided in f32 wherever f32 can decide it.  The offsets are formed in f64 as
      // above and THEN rounded; with dy2 = fl(dyf^2), dzt = fl(fl(dzf^2) - fl(t2)) kept in registers (cells past the range carry +inf),
      // a cell costs d = fl(fl(dx2 + dy2) + dzt), its sign bit and a running minimum of |d| -- no compare-to-scalar, no dependent LDS
      // read (the plain loops ran at ~ 380 cycles per test per wave).  |d - (R - t2)| <= 2^-24 (6 R + 2 t2) for the exact squared
      // distance R (three squares of values with 2^-24 relative error, their product roundings, two sums, the rounding of t2 and of the
      // difference), and the f64 value of the predicate differs from R by 2^-51 R: below R = 2 t2 that is < 2^-20 t2, above it far less
      // than R - t2.  So sign(d) IS the predicate whenever |d| > 2^-18 t2; a plane holding a cell inside that band (about one plane in
      // 10^4 per lane) is recounted exactly.
      float dy2[8], dzt[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dy = (float)(cen[R + min(lo[1] + k, R - 1)] - qc[1]), dz = (float)(cen[2 * R + min(lo[2] + k, R - 1)] - qc[2]);
        dy2[k] = k < min(n[1], W) ? dy * dy : __builtin_inff();          // (a hit lies in the first W cells of a range: see below)
        dzt[k] = k < min(n[2], W) ? dz * dz - t2_f : __builtin_inff();
      }
      for (int ix = lo[0]; ix < lo[0] + n[0]; ++ix) {
        const float dx = (float)(cen[ix] - qc[0]);
        const float dx2 = dx * dx;
        unsigned cnt = 0;
        float near = __builtin_inff();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dxy = dx2 + dy2[j];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float d = dxy + dzt[k];
            cnt += __float_as_uint(d) >> 31;
            near = fminf(near, fabsf(d));
          }
        }
        hits += near > t2_band ? cnt : plane_exact(ix);
      }
    } else {
      for (int ix = lo[0]; ix < lo[0] + n[0]; ++ix) hits += plane_exact(ix);
    }
    // hits can only lie in the first W cells of a conservative range (it is at most one cell wider than W on its far side)
    for (int ix = lo[0]; ix < lo[0] + min(n[0], W); ++ix) atomicAdd(&hist[ix], 1u);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) hits += __shfl_xor(hits, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    rowsum[h] = (float)((part[0] + part[1]) + (part[2] + part[3]));   // exact below 2^24, like the f32 row sum
    unsigned acc = 0;
    for (int p = 0; p < R; ++p) { base[p] = acc; acc += hist[p]; }
    base[R] = acc;
  }
  __syncthreads();
  for (int p = threadIdx.x; p <= R; p += 256) plane_off[(int64_t)h * (R + 1) + p] = base[p];
  hist[threadIdx.x] = 0u;                                         // now the per-plane cursors
  __syncthreads();
  OccItem* row_items = items + (int64_t)h * S * W;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float* qp = q + ((int64_t)s * H + h) * 3;
    const float fx = qp[0], fy = qp[1], fz = qp[2];
    int lo[3], n[3];
    axis_range((double)fx, thres, cen[0], inv, R, lo[0], n[0]);
    axis_range((double)fy, thres, cen[R], inv, R, lo[1], n[1]);
    axis_range((double)fz, thres, cen[2 * R], inv, R, lo[2], n[2]);
    const bool empty = n[0] <= 0 || n[1] <= 0 || n[2] <= 0;
    if (empty) continue;
    // one complete 16-byte incidence per (sample, x-plane): the fused pass reads its slab's bucket with ONE coalesced load per item
    // (r3 first kept 4-byte (sample | plane) words + one record per (row, sample): a dependent gather, i.e. one more global latency per row)
    const unsigned yz = (unsigned)lo[1] | ((unsigned)lo[2] << 8);
    for (int ix = lo[0]; ix < lo[0] + min(n[0], W); ++ix) row_items[base[ix] + atomicAdd(&hist[ix], 1u)] = OccItem{fx, fy, fz, yz | ((unsigned)ix << 16)};
  }
}

// pass 2.  Counters are 16-bit halves of LDS words (a cell sees at most S < 65536 hits per row), so a 128 x 128 plane is 32 KB
// and three workgroups share a CU: one's candidate tests (VALU / LDS) run under another's slab stores (HBM).
__global__ __launch_bounds__(kFusedThreads, kFusedWavesPerSimd) void occupancy_fused_kernel(
    const OccItem* __restrict__ items, const unsigned* __restrict__ plane_off, const float* __restrict__ rowsum,
    const uint8_t* __restrict__ select, int S, int H, int R, int P, int W, int groups, int write_raw,
    const double* __restrict__ centers, double voxel, double thres, double t2, float* __restrict__ counts,
    float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int RR = R * R;
  const int x0 = blockIdx.x * P;
  const int np = min(P, R - x0);                                 // planes of this slab
  const int cells = np * RR;                                     // multiple of 4 whenever RR is (checked on the host)
  // two counter buffers: while the slab of row h is swept out of one, the candidate tests of row h + 1 run into the other
  unsigned* cnt_cur = reinterpret_cast<unsigned*>(smem);
  const size_t cnt_bytes = (((size_t)P * RR * 2 + 15) / 16) * 16;
  unsigned* cnt_nxt = reinterpret_cast<unsigned*>(smem + cnt_bytes);
  OccItem* list = reinterpret_cast<OccItem*>(smem + 2 * cnt_bytes);
  double* cen = reinterpret_cast<double*>(smem + 2 * cnt_bytes + (size_t)kFusedListCap * sizeof(OccItem));   // [3][R]
  // quotients count / rowsum of the row being swept for counts 0 .. 255 (nearly every cell): the sweep looked like a store loop but
  // was bound by its 32 correctly rounded f32 divisions per thread and row; one division per thread and a table read per cell instead
  float* qt = reinterpret_cast<float*>(smem + 2 * cnt_bytes + (size_t)kFusedListCap * sizeof(OccItem) + (size_t)3 * R * sizeof(double));
  for (int i = threadIdx.x; i < 3 * R; i += kFusedThreads) cen[i] = centers[i];
  const int g = blockIdx.y;
  const int h_lo = (int)((int64_t)H * g / groups), h_hi = (int)((int64_t)H * (g + 1) / groups);
  const int n4 = cells >> 2;                                     // quads of cells (cells % 4 == 0, checked on the host)
  float mx[kFusedMaxChunks][8];
#pragma unroll
  for (int k = 0; k < kFusedMaxChunks; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) mx[k][e] = -__builtin_inff();
  for (int i = threadIdx.x; i < 2 * (int)(cnt_bytes / 16); i += kFusedThreads) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  const int WW = W * W;
  const int w_shift = __builtin_ctz(W);
  const float t2_f = (float)t2, t2_band = (float)(t2 * 0x1p-18);      // f32 pre-decision of the candidate tests (see occupancy_rowprep_kernel)
  const bool fast = W == 8 && t2 > 1e-30 && t2 < 1e30;
  static_assert(kFusedListCap == kFusedThreads, "one incidence per thread and round");
  OccItem nxt = {0.f, 0.f, 0.f, 0u};
  unsigned o0 = 0, o1 = 0;                                       // [o0, o1): incidences of (row, slab) inside the row's list
  auto bounds = [&](int hh) {
    o0 = plane_off[(int64_t)hh * (R + 1) + x0];
    o1 = plane_off[(int64_t)hh * (R + 1) + x0 + np];
  };
  auto fetch = [&](int hh, unsigned i0) {
    const unsigned i = i0 + threadIdx.x;
    if (i < o1) {
      const float4 v = reinterpret_cast<const float4*>(items)[(int64_t)hh * S * W + i];
      nxt.x = v.x; nxt.y = v.y; nxt.z = v.z; nxt.pack = __float_as_uint(v.w);
    }
  };
  // Everything a row needs from global memory is requested one or two rows early -- the chain plane_off -> items is two dependent
  // loads, and with the store queues full each costs microseconds: (p0, p1) are the bounds of the row after the one (o0, o1) describe.
  unsigned p0 = 0, p1 = 0;
  auto bounds_ahead = [&](int hh) {
    p0 = plane_off[(int64_t)hh * (R + 1) + x0];
    p1 = plane_off[(int64_t)hh * (R + 1) + x0 + np];
  };
  // candidate tests of the staged incidences list[0 .. n_items) into counter buffer `c`, trips t = t0, t0 + dt, ... of kFusedThreads lanes
  auto tests = [&](unsigned* c, int n_items, int t0, int dt) {
    if (fast) {
      // usual window: a lane takes one (incidence, y-row) = 8 consecutive z-cells.  The x / y offsets are formed once, the predicate
      // is decided in f32 exactly as in the preparation pass (offsets in f64, then rounded; sign of fl(fl(dx2 + dy2) + fl(dz2 - t2))
      // unless it lies inside the 2^-18 t2 band, where the cell is re-evaluated in f64), and the hits of two cells that share a
      // counter word leave as ONE LDS atomic: 10 LDS reads and <= 5 atomics per 8 cells instead of 32 and <= 8.
      const int total = n_items << 3;
      for (int w = threadIdx.x + t0 * kFusedThreads; w < total; w += dt * kFusedThreads) {
        const OccItem m = list[w >> 3];
        const int iy = (int)(m.pack & 0xffu) + (w & 7), iz0 = (int)((m.pack >> 8) & 0xffu), ix = (int)(m.pack >> 16);
        if (iy >= R) continue;
        const double dx = cen[ix] - (double)m.x, dy = cen[R + iy] - (double)m.y;
        const float dxf = (float)dx, dyf = (float)dy;
        const float dxy = dxf * dxf + dyf * dyf;
        unsigned hm = 0, amb = 0;
        const double mz = (double)m.z;
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                              // branch-free: the eight LDS reads and chains overlap
          const float dzf = (float)(cen[2 * R + min(iz0 + k, R - 1)] - mz);
          const float d = dxy + (dzf * dzf - t2_f);
          hm |= (__float_as_uint(d) >> 31) << k;
          amb |= (fabsf(d) <= t2_band ? 1u : 0u) << k;
        }
        if (amb) {                                                                  // (rare) the reference's own arithmetic for those cells
          for (int k = 0; k < 8; ++k)
            if ((amb >> k) & 1u) {
              const double dz = cen[2 * R + min(iz0 + k, R - 1)] - mz;
              hm = (hm & ~(1u << k)) | ((((dx * dx + dy * dy) + dz * dz) < t2 ? 1u : 0u) << k);
            }
        }
        hm &= iz0 + 8 <= R ? 0xffu : (0xffu >> (iz0 + 8 - R));                      // cells past the grid edge
        if (hm) {
          const int cell = (ix - x0) * RR + iy * R + iz0;
          const unsigned hs = hm << (cell & 1);                                      // bit 2 j (+ 1): low (high) half of word j
          unsigned* wp = c + (cell >> 1);
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const unsigned v = ((hs >> (2 * j)) & 1u) | (((hs >> (2 * j + 1)) & 1u) << 16);
            if (v) atomicAdd(wp + j, v);
          }
        }
      }
    } else {
      // (incidence, window cell), every lane busy; W is a power of two (host), so the decode is shifts
      const int total = n_items << (2 * w_shift);
      for (int w = threadIdx.x + t0 * kFusedThreads; w < total; w += dt * kFusedThreads) {
        const OccItem m = list[w >> (2 * w_shift)];
        const int wc = w & (WW - 1);
        const int iy = (int)(m.pack & 0xffu) + (wc >> w_shift), iz = (int)((m.pack >> 8) & 0xffu) + (wc & (W - 1));
        if (iy < R && iz < R) {
          const int ix = (int)(m.pack >> 16);
          const double dx = cen[ix] - (double)m.x, dy = cen[R + iy] - (double)m.y, dz = cen[2 * R + iz] - (double)m.z;
          if (((dx * dx + dy * dy) + dz * dz) < t2) {
            const int cell = (ix - x0) * RR + iy * R + iz;
            atomicAdd(&c[cell >> 1], 1u << ((cell & 1) * 16));
          }
        }
      }
    }
  };
  // every round of row hh ([b0, b1), the first kFusedListCap incidences already in `nxt`) into `c`, nothing overlapped
  auto all_rounds = [&](unsigned* c, int hh, unsigned b0, unsigned b1) {
    for (unsigned i0 = b0; i0 < b1; i0 += kFusedListCap) {
      if (i0 > b0) fetch(hh, i0);
      const int n_items = (int)min((unsigned)kFusedListCap, b1 - i0);
      if ((int)threadIdx.x < n_items) list[threadIdx.x] = nxt;
      __syncthreads();
      tests(c, n_items, 0, 1);
      __syncthreads();
    }
  };
  float rs_next = 1.0f;
  bool sel_next = false;
  if (h_lo < h_hi) {
    bounds(h_lo);
    fetch(h_lo, o0);
    if (h_lo + 1 < h_hi) bounds_ahead(h_lo + 1);
    rs_next = rowsum[h_lo];                                      // row sum and selection flag: requested a row early as well
    sel_next = !select || select[h_lo];
  }
  __syncthreads();
  if (h_lo < h_hi) {
    all_rounds(cnt_cur, h_lo, o0, o1);                           // the first row has nothing to hide under
    if (h_lo + 1 < h_hi) { o0 = p0; o1 = p1; fetch(h_lo + 1, o0); }
    if (h_lo + 2 < h_hi) bounds_ahead(h_lo + 2);
  }
  // Invariant at the top of iteration h: cnt_cur holds the complete counts of row h and cnt_nxt is zero; (o0, o1) and `nxt` describe
  // row h + 1 (its bounds and its first incidences), (p0, p1) row h + 2.
  for (int h = h_lo; h < h_hi; ++h) {
    const bool have_next = h + 1 < h_hi;
    const unsigned nb0 = o0, nb1 = o1;
    // A row whose slab bucket fits one round (the usual case) is tested UNDER this row's sweep: a wave that has handed its stores to a
    // full memory pipeline stalls at the next store, so test trips are placed between the store chunks of the same wave -- the queue
    // drains while the wave computes.  (The sweep alone ran at the store rate, the tests came on top: 1.7 M + 3.7 M cycles per
    // workgroup at the config-5 share, profiles/r03_notes.md 13.)
    const bool overlap = have_next && nb1 - nb0 <= (unsigned)kFusedListCap;
    const int n_over = overlap ? (int)(nb1 - nb0) : 0;
    if ((int)threadIdx.x < n_over) list[threadIdx.x] = nxt;
    if (threadIdx.x < 256) qt[threadIdx.x] = (float)threadIdx.x / rs_next;      // rs_next is still row h's sum here (0 / 0 = NaN as in the reference)
    __syncthreads();
    if (overlap) {
      if (h + 2 < h_hi) { o0 = p0; o1 = p1; fetch(h + 2, o0); }
      if (h + 3 < h_hi) bounds_ahead(h + 3);
    }
    // ---- sweep: normalise, store the slab of row h, fold into the running maximum, leave the counters zero
    const float rs = rs_next;
    const bool sel = sel_next;
    if (have_next) { rs_next = rowsum[h + 1]; sel_next = !select || select[h + 1]; }
    float4* dst = reinterpret_cast<float4*>(counts + (int64_t)h * R * RR + (int64_t)x0 * RR);
#pragma unroll
    for (int k = 0; k < kFusedMaxChunks; ++k) {
      // a thread owns two QUADS of cells per chunk, a = tid + 1024 k and b = a + 512: every store instruction of a wave then covers
      // 1 KB of contiguous memory (8 consecutive cells per thread made two half-used instructions of it)
      const int a = threadIdx.x + k * 2 * kFusedThreads, b = a + kFusedThreads;
      if (a < n4) {
        const bool second = b < n4;
        const uint2 ca = reinterpret_cast<uint2*>(cnt_cur)[a];
        const uint2 cb = second ? reinterpret_cast<uint2*>(cnt_cur)[b] : make_uint2(0, 0);
        reinterpret_cast<uint2*>(cnt_cur)[a] = make_uint2(0, 0);
        if (second) reinterpret_cast<uint2*>(cnt_cur)[b] = make_uint2(0, 0);
        const unsigned w[4] = {ca.x, ca.y, cb.x, cb.y};
        unsigned cu[8];
        float c[8], v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cu[2 * e] = w[e] & 0xffffu; cu[2 * e + 1] = w[e] >> 16; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          c[e] = (float)cu[e];
          // the same correctly rounded quotient either way
          v[e] = cu[e] < 256u ? qt[cu[e]] : c[e] / rs;
        }
        dst[a] = write_raw ? make_float4(c[0], c[1], c[2], c[3]) : make_float4(v[0], v[1], v[2], v[3]);
        if (second) dst[b] = write_raw ? make_float4(c[4], c[5], c[6], c[7]) : make_float4(v[4], v[5], v[6], v[7]);
        if (sel) {
#pragma unroll
          for (int e = 0; e < 8; ++e) mx[k][e] = (v[e] > mx[k][e] || v[e] != v[e]) ? v[e] : mx[k][e];
        }
      }
      if (n_over) tests(cnt_nxt, n_over, k, kFusedMaxChunks);      // trips k, k + 4, ... of the next row
    }
    __syncthreads();
    if (have_next && !overlap) {                                   // a crowded (row, slab): all its rounds now, as for the first row
      all_rounds(cnt_nxt, h + 1, nb0, nb1);
      if (h + 2 < h_hi) { o0 = p0; o1 = p1; fetch(h + 2, o0); }
      if (h + 3 < h_hi) bounds_ahead(h + 3);
    }
    unsigned* t = cnt_cur; cnt_cur = cnt_nxt; cnt_nxt = t;
  }
  float4* pout = reinterpret_cast<float4*>(partial + (int64_t)g * R * RR + (int64_t)x0 * RR);
#pragma unroll
  for (int k = 0; k < kFusedMaxChunks; ++k) {
    const int a = threadIdx.x + k * 2 * kFusedThreads, b = a + kFusedThreads;
    if (a < n4) pout[a] = make_float4(mx[k][0], mx[k][1], mx[k][2], mx[k][3]);
    if (b < n4) pout[b] = make_float4(mx[k][4], mx[k][5], mx[k][6], mx[k][7]);
  }
}

__global__ __launch_bounds__(256) void occupancy_groupmax_kernel(const float* __restrict__ partial, int groups, int64_t R3,
                                                                 float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R3) return;
  float m = -__builtin_inff();
  for (int g = 0; g < groups; ++g) {
    const float v = partial[(int64_t)g * R3 + i];
    m = (v > m || v != v || m != m) ? ((m != m) ? m : v) : m;      // NaN sticks
  }
  out[i] = m;
}

}  // namespace coma

using namespace coma;

extern "C" int coma_occupancy_splat(const float* q, int S, int H, int R, const double* centers,
                                    double voxel, double thres, float* counts, void* stream) {
  if (!q || !centers || !counts) return fail(COMA_E_INVALID, "coma_occupancy_splat: null pointer");
  if (S < 0 || H <= 0 || R <= 0 || !(voxel > 0.0) || !(thres > 0.0))
    return fail(COMA_E_INVALID, "coma_occupancy_splat: bad sizes S=%d H=%d R=%d", S, H, R);
  if (S == 0) return COMA_OK;
  hipLaunchKernelGGL(occupancy_splat_kernel, dim3((unsigned)H), dim3(kSplatWaves * kWave), 0,
                     (hipStream_t)stream, q, S, H, R, centers, voxel, thres, counts);
  return check_launch("occupancy_splat_kernel");
}

extern "C" int coma_occupancy_reduce(float* counts, const uint8_t* select, int H, int64_t R3,
                                     float* rowsum, float* out, void* stream) {
  if (!counts || !rowsum || !out) return fail(COMA_E_INVALID, "coma_occupancy_reduce: null pointer");
  if (H <= 0 || R3 <= 0) return fail(COMA_E_INVALID, "coma_occupancy_reduce: bad sizes");
  hipLaunchKernelGGL(occupancy_rowsum_kernel, dim3((unsigned)H), dim3(256), 0, (hipStream_t)stream, counts,
                     R3, rowsum);
  if ((R3 & 3) == 0) {
    const int64_t blocks = (R3 / 4 + 255) / 256;
    hipLaunchKernelGGL(occupancy_norm_max4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       counts, select, rowsum, H, R3, out);
  } else {
    const int64_t blocks = (R3 + 255) / 256;
    hipLaunchKernelGGL(occupancy_norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       counts, select, rowsum, H, R3, out);
  }
  return check_launch("occupancy reduce kernels");
}


// workspace layout: [items: H * S * window 16-byte incidences | plane_off: H * (R + 1) words | partial maxima: groups * R^3 f32]
static size_t occ_ws_items(int S, int H, int window) { return occ_align256((size_t)S * H * window * sizeof(OccItem)); }
static size_t occ_ws_off(int H, int R) { return occ_align256((size_t)H * (R + 1) * sizeof(unsigned)); }
static int occ_pow2_window(int window) {
  if (window > 2 && (window & (window - 1))) { int w2 = 1; while (w2 < window) w2 <<= 1; window = w2; }   // cell decode by shifts
  return window;
}

extern "C" size_t coma_occupancy_fused_workspace_bytes(int S, int H, int R, int window) {
  window = occ_pow2_window(window);
  if (S <= 0 || H <= 0 || R <= 0 || window < 2 || window > kFusedMaxWindow) return 0;
  const int64_t R3 = (int64_t)R * R * R;
  const int P = kFusedMaxCells / (R * R);
  if (P < 1) return 0;
  const int slabs = (R + P - 1) / P;
  int groups = (kFusedResident + slabs - 1) / slabs;   // every workgroup resident at once: one round, no tail
  if (groups > H) groups = H;
  return occ_ws_items(S, H, window) + occ_ws_off(H, R) + (size_t)groups * R3 * sizeof(float) + 256;
}

extern "C" int coma_occupancy_fused(const float* q, int S, int H, int R, const double* centers, double voxel, double thres,
                                    double thres_sq_cut, int window, const uint8_t* select, int write_raw, float* counts, float* rowsum,
                                    float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!q || !centers || !counts || !rowsum || !out || !workspace) return fail(COMA_E_INVALID, "coma_occupancy_fused: null pointer");
  window = occ_pow2_window(window);
  if (S <= 0 || H <= 0 || R <= 0 || R > 255 || (R * R) % 4 || !(voxel > 0.0) || !(thres > 0.0) || window < 2 || window > 16)
    return fail(COMA_E_INVALID, "coma_occupancy_fused: bad sizes S=%d H=%d R=%d window=%d (R*R must be a multiple of 4, R <= 255)", S, H, R, window);
  const int RR = R * R;
  const int P = kFusedMaxCells / RR;
  if (P < 1) return fail(COMA_E_INVALID, "coma_occupancy_fused: R=%d too large for an LDS-resident plane (use splat + reduce)", R);
  const size_t need = coma_occupancy_fused_workspace_bytes(S, H, R, window);
  if (workspace_bytes < need) return fail(COMA_E_INVALID, "coma_occupancy_fused: workspace %zu < %zu bytes", workspace_bytes, need);
  const int slabs = (R + P - 1) / P;
  int groups = (kFusedResident + slabs - 1) / slabs;   // every workgroup resident at once: one round, no tail
  if (groups > H) groups = H;
  const size_t lds = 2 * ((((size_t)P * RR * 2 + 15) / 16) * 16) + (size_t)kFusedListCap * sizeof(OccItem) + (size_t)3 * R * sizeof(double) + 256 * sizeof(float);   // two counter buffers, quotient table
  if (S >= 65536) return fail(COMA_E_INVALID, "coma_occupancy_fused: S=%d >= 65536 samples per call (16-bit counters)", S);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
  OccItem* items = reinterpret_cast<OccItem*>(wsb);
  unsigned* plane_off = reinterpret_cast<unsigned*>(wsb + occ_ws_items(S, H, window));
  float* partial = reinterpret_cast<float*>(wsb + occ_ws_items(S, H, window) + occ_ws_off(H, R));
  hipLaunchKernelGGL(occupancy_rowprep_kernel, dim3((unsigned)H), dim3(256), 0, st, q, S, H, R, window, centers, voxel, thres, thres_sq_cut,
                     items, plane_off, rowsum);
  static coma::LdsOptIn lds_opt;                                 // the attribute is per function AND per device: set it when it grows
  if (int rc = coma::opt_in_lds(lds_opt, reinterpret_cast<const void*>(occupancy_fused_kernel), lds, "coma_occupancy_fused")) return rc;
  hipLaunchKernelGGL(occupancy_fused_kernel, dim3((unsigned)slabs, (unsigned)groups), dim3(kFusedThreads), lds, st, items, plane_off, rowsum, select,
                     S, H, R, P, window, groups, write_raw, centers, voxel, thres, thres_sq_cut, counts, partial);
  const int64_t R3 = (int64_t)R * RR;
  hipLaunchKernelGGL(occupancy_groupmax_kernel, dim3((unsigned)((R3 + 255) / 256)), dim3(256), 0, st, partial, groups, R3, out);
  return check_launch("occupancy fused kernels");
}
