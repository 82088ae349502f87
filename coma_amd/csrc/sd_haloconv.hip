// sd_haloconv.hip -- GroupNorm affine + SiLU + 3x3 / stride 1 / pad 1 convolution as a halo-patch ("direct") convolution on gfx950, in
// slices of 128 OUTPUT channels.  Shipped envelope: n in {128, 256, 384, 512} as n / 128 workgroups per 16 x 16 tile (each re-fetches the
// patch, which stays in L2), c a multiple of 64 up to 512, h and w multiples of 16, one input sample below 2 GiB (32-bit byte offsets
// against a 0x7fffffff-byte buffer resource; refused otherwise -- the callers fall back to the implicit GEMM).  The description below is
// of one 128-channel slice.  First users: the ResNet layers of the VAE at 512 x 512 (decoder up_blocks.3, encoder down_blocks.0;
// self.vae.decode / self.vae.encode, utils/adaptive_mask_inpainting.py:1086, :1112, :677-680).
//
// Why not the implicit GEMM (sd_gemm.hip): with N = 128 every activation row is re-staged through L2 -> LDS nine times (once per tap)
// for only 128 output columns -- 756 TF/s isolated, 670 TF/s inside the VAE graphs -- and every such layer sits behind a GroupNorm
// apply pass that reads and writes the 0.5 GB tensor once more (0.26 ms).  Here a workgroup (4 waves, two workgroups per CU) owns a
// 16 x 16 pixel tile x all 128 output channels:
//   * per 64-channel chunk the (18 x 18) halo patch is fetched ONCE, normalised (per-(sample, channel) affine table of the GroupNorm,
//     statistics from the producer's column sums), activated, rounded to fp16 -- exactly what the GroupNorm kernel would have stored --
//     and written to LDS in XOR-swizzled 16-byte slots; the nine taps read it at shifted positions (one conflict-free ds_read_b128 per
//     MFMA operand); the next chunk's global loads are in flight under this chunk's MFMAs;
//   * the weights stream through two LDS stages per (tap, chunk) slice [128][64] by LDS-DMA (`buffer_load ... lds`, source-side XOR
//     swizzle, counted vmcnt, one raw s_barrier per slice) like the implicit GEMM's W operand;
//   * `v_mfma_f32_16x16x32_f16`, D = W_frag . pixel_frag^T: a wave owns 4 tile rows (64 pixels) x 128 channels = 32 accumulator tiles,
//     64 MFMAs per slice against 24 LDS fragment reads;
//   * epilogue through LDS in two channel halves (fp32 staging): bias, residual, ONE rounding to fp16, 16-byte coalesced stores and, on
//     request, the column sums / sums of squares of the stored tensor for the consumer's GroupNorm, ONE slot per workgroup / tile
//     (sd_groupnorm_table_f16 with rows_per_slot = 256: 8 x fewer slots than the GEMM epilogue's, whose 67 MB at 512 x 512 cost the
//     consumer's table launch 84-120 us).
// Algorithmic bytes: the input read once (+ 27 % halo), the output written once, the residual read once.
// Measured alternatives that were NOT kept (profiles/r05_notes.md 1): persistent workgroups with the next tile's patch prefetched inside
// the epilogue (spills: ~ 35 registers short), the four waves splitting the output channels with their weight fragments loaded straight
// from L2 into registers (no weight staging, no barrier per slice: 739 vs 736 us), and that form on 16 x 8 tiles with three workgroups
// per CU (146 registers, 37 KB of LDS: 808-823 vs 730-737 us).
#include <hip/hip_fp16.h>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {
namespace hc {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kTile = 16, kHalo = 18, kChunk = 64, kN = 128;
constexpr int kPatchBytes = kHalo * kHalo * 8 * 16;      // 41 472: 324 pixels x 8 octets of 8 channels
constexpr int kWStage = kN * kChunk * 2;                 // 16 384: one (tap, chunk) weight slice [128][64]
constexpr int kAffOff = kPatchBytes + 2 * kWStage;       // 74 240
constexpr int kMaxC = 512;
constexpr int kLds = kAffOff + kMaxC * 8;                // 76 288 bytes: two workgroups per CU
constexpr int kStageRow = 64 * 4 + 16;                   // epilogue staging: 64 fp32 channels + 16 bytes of padding per pixel
static_assert(4 * 64 * kStageRow + 4 * 2 * 2 * 64 * 4 <= kAffOff, "epilogue staging + the waves' column sums overlay the patch and the weight stages");

// slot of 16-byte channel octet v (0..7) of halo pixel (row r, column c): 16 consecutive pixels of a row x one octet hit 16 distinct bank
// groups (18 is even, so pixel parity = column parity; (c >> 1) & 7 takes 8 values over 16 consecutive columns, each at both parities).
// The swizzle depends on the COLUMN only, so a tap's row shift is a constant byte offset of the fragment address.
__device__ __forceinline__ int pslot(int r, int c, int v) { return (r * kHalo + c) * 8 + (v ^ ((c >> 1) & 7)); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct HaloArgs {
  const _Float16* x;          // NHWC [batch][H][W][C]
  const float* affine;        // fp32 [batch][C][2] = (scale, shift) or nullptr
  int silu;
  const _Float16* w;          // [128][9][C]
  const _Float16* bias;       // [128] or nullptr
  const _Float16* res;        // [batch*H*W][ldr] or nullptr
  int ldr;
  int C, H, W;
  _Float16* out;              // [batch*H*W][ldo]
  int ldo;
  float* colstats;            // fp32 [batch*H*W/256][2][ntot] (one slot per 16 x 16 tile) or nullptr
  int tiles_x, tiles_y;
  int ntot;                   // output channels of the layer (128 or 256): block id % (ntot / 128) selects this workgroup's 128
};

#ifdef HALO_DBG
// tuning build (-DHALO_DBG, loaded through COMA_HIP_LIB): per workgroup, wave 0 records shader-clock stamps -- entry, first commit done,
// end of the K loop, end of the epilogue -- and the cycles it spent in the per-slice wait + barrier and in the commits; sd_halo_debug()
constexpr int kDbgBlocks = 8192;
__device__ unsigned long long g_halo_dbg[kDbgBlocks * 8];
#define HALO_STAMP(k) do { if (dbg_on) g_halo_dbg[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#ifdef HALO_DBG_LIGHT                                      // stamps only (the accumulating counters cost a global read-modify-write per slice)
#define HALO_ACC(k, t0) do {} while (0)
#else
#define HALO_ACC(k, t0) do { if (dbg_on) g_halo_dbg[blockIdx.x * 8 + (k)] += __builtin_readcyclecounter() - (t0); } while (0)
#endif
#else
#define HALO_STAMP(k) do {} while (0)
#define HALO_ACC(k, t0) do {} while (0)
#endif

__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kLds];
  half8* const tile = reinterpret_cast<half8*>(lds);
  unsigned char* const wst = lds + kPatchBytes;
  float2* const aff = reinterpret_cast<float2*>(lds + kAffOff);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef HALO_DBG
  const bool dbg_on = tid == 0 && blockIdx.x < kDbgBlocks;
  if (dbg_on) { g_halo_dbg[blockIdx.x * 8 + 4] = 0; g_halo_dbg[blockIdx.x * 8 + 5] = 0; }
  unsigned long long tdbg = 0;
#endif
  HALO_STAMP(0);
  const int lp = lane & 15, lo = lane >> 4;
  const int C = a.C, nchunk = C / kChunk, nkt = 9 * nchunk;
  const int nsplit = a.ntot / kN;

  auto make_rsrc = [](const void* p) {
    const unsigned long long q = reinterpret_cast<unsigned long long>(p);
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)q), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi32 << 32) | lo32), 0, 0x7fffffff, 0x00020000);
  };
  // ---- weight slices by LDS-DMA: instruction j of this wave covers rows (wave * 4 + j) * 8 .. + 7 of the slice; lane -> row + lane / 8,
  // LDS slot lane % 8, which must receive K octet slot ^ ((row >> 1) & 7)
  // (row r + 8 j: the swizzle term (r >> 1) & 7 only flips bit 2 for odd j -> two lane registers serve the four instructions)
  unsigned w_off2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);
    w_off2[j] = (unsigned)(r * 9 * C) * 2u + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
  }
  const int w_jstep = __builtin_amdgcn_readfirstlane(16 * 9 * C * 2);       // two instructions further = 16 rows further
  // W fragment: row = 16 j + (lane & 15), K octet = 4 kh + (lane >> 4); the swizzle term looks at row bits 1-3 only, so tile j is a
  // constant byte offset
  int w_fo[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) w_fo[kh] = lp * 128 + (((kh * 4 + lo) ^ ((lp >> 1) & 7)) * 16);
  // pixel fragment of tap (dy, dx), tile row t, K half kh: patch pixel (4 wave + t + dy, lp + dx), octet 4 kh + lo.  The swizzle looks at
  // the patch COLUMN only (pslot), so per (dx, kh) the lane part of the address is one register and rows are constant offsets.
  int pf_lane[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) pf_lane[dx][kh] = (pslot(0, lp + dx, kh * 4 + lo) + wave * 4 * kHalo * 8) * 16;
  // epilogue read-back role (also the role in which the residual is prefetched): pixel 8 it + ps of this wave's 64, channel octet c8
  const int ps = lane >> 3, c8 = (lane & 7) * 8;
  unsigned char* const stg = lds + wave * (64 * kStageRow);
  float* const wsum = reinterpret_cast<float*>(lds + 4 * 64 * kStageRow);               // [wave][half][sum | sumsq][64]: 4 KB behind the staging areas

  // ---- halo patch: fetch = all of a chunk's global loads of this thread (11 independent 16-byte buffer loads, every wave issues all 11 so
  // that the counted vmcnt below means the same in every wave; one 32-bit offset per load, pad pixels carry an out-of-range offset and
  // come back as zeros), commit = normalise + activate + LDS write.
  constexpr int kVec = kHalo * kHalo * 8, kPer = (kVec + 255) / 256;      // 2592 vectors, 11 per thread
  // Offsets, slots and validity are recomputed at every fetch / commit from an opaque copy of the thread id: kept live across the K loop
  // they would cost 20-30 registers next to the 128 accumulators + the 44 patch registers (and spill).
  half8 q[kPer];
  auto fetch = [&](__amdgpu_buffer_rsrc_t x_rsrc, int ty0, int tx0, int chunk) {
#if defined(__HIP_DEVICE_COMPILE__)
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int soff = __builtin_amdgcn_readfirstlane(chunk * kChunk * 2);
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int idx = t_ + it * 256;
      const int p = idx >> 3, v = idx & 7;
      const int r = p / kHalo, c = p - r * kHalo;
      const int y = ty0 + r - 1, xx = tx0 + c - 1;
      const bool ok = idx < kVec && y >= 0 && y < a.H && xx >= 0 && xx < a.W;
      const unsigned xo = ok ? (unsigned)((y * a.W + xx) * C + v * 8) * 2u : 0x80000000u;
      q[it] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xo, soff, 0));
    }
#endif
  };
  auto commit = [&](int ty0, int tx0, int chunk) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));                           // opaque: nothing below is loop-invariant as far as the compiler can tell
    // idx = tid + 256 it: the channel octet v = tid & 7 is the same for all of a thread's vectors -> its 8 (scale, shift) pairs are read
    // from LDS ONCE per chunk
    float2 sc8[8];
    if (a.affine) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sc8[e] = aff[chunk * kChunk + (t_ & 7) * 8 + e];
    }
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int idx = t_ + it * 256;
      if (idx < kVec) {
        const int p = idx >> 3, v = idx & 7;
        const int r = p / kHalo, c = p - r * kHalo;
        const int y = ty0 + r - 1, xx = tx0 + c - 1;
        half8 z = q[it];
        if (a.affine && y >= 0 && y < a.H && xx >= 0 && xx < a.W) {     // zero padding applies to the ACTIVATED tensor: pad pixels stay 0
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)z[e], sc8[e].x, sc8[e].y);
            if (a.silu) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));
            z[e] = (_Float16)f;
          }
        }
        tile[pslot(r, c, v)] = z;
      }
    }
  };

  // ---- one workgroup per (tile, 128-channel slice).  n > 128: the slices of a tile are consecutive in dispatch order (the later ones find
  // the patch in the Infinity Cache); the patch is staged (and activated) once per workgroup.
  // (A PERSISTENT form -- two workgroups per CU walking the tile list, the next tile's first patch requested in the middle of the epilogue
  // -- was written and measured to be worth having: the phase stamps show a CU slot empty for ~ 30 % of the kernel between workgroups and
  // 4 us of first-patch latency per tile.  It does not fit: the loop-carried state costs ~ 35 registers more than the 256 a wave has at
  // two workgroups per CU, and the library refuses kernels that touch scratch; profiles/r05_notes.md.)
  int id = blockIdx.x;
  const int n_off = __builtin_amdgcn_readfirstlane((id % nsplit) * kN);
  id /= nsplit;
  const int tx = __builtin_amdgcn_readfirstlane(id % a.tiles_x);
  id /= a.tiles_x;
  const int ty = __builtin_amdgcn_readfirstlane(id % a.tiles_y);
  const int b = __builtin_amdgcn_readfirstlane(id / a.tiles_y);
  const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(a.x + (size_t)b * a.H * a.W * C);
  fetch(x_rsrc, ty * kTile, tx * kTile, 0);
  {
    const int ty0 = ty * kTile, tx0 = tx * kTile;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(a.w + (size_t)n_off * 9 * C);
    auto issue_w = [&](int kt) {                          // slice kt = chunk * 9 + tap into stage kt & 1
#if defined(__HIP_DEVICE_COMPILE__)
      const int chunk = kt / 9, tap = kt - chunk * 9;
      const int soff = __builtin_amdgcn_readfirstlane((tap * C + chunk * kChunk) * 2);
      unsigned char* dst = wst + (kt & 1) * kWStage + wave * 4096;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(dst + j * 1024), 16, w_off2[j & 1], soff + (j >> 1) * w_jstep, 0, 0);
#endif
    };
    const size_t m_base = ((size_t)b * a.H + ty0 + 4 * wave) * a.W + tx0;               // pixel (t, col) of this wave = m_base + t * W + col
    const _Float16* const resp = a.res ? a.res + n_off : nullptr;

    issue_w(0);
    if (a.affine)
      for (int i = tid; i < C; i += 256) aff[i] = reinterpret_cast<const float2*>(a.affine)[(size_t)b * C + i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // `aff` is visible

    float4v acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = float4v{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      if (chunk > 0) __builtin_amdgcn_s_barrier();         // every wave is done reading the previous chunk's patch
#ifdef HALO_DBG
      tdbg = __builtin_readcyclecounter();
#endif
      commit(ty0, tx0, chunk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HALO_ACC(5, tdbg);
      if (chunk == 0) HALO_STAMP(1);
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        const int kt = chunk * 9 + tap;
        // slice kt has landed (this wave's share); the next chunk's patch loads, issued during tap 0 BEHIND slice kt + 1, may stay in flight
        // (in the LAST chunk the idle patch registers take the residual rows of the epilogue's first channel half instead: 8 loads)
#ifdef HALO_DBG
        tdbg = __builtin_readcyclecounter();
#endif
        if (tap == 1 && chunk + 1 < nchunk) wait_vmcnt<kPer>();
        else if (tap == 1 && resp) wait_vmcnt<8>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                      // ... everyone's share; and everyone is done with stage (kt + 1) & 1 and has committed
        HALO_ACC(4, tdbg);
        if (kt + 1 < nkt) issue_w(kt + 1);
        asm volatile("" ::: "memory");                     // the loads below must be issued BEHIND the slice: the counted vmcnt relies on it
        if (tap == 0 && chunk + 1 < nchunk) fetch(x_rsrc, ty0, tx0, chunk + 1);
        if (tap == 0 && chunk + 1 == nchunk && resp) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int pw = it * 8 + ps;
            q[it] = *reinterpret_cast<const half8*>(resp + (m_base + (size_t)(pw >> 4) * a.W + (pw & 15)) * a.ldr + c8);
          }
        }
        const unsigned char* ws = wst + (kt & 1) * kWStage;
        const int dy = tap / 3, dx = tap - dy * 3;
        const int prow = __builtin_amdgcn_readfirstlane(dy * kHalo * 128);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const unsigned char* pb = lds + (dx == 0 ? pf_lane[0][kh] : (dx == 1 ? pf_lane[1][kh] : pf_lane[2][kh])) + prow;
          half8 pf[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) pf[t] = *reinterpret_cast<const half8*>(pb + t * (kHalo * 128));
          // the W fragments come in two halves of four column tiles: 16 fewer registers live next to the 128 accumulators + the prefetched patch
#pragma unroll
          for (int jh = 0; jh < 2; ++jh) {
            half8 wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const half8*>(ws + w_fo[kh] + (jh * 4 + j) * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[t][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], pf[t], acc[t][jh * 4 + j], 0, 0, 0);
          }
        }
      }
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // every wave is done with the patch and the weight stages
    HALO_STAMP(2);

    // ---- epilogue.  acc[t][j][r] = output channel 16 j + 4 lo + r of pixel (tile row 4 wave + t, column lp).  Two channel halves through
    // this wave's fp32 staging area [64 pixels][64 channels (+ pad)], read back as (pixel = 8 it + lane / 8, 8 channels = lane % 8).
    // residual rows: the first channel half was prefetched under the last chunk's MFMAs (into the idle patch registers), the second half
    // is requested as soon as the first half's accumulators have been staged (their registers are free then)
    // (opaque copies of the lane roles: hoisted out of the persistent loop, the epilogue's ~40 lane-dependent addresses would stay live
    // across the K loop next to the accumulators -- and spill)
    int lp_e = lp, lo_e = lo, ps_e = ps, c8_e = c8;
    asm volatile("" : "+v"(lp_e), "+v"(lo_e), "+v"(ps_e), "+v"(c8_e));
    _Float16* const outp = a.out + n_off;
    const _Float16* const biasp = a.bias ? a.bias + n_off : nullptr;
    const size_t slot_id = ((size_t)b * a.tiles_y + ty) * a.tiles_x + tx;
    half8 rres[2][8];
    if (resp) {
#pragma unroll
      for (int it = 0; it < 8; ++it) rres[0][it] = q[it];
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4v*>(stg + (t * 16 + lp_e) * kStageRow + (j * 16 + lo_e * 4) * 4) = acc[t][hf * 4 + j];
      asm volatile("" ::: "memory");                       // the loads below must not be scheduled above the staging writes (accumulators still live there)
      if (hf == 0 && resp) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int pw = it * 8 + ps_e;
          rres[1][it] = *reinterpret_cast<const half8*>(resp + (m_base + (size_t)(pw >> 4) * a.W + (pw & 15)) * a.ldr + 64 + c8_e);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = 0.0f;
      if (biasp) {
        const half8 bq = *reinterpret_cast<const half8*>(biasp + hf * 64 + c8_e);
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (float)bq[e];
      }
      // statistics: ONE fold per channel half over this wave's 64 pixels; the four waves' sums meet in LDS below (one slot per tile)
      float cs[8], cq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.0f;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int pw = it * 8 + ps_e;
        const float4v v0 = *reinterpret_cast<const float4v*>(stg + pw * kStageRow + c8_e * 4);
        const float4v v1 = *reinterpret_cast<const float4v*>(stg + pw * kStageRow + c8_e * 4 + 16);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = v[e] + bv[e];
          if (resp) f += (float)rres[hf][it][e];
          o[e] = (_Float16)f;
          const float g = (float)o[e];                     // statistics of the stored (fp16-rounded) tensor
          cs[e] += g;
          cq[e] += g * g;
        }
        *reinterpret_cast<half8*>(outp + (m_base + (size_t)(pw >> 4) * a.W + (pw & 15)) * a.ldo + hf * 64 + c8_e) = o;
      }
      if (a.colstats) {
        // fold the 8 pixel lanes that share a channel octet (lane bits 3-5), fixed order -> reproducible
#pragma unroll
        for (int e = 0; e < 8; ++e) {                       // lane ^ 8 inside a row of 16 lanes: a DPP row rotation by 8 (VALU speed)
          cs[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cs[e]), 0x128, 0xf, 0xf, false));
          cq[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cq[e]), 0x128, 0xf, 0xf, false));
        }
#pragma unroll
        for (int mask = 16; mask < 64; mask <<= 1)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], mask);
            cq[e] += __shfl_xor(cq[e], mask);
          }
        if (lane < 8) {                                     // this wave's sums of the half -> LDS (behind the four staging areas)
          float* dst = wsum + ((wave * 2 + hf) * 2) * 64 + c8_e;
          *reinterpret_cast<float4v*>(dst) = float4v{cs[0], cs[1], cs[2], cs[3]};
          *reinterpret_cast<float4v*>(dst + 4) = float4v{cs[4], cs[5], cs[6], cs[7]};
          *reinterpret_cast<float4v*>(dst + 64) = float4v{cq[0], cq[1], cq[2], cq[3]};
          *reinterpret_cast<float4v*>(dst + 64 + 4) = float4v{cq[4], cq[5], cq[6], cq[7]};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    HALO_STAMP(3);
    if (a.colstats) {
      __syncthreads();
      // one statistics slot per workgroup (= 16 x 16 pixel tile): the four waves' sums added in wave order (fixed -> reproducible)
      const int hf = tid >> 7, which = (tid >> 6) & 1, c = tid & 63;             // 256 threads = [half][sum | sumsq][64 channels]
      float r = 0.0f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) r += wsum[((wv * 2 + hf) * 2 + which) * 64 + c];
      a.colstats[(slot_id * 2 + which) * a.ntot + n_off + hf * 64 + c] = r;
    }
  }
}


}  // namespace hc
}  // namespace sd

extern "C" int sd_conv3x3_halo_f16(const void* x, int c, const float* gn_affine, int silu, const void* w, const void* bias, const void* res,
                                   int ldr, int batch, int h, int w_, int n, void* out, int ldo, float* colstats, void* stream) {
  using namespace sd::hc;
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_CONV_HALO;
    r.p[0] = (void*)x; r.p[1] = (void*)gn_affine; r.p[2] = (void*)w; r.p[3] = (void*)bias; r.p[4] = (void*)res; r.p[5] = out; r.p[6] = colstats;
    r.i[0] = c; r.i[1] = silu; r.i[2] = ldr; r.i[3] = batch; r.i[4] = h; r.i[5] = w_; r.i[6] = n; r.i[7] = ldo;
    return sd::plan_record(r);
  }
  if (!x || !w || !out) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: null pointer");
  if (n <= 0 || n % kN || n > 4 * kN) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: n = %d (built for 128, 256, 384 and 512 output channels)", n);
  if (c <= 0 || c % kChunk || c > kMaxC) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: c = %d (a multiple of 64, at most %d)", c, kMaxC);
  if (batch <= 0 || h <= 0 || w_ <= 0 || h % kTile || w_ % kTile)
    return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: batch=%d h=%d w=%d (h, w multiples of 16)", batch, h, w_);
  if (silu && !gn_affine) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: silu is applied with the GroupNorm affine (gn_affine is NULL)");
  if (ldo == 0) ldo = n;
  if (ldr == 0) ldr = n;
  if (ldo < n || ldo % 8 || (res && (ldr < n || ldr % 8))) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: ldo = %d, ldr = %d", ldo, ldr);
  if ((long long)n * 9 * c * 2 >= 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: weights too large");
  // the kernel addresses one sample of x with a 32-bit byte offset against a 0x7fffffff-byte buffer resource: refuse what it cannot reach
  if ((long long)h * w_ * c * 2 >= 0x7fffffffLL)
    return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: one input sample is %lld bytes (must stay below 2 GiB)", (long long)h * w_ * c * 2);
  const long long tiles = (long long)batch * (h / kTile) * (w_ / kTile);
  if (tiles > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: grid too large");
  HaloArgs a;
  a.x = (const _Float16*)x; a.affine = gn_affine; a.silu = silu; a.w = (const _Float16*)w; a.bias = (const _Float16*)bias;
  a.res = (const _Float16*)res; a.ldr = ldr; a.C = c; a.H = h; a.W = w_; a.out = (_Float16*)out; a.ldo = ldo; a.colstats = colstats;
  a.tiles_x = w_ / kTile; a.tiles_y = h / kTile; a.ntot = n;
  if (tiles * (n / kN) > 0x7fffffffLL) return fail(COMA_E_INVALID, "sd_conv3x3_halo_f16: grid too large");
  hipLaunchKernelGGL(conv3x3_halo_kernel, dim3((unsigned)(tiles * (n / kN))), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("conv3x3_halo_kernel");
}

#ifdef HALO_DBG
extern "C" int sd_halo_debug(unsigned long long* host_dst, int n_blocks) {
  if (!host_dst || n_blocks <= 0 || n_blocks > sd::hc::kDbgBlocks) return -1;
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(sd::hc::g_halo_dbg), (size_t)n_blocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
