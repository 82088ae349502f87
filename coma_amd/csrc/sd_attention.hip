// sd_attention.hip -- fused multi-head attention with online softmax for the UNet transformer blocks (gfx950).
//
// replaces: diffusers' Attention processor (softmax(q k^T / sqrt(d)) v) reached from the reference through
// self.unet(...) (utils/adaptive_mask_inpainting.py:1001-1007); self-attention at L = 4096/1024/256/64 with
// d = 40/80/160/160 and cross-attention over 77 text tokens.
//
// Formulation ("swapped", so that softmax state is lane-local): with v_mfma_f32_32x32x16_f16
//     S^T[key, q] = K[key, :] . Q[q, :]        A = K tile (LDS, ds_read_b128), B = Q (registers, loaded once)
//     O^T[dd, q] += V^T[dd, key] . P^T[key, q]  A = V^T tile (LDS, 2 x ds_read_b64), B = P (from the S accumulators)
// the C layout puts query q = lane & 31 in every accumulator register of a lane, so the running max, the
// running sum and the rescale factor are one scalar per lane (two lanes share a query and exchange one value
// per tile).  The contraction index of the second product is the key; because a sum is invariant under a
// permutation of its index, P is fed to the MFMA in the order the accumulator registers already hold it and the
// V^T fragment is read in the matching order (keys {0-3, 8-11}+4*half per 16-key step) -- no cross-lane traffic.
// V must be supplied TRANSPOSED ([heads*d, keys]); the projection GEMM produces that layout directly
// (sd_conv_gemm_f16 with the weight as the A operand), so no transpose pass exists anywhere.
//
// Block = 4 waves x 32 (or 64) queries of one (batch, head); K / V^T tiles of 64 keys reach LDS by LDS-DMA into
// XOR-swizzled 128-byte rows (conflict-free b128 / b64 fragment reads), two stages.
// Measured on MI355X (scripts/probes/mfma_valu.hip): MFMA and VALU time of the waves of one SIMD ADD UP rather than
// overlap, so per 32 x 64 score tile the kernel pays 14 MFMAs (~450 cycles) plus the softmax VALU work (32 exp2 at
// quarter rate ~510 cycles, ~75 more full-rate instructions); the staging is kept off the VALU entirely.
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int BKV = 64;            // keys per tile

struct AttnArgs {
  const _Float16* q;
  const _Float16* k;
  const _Float16* vt;
  _Float16* out;
  int heads, lq, lk, d;
  int ldq, ldk, ldv, ldo;
  float scale_log2;   // scale * log2(e)
  int no_spec;        // attention_sp_kernel: track the maximum on every key tile from the start (A/B aid, SD_ATTN_SPEC=0)
};

typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte chunk slot of K-chunk `chunk` in a 128-byte LDS row (same XOR swizzle as the GEMM's BK = 64 tiles): the rows
// read by one ds_read lane group land on distinct bank slots for both the b128 K reads and the b64 V^T reads
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// QT = 32-query tiles per wave (QT = 2: each K / V^T fragment read feeds two MFMAs).
// ONES >= 0: V^T has spare padded rows (d < DVT*32); row ONES (= d) of both LDS stages is written once with ones, so the
// softmax denominator falls out of the same MFMAs as the numerator (sum of the SAME fp16-rounded weights) and the
// per-tile VALU row sums disappear.  ONES < 0: the denominator is summed on the VALU.
//
// Staging: K / V^T tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), two stages, tile t+1 in flight
// while tile t is multiplied.  Per lane the source offset is fixed up to a per-tile increment, zero padding (head dim
// beyond d, keys beyond lk, V^T rows beyond d) comes from the buffer range check, and the VALU -- the unit that limits
// this kernel (exp2, max, convert) -- spends nothing on the copy.
//   K  stage: KP panels of [64 keys][64 halves] (panel p = head dims 64p .. 64p+63), rows of 128 B, swizzled
//   V^T stage: [DV rows][64 keys], rows of 128 B, swizzled
template <int KS, int DVT, int QT, int ONES, bool VPERM = false>
__global__ __launch_bounds__(256, 2) void attention_kernel(AttnArgs a) {
  constexpr int KP = (KS * 16 + 63) / 64;
  constexpr int DV = DVT * 32;
  constexpr int K_STAGE = KP * BKV * 64, V_STAGE = DV * 64;           // halves
  __shared__ __attribute__((aligned(1024))) _Float16 Kbuf[2][K_STAGE];
  __shared__ __attribute__((aligned(1024))) _Float16 Vbuf[2][V_STAGE];
  constexpr int K_DMA = KP * 8 / 4, V_DMA = DV / 8 / 4;                // DMA instructions per wave per tile
  static_assert(DV % 32 == 0, "V^T rows come in groups of 32");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (128 * QT) + wave * (32 * QT);
  const int ql = lane & 31, hh = lane >> 5;
  const int d = a.d;

  // Q fragments (B operand): lane (query ql, half hh) holds dd = ks*16 + hh*8 .. +7
  half8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = q0 + qt * 32 + ql;
    const _Float16* qp = a.q + ((long long)b * a.lq + (qi < a.lq ? qi : 0)) * a.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = ks * 16 + hh * 8;
      if (qi < a.lq && dd < d) {
        qf[qt][ks] = *reinterpret_cast<const half8*>(qp + dd);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (_Float16)0.0f;
      }
    }
  }

  float16v o[QT][DVT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][t][r] = 0.0f;
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -__builtin_inff(); l_run[qt] = 0.0f; }

  // ---- DMA descriptors.  K: rows = keys of this (batch, head) slice, valid bytes end with key lk-1; V^T: rows = head
  // dims, valid bytes end with row d-1.  Anything beyond fails the range check and lands in LDS as zeros.
  const _Float16* kbase = a.k + (long long)b * a.lk * a.ldk + h * d;
  const _Float16* vbase = a.vt + ((long long)b * a.heads + h) * d * (long long)a.ldv;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kbase, (unsigned)(((long long)(a.lk - 1) * a.ldk + d) * 2));
  const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vbase, (unsigned)((long long)d * a.ldv * 2));
  constexpr unsigned OOB = 0x80000000u;
  const int l_row = lane >> 3, l_slot = lane & 7;
  unsigned k_off[K_DMA], v_off[V_DMA];
#pragma unroll
  for (int j = 0; j < K_DMA; ++j) {
    const int idx = wave * K_DMA + j;                 // instruction index inside the stage: panel = idx / 8
    const int panel = idx >> 3, row = (idx & 7) * 8 + l_row;
    const int chunk = panel * 8 + swz64(row, l_slot); // 8-half chunk of the head dimension
    k_off[j] = chunk * 8 < d ? (unsigned)(row * a.ldk * 2 + chunk * 16) : OOB;
  }
#pragma unroll
  for (int j = 0; j < V_DMA; ++j) {
    const int row = (wave * V_DMA + j) * 8 + l_row;
    v_off[j] = (unsigned)(row * a.ldv * 2 + swz64(row, l_slot) * 16);
  }
  const unsigned k_step = (unsigned)(BKV * a.ldk * 2), v_step = BKV * 2;   // bytes per key tile
  auto issue_tile = [&](int buf, int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16* kd = Kbuf[buf] + wave * (K_DMA * 512);
    _Float16* vd = Vbuf[buf] + wave * (V_DMA * 512);
#pragma unroll
    for (int j = 0; j < K_DMA; ++j) {
      const unsigned off = k_off[j] == OOB ? OOB : k_off[j] + tile * k_step;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(kd + j * 512), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < V_DMA; ++j) {
      const int row = (wave * V_DMA + j) * 8 + l_row;
      if (ONES < 0 || row != ONES)       // the ones row is written once below and never overwritten
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(vd + j * 512), 16, v_off[j] + tile * v_step, 0, 0, 0);
    }
#endif
  };
  if (ONES >= 0 && tid < 16) {           // 2 stages x 8 chunks of the ones row
    half8 one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one[j] = (_Float16)1.0f;
    *reinterpret_cast<half8*>(&Vbuf[tid >> 3][(ONES >= 0 ? ONES : 0) * 64 + (tid & 7) * 8]) = one;
  }

  const int ntiles = (a.lk + BKV - 1) / BKV;
  issue_tile(0, 0);
  // the key-tile body exists twice: full tiles carry no masking code at all (left inside one body the compiler hoists
  // the 32 key-index adds and compares of the ragged case out of their wave-uniform branch, +64 VALU per tile)
  auto process_tile = [&](int tile, auto ragged_c) {
    constexpr bool RAGGED = decltype(ragged_c)::value;
    const int stage = tile & 1, key0 = tile * BKV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile `tile` has landed
    __builtin_amdgcn_s_barrier();                        // ... and everyone's; all waves are done reading stage^1
    if (tile + 1 < ntiles) issue_tile(stage ^ 1, tile + 1);
    const _Float16* Ks = Kbuf[stage];
    const _Float16* Vs = Vbuf[stage];

    // ---- S^T = K Q^T  (two 32-key tiles x QT query tiles; each K fragment is read once)
    float16v s[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qt][t][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int row = t * 32 + ql;
        const half8 kf = *reinterpret_cast<const half8*>(&Ks[(ks >> 2) * (BKV * 64) + row * 64 + swz64(row, 2 * (ks & 3) + hh) * 8]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qt][ks], s[qt][t], 0, 0, 0);
      }
    // ---- online softmax (per-lane scalars; the partner lane holds the other 32 keys of this query).
    // Scores stay RAW in the accumulators; scale*log2(e) rides in the FMA that forms the exp2 argument.  The
    // running max is only raised when it would grow by more than RESCALE_THR (log2 units): P may then reach
    // 2^THR (fine in fp16/fp32) and the O / l rescale becomes a rare, wave-uniformly skipped branch.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    half8 pf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if constexpr (RAGGED) {    // only the last tile can be ragged
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            s[qt][t][r] = key < a.lk ? s[qt][t][r] : -__builtin_inff();
          }
      }
      float mx = fmaxf(s[qt][0][0], s[qt][1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[qt][0][r]), s[qt][1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;   // scale > 0: max commutes with the scaling
      constexpr float RESCALE_THR = 6.0f;
      const bool need = mx > m_run[qt] + RESCALE_THR;      // first tile: m_run = -inf -> true
      if (__any(need)) {
        const float m_new = need ? mx : m_run[qt];
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);   // 1 when unchanged, 0 on the first tile
        m_run[qt] = m_new;
        l_run[qt] *= alpha;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qt][t][r] *= alpha;
      }
      const f32x2 c2 = {a.scale_log2, a.scale_log2}, nm2 = {-m_run[qt], -m_run[qt]};
      f32x2 ps2 = {0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 sv = {s[qt][t][r], s[qt][t][r + 1]};
          const f32x2 e = __builtin_elementwise_fma(sv, c2, nm2);
          const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          if constexpr (ONES < 0) ps2 += p;
          pf[qt][t * 2 + (r >> 3)][r & 7] = (_Float16)p.x;
          pf[qt][t * 2 + (r >> 3)][(r & 7) + 1] = (_Float16)p.y;
        }
      if constexpr (ONES < 0) l_run[qt] += ps2.x + ps2.y;
    }
    // ---- O^T += V^T P^T, 16 keys per MFMA, keys in accumulator-register order {0-3, 8-11} + 4*hh of each 16-key
    // step: two 8-byte reads from the chunks 2*st and 2*st+1 of the row; each V^T fragment feeds QT MFMAs
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
      for (int t = 0; t < DVT; ++t) {
        const int row = t * 32 + ql;
        half8 vf;
        if constexpr (VPERM) {
          // keys stored as (0-3, 8-11 | 4-7, 12-15) per 16: chunk 2*st + hh IS this lane's operand -- one b128 read, and the 16
          // lanes of a read group hit 16 distinct bank slots (the two b64 reads below collide two-way: a group of 32 lanes
          // shares hh and can only reach half of the 8-byte slots)
          vf = *reinterpret_cast<const half8*>(&Vs[row * 64 + swz64(row, 2 * st + hh) * 8]);
        } else {
          const _Float16* vr = &Vs[row * 64 + 4 * hh];
          const half4 lo = *reinterpret_cast<const half4*>(vr + swz64(row, 2 * st) * 8);
          const half4 hi = *reinterpret_cast<const half4*>(vr + swz64(row, 2 * st + 1) * 8);
          vf = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qt][st], o[qt][t], 0, 0, 0);
      }
    }
  };
  const int nfull = a.lk / BKV;
  for (int tile = 0; tile < nfull; ++tile) process_tile(tile, std::false_type{});
  if (nfull < ntiles) process_tile(nfull, std::true_type{});

  // ---- normalise and store: lane (query, half) owns dd = t*32 + 8*(r>>2) + 4*hh + (r&3)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l_tot;
    if constexpr (ONES >= 0) {
      // row ONES of O^T is the denominator: it sits in register (ONES%32 / 8)*4 of the hh = 0 lane of each query
      static_assert(ONES % 8 == 0, "the ones row must start an 8-row group");
      l_tot = __shfl(o[qt][ONES / 32][((ONES % 32) / 8) * 4], ql);
    } else {
      l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
    }
    const float inv = 1.0f / l_tot;
    const int qi = q0 + qt * 32 + ql;
    if (qi < a.lq) {
      _Float16* op = a.out + ((long long)b * a.lq + qi) * a.ldo + h * d;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = t * 32 + 8 * g + 4 * hh;
          if (dd < d) {
            half4 v = {(_Float16)(o[qt][t][g * 4 + 0] * inv), (_Float16)(o[qt][t][g * 4 + 1] * inv),
                       (_Float16)(o[qt][t][g * 4 + 2] * inv), (_Float16)(o[qt][t][g * 4 + 3] * inv)};
            *reinterpret_cast<half4*>(op + dd) = v;
          }
        }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// attention_sp_kernel<HQ> -- software-pipelined self-attention for d = 40 and whole 64-key tiles (the 64 x 64 level: 83 % of
// the attention time of a forward).  The kernel is bound by the VALU, not by the matrix pipe: per 64-query x 64-key tile a wave
// issues 28 MFMAs (896 cycles) and 64 v_exp_f32 (9-11 cycles each) + 32 conversions + 34 max -- measured on the whole chip
// (profiles/r03_notes.md): MFMAs alone 292 us, VALU alone 348 us, together in ONE wave per SIMD 616 us (the sum: an MFMA and the
// VALU instructions behind it do not overlap inside a wave here, whatever the placement), two waves per SIMD 487-500 us, the
// previous kernel 542 us.  So the levers are fewer VALU instructions and two resident workgroups per CU:
//   * no exp2-argument arithmetic at all: Q is pre-multiplied by scale*log2(e) (fp16), and the running maximum rides in the
//     product itself -- head dim 40 pads to 48, so K gets a constant-one column at dd = 40 and Q carries -m there: the
//     accumulator comes out as the exp2 argument (-32 v_pk_fma_f32 per tile).  m is kept on fp16 values, so numerator and
//     denominator (the ones row of V^T) see the same factor and it cancels exactly;
//   * a wave owns two halves (A, B) of HQ 32-query sub-tiles each; the softmax of one half is issued between the MFMAs of the
//     other, slot by slot in source order (a sched_barrier per MFMA slot), so that the co-resident wave of the other workgroup
//     always finds matrix work and VALU work mixed:
//         segment 1 of tile t:  softmax A(t)   ||  S_B(t) = K(t) Q_B^T,  O_B += V(t-1)^T P_B(t-1)
//         segment 2 of tile t:  softmax B(t)   ||  S_A(t+1) = K(t+1) Q_A^T,  O_A += V(t)^T P_A(t)
//     and the maximum of a fresh S tile is taken under the second half of the segment that produced it (v_permlane32_swap
//     instead of an LDS round trip for the partner lane);
//   * three LDS stages for K and V^T (48 KB: two workgroups per CU) and ONE barrier per key tile.
template <int HQ>
__global__ __launch_bounds__(256, 1) void attention_sp_kernel(AttnArgs a) {
  constexpr int NQ = 2 * HQ;                       // 32-query sub-tiles per wave
  constexpr int D = 40, STG = BKV * 64;            // halves per K / V^T stage (64 rows of 128 B)
  __shared__ __attribute__((aligned(1024))) _Float16 Kbuf[3][STG];
  __shared__ __attribute__((aligned(1024))) _Float16 Vbuf[3][STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (128 * NQ) + wave * (32 * NQ);
  const int ql = lane & 31, hh = lane >> 5;

  // ---- Q fragments, pre-scaled; element 0 of the (ks = 2, hh = 1) fragment is head dim 40: the slot that carries -m
  half8 qf[NQ][3];
#pragma unroll
  for (int sq = 0; sq < NQ; ++sq) {
    const int qi = q0 + sq * 32 + ql;
    const _Float16* qp = a.q + ((long long)b * a.lq + (qi < a.lq ? qi : 0)) * a.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int dd = ks * 16 + hh * 8;
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
      if (qi < a.lq && dd < D) v = *reinterpret_cast<const half8*>(qp + dd);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[sq][ks][j] = (_Float16)((float)v[j] * a.scale_log2);
    }
  }
  float16v o[NQ][2], s[NQ][2];
  half8 pf[NQ][4];
  float m_run[NQ];
#pragma unroll
  for (int sq = 0; sq < NQ; ++sq) {
    m_run[sq] = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[sq][t][r] = 0.0f; s[sq][t][r] = 0.0f; }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[sq][st][j] = (_Float16)0.0f;
  }

  // ---- LDS-DMA set-up (as in attention_kernel): K rows = keys, 5 valid 16-byte chunks per row; V^T rows = head dims
  const _Float16* kbase = a.k + (long long)b * a.lk * a.ldk + h * D;
  const _Float16* vbase = a.vt + ((long long)b * a.heads + h) * D * (long long)a.ldv;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kbase, (unsigned)(((long long)(a.lk - 1) * a.ldk + D) * 2));
  const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vbase, (unsigned)((long long)D * a.ldv * 2));
  const int l_row = lane >> 3, l_slot = lane & 7;
  // per wave and tile: 2 K instructions (16 keys) and 2 V^T instructions (16 head-dim rows)
  unsigned k_off[2], v_off[2];
  bool k_on[2], v_on[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + l_row;
    const int chunk = swz64(row, l_slot);
    k_on[j] = chunk < 5;                          // chunk 5 holds the constant-one column, chunks 6-7 are never read
    k_off[j] = (unsigned)(row * a.ldk * 2 + chunk * 16);
    v_on[j] = row < D;                            // row 40 = ones (the softmax denominator), rows 41-63 stay zero
    v_off[j] = (unsigned)(row * a.ldv * 2 + swz64(row, l_slot) * 16);
  }
  const unsigned k_step = (unsigned)(BKV * a.ldk * 2), v_step = BKV * 2;
  auto issue_k = [&](int stage, int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16* kd = Kbuf[stage] + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (k_on[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(kd + j * 512), 16, k_off[j] + tile * k_step, 0, 0, 0);
#endif
  };
  auto issue_v = [&](int stage, int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    _Float16* vd = Vbuf[stage] + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (v_on[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(vd + j * 512), 16, v_off[j] + tile * v_step, 0, 0, 0);
#endif
  };
  // constant parts of the stages, written once: K chunk 5 = (1, 0, ..., 0), V^T row 40 = ones, rows 41-63 (and all of the
  // stage the first O_B product reads before any V^T tile landed there) = zero
  for (int i = tid; i < 3 * STG / 8; i += 256) reinterpret_cast<uint4*>(&Vbuf[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  if (tid < 3 * 64) {
    const int stage = tid >> 6, row = tid & 63;
    half8 e;
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (_Float16)(j == 0 ? 1.0f : 0.0f);
    *reinterpret_cast<half8*>(&Kbuf[stage][row * 64 + swz64(row, 5) * 8]) = e;
  }
  if (tid < 3 * 8) {
    half8 one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one[j] = (_Float16)1.0f;
    *reinterpret_cast<half8*>(&Vbuf[tid >> 3][D * 64 + (tid & 7) * 8]) = one;
  }

  const int ntiles = a.lk / BKV;

  // ---- per-lane LDS offsets (halves): row ql, chunk (2ks + hh) / (2st + hh), swizzled; stages and the 32-row tile add immediates
  int k_lane[3], v_lane[4];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) k_lane[ks] = ql * 64 + swz64(ql, 2 * ks + hh) * 8;
#pragma unroll
  for (int st = 0; st < 4; ++st) v_lane[st] = ql * 64 + swz64(ql, 2 * st + hh) * 8;
  float16v zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
  float mxs[NQ];                                    // max of the S' tile in flight, per sub-tile

  // A segment = 14 slots.  Slot g: fetch the fragment of slot g+1, HQ MFMAs on the MATRIX half (slots 0-5: S' = K Q^T, slots
  // 6-13: O^T += V^T P^T), and on the VALU half a share of the 32 HQ exp2 + the conversions of the pairs finished one slot
  // earlier (an exp2 result needs a wait state before a VALU may read it); from slot 8 on, the maximum of the S' tile the
  // slots 0-5 produced.  The order is the source order: a sched_barrier closes every slot.
  auto frag = [&](int g, const _Float16* Ks, const _Float16* Vs) -> half8 {
    return g < 6 ? *reinterpret_cast<const half8*>(&Ks[(g & 1) * 32 * 64 + k_lane[g >> 1]])
                 : *reinterpret_cast<const half8*>(&Vs[((g - 6) & 1) * 32 * 64 + v_lane[(g - 6) >> 1]]);
  };
  auto segment = [&](auto hvc, const _Float16* Ks, const _Float16* Vs, auto maxc) {
    constexpr int HV = decltype(hvc)::value, HM = 1 - HV;       // VALU half, matrix half
    constexpr bool MAXON = decltype(maxc)::value;               // take the maximum of the S' tile this segment produces
    constexpr int NE = 32 * HQ;                                  // exp2 per segment
    half8 fr[3];                                                 // fragments are fetched two slots ahead
    fr[0] = frag(0, Ks, Vs);
    fr[1] = frag(1, Ks, Vs);
    auto slot = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      if constexpr (G + 2 < 14) fr[(G + 2) % 3] = frag(G + 2, Ks, Vs);
#pragma unroll
      for (int j = 0; j < HQ; ++j) {
        const int sq = HM * HQ + j;
        if constexpr (G < 6) {
          constexpr int ks = G >> 1, t = G & 1;
          s[sq][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[G % 3], qf[sq][ks], ks == 0 ? zero16 : s[sq][t], 0, 0, 0);
        } else {
          constexpr int st = (G - 6) >> 1, t = (G - 6) & 1;
          o[sq][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[G % 3], pf[sq][st], o[sq][t], 0, 0, 0);
        }
      }
      constexpr int E0 = G * NE / 14, E1 = (G + 1) * NE / 14, EP = G == 0 ? 0 : (G - 1) * NE / 14;
#pragma unroll
      for (int e = E0; e < E1; ++e) {               // P = exp2(S'), written back over S'
        const int sq = HV * HQ + e / 32, t = (e % 32) / 16, r = e % 16;
        s[sq][t][r] = __builtin_amdgcn_exp2f(s[sq][t][r]);
      }
#pragma unroll
      for (int pr = (EP + 1) / 2; pr < (G == 13 ? NE / 2 : (E0 + 1) / 2); ++pr) {   // pairs complete before this slot's exp2
        const int e = 2 * pr, sq = HV * HQ + e / 32, t = (e % 32) / 16, r = e % 16;
        pf[sq][t * 2 + (r >> 3)][r & 7] = (_Float16)s[sq][t][r];
        pf[sq][t * 2 + (r >> 3)][(r & 7) + 1] = (_Float16)s[sq][t][r + 1];
      }
      if constexpr (G >= 8 && MAXON) {              // 16 HQ max3 over the fresh S' of the matrix half, 6 slots
        constexpr int M0 = (G - 8) * 16 * HQ / 6, M1 = (G - 7) * 16 * HQ / 6;
#pragma unroll
        for (int mi = M0; mi < M1; ++mi) {
          const int j = mi / 16, pr = mi % 16, sq = HM * HQ + j;        // pair pr: (t = pr / 8, r = 2 (pr % 8))
          const float x0 = s[sq][pr >> 3][2 * (pr & 7)], x1 = s[sq][pr >> 3][2 * (pr & 7) + 1];
          mxs[sq] = pr == 0 ? fmaxf(x0, x1) : fmaxf(fmaxf(mxs[sq], x0), x1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... G>(std::integer_sequence<int, G...>) { (slot(std::integral_constant<int, G>{}), ...); }(std::make_integer_sequence<int, 14>{});
    // the fp16 weights are only read by the NEXT segment's MFMAs: without a use here the compiler sinks the whole exp2 / convert
    // chain of this segment into the next basic block (behind the rescale branch) and the interleave is gone
#pragma unroll
    for (int j = 0; j < HQ; ++j)
#pragma unroll
      for (int st = 0; st < 4; ++st) asm volatile("" : "+v"(pf[HV * HQ + j][st]));
    if constexpr (MAXON) {
#pragma unroll
      for (int j = 0; j < HQ; ++j) {                // the partner lane holds the other 32 keys of the query
        const int sq = HM * HQ + j;
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxs[sq]), __float_as_uint(mxs[sq]), false, false);
        mxs[sq] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
    }
  };
  // running maximum: the accumulators already hold score - m.  m only moves when a score exceeds it by more than 2^6 (or on
  // the first tile): then the tile in flight, O and the -m slot of Q are shifted by delta (wave-uniform rare branch)
  auto maxfix = [&](int half, bool first) {
#pragma unroll
    for (int j = 0; j < HQ; ++j) {
      const int sq = half * HQ + j;
      const bool need = first || mxs[sq] > 6.0f;
      if (__any(need)) {
        const float m_new = need ? (float)(_Float16)(m_run[sq] + mxs[sq]) : m_run[sq];      // stays an fp16 value
        const float delta = m_new - m_run[sq];
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[sq] = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[sq][t][r] -= delta; o[sq][t][r] *= alpha; }
        if (hh) qf[sq][2][0] = (_Float16)(-m_new);
      }
    }
  };

  // One pass over all key tiles.  FULL = false (first attempt): the running maximum is set from key tile 0 and then left alone --
  // no max3 / lane exchange / rescale test on the other 63 tiles (34 of ~165 VALU instructions per tile in a VALU-bound kernel).
  // That is exact unless a later score exceeds tile 0's maximum by 2^16 and the fp16 weight overflows; then the denominator comes
  // out non-finite, the workgroup notices (below) and repeats the pass with FULL = true: maximum tracked on every tile.
  auto run_pass = [&](auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;
    issue_k(0, 0);
    issue_v(0, 0);
    if (ntiles > 1) issue_k(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // S'_A(0) and its maximum, outside the pipeline
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const half8 kf = *reinterpret_cast<const half8*>(&Kbuf[0][t * 32 * 64 + k_lane[ks]]);
#pragma unroll
        for (int j = 0; j < HQ; ++j) s[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[j][ks], ks == 0 ? zero16 : s[j][t], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < HQ; ++j) {
      float mx = fmaxf(s[j][0][0], s[j][1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[j][0][r]), s[j][1][r]);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mxs[j] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    auto tile_body = [&](int t, auto s0c, auto firstc) {   // K(t), V^T(t) live in stage S0 = t % 3
      constexpr int S0 = decltype(s0c)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;
      constexpr bool FIRST = decltype(firstc)::value;      // tile 0: the maximum of both halves is always taken
      constexpr bool MAXA = FULL;                           // S'_A(t+1), produced in segment 2
      constexpr bool MAXB = FULL || FIRST;                  // S'_B(t), produced in segment 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of K(t+1), V^T(t) (issued one tile ago) have landed
      __builtin_amdgcn_s_barrier();                     // ... everyone's; and every wave is done with K(t-1), V^T(t-2)
      if (t + 2 < ntiles) issue_k(S2, t + 2);
      if (t + 1 < ntiles) issue_v(S1, t + 1);
      if constexpr (FULL || FIRST) maxfix(0, FIRST);
      __builtin_amdgcn_sched_barrier(0);
      // segment 1: softmax A(t)  ||  S'_B(t) = K(t) Q_B^T, O_B += V(t-1)^T P_B(t-1)   (t = 0: P_B = 0 and the stage holds finite data)
      segment(std::integral_constant<int, 0>{}, Kbuf[S0], Vbuf[S2], std::integral_constant<bool, MAXB>{});
      if constexpr (FULL || FIRST) maxfix(1, FIRST);
      __builtin_amdgcn_sched_barrier(0);
      // segment 2: softmax B(t)  ||  S'_A(t+1) = K(t+1) Q_A^T, O_A += V(t)^T P_A(t)    (after the last tile S'_A is never used)
      segment(std::integral_constant<int, 1>{}, Kbuf[S1], Vbuf[S0], std::integral_constant<bool, MAXA>{});
    };
    tile_body(0, std::integral_constant<int, 0>{}, std::true_type{});
    for (int t = 1; t < ntiles; t += 3) {
      tile_body(t, std::integral_constant<int, 1>{}, std::false_type{});
      if (t + 1 < ntiles) tile_body(t + 1, std::integral_constant<int, 2>{}, std::false_type{});
      if (t + 2 < ntiles) tile_body(t + 2, std::integral_constant<int, 0>{}, std::false_type{});
    }
    // O_B of the last tile
    {
      const int ls = (ntiles - 1) % 3;
      const _Float16* Vs = ls == 0 ? Vbuf[0] : ls == 1 ? Vbuf[1] : Vbuf[2];
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 vf = *reinterpret_cast<const half8*>(&Vs[t * 32 * 64 + v_lane[st]]);
#pragma unroll
          for (int j = 0; j < HQ; ++j) o[HQ + j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[HQ + j][st], o[HQ + j][t], 0, 0, 0);
        }
    }
  };
  if (a.no_spec) {
    run_pass(std::true_type{});
  } else {
    run_pass(std::false_type{});
    bool bad = false;
#pragma unroll
    for (int sq = 0; sq < NQ; ++sq) bad |= !(o[sq][1][4] < 3.0e38f);       // inf / NaN denominator (row 40 of O^T; every lane checks its own)
    if (__syncthreads_or(bad ? 1 : 0)) {
      // a weight overflowed fp16 somewhere in this workgroup: start over with the maximum tracked on every tile
#pragma unroll
      for (int sq = 0; sq < NQ; ++sq) {
        m_run[sq] = 0.0f;
        if (hh) qf[sq][2][0] = (_Float16)0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[sq][t][r] = 0.0f; s[sq][t][r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[sq][st][j] = (_Float16)0.0f;
      }
      run_pass(std::true_type{});
    }
  }

  // ---- normalise by the ones row (row 40 = register 4 of the second 32-row tile, hh = 0 lane) and store
#pragma unroll
  for (int sq = 0; sq < NQ; ++sq) {
    const float l_tot = __shfl(o[sq][1][4], ql);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + sq * 32 + ql;
    if (qi < a.lq) {
      _Float16* op = a.out + ((long long)b * a.lq + qi) * a.ldo + h * D;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = t * 32 + 8 * g + 4 * hh;
          if (dd < D) {
            half4 v = {(_Float16)(o[sq][t][g * 4 + 0] * inv), (_Float16)(o[sq][t][g * 4 + 1] * inv),
                       (_Float16)(o[sq][t][g * 4 + 2] * inv), (_Float16)(o[sq][t][g * 4 + 3] * inv)};
            *reinterpret_cast<half4*>(op + dd) = v;
          }
        }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// attention_wide_kernel<DT> -- wide heads (d = 32 DT <= 512): the VAE mid-block attention (one head, d = 512, 4096 tokens).
// Un-fused (r2) this was QK^T GEMM -> a 4096 x 4096 fp16 score matrix through HBM -> row softmax in place -> PV GEMM: 1.3 ms of
// every decode / encode, i.e. of every mask re-estimation.  O^T of 512 x 32 queries is 256 accumulator registers on the 32x32
// MFMA, so a wave owns 16 queries on v_mfma_f32_16x16x32_f16 instead:
//     S^T[16 keys, 16 q] += K[keys, 32 dd] . Q[q, 32 dd]^T        A = K fragment (LDS), B = Q (64 registers, loaded once)
//     O^T[16 dd, 16 q]  += V^T[dd, 32 keys] . P^T[32 keys, q]     A = V^T fragment (LDS), B = P straight from the S accumulators
// The C layout puts query (lane & 15) in all 4 registers of a lane and keys 4 (lane >> 4) + i in register i, so the softmax
// state is per lane (two cross-lane exchanges per tile for the maximum) and the P operand of a 32-key step is the lane's own
// 8 values of two key blocks -- keys {4g + i, 16 + 4g + i}; V^T is stored in exactly that key order (SD_EPI_PERM32_N of the
// projection GEMM), so its operand is ONE conflict-free 16-byte LDS read.  K and V^T tiles of 64 keys are 64 KB each: they live
// in two single-buffered LDS regions and alternate -- the next K tile streams in (LDS-DMA) while P.V runs, the next V^T tile
// while Q.K^T runs.  128 MFMAs (2048 cycles) against 16 exp2 per lane and tile: matrix-bound; every fragment feeds one MFMA,
// so the LDS read port (128 B / clk / CU) is the practical limit.
// NW = 8 waves (128 queries per workgroup, two waves per SIMD) when that still fills the chip: a wave's fragment reads and its four barrier waits per
// key tile run under its partner's MFMAs, and a K / V^T tile is fetched once per 128 queries (profiles/r05_notes.md 9).
template <int DT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attention_wide_kernel(AttnArgs a) {
  constexpr int D = 32 * DT;                       // head dim
  constexpr int KROW = D;                          // halves per K row in LDS (16-byte chunks: D / 8, swizzled with the key index)
  extern __shared__ __attribute__((aligned(1024))) _Float16 wide_smem[];
  _Float16* Kbuf = wide_smem;                      // [64 keys][D]
  _Float16* Vbuf = wide_smem + BKV * D;            // [D rows][64 keys]
  typedef float float4v __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (NW * 16) + wave * 16;
  const int n = lane & 15, g = lane >> 4;

  // Q fragments (B operand): lane (query n, chunk g) holds dd = 32 j + 8 g .. + 7
  half8 qf[DT];
  {
    const int qi = q0 + n;
    const _Float16* qp = a.q + ((long long)b * a.lq + (qi < a.lq ? qi : 0)) * a.ldq + h * D;
#pragma unroll
    for (int j = 0; j < DT; ++j) {
      if (qi < a.lq) {
        qf[j] = *reinterpret_cast<const half8*>(qp + 32 * j + 8 * g);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[j][e] = (_Float16)0.0f;
      }
    }
  }
  float4v o[2 * DT];                               // O^T blocks of 16 head dims
#pragma unroll
  for (int db = 0; db < 2 * DT; ++db) o[db] = float4v{0.f, 0.f, 0.f, 0.f};
  float m_run = -__builtin_inff(), l_run = 0.0f;

  // ---- LDS-DMA: one instruction = 1 KiB.  K: D * 2 bytes per key row -> D / 512 instructions per key, chunk c of key r lands
  // in slot c ^ (r & 15) of its 256-byte group (the 16 keys of a fragment read then hit 16 distinct bank groups).
  // V^T: 8 rows of 128 bytes per instruction, swz64 as in the other kernels.
  const _Float16* kbase = a.k + (long long)b * a.lk * a.ldk + h * D;
  const _Float16* vbase = a.vt + ((long long)b * a.heads + h) * D * (long long)a.ldv;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kbase, (unsigned)(((long long)(a.lk - 1) * a.ldk + D) * 2));
  const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vbase, (unsigned)((long long)D * a.ldv * 2));
  constexpr int KPI = D / 512 > 0 ? D / 512 : 1;   // K instructions per key row (D = 512: 1); smaller D: several rows per instruction
  constexpr int K_INSTR = BKV * D * 2 / 1024 / NW;  // per wave and tile
  constexpr int V_INSTR = D * 128 / 1024 / NW;
  static_assert(K_INSTR >= 1 && V_INSTR >= 1 && K_INSTR * NW * 1024 == BKV * D * 2 && V_INSTR * NW * 1024 == D * 128, "DMA pieces per wave");
  static_assert(D % 64 == 0 || D == 32, "row pitch");
  const unsigned k_step = (unsigned)(BKV * a.ldk * 2), v_step = BKV * 2;
  auto issue_k = [&](int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    int ln = lane;
    asm volatile("" : "+v"(ln));                   // the piece offsets are recomputed per tile, not carried through the loop (registers)
#pragma unroll
    for (int i = 0; i < K_INSTR; ++i) {
      const int idx = wave * K_INSTR + i;          // 1 KiB piece of the tile
      const int e16 = idx * 64 + ln;               // 16-byte element of the tile
      const int row = e16 / (D / 8), slot = e16 % (D / 8);
      const int chunk = (slot & ~15) | ((slot ^ row) & 15);
      const unsigned off = (unsigned)(row * a.ldk * 2 + chunk * 16) + tile * k_step;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lptr_t)(Kbuf + idx * 512), 16, off, 0, 0, 0);
    }
#endif
  };
  auto issue_v = [&](int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < V_INSTR; ++i) {
      const int idx = wave * V_INSTR + i;
      const int row = idx * 8 + (ln >> 3), slot = ln & 7;
      const unsigned off = (unsigned)(row * a.ldv * 2 + swz64(row, slot) * 16) + tile * v_step;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lptr_t)(Vbuf + idx * 512), 16, off, 0, 0, 0);
    }
#endif
  };
  (void)KPI;

  int koff[4], voff[2];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) koff[jj] = n * KROW + (((4 * jj + g) ^ n) & 15) * 8;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) voff[ks] = n * 64 + swz64(n, 4 * ks + g) * 8;      // (swz64 looks at the low row bits only: 16 db drops out)
  const int ntiles = a.lk / BKV;
  issue_k(0);
  issue_v(0);
  for (int t = 0; t < ntiles; ++t) {
    // K(t) was issued before V^T(t): all but the youngest V_INSTR pieces of this wave have landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V_INSTR) : "memory");
    __builtin_amdgcn_s_barrier();
    // ---- S^T = K Q^T, 4 key blocks of 16
    float4v s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s[kb] = float4v{0.f, 0.f, 0.f, 0.f};
    // fragment (key block kb, dd step j) of lane (n, g): row 16 kb + n, chunk c = 4 j + g in slot (c & ~15) | ((c ^ row) & 15); the swizzled
    // part only depends on j & 3, so FOUR lane offsets serve all 64 reads of a tile (the rest is an immediate)
#pragma unroll
    for (int j = 0; j < DT; ++j)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const half8 kf = *reinterpret_cast<const half8*>(Kbuf + koff[j & 3] + kb * 16 * KROW + (j >> 2) * 128);
        s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[j], s[kb], 0, 0, 0);
      }
    __builtin_amdgcn_s_barrier();                    // every wave is done with K(t)
    if (t + 1 < ntiles) issue_k(t + 1);
    // ---- online softmax: per lane 16 scores of query n (keys 16 kb + 4 g + i); the other 3 key quarters sit in lanes n + 16 g'
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
    for (int kb = 1; kb < 4; ++kb) mx = fmaxf(mx, fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;
    if (mx > m_run) {                                // lane-divergent is fine: all state here is per lane
      const float alpha = __builtin_amdgcn_exp2f(m_run - mx);    // 0 on the first tile
      m_run = mx;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2 * DT; ++db) o[db] *= alpha;
    }
    half8 pf[2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], a.scale_log2, -m_run));
        l_run += p;
        pf[kb >> 1][(kb & 1) * 4 + i] = (_Float16)p;
      }
    // V^T(t) has landed (the K(t+1) pieces issued above may still be in flight)
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K_INSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- O^T += V^T P^T: 2 key steps of 32, 2 DT blocks of 16 head dims
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2 * DT; ++db) {
        const half8 vf = *reinterpret_cast<const half8*>(Vbuf + voff[ks] + db * 16 * 64);
        o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[ks], o[db], 0, 0, 0);
      }
    __builtin_amdgcn_s_barrier();                    // every wave is done with V^T(t)
    if (t + 1 < ntiles) issue_v(t + 1);
  }
  // ---- denominator over the 4 key quarters, normalise, store: lane (query n, quarter g) owns dd = 16 db + 4 g + i
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_run;
  const int qi = q0 + n;
  if (qi < a.lq) {
    _Float16* op = a.out + ((long long)b * a.lq + qi) * a.ldo + h * D;
#pragma unroll
    for (int db = 0; db < 2 * DT; ++db) {
      half4 v = {(_Float16)(o[db][0] * inv), (_Float16)(o[db][1] * inv), (_Float16)(o[db][2] * inv), (_Float16)(o[db][3] * inv)};
      *reinterpret_cast<half4*>(op + db * 16 + 4 * g) = v;
    }
  }
}

}  // namespace sd

using namespace sd;

extern "C" int sd_attention_f16(const void* q, const void* k, const void* vt, void* out, int batch, int heads, int lq,
                                int lk, int d, int ldq, int ldk, int ldv, int ldo, float scale, int vt_perm16, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_ATTN;
    r.p[0] = (void*)q; r.p[1] = (void*)k; r.p[2] = (void*)vt; r.p[3] = out;
    const int64_t is[10] = {batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, vt_perm16};
    for (int j = 0; j < 10; ++j) r.i[j] = is[j];
    r.f[0] = scale;
    return sd::plan_record(r);
  }
  if (!q || !k || !vt || !out) return fail(COMA_E_INVALID, "sd_attention_f16: null pointer");
  if (batch <= 0 || heads <= 0 || lq <= 0 || lk <= 0) return fail(COMA_E_INVALID, "sd_attention_f16: bad sizes");
  if (d % 8 || d <= 0 || d > 160) return fail(COMA_E_INVALID, "sd_attention_f16: head dim %d unsupported (multiple of 8, <= 160)", d);
  if (ldq < heads * d || ldk < heads * d || ldo < heads * d || ldv < ((lk + 7) & ~7) || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
    return fail(COMA_E_INVALID, "sd_attention_f16: bad leading dimensions");
  if (vt_perm16 && (ldv % 16 || ldv < ((lk + 15) & ~15))) return fail(COMA_E_INVALID, "sd_attention_f16: vt_perm16 needs ldv a multiple of 16 covering lk");
  if ((long long)lk * ldk * 2 >= 0x80000000LL || (long long)d * ldv * 2 >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_attention_f16: K / V^T slice of one (batch, head) exceeds 2 GiB");
  AttnArgs a;
  a.q = (const _Float16*)q; a.k = (const _Float16*)k; a.vt = (const _Float16*)vt; a.out = (_Float16*)out;
  a.heads = heads; a.lq = lq; a.lk = lk; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.scale_log2 = scale * 1.4426950408889634f;
  static const int spec_env = [] { const char* e = getenv("SD_ATTN_SPEC"); return e ? atoi(e) : 1; }();
  a.no_spec = !spec_env;
  hipStream_t s = (hipStream_t)stream;
  // software-pipelined kernel: d = 40, whole key tiles (at least two), V^T key-permuted; by default only when 256-query blocks
  // fill the chip.  vt_perm16 bit 1 forces it wherever it is legal, bit 2 forbids it (tests / A-B timing); SD_ATTN_V=1 forbids
  // it process-wide.
  static const int sp_mode = [] { const char* e = getenv("SD_ATTN_V"); return e ? atoi(e) : 2; }();
  const bool sp_legal = d == 40 && (vt_perm16 & 1) && lk % BKV == 0 && lk >= 2 * BKV;
  // (r5: from ONE block per CU on -- UNet batch 2, one image per pipeline call: 75 us against 97 us for the r2 kernel; it was two per CU)
  const bool sp_auto = sp_mode >= 2 && !(vt_perm16 & 4) && (long long)batch * heads * ((lq + 255) / 256) >= 256;
  if (sp_legal && ((vt_perm16 & 2) || sp_auto)) {
    // (HQ = 2, 128 queries per wave, measured slower -- 585 vs 487 us -- and is not instantiated: it needs 460 registers)
    dim3 g2((unsigned)((lq + 255) / 256), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL((attention_sp_kernel<1>), g2, dim3(256), 0, s, a);
    return check_launch("attention_sp_kernel");
  }
  vt_perm16 &= 1;
  // 64 queries per wave once there are enough blocks to fill the chip -- unless there are only a few key tiles (the 77-token cross
  // attention): then a block is all prologue / epilogue latency and more resident blocks (QT = 1: ~100 registers) hide it better
  static const int xqt1 = [] { const char* e = getenv("SD_ATTN_XQT1"); return e ? atoi(e) : 1; }();   // 38.5 -> 34.9 us at lq = 4096, lk = 77
  // r5: ... and at d = 80 (the 32 x 32 level: 73.4 -> 64.4 us; 254 registers) once 256-query blocks fill the chip twice
  const bool two80 = d == 80 && lq >= 1024 && lk >= 256 && (long long)batch * heads * ((lq + 255) / 256) >= 512;
  const bool two = (lq >= 1024 && d == 40 && !(xqt1 && lk <= 128)) || two80;
  dim3 grid((unsigned)((lq + (two ? 255 : 127)) / (two ? 256 : 128)), (unsigned)heads, (unsigned)batch);
#define SD_ATTN_LAUNCH(KS, DVT, QT, ONES)                                                                       \
  do {                                                                                                          \
    if (vt_perm16) hipLaunchKernelGGL((attention_kernel<KS, DVT, QT, ONES, true>), grid, dim3(256), 0, s, a);  \
    else hipLaunchKernelGGL((attention_kernel<KS, DVT, QT, ONES, false>), grid, dim3(256), 0, s, a);           \
  } while (0)
  if (d == 40 && two) SD_ATTN_LAUNCH(3, 2, 2, 40);
  else if (d == 40) SD_ATTN_LAUNCH(3, 2, 1, 40);
  else if (d <= 48) SD_ATTN_LAUNCH(3, 2, 1, -1);
  else if (d <= 64) SD_ATTN_LAUNCH(4, 2, 1, -1);
  else if (two80) SD_ATTN_LAUNCH(5, 3, 2, 80);
  else if (d == 80) SD_ATTN_LAUNCH(5, 3, 1, 80);     // r5: V^T rows 80-95 are padding at d = 80 too -> the ones row gives the denominator (C = 640 level)
  else if (d <= 80) SD_ATTN_LAUNCH(5, 3, 1, -1);
  else if (d <= 96) SD_ATTN_LAUNCH(6, 3, 1, -1);
  else if (d <= 128) SD_ATTN_LAUNCH(8, 4, 1, -1);
  else SD_ATTN_LAUNCH(10, 5, 1, -1);
#undef SD_ATTN_LAUNCH
  return check_launch("attention_kernel");
}

extern "C" int sd_attention_wide_f16(const void* q, const void* k, const void* vt, void* out, int batch, int heads, int lq, int lk, int d,
                                     int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  if (sd::plan_recording()) {
    sd::PlanRec r{};
    r.kind = sd::PK_ATTN_WIDE;
    r.p[0] = (void*)q; r.p[1] = (void*)k; r.p[2] = (void*)vt; r.p[3] = out;
    const int64_t is[9] = {batch, heads, lq, lk, d, ldq, ldk, ldv, ldo};
    for (int j = 0; j < 9; ++j) r.i[j] = is[j];
    r.f[0] = scale;
    return sd::plan_record(r);
  }
  if (!q || !k || !vt || !out) return fail(COMA_E_INVALID, "sd_attention_wide_f16: null pointer");
  if (batch <= 0 || heads <= 0 || lq <= 0 || lk <= 0 || lk % BKV) return fail(COMA_E_INVALID, "sd_attention_wide_f16: bad sizes (lk must be a multiple of 64)");
  if (d != 512 && d != 256 && d != 128) return fail(COMA_E_INVALID, "sd_attention_wide_f16: head dim %d unsupported (128, 256 or 512)", d);
  if (ldq < heads * d || ldk < heads * d || ldo < heads * d || ldv < lk || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
    return fail(COMA_E_INVALID, "sd_attention_wide_f16: bad leading dimensions");
  if ((long long)lk * ldk * 2 >= 0x80000000LL || (long long)d * ldv * 2 >= 0x80000000LL)
    return fail(COMA_E_INVALID, "sd_attention_wide_f16: K / V^T slice of one (batch, head) exceeds 2 GiB");
  AttnArgs a;
  a.q = (const _Float16*)q; a.k = (const _Float16*)k; a.vt = (const _Float16*)vt; a.out = (_Float16*)out;
  a.heads = heads; a.lq = lq; a.lk = lk; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.no_spec = 0;
  const size_t lds = (size_t)2 * BKV * d * sizeof(_Float16);          // K tile + V^T tile
  hipStream_t s = (hipStream_t)stream;
  // 128 queries per workgroup (8 waves) from one full round of such workgroups on (VAE batch 8: 470 -> 300 us; half a round -- batch 4 -- is a tie,
  // below it the 64-query form wins by 8 - 10 %); SD_WIDE_NW = 4 / 8 forces either (A/B timing)
  static const int nw_env = [] { const char* e = getenv("SD_WIDE_NW"); return e ? atoi(e) : 0; }();
  const bool eight = nw_env ? nw_env == 8 : (long long)batch * heads * ((lq + 127) / 128) >= 256;
#define SD_WIDE_LAUNCH(DT, NW)                                                                                                \
  do {                                                                                                                        \
    static coma::LdsOptIn lds_opt;                                                                                            \
    if (int rc = coma::opt_in_lds(lds_opt, reinterpret_cast<const void*>(attention_wide_kernel<DT, NW>), lds, "sd_attention_wide_f16")) return rc; \
    dim3 grid((unsigned)((lq + NW * 16 - 1) / (NW * 16)), (unsigned)heads, (unsigned)batch);                                  \
    hipLaunchKernelGGL((attention_wide_kernel<DT, NW>), grid, dim3(NW * 64), lds, s, a);                                      \
  } while (0)
  if (d == 512) { if (eight) SD_WIDE_LAUNCH(16, 8); else SD_WIDE_LAUNCH(16, 4); }
  else if (d == 256) { if (eight) SD_WIDE_LAUNCH(8, 8); else SD_WIDE_LAUNCH(8, 4); }
  else { if (eight) SD_WIDE_LAUNCH(4, 8); else SD_WIDE_LAUNCH(4, 4); }
#undef SD_WIDE_LAUNCH
  return check_launch("attention_wide_kernel");
}
