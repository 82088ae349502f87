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
// Block = 4 waves x 32 queries = 128 queries of one (batch, head); K / V^T tiles of 64 keys staged in LDS with
// padded rows (stride/16 B odd for the b128 reads, stride/8 B odd for the b64 reads -> conflict-free).
#include <hip/hip_fp16.h>

#include "common.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::check_launch;
using coma::fail;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int BKV = 64;            // keys per tile
constexpr int VT_STRIDE = BKV + 4; // halves (136 B rows)

struct AttnArgs {
  const _Float16* q;
  const _Float16* k;
  const _Float16* vt;
  _Float16* out;
  int heads, lq, lk, d;
  int ldq, ldk, ldv, ldo;
  float scale_log2;   // scale * log2(e)
};

template <int KS, int DVT>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  constexpr int DQK = KS * 16;
  constexpr int K_STRIDE = DQK + 8;     // halves
  constexpr int DV = DVT * 32;
  // double-buffered tiles: one barrier per key tile (stage t+1 is written while stage t is still being read)
  __shared__ __attribute__((aligned(16))) _Float16 Kbuf[2][BKV * K_STRIDE];
  __shared__ __attribute__((aligned(16))) _Float16 Vbuf[2][DV * VT_STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int ql = lane & 31, hh = lane >> 5;
  const int d = a.d;

  // every K slot [64][DQK] and V^T slot [DV][64] is rewritten by each tile's staging pass (zeros beyond d / lk),
  // so pad columns and pad rows always hold finite values

  // Q fragments (B operand): lane (query ql, half hh) holds dd = ks*16 + hh*8 .. +7
  half8 qf[KS];
  {
    const int qi = q0 + ql;
    const _Float16* qp = a.q + ((long long)b * a.lq + (qi < a.lq ? qi : 0)) * a.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = ks * 16 + hh * 8;
      if (qi < a.lq && dd < d) {
        qf[ks] = *reinterpret_cast<const half8*>(qp + dd);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (_Float16)0.0f;
      }
    }
  }

  float16v o[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m_run = -__builtin_inff(), l_run = 0.0f;

  const _Float16* kbase = a.k + (long long)b * a.lk * a.ldk + h * d;
  const _Float16* vbase = a.vt + ((long long)b * a.heads + h) * d * (long long)a.ldv;
  const int kchunks = d / 8;            // 16-byte chunks per K row
  __syncthreads();

  // register-staged software pipeline: the global loads of tile t+1 are in flight while tile t is multiplied
  constexpr int KCH = DQK / 8;                       // 16-byte chunk slots per K row (>= d/8)
  constexpr int K_ITEMS = (BKV * KCH + 255) / 256;
  constexpr int V_ITEMS = DV * (BKV / 8) / 256;      // = DVT
  uint4 rk[K_ITEMS];
  half8 rv[V_ITEMS];
  auto load_regs = [&](int key0) {
#pragma unroll
    for (int i = 0; i < K_ITEMS; ++i) {
      const int it = tid + i * 256;
      const int row = it / KCH, ch = it - row * KCH;
      const int key = key0 + row;
      rk[i] = (row < BKV && ch < kchunks && key < a.lk)
                  ? *reinterpret_cast<const uint4*>(kbase + (long long)key * a.ldk + ch * 8)
                  : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < V_ITEMS; ++i) {
      const int it = tid + i * 256;
      const int row = it >> 3, ch = it & 7;
      const int key = key0 + ch * 8;
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
      if (row < d) {
        if (key + 8 <= a.lk) {
          v = *reinterpret_cast<const half8*>(vbase + (long long)row * a.ldv + key);
        } else if (key < a.lk) {
          const _Float16* p = vbase + (long long)row * a.ldv + key;
          for (int j = 0; j < a.lk - key; ++j) v[j] = p[j];
        }
      }
      rv[i] = v;
    }
  };
  auto store_regs = [&](int buf) {
    _Float16* Ks = Kbuf[buf];
    _Float16* Vs = Vbuf[buf];
#pragma unroll
    for (int i = 0; i < K_ITEMS; ++i) {
      const int it = tid + i * 256;
      const int row = it / KCH, ch = it - row * KCH;
      if (row < BKV) *reinterpret_cast<uint4*>(&Ks[row * K_STRIDE + ch * 8]) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < V_ITEMS; ++i) {
      const int it = tid + i * 256;
      const int row = it >> 3, ch = it & 7;
      // 8 halves = two 8-byte LDS writes (rows are 8-byte aligned, not 16)
      half4 lo = {rv[i][0], rv[i][1], rv[i][2], rv[i][3]}, hi = {rv[i][4], rv[i][5], rv[i][6], rv[i][7]};
      *reinterpret_cast<half4*>(&Vs[row * VT_STRIDE + ch * 8]) = lo;
      *reinterpret_cast<half4*>(&Vs[row * VT_STRIDE + ch * 8 + 4]) = hi;
    }
  };

  load_regs(0);
  int stage = 0;
  for (int key0 = 0; key0 < a.lk; key0 += BKV, stage ^= 1) {
    store_regs(stage);
    __syncthreads();            // stage `stage` is complete; every wave finished reading stage^1 one iteration ago
    if (key0 + BKV < a.lk) load_regs(key0 + BKV);
    const _Float16* Ks = Kbuf[stage];
    const _Float16* Vs = Vbuf[stage];

    // ---- S^T = K Q^T  (two 32-key tiles)
    float16v s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half8 kf = *reinterpret_cast<const half8*>(&Ks[(t * 32 + ql) * K_STRIDE + ks * 16 + hh * 8]);
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[t], 0, 0, 0);
      }
    }
    // ---- online softmax (per-lane scalars; the partner lane holds the other 32 keys of this query).
    // Scores stay RAW in the accumulators; scale*log2(e) rides in the FMA that forms the exp2 argument.  The
    // running max is only raised when it would grow by more than RESCALE_THR (log2 units): P may then reach
    // 2^THR (fine in fp16/fp32) and the O / l rescale becomes a rare, wave-uniformly skipped branch.
    if (key0 + BKV > a.lk) {   // wave-uniform: only the last tile can be ragged
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          s[t][r] = key < a.lk ? s[t][r] : -__builtin_inff();
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2;   // scale > 0: max commutes with the scaling
    constexpr float RESCALE_THR = 6.0f;
    const bool need = mx > m_run + RESCALE_THR;          // first tile: m_run = -inf -> true
    if (__any(need)) {
      const float m_new = need ? mx : m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 1 when unchanged, 0 on the first tile
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 c2 = {a.scale_log2, a.scale_log2}, nm2 = {-m_run, -m_run};
    f32x2 ps2 = {0.0f, 0.0f};
    half8 pf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s[t][r], s[t][r + 1]};
        const f32x2 e = __builtin_elementwise_fma(sv, c2, nm2);
        const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
        ps2 += p;
        pf[t * 2 + (r >> 3)][r & 7] = (_Float16)p.x;
        pf[t * 2 + (r >> 3)][(r & 7) + 1] = (_Float16)p.y;
      }
    l_run += ps2.x + ps2.y;
    // ---- O^T += V^T P^T, 16 keys per MFMA, keys in accumulator-register order
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int base = (st >> 1) * 32 + (st & 1) * 16 + 4 * hh;
#pragma unroll
      for (int t = 0; t < DVT; ++t) {
        const _Float16* vp = &Vs[(t * 32 + ql) * VT_STRIDE + base];
        half4 lo = *reinterpret_cast<const half4*>(vp);
        half4 hi = *reinterpret_cast<const half4*>(vp + 8);
        half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], o[t], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane (query, half) owns dd = t*32 + 8*(r>>2) + 4*hh + (r&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + ql;
  if (qi < a.lq) {
    _Float16* op = a.out + ((long long)b * a.lq + qi) * a.ldo + h * d;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = t * 32 + 8 * g + 4 * hh;
        if (dd < d) {
          half4 v = {(_Float16)(o[t][g * 4 + 0] * inv), (_Float16)(o[t][g * 4 + 1] * inv),
                     (_Float16)(o[t][g * 4 + 2] * inv), (_Float16)(o[t][g * 4 + 3] * inv)};
          *reinterpret_cast<half4*>(op + dd) = v;
        }
      }
  }
}

}  // namespace sd

using namespace sd;

extern "C" int sd_attention_f16(const void* q, const void* k, const void* vt, void* out, int batch, int heads, int lq,
                                int lk, int d, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  if (!q || !k || !vt || !out) return fail(COMA_E_INVALID, "sd_attention_f16: null pointer");
  if (batch <= 0 || heads <= 0 || lq <= 0 || lk <= 0) return fail(COMA_E_INVALID, "sd_attention_f16: bad sizes");
  if (d % 8 || d <= 0 || d > 160) return fail(COMA_E_INVALID, "sd_attention_f16: head dim %d unsupported (multiple of 8, <= 160)", d);
  if (ldq < heads * d || ldk < heads * d || ldo < heads * d || ldv < lk || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
    return fail(COMA_E_INVALID, "sd_attention_f16: bad leading dimensions");
  AttnArgs a;
  a.q = (const _Float16*)q; a.k = (const _Float16*)k; a.vt = (const _Float16*)vt; a.out = (_Float16*)out;
  a.heads = heads; a.lq = lq; a.lk = lk; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.scale_log2 = scale * 1.4426950408889634f;
  dim3 grid((unsigned)((lq + 127) / 128), (unsigned)heads, (unsigned)batch);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 48) hipLaunchKernelGGL((attention_kernel<3, 2>), grid, dim3(256), 0, s, a);
  else if (d <= 64) hipLaunchKernelGGL((attention_kernel<4, 2>), grid, dim3(256), 0, s, a);
  else if (d <= 80) hipLaunchKernelGGL((attention_kernel<5, 3>), grid, dim3(256), 0, s, a);
  else if (d <= 96) hipLaunchKernelGGL((attention_kernel<6, 3>), grid, dim3(256), 0, s, a);
  else if (d <= 128) hipLaunchKernelGGL((attention_kernel<8, 4>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attention_kernel<10, 5>), grid, dim3(256), 0, s, a);
  return check_launch("attention_kernel");
}
