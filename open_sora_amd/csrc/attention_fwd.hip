// Flash-attention forward (non-causal) for gfx950, bf16 in/out, f32 online softmax.
//
// Workgroup = 8 waves = 256 query rows (32 per wave); KV tile = 64 keys; one barrier per tile.
// Both matrix products are issued "swapped" on v_mfma_f32_32x32x16_bf16 so that a lane owns ONE query:
//     S^T[key][q]  = K[key][:]  . Q^T      (A = K fragment from LDS,   B = Q fragment in registers)
//     O^T[d][q]   += V^T[d][key] . P^T     (A = V^T fragment from LDS, B = P in registers)
// -> the 32 scores a lane holds all belong to its query: row max / row sum / rescale are lane-local
//    plus ONE exchange with lane^32;  the f32->bf16 P values are already in the register order the
//    second MFMA wants for its B operand PROVIDED the V^T rows are read in the same key order.  That
//    order (k0-3, k8-11 | k4-7, k12-15 per 16 keys) is baked into the VT buffer by osk_v_transpose_bf16,
//    so there is no cross-lane shuffle, no LDS round trip for P and no transposed LDS read.
//
// LDS: K tile [64][hd(+pad)] and V^T tile [hd][64 keys], row stride padded by 16 B so that the 32-row x
// 16-B fragment reads (ds_read_b128) are bank-conflict free; double buffered; tiles are staged through
// registers (global_load_dwordx4 issued before the tile's math, ds_write_b128 after it: HBM/L2 latency
// hides under the MFMAs, guide T14).
//
// head_dim 72 (DiT-XL geometry) is zero-padded to 80 for the QK^T contraction (5 MFMA k-steps) and to 96
// output rows for PV (3 MFMA row tiles).
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 4 * B * H * Lq * Lk * hd (QK^T + PV, not halved).
#include "osk_common.h"
#include "../../include/osk.h"

namespace {

template <int HD>
struct Cfg {
  static constexpr int HDP = (HD + 15) / 16 * 16;
  static constexpr int HDV = (HD + 31) / 32 * 32;
  static constexpr int NKS = HDP / 16;
  static constexpr int NDT = HDV / 32;
  static constexpr int KROW = HDP * 2 + 16;
  static constexpr int VROW = 64 * 2 + 16;
  static constexpr int KTILE = 64 * KROW;
  static constexpr int VTILE = HDV * VROW;
  static constexpr int BUF = KTILE + VTILE;
  static constexpr int SMEM = 2 * BUF;
  static constexpr int CPR = HD / 8;
  static constexpr int NKC = 64 * CPR;              // 16-B chunks in a K tile
  static constexpr int NVC = HD * 8;                // 16-B chunks in a V^T tile
  static constexpr int KIT = (NKC + 511) / 512;
  static constexpr int VIT = (NVC + 511) / 512;
};

struct AttnParams {
  const unsigned short* q;
  int64_t qbs, qrs;
  const unsigned short* k;
  int64_t kss, kbs, krs;
  const unsigned short* vt;
  int64_t vtss;
  unsigned short* out;
  int64_t obs, ors;
  float* lse;
  int B, H, Lq, n_seg, seg_len, seg_lp, tps;
  float sc;  // softmax scale * log2(e)
};

template <int HD>
__global__ void __launch_bounds__(512) attn_fwd_kernel(const AttnParams p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int nbh = p.B * p.H;
  const int bh = blockIdx.x % nbh, qb = blockIdx.x / nbh;
  const int b = bh / p.H, h = bh - b * p.H;

  // zero the whole LDS once: pad chunks / pad rows are never overwritten by the staging below
  for (int i = tid; i < C::SMEM / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (B operand): Q[q][ks*16 + hi*8 .. +8]
  const int qi = qb * 256 + wave * 32 + l31;
  const int qc = qi < p.Lq ? qi : p.Lq - 1;
  const unsigned short* qrow = p.q + b * p.qbs + (int64_t)qc * p.qrs + h * HD;
  bf16x8_t qf[C::NKS];
#pragma unroll
  for (int ks = 0; ks < C::NKS; ++ks) {
    const int e0 = ks * 16 + hi * 8;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (e0 < HD) u = *reinterpret_cast<const uint4*>(qrow + e0);
    qf[ks] = __builtin_bit_cast(bf16x8_t, u);
  }

  // ---- staging slots of this thread
  int k_row[C::KIT], k_c[C::KIT], v_d[C::VIT], v_c[C::VIT];
#pragma unroll
  for (int it = 0; it < C::KIT; ++it) {
    const int i = tid + it * 512;
    k_row[it] = i / C::CPR;
    k_c[it] = i - k_row[it] * C::CPR;
  }
#pragma unroll
  for (int it = 0; it < C::VIT; ++it) {
    const int i = tid + it * 512;
    v_d[it] = i >> 3;
    v_c[it] = i & 7;
  }
  const unsigned short* kbase_b = p.k + b * p.kbs + h * HD;
  const unsigned short* vbase_bh = p.vt + (int64_t)bh * HD * p.seg_lp;

  // staging registers as named scalars (arrays written under a divergent guard were placed in scratch)
  uint4 rk0 = make_uint4(0, 0, 0, 0), rk1 = rk0, rv0 = rk0, rv1 = rk0;
  static_assert(C::KIT <= 2 && C::VIT <= 2, "staging assumes <= 1024 chunks per tile");
  constexpr bool K1_FULL = C::NKC >= 1024, V1_FULL = C::NVC >= 1024;  // slot 1 unguarded?
  const bool k1_on = K1_FULL || (tid + 512 < C::NKC);
  const bool v1_on = V1_FULL || (tid + 512 < C::NVC);
#define STAGE_ISSUE(T)                                                                               \
  {                                                                                                  \
    const int s_ = (T) / p.tps, tt_ = (T) - s_ * p.tps;                                              \
    const int key0_ = tt_ * 64;                                                                      \
    const int last_ = p.seg_len - 1;                                                                 \
    const unsigned short* kb_ = kbase_b + s_ * p.kss;                                                \
    const unsigned short* vb_ = vbase_bh + s_ * p.vtss + key0_;                                      \
    {                                                                                                \
      int key_ = key0_ + k_row[0];                                                                   \
      key_ = key_ < last_ ? key_ : last_;                                                            \
      rk0 = *reinterpret_cast<const uint4*>(kb_ + (int64_t)key_ * p.krs + k_c[0] * 8);               \
      rv0 = *reinterpret_cast<const uint4*>(vb_ + (int64_t)v_d[0] * p.seg_lp + v_c[0] * 8);         \
    }                                                                                                \
    if constexpr (C::KIT > 1) {                                                                      \
      if (k1_on) {                                                                                   \
        int key_ = key0_ + k_row[C::KIT - 1];                                                        \
        key_ = key_ < last_ ? key_ : last_;                                                          \
        rk1 = *reinterpret_cast<const uint4*>(kb_ + (int64_t)key_ * p.krs + k_c[C::KIT - 1] * 8);    \
      }                                                                                              \
    }                                                                                                \
    if constexpr (C::VIT > 1) {                                                                      \
      if (v1_on)                                                                                     \
        rv1 = *reinterpret_cast<const uint4*>(vb_ + (int64_t)v_d[C::VIT - 1] * p.seg_lp +            \
                                              v_c[C::VIT - 1] * 8);                                  \
    }                                                                                                \
  }
#define STAGE_COMMIT(BUFI)                                                                           \
  {                                                                                                  \
    unsigned char* kb_ = smem + (BUFI) * C::BUF;                                                     \
    unsigned char* vb_ = kb_ + C::KTILE;                                                             \
    *reinterpret_cast<uint4*>(kb_ + k_row[0] * C::KROW + k_c[0] * 16) = rk0;                         \
    *reinterpret_cast<uint4*>(vb_ + v_d[0] * C::VROW + v_c[0] * 16) = rv0;                           \
    if constexpr (C::KIT > 1) {                                                                      \
      if (k1_on)                                                                                     \
        *reinterpret_cast<uint4*>(kb_ + k_row[C::KIT - 1] * C::KROW + k_c[C::KIT - 1] * 16) = rk1;   \
    }                                                                                                \
    if constexpr (C::VIT > 1) {                                                                      \
      if (v1_on)                                                                                     \
        *reinterpret_cast<uint4*>(vb_ + v_d[C::VIT - 1] * C::VROW + v_c[C::VIT - 1] * 16) = rv1;     \
    }                                                                                                \
  }

  f32x16_t o[C::NDT];
#pragma unroll
  for (int d = 0; d < C::NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nt = p.n_seg * p.tps;
  __syncthreads();  // zero-fill done
  STAGE_ISSUE(0);
  STAGE_COMMIT(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nt;
    if (more) STAGE_ISSUE(t + 1);
    const unsigned char* kb = smem + cur * C::BUF;
    const unsigned char* vb = kb + C::KTILE;

    // ---- S^T = K . Q^T : two 32-key sub-tiles
    f32x16_t s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + (t2 * 32 + l31) * C::KROW + (ks * 2 + hi) * 16);
        s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t2], 0, 0, 0);
      }
    }
    // ---- mask the ragged tail of a segment
    {
      const int sidx = t / p.tps, tt = t - sidx * p.tps;
      const int valid = p.seg_len - tt * 64;
      if (valid < 64) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kl >= valid) s[t2][r] = -INFINITY;
          }
      }
    }
    // ---- online softmax (lane-local; one exchange with the other half-wave)
    float mt = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
    const float msc = m_new * p.sc;
    float rs = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t2][r] * p.sc - msc);
        s[t2][r] = e;
        rs += e;
      }
    l_run = l_run * alpha + rs;
    if (!__all(m_new == m_run)) {
#pragma unroll
      for (int d = 0; d < C::NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    m_run = m_new;
    // ---- P^T fragments (B operand): group g = 16 keys = regs [8*(g&1), +8) of sub-tile g>>1
    bf16x8_t pb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int t2 = g >> 1, r0 = (g & 1) * 8;
      uint4 u;
      u.x = pack_bf16x2(s[t2][r0 + 0], s[t2][r0 + 1]);
      u.y = pack_bf16x2(s[t2][r0 + 2], s[t2][r0 + 3]);
      u.z = pack_bf16x2(s[t2][r0 + 4], s[t2][r0 + 5]);
      u.w = pack_bf16x2(s[t2][r0 + 6], s[t2][r0 + 7]);
      pb[g] = __builtin_bit_cast(bf16x8_t, u);
    }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int d = 0; d < C::NDT; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vb + (d * 32 + l31) * C::VROW + (g * 2 + hi) * 16);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[g], o[d], 0, 0, 0);
      }
    }
    if (more) STAGE_COMMIT(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qi < p.Lq) {
    unsigned short* orow = p.out + b * p.obs + (int64_t)qi * p.ors + h * HD;
#pragma unroll
    for (int d = 0; d < C::NDT; ++d) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d0 = d * 32 + qd * 8 + hi * 4;
        if (d0 < HD) {
          uint2 u;
          u.x = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
          u.y = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d0) = u;
        }
      }
    }
    if (p.lse && hi == 0)
      p.lse[(int64_t)bh * p.Lq + qi] = (m_run * p.sc + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
}

template <int HD>
int launch(const AttnParams& p, hipStream_t st) {
  using C = Cfg<HD>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<HD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nqb = (p.Lq + 255) / 256;
  dim3 grid(nqb * p.B * p.H), block(512);
  hipLaunchKernelGGL(attn_fwd_kernel<HD>, grid, block, C::SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int osk_attention_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                      const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                      int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                      void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                      float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                      float scale, void* stream) {
  if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0) return OSK_EINVAL;
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_seg_stride & 7) || (k_batch_stride & 7) ||
      (k_row_stride & 7) || (vt_seg_stride & 7) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7)) return OSK_EINVAL;
  AttnParams p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kss = k_seg_stride; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = (const unsigned short*)vt; p.vtss = vt_seg_stride;
  p.out = (unsigned short*)out; p.obs = o_batch_stride; p.ors = o_row_stride;
  p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.sc = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 64: return launch<64>(p, st);
    case 72: return launch<72>(p, st);
    case 128: return launch<128>(p, st);
    default: return OSK_EUNSUPPORTED;
  }
}
