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
#include "attention_params.h"
#include "../../include/osk.h"

namespace {
using osk_attn::AttnParams;

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

// =====================================================================================================
// Software-pipelined (head_dim 64; head_dim 72 / 128 run the hand-scheduled kernels of attention_asm72.hip / attention_asm128.hip).  In iteration t a wave issues the QK^T MFMAs of tile t+1 BEFORE the softmax of
// tile t, so the softmax VALU work (max / exp2 / pack) of one tile runs in the shadow of the next tile's
// matrix work instead of both waves of a SIMD alternating "all-MFMA" and "all-VALU" phases in lockstep
// behind the per-tile barrier (v1: MFMA pipe idle during every softmax).  K is therefore staged two tiles
// ahead, V^T one tile ahead; still one barrier per tile, same LDS footprint.
// Row max is exchanged with v_permlane32_swap (VALU) instead of ds_bpermute.  For head_dim 72 the row sum
// comes out of the PV MFMA for free: padding row 72 of the V^T tile is set to 1.0, so accumulator row 72
// of O^T is sum_k P[q][k] (of the bf16-rounded P, i.e. consistent with the numerator).
// =====================================================================================================
template <int HD, int HINTS>
__global__ void __launch_bounds__(512) attn_fwd_kernel(const AttnParams p) {
  using C = Cfg<HD>;
  constexpr bool ONES_ROW = (HD % 32) != 0;  // a free padding row exists -> row sum from the MFMA
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  int bh, qb;
  osk_attn::block_to_work(p, (p.Lq + 255) / 256, bh, qb);
  const int b = bh / p.H, h = bh - b * p.H;

  for (int i = tid; i < C::SMEM / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if constexpr (ONES_ROW) {
    __syncthreads();
    if (tid < 64) {  // V^T row HD (first padding row) = 1.0 in both buffers
      const unsigned one2 = 0x3F803F80u;
      reinterpret_cast<unsigned*>(smem + C::KTILE + HD * C::VROW)[tid & 31] = one2;
      reinterpret_cast<unsigned*>(smem + C::BUF + C::KTILE + HD * C::VROW)[tid & 31] = one2;
    }
  }

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

  // ---- staging: <= 2 16-B chunks of K and of V^T per thread; running pointers, no per-tile index math
  static_assert(C::KIT <= 2 && C::VIT <= 2, "staging assumes <= 1024 chunks per tile");
  const int kr0 = tid / C::CPR, kc0 = tid - kr0 * C::CPR;
  const int kr1 = (tid + 512) / C::CPR, kc1 = (tid + 512) - kr1 * C::CPR;
  const int vd0 = tid >> 3, vc0 = tid & 7, vd1 = (tid + 512) >> 3;
  constexpr bool K1_FULL = C::NKC >= 1024, V1_FULL = C::NVC >= 1024;
  const bool k1_on = C::KIT > 1 && (K1_FULL || (tid + 512 < C::NKC));
  const bool v1_on = C::VIT > 1 && (V1_FULL || (tid + 512 < C::NVC));
  const int bkv = b % p.Bkv;                                        // key / value batch of this query batch
  const unsigned short* kseg = p.k + bkv * p.kbs + h * HD;         // segment base of the K loader
  const unsigned short* kp0 = kseg + (int64_t)kr0 * p.krs + kc0 * 8;  // this thread's chunk in the loader's tile
  const unsigned short* kp1 = kseg + (int64_t)kr1 * p.krs + kc1 * 8;
  const unsigned short* vp0 = p.vt + ((int64_t)(bkv * p.H + h) * HD + vd0) * p.seg_lp + vc0 * 8;
  const unsigned short* vp1 = p.vt + ((int64_t)(bkv * p.H + h) * HD + vd1) * p.seg_lp + vc0 * 8;
  const int64_t k_tile_step = (int64_t)64 * p.krs;
  const int64_t k_seg_jump = p.kss - (int64_t)p.tps * 64 * p.krs;  // from past-the-last tile of a segment to the next
  const int64_t v_seg_jump = p.vtss - (int64_t)p.tps * 64;
  const int last_valid = p.seg_len - (p.tps - 1) * 64;              // keys in the last tile of a segment (1..64)
  int ktt = 0, vtt = 0;                                             // tile-in-segment counters of the two loaders
  uint4 rk0 = make_uint4(0, 0, 0, 0), rk1 = rk0, rv0 = rk0, rv1 = rk0;
  const int kbyte0 = kr0 * C::KROW + kc0 * 16, kbyte1 = kr1 * C::KROW + kc1 * 16;
  const int vbyte0 = vd0 * C::VROW + vc0 * 16, vbyte1 = vd1 * C::VROW + vc0 * 16;

  // load the K loader's current tile into registers and advance it (ragged last tile: clamp rows to the segment)
#define K_ISSUE()                                                                                   \
  {                                                                                                 \
    if (ktt == p.tps - 1 && last_valid < 64) {                                                      \
      const int64_t c0_ = kr0 < last_valid ? 0 : (int64_t)(last_valid - 1 - kr0) * p.krs;           \
      const int64_t c1_ = kr1 < last_valid ? 0 : (int64_t)(last_valid - 1 - kr1) * p.krs;           \
      rk0 = *reinterpret_cast<const uint4*>(kp0 + c0_);                                             \
      if (k1_on) rk1 = *reinterpret_cast<const uint4*>(kp1 + c1_);                                  \
    } else {                                                                                        \
      rk0 = *reinterpret_cast<const uint4*>(kp0);                                                   \
      if (k1_on) rk1 = *reinterpret_cast<const uint4*>(kp1);                                        \
    }                                                                                               \
    kp0 += k_tile_step;                                                                             \
    kp1 += k_tile_step;                                                                             \
    if (++ktt == p.tps) {                                                                           \
      ktt = 0;                                                                                      \
      kp0 += k_seg_jump;                                                                            \
      kp1 += k_seg_jump;                                                                            \
    }                                                                                               \
  }
#define V_ISSUE()                                                                                   \
  {                                                                                                 \
    rv0 = *reinterpret_cast<const uint4*>(vp0);                                                     \
    if (v1_on) rv1 = *reinterpret_cast<const uint4*>(vp1);                                          \
    vp0 += 64;                                                                                      \
    vp1 += 64;                                                                                      \
    if (++vtt == p.tps) {                                                                           \
      vtt = 0;                                                                                      \
      vp0 += v_seg_jump;                                                                            \
      vp1 += v_seg_jump;                                                                            \
    }                                                                                               \
  }
#define K_COMMIT(BUFI)                                                                              \
  {                                                                                                 \
    unsigned char* kb_ = smem + (BUFI) * C::BUF;                                                    \
    *reinterpret_cast<uint4*>(kb_ + kbyte0) = rk0;                                                  \
    if (k1_on) *reinterpret_cast<uint4*>(kb_ + kbyte1) = rk1;                                       \
  }
#define V_COMMIT(BUFI)                                                                              \
  {                                                                                                 \
    unsigned char* vb_ = smem + (BUFI) * C::BUF + C::KTILE;                                         \
    *reinterpret_cast<uint4*>(vb_ + vbyte0) = rv0;                                                  \
    if (v1_on) *reinterpret_cast<uint4*>(vb_ + vbyte1) = rv1;                                       \
  }
#define QK_TILE(SDST, BUFI)                                                                         \
  {                                                                                                 \
    const unsigned char* kb_ = smem + (BUFI) * C::BUF;                                              \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { SDST[0][r] = 0.f; SDST[1][r] = 0.f; }          \
    _Pragma("unroll") for (int ks = 0; ks < C::NKS; ++ks) {                                         \
      _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                            \
        const bf16x8_t kf_ = *reinterpret_cast<const bf16x8_t*>(kb_ + (t2 * 32 + l31) * C::KROW +   \
                                                                (ks * 2 + hi) * 16);                \
        SDST[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[ks], SDST[t2], 0, 0, 0);         \
      }                                                                                             \
    }                                                                                               \
  }
  // scores of keys >= VALID inside a ragged last tile -> -inf
#define MASK_TILE(S, VALID)                                                                         \
  {                                                                                                 \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) _Pragma("unroll") for (int r = 0; r < 16; ++r) { \
      const int kl_ = t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                    \
      if (kl_ >= (VALID)) S[t2][r] = -INFINITY;                                                     \
    }                                                                                               \
  }
  // row max of a score tile (32 own values as max3 chains, then the other half-wave through v_permlane32_swap),
  // the new running max, the accumulator rescale factor and whether any lane of the wave needs the rescale
#define ROW_MAX(S)                                                                                  \
  {                                                                                                 \
    float m0_ = fmaxf(fmaxf(S[0][0], S[0][1]), S[0][2]);                                            \
    float m1_ = fmaxf(fmaxf(S[1][0], S[1][1]), S[1][2]);                                            \
    _Pragma("unroll") for (int r = 3; r < 15; r += 2) {                                             \
      m0_ = fmaxf(fmaxf(m0_, S[0][r]), S[0][r + 1]);                                                \
      m1_ = fmaxf(fmaxf(m1_, S[1][r]), S[1][r + 1]);                                                \
    }                                                                                               \
    float mt_ = fmaxf(fmaxf(m0_, m1_), fmaxf(S[0][15], S[1][15]));                                  \
    const unsigned mu_ = __float_as_uint(mt_);                                                      \
    auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                            \
    mt_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                  \
    m_new = fmaxf(m_run, mt_);                                                                      \
    alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                                         \
    resc = !__all(m_new == m_run);                                                                  \
  }

  f32x16_t o[C::NDT];
#pragma unroll
  for (int d = 0; d < C::NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nt = p.n_seg * p.tps;

  __syncthreads();  // LDS init done
  K_ISSUE();
  V_ISSUE();
  K_COMMIT(0);
  V_COMMIT(0);
  if (nt > 1) {
    K_ISSUE();
    K_COMMIT(1);
  }
  __syncthreads();

  f32x16_t sa_[2], sb_[2];  // score tiles: roles (current / next) alternate every iteration, no copies
  float m_new, alpha;
  int resc;
  int ctt = 0;              // tile-in-segment counter of the tile whose scores are "current"
  QK_TILE(sa_, 0);
  if (p.tps == 1 && last_valid < 64) MASK_TILE(sa_, last_valid);
  ROW_MAX(sa_);

  // one iteration: SC = scores of tile t (masked, max known), SN receives tile t+1
#define ITERATION(SC, SN, CUR)                                                                      \
  {                                                                                                 \
    const bool has1_ = t + 1 < nt, has2_ = t + 2 < nt;                                              \
    if (has2_) K_ISSUE();                                                                           \
    if (has1_) V_ISSUE();                                                                           \
    /* block 1: (rare) rescale of the accumulators for the new running max of tile t */            \
    if (resc) {                                                                                     \
      _Pragma("unroll") for (int d = 0; d < C::NDT; ++d) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[d][r] *= alpha; \
    }                                                                                               \
    if constexpr (!ONES_ROW) l_run *= alpha;                                                        \
    m_run = m_new;                                                                                  \
    /* block 2: next tile's QK^T (MFMA) || exp2 + pack of this tile (VALU).  QK is unconditional: on the   \
       last tile it multiplies stale-but-finite LDS data and the result is dropped (one basic block). */   \
    const float msc_ = m_run * p.sc;                                                                \
    bf16x8_t pb_[4];                                                                                \
    float rs_ = 0.f;                                                                                \
    if constexpr (HINTS == 2) {                                                                     \
      /* hand-ordered block 2: [K-fragment read two slots ahead] [MFMA] [a slice of exp2 + pack], pinned with     \
         sched_barrier so the VALU work sits in the shadow of the 32-cycle MFMAs instead of behind all of them */ \
      constexpr int NM_ = 2 * C::NKS;                                                               \
      const unsigned char* kb_ = smem + ((CUR) ^ 1) * C::BUF;                                       \
      bf16x8_t kfr_[NM_];                                                                           \
      unsigned w_[16];                                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        kfr_[i] = *reinterpret_cast<const bf16x8_t*>(kb_ + ((i & 1) * 32 + l31) * C::KROW + ((i >> 1) * 2 + hi) * 16); \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                             \
        if (i + 2 < NM_)                                                                            \
          kfr_[i + 2] = *reinterpret_cast<const bf16x8_t*>(kb_ + (((i + 2) & 1) * 32 + l31) * C::KROW + (((i + 2) >> 1) * 2 + hi) * 16); \
        if (i < 2) {                                                                                \
          f32x16_t z_;                                                                              \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) z_[r] = 0.f;                               \
          SN[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr_[i], qf[i >> 1], z_, 0, 0, 0);    \
        } else {                                                                                    \
          SN[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr_[i], qf[i >> 1], SN[i & 1], 0, 0, 0); \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        _Pragma("unroll") for (int pp = (i * 16) / NM_; pp < ((i + 1) * 16) / NM_; ++pp) {          \
          const int t2 = pp >> 3, r = (pp & 7) * 2;                                                 \
          const float e0_ = __builtin_amdgcn_exp2f(SC[t2][r] * p.sc - msc_);                        \
          const float e1_ = __builtin_amdgcn_exp2f(SC[t2][r + 1] * p.sc - msc_);                    \
          if constexpr (!ONES_ROW) rs_ += e0_ + e1_;                                                \
          w_[pp] = pack_bf16x2(e0_, e1_);                                                           \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
      }                                                                                             \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                               \
        uint4 u_ = make_uint4(w_[4 * g], w_[4 * g + 1], w_[4 * g + 2], w_[4 * g + 3]);              \
        pb_[g] = __builtin_bit_cast(bf16x8_t, u_);                                                  \
      }                                                                                             \
    } else {                                                                                        \
    QK_TILE(SN, (CUR) ^ 1);                                                                         \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) _Pragma("unroll") for (int r = 0; r < 16; ++r) { \
      const float e_ = __builtin_amdgcn_exp2f(SC[t2][r] * p.sc - msc_);                             \
      SC[t2][r] = e_;                                                                               \
      if constexpr (!ONES_ROW) rs_ += e_;                                                           \
    }                                                                                               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
      const int t2 = g >> 1, r0 = (g & 1) * 8;                                                      \
      uint4 u_;                                                                                     \
      u_.x = pack_bf16x2(SC[t2][r0 + 0], SC[t2][r0 + 1]);                                           \
      u_.y = pack_bf16x2(SC[t2][r0 + 2], SC[t2][r0 + 3]);                                           \
      u_.z = pack_bf16x2(SC[t2][r0 + 4], SC[t2][r0 + 5]);                                           \
      u_.w = pack_bf16x2(SC[t2][r0 + 6], SC[t2][r0 + 7]);                                           \
      pb_[g] = __builtin_bit_cast(bf16x8_t, u_);                                                    \
    }                                                                                               \
    }                                                                                               \
    if constexpr (!ONES_ROW) l_run += rs_;                                                          \
    {                                                                                               \
      /* anchor: all 16 packed P words must exist HERE, otherwise hipcc sinks the exp2 / pack work below the \
         mask branch into block 3 and block 2 degenerates to bare MFMAs */                           \
      const uint4 a0 = __builtin_bit_cast(uint4, pb_[0]), a1 = __builtin_bit_cast(uint4, pb_[1]);   \
      const uint4 a2 = __builtin_bit_cast(uint4, pb_[2]), a3 = __builtin_bit_cast(uint4, pb_[3]);   \
      asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), \
                   "v"(a1.w), "v"(a2.x), "v"(a2.y), "v"(a2.z), "v"(a2.w), "v"(a3.x), "v"(a3.y),     \
                   "v"(a3.z), "v"(a3.w));                                                           \
    }                                                                                               \
    if constexpr (HINTS == 1) {                                                                     \
      /* shape block 2: 1 K-fragment read : 1 MFMA : a slice of the exp2/pack VALU work, 2*NKS times */ \
      constexpr int NM_ = 2 * C::NKS, NV_ = ONES_ROW ? 80 : 112;                                    \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x002, (NV_ + NM_ - 1) / NM_, 0);                      \
      }                                                                                             \
    }                                                                                               \
    /* ragged tail of the NEXT tile, then block 3: O^T += V^T . P^T (MFMA) || row max of the next tile (VALU) */ \
    if (++ctt == p.tps) ctt = 0;                                                                    \
    if (has1_ && ctt == p.tps - 1 && last_valid < 64) MASK_TILE(SN, last_valid);                    \
    if constexpr (HINTS == 2) {                                                                     \
      /* hand-ordered block 3: [V^T fragment read two slots ahead] [MFMA] [two max3 of the next tile's row max] */ \
      constexpr int NM_ = 4 * C::NDT;                                                               \
      const unsigned char* vb_ = smem + (CUR) * C::BUF + C::KTILE;                                  \
      bf16x8_t vfr_[NM_];                                                                           \
      float mx_[2] = {SN[0][0], SN[1][0]};                                                          \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        vfr_[i] = *reinterpret_cast<const bf16x8_t*>(vb_ + ((i % C::NDT) * 32 + l31) * C::VROW + ((i / C::NDT) * 2 + hi) * 16); \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                             \
        if (i + 2 < NM_)                                                                            \
          vfr_[i + 2] = *reinterpret_cast<const bf16x8_t*>(vb_ + (((i + 2) % C::NDT) * 32 + l31) * C::VROW + (((i + 2) / C::NDT) * 2 + hi) * 16); \
        o[i % C::NDT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr_[i], pb_[i / C::NDT], o[i % C::NDT], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        /* 30 remaining score values in max3 steps of 2, spread over the first MFMAs */            \
        _Pragma("unroll") for (int j = (i * 16) / NM_; j < ((i + 1) * 16) / NM_; ++j) {             \
          if (j < 15) {                                                                             \
            const int t2 = j & 1, r = 1 + 2 * (j >> 1);                                             \
            if (r + 1 < 16) mx_[t2] = fmaxf(fmaxf(mx_[t2], SN[t2][r]), SN[t2][r + 1]);              \
            else mx_[t2] = fmaxf(mx_[t2], SN[t2][r]);                                               \
          }                                                                                         \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
      }                                                                                             \
      {                                                                                             \
        float mt_ = fmaxf(mx_[0], mx_[1]);                                                          \
        mt_ = fmaxf(mt_, fmaxf(SN[0][15], SN[1][15]));                                              \
        const unsigned mu_ = __float_as_uint(mt_);                                                  \
        auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                        \
        mt_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                              \
        m_new = fmaxf(m_run, mt_);                                                                  \
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                                     \
        resc = !__all(m_new == m_run);                                                              \
      }                                                                                             \
    } else {                                                                                        \
      const unsigned char* vb_ = smem + (CUR) * C::BUF + C::KTILE;                                  \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                               \
        _Pragma("unroll") for (int d = 0; d < C::NDT; ++d) {                                        \
          const bf16x8_t vf_ = *reinterpret_cast<const bf16x8_t*>(vb_ + (d * 32 + l31) * C::VROW + (g * 2 + hi) * 16); \
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_, pb_[g], o[d], 0, 0, 0);               \
        }                                                                                           \
      }                                                                                             \
      ROW_MAX(SN);                                                                                  \
    }                                                                                               \
    if constexpr (HINTS == 1) {                                                                     \
      constexpr int NM_ = 4 * C::NDT;                                                               \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);                                          \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 1);                                          \
      }                                                                                             \
    }                                                                                               \
    if (has2_) K_COMMIT(CUR);        /* K_{t+2} -> the buffer K_t lived in (last read in iteration t-1) */ \
    if (has1_) V_COMMIT((CUR) ^ 1);  /* V_{t+1} */                                                   \
    __syncthreads();                                                                                \
  }

  for (int t = 0; t < nt; t += 2) {
    ITERATION(sa_, sb_, 0);
    ++t;
    if (t < nt) ITERATION(sb_, sa_, 1);
    --t;
  }
#undef ITERATION
#undef MASK_TILE
#undef ROW_MAX

  // ---- epilogue
  if constexpr (ONES_ROW) {
    // row HD of O^T = sum_k P: lives in lanes hi == 0, register (HD % 32) -> index of d = HD
    constexpr int dloc = HD % 32;                 // 8 for hd 72
    constexpr int rr = (dloc & 3) + 4 * (dloc >> 3);  // d_local = (r&3) + 8*(r>>2) + 4*hi, hi = 0
    static_assert(((dloc >> 2) & 1) == 0, "ones row must sit in the hi == 0 half");
    l_run = o[HD / 32][rr];
    const unsigned lu = __float_as_uint(l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    l_run = __uint_as_float(sw[0]);  // lanes 0-31 keep their own, lanes 32-63 receive lanes 0-31
  }
  float l_tot = l_run;
  if constexpr (!ONES_ROW) {
    const unsigned lu = __float_as_uint(l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
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
#undef K_ISSUE
#undef V_ISSUE
#undef K_COMMIT
#undef V_COMMIT

template <typename KernelT>
int launch_kernel(KernelT kernel, int smem, const AttnParams& p, hipStream_t st) {
  OSK_ENSURE_MAX_SMEM(kernel, smem);   // one instantiation of this template per kernel: attn_fwd_kernel<64, 0>
  const int nqb = (p.Lq + 255) / 256;
  dim3 grid(nqb * p.B * p.H), block(512);
  hipLaunchKernelGGL(kernel, grid, block, smem, st, p);
  return (int)hipGetLastError();
}

// ONE kernel per head_dim: 72 and 128 -> the hand-scheduled kernels (tail units split along the keys + merge when the caller
// handed over a workspace); 64 -> the compiler-scheduled pipeline of this file.  (The A/B variants of rounds 1-2 -- 8 waves x
// 32 rows, the compiler-scheduled 4 x 64 layout, scheduling hints -- lost their comparisons and are generator / git history.)
template <int HD>
int launch(const AttnParams& p, hipStream_t st) {
  if constexpr (HD == 72) {
    const int rc = p.rows == 512 ? osk_attn::launch_asm72w(p, st) : osk_attn::launch_asm72(p, st);
    return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, HD, st);
  } else if constexpr (HD == 128) {
    const int rc = osk_attn::launch_asm128(p, st);
    return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, HD, st);
  } else {
    return launch_kernel(attn_fwd_kernel<HD, 0>, Cfg<HD>::SMEM, p, st);
  }
}

}  // namespace

namespace osk_attn {
namespace {

// combine the key parts of the split tail units: out = sum_p 2^(lse_p - lse) O_p, lse = log2 sum_p 2^lse_p.
// grid (tail units, 4): a block owns 64 query rows; thread = (row, 16-byte chunk of the head dim), so the partial rows
// -- contiguous [row][hd] f32 in the workspace -- are read as one linear stream per part.  A few tens of MB in all.
template <int HD>
__global__ void __launch_bounds__(256) attn_merge_kernel(const AttnParams p) {
  constexpr int CPR = HD / 4;                    // float4 chunks per row
  const int nqb = (p.Lq + p.rows - 1) / p.rows;
  int bh, qb;
  unit_to_work(p, nqb, p.tail_first + (int)blockIdx.x, bh, qb);
  const int b = bh / p.H, h = bh - b * p.H;
  const int S = p.tail_split;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int r = blockIdx.y * 64 + i / CPR, c = i % CPR;
    const int row = qb * p.rows + r;
    if (row >= p.Lq) continue;
    const int64_t slot0 = (int64_t)blockIdx.x * S * p.rows + r;
    float l[8], m = -INFINITY;
    for (int s = 0; s < S; ++s) { l[s] = p.ws_lse[slot0 + s * p.rows]; m = fmaxf(m, l[s]); }
    float tot = 0.f;
    for (int s = 0; s < S; ++s) { l[s] = exp2f(l[s] - m); tot += l[s]; }
    const float inv = 1.0f / tot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(p.ws_o + (slot0 + (int64_t)s * p.rows) * HD + c * 4);
      acc.x += l[s] * v.x; acc.y += l[s] * v.y; acc.z += l[s] * v.z; acc.w += l[s] * v.w;
    }
    uint2 w2;
    w2.x = pack_bf16x2(acc.x * inv, acc.y * inv);
    w2.y = pack_bf16x2(acc.z * inv, acc.w * inv);
    *reinterpret_cast<uint2*>(p.out + b * p.obs + (int64_t)row * p.ors + h * HD + c * 4) = w2;
    if (p.lse && c == 0) p.lse[(int64_t)bh * p.Lq + row] = (m + log2f(tot)) * 0.6931471805599453f;
  }
}

}  // namespace

// A workgroup owns a CU for its whole key loop, so a launch takes ceil(units / CUs) rounds and the last round may run
// a fraction of the chip (3168 units on 256 CUs: 12.4 -> 13; sequence-parallel ranks: 1.7 -> 2, 3.2 -> 4).  With a
// workspace the units of that last round are cut into `s` key parts (whole key segments when there are several,
// else runs of 64-key tiles) so that the round costs ceil(R s / CUs) / s instead of 1; only those units pay the
// partial-result traffic.  s <= 8, chosen to minimise that cost; no split when it saves < 15 % of a round.
void split_tail(AttnParams& p, int units, int hd, void* workspace, int64_t workspace_bytes) {
  p.tail_split = 1;
  p.tail_first = 0x7fffffff;
  if (!workspace) return;
  const int cus = osk_device_cus();
  const int R = units % cus;
  if (R == 0) return;
  auto cost = [&](int s) { return (double)((R * s + cus - 1) / cus) / s; };
  int best = 1;
  for (int s = 2; s <= 8; ++s) {
    if (p.n_seg > 1 ? (p.n_seg % s != 0) : (s > p.tps)) continue;
    const int64_t need = (int64_t)R * s * p.rows * (hd + 1) * 4;
    if (need > workspace_bytes) continue;
    if (cost(s) < cost(best) - 1e-9) best = s;
  }
  if (cost(1) - cost(best) < 0.15) return;
  p.tail_split = best;
  p.tail_first = units - R;
  p.ws_o = (float*)workspace;
  p.ws_lse = p.ws_o + (int64_t)R * best * p.rows * hd;
}

int launch_merge(const AttnParams& p, int hd, hipStream_t st) {
  const int units = ((p.Lq + p.rows - 1) / p.rows) * p.B * p.H;
  dim3 grid(units - p.tail_first, p.rows / 64), block(256);
  if (hd == 72) hipLaunchKernelGGL(attn_merge_kernel<72>, grid, block, 0, st, p);
  else if (hd == 128) hipLaunchKernelGGL(attn_merge_kernel<128>, grid, block, 0, st, p);
  else return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

}  // namespace osk_attn

extern "C" int osk_attention_tail_split_factor(int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                               int64_t workspace_bytes) {
  if (B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0 || (hd != 72 && hd != 128)) return 1;
  osk_attn::AttnParams p{};
  p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  static char dummy[16] __attribute__((aligned(16)));
  osk_attn::split_tail(p, ((Lq + 255) / 256) * B * H, hd, dummy, workspace_bytes);
  return p.tail_split;
}

extern "C" int64_t osk_attention_workspace_bytes(void) {
  // enough for 512 key parts (two rounds of a 256-CU chip) of 256 rows at head_dim 128: partial O + LSE in f32
  return (int64_t)512 * 256 * (128 + 1) * 4;
}

extern "C" int osk_attention_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                      const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                      int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                      void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                      float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                      float scale, int q_prescaled, int kv_batches, void* stream) {
  return osk_attention_fwd_ws_bf16(q, q_batch_stride, q_row_stride, k, k_seg_stride, k_batch_stride, k_row_stride, vt,
                                   vt_seg_stride, out, o_batch_stride, o_row_stride, lse, B, H, Lq, n_seg, seg_len, hd,
                                   scale, q_prescaled, kv_batches, nullptr, 0, stream);
}

extern "C" int osk_attention_fwd_bounded_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                         const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                         int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                         void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                         float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                         float scale, int q_prescaled, int kv_batches, float score_bound,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
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
  p.sc = q_prescaled ? 1.0f : scale * 1.4426950408889634f;  // log2 units; 1: q already carries it
  p.q_prescaled = q_prescaled;
  if (!(score_bound >= 0.f)) return OSK_EINVAL;   // (also rejects NaN)
  {
    // the kernels keep the bound in a bf16 field of Q's padding dim: round it UP to the next bf16 value (it stays a bound)
    const unsigned bits = __builtin_bit_cast(unsigned, score_bound);
    p.bound = __builtin_bit_cast(float, (bits + 0xFFFFu) & 0xFFFF0000u);
  }
  if (kv_batches < 0 || kv_batches > B) return OSK_EINVAL;
  p.Bkv = kv_batches > 0 ? kv_batches : B;
  p.map = 1;   // XCD-contiguous work order (each XCD walks one head's K / V^T stream)
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  if (osk_attn::attn_wide_path(p, hd)) p.rows = 512;
  if (hd == 72 || hd == 128) osk_attn::split_tail(p, ((Lq + p.rows - 1) / p.rows) * B * H, hd, workspace, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 64: return launch<64>(p, st);
    case 72: return launch<72>(p, st);
    case 128: return launch<128>(p, st);
    default: return OSK_EUNSUPPORTED;
  }
}

extern "C" int osk_attention_fwd_ws_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                         const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                         int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                         void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                         float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                         float scale, int q_prescaled, int kv_batches, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  return osk_attention_fwd_bounded_bf16(q, q_batch_stride, q_row_stride, k, k_seg_stride, k_batch_stride, k_row_stride, vt,
                                        vt_seg_stride, out, o_batch_stride, o_row_stride, lse, B, H, Lq, n_seg, seg_len, hd,
                                        scale, q_prescaled, kv_batches, 0.0f, workspace, workspace_bytes, stream);
}

extern "C" const char* osk_attention_kernel_name(int hd, int seg_len) {
  (void)seg_len;
  return hd == 72 ? "attn_asm72_kernel" : hd == 128 ? "attn_asm128_kernel" : "attn_fwd_kernel";
}

extern "C" const char* osk_attention_body_name(int hd, int n_seg, int seg_len, float score_bound) {
  if (hd != 72 && hd != 128) return hd == 64 ? "attn_fwd_kernel<64>" : "unsupported";
  AttnParams p{};
  p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  if (score_bound > 0.f) {   // rounded up to bf16 as osk_attention_fwd_bounded_bf16 does
    const unsigned bits = __builtin_bit_cast(unsigned, score_bound);
    p.bound = __builtin_bit_cast(float, (bits + 0xFFFFu) & 0xFFFF0000u);
  }
  const bool fast = osk_attn::attn_fast_path(p);
  if (hd == 72) return fast ? "attn_asm72_kernel<FAST>" : "attn_asm72_kernel<general>";   // (Lq >= 1024: the FAST body's wide layout, attn_asm72w_kernel)
  return fast ? "attn_asm128_kernel<FAST>" : "attn_asm128_kernel<general>";
}


// =============================================================================================
// fp8 P.V variant: V -> e4m3 V^T with the key order, ones row and zero rows the kernels expect; entry point
// =============================================================================================
namespace {

// V [B, L, H, hd] bf16 (strided) -> vt8 [B, H, RP, Lp] e4m3 bytes, RP = (hd + 1) rounded up to 16.
// One block = one 64-key tile of one head.  Row d < hd: byte hi*32 + t2*16 + j*4 + i of the tile holds key
// 32 t2 + 8 j + 4 hi + i (the order in which a lane of the score accumulators holds its 32 P values), quantised
// with the (batch, head) scale; row hd: 1.0 for keys < L, 0 behind them (softmax denominator / key validity);
// rows hd+1 .. RP-1: zero.
template <int HD>
__global__ void __launch_bounds__(256) v_transpose_fp8_kernel(const unsigned short* __restrict__ v, int64_t bs,
                                                              int64_t rs, const float* __restrict__ scales,
                                                              unsigned char* __restrict__ vt8, int L, int Lp, int H) {
  constexpr int RP = (HD + 1 + 15) / 16 * 16;
  __shared__ unsigned short tile[64][HD + 2];
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int key0 = kt * 64;
  constexpr int CPR = HD / 8;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int r = i / CPR, c = i % CPR;
    const int key = key0 + r;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (key < L) u = *reinterpret_cast<const uint4*>(v + b * bs + (int64_t)key * rs + h * HD + c * 8);
    unsigned* dst = reinterpret_cast<unsigned*>(&tile[r][c * 8]);
    dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
  }
  __syncthreads();
  const float inv = 1.0f / scales[b * H + h];
  unsigned char* obase = vt8 + ((int64_t)(b * H + h) * RP) * Lp + key0;
  // thread = (row, 16-byte chunk of the 64-byte row)
  for (int i = threadIdx.x; i < RP * 4; i += 256) {
    const int d = i >> 2, c = i & 3;
    unsigned w[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {          // dword q4 of the chunk: bytes pos = c*16 + q4*4 + 0..3
      const int pos = c * 16 + q4 * 4;
      const int hi = pos >> 5, s5 = pos & 31, t2 = s5 >> 4, j = (s5 & 15) >> 2;
      const int k0 = 32 * t2 + 8 * j + 4 * hi;   // keys k0 .. k0 + 3
      if (d < HD) {
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          f[e] = fminf(fmaxf(bf16_bits_to_f32(tile[k0 + e][d]) * inv, -448.0f), 448.0f);
        int r = 0;
        r = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r, true);
        w[q4] = (unsigned)r;
      } else if (d == HD) {
        unsigned r = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) r |= (key0 + k0 + e < L ? 0x38u : 0u) << (8 * e);
        w[q4] = r;
      } else {
        w[q4] = 0;
      }
    }
    *reinterpret_cast<uint4*>(obase + (int64_t)d * Lp + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace

namespace {

// absmax of V per (batch, head) -> e4m3 scale: bit patterns of |bf16| order like the values, so the reduction runs on
// integers.  grid (row slabs, B); a thread owns 16-byte chunks of the [H*hd] row (one head each), walks the slab's rows,
// then one LDS atomic per chunk and one global atomic per (block, head): few, fat blocks (same-address atomics are slow).
__global__ void __launch_bounds__(256) v_absmax_kernel(const unsigned short* __restrict__ v, int64_t bs, int64_t rs,
                                                       int L, int H, int hd, int rows_per_block,
                                                       unsigned* __restrict__ amax_bits) {
  __shared__ unsigned smax[256];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < H; i += 256) smax[i] = 0;
  __syncthreads();
  const int cpr = (H * hd) >> 3;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = r0 + rows_per_block < L ? r0 + rows_per_block : L;
  for (int c = threadIdx.x; c < cpr; c += 256) {
    const unsigned short* p = v + b * bs + c * 8;
    unsigned m = 0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      uint4 u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const uint4*>(p + (int64_t)(r + i) * rs);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned w[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned a = w[j] & 0x7FFF7FFFu;
          m = max(m, max(a & 0xFFFFu, a >> 16));
        }
      }
    }
    for (; r < r1; ++r) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + (int64_t)r * rs);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned a = w[j] & 0x7FFF7FFFu;
        m = max(m, max(a & 0xFFFFu, a >> 16));
      }
    }
    atomicMax(&smax[(c * 8) / hd], m);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256)
    if (smax[i]) atomicMax(&amax_bits[b * H + i], smax[i]);
}

__global__ void v_scale_finalize_kernel(unsigned* __restrict__ amax_bits, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float amax = __uint_as_float(amax_bits[i] << 16);
  reinterpret_cast<float*>(amax_bits)[i] = amax > 0.f ? amax / 448.0f : 1.0f;
}

}  // namespace

extern "C" int osk_v_scale_fp8(const void* v, int64_t bs, int64_t rs, float* scales, int B, int L, int H, int hd,
                               void* stream) {
  if (!v || !scales || B <= 0 || L <= 0 || H <= 0 || H > 256 || (hd & 7) || (bs & 7) || (rs & 7) || ((uintptr_t)v & 15))
    return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(scales, 0, sizeof(float) * B * H, st);
  if (e != hipSuccess) return (int)e;
  int rows = (L + 127) / 128;            // about 128 blocks per batch item
  if (rows < 16) rows = 16;
  dim3 grid((L + rows - 1) / rows, B), block(256);
  hipLaunchKernelGGL(v_absmax_kernel, grid, block, 0, st, (const unsigned short*)v, bs, rs, L, H, hd, rows,
                     reinterpret_cast<unsigned*>(scales));
  hipLaunchKernelGGL(v_scale_finalize_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<unsigned*>(scales), B * H);
  return (int)hipGetLastError();
}

extern "C" int osk_v_transpose_fp8(const void* v, int64_t bs, int64_t rs, const float* scales, void* vt8, int B, int L,
                                   int H, int hd, void* stream) {
  if (!v || !vt8 || !scales || B <= 0 || L <= 0 || H <= 0 || (bs & 7) || (rs & 7) || ((uintptr_t)vt8 & 15)) return OSK_EINVAL;
  const int Lp = (L + 63) / 64 * 64;
  dim3 grid(Lp / 64, H, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (hd == 72)
    hipLaunchKernelGGL(v_transpose_fp8_kernel<72>, grid, block, 0, st, (const unsigned short*)v, bs, rs, scales, (unsigned char*)vt8, L, Lp, H);
  else if (hd == 128)
    hipLaunchKernelGGL(v_transpose_fp8_kernel<128>, grid, block, 0, st, (const unsigned short*)v, bs, rs, scales, (unsigned char*)vt8, L, Lp, H);
  else
    return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

extern "C" int osk_attention_fwd_pv8_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                          const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                          int64_t k_row_stride, const void* vt8, int64_t vt8_seg_stride,
                                          const float* v_scale, void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                          float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                          float scale, int q_prescaled, int kv_batches, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (!q || !k || !vt8 || !v_scale || !out || B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0) return OSK_EINVAL;
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_seg_stride & 7) || (k_batch_stride & 7) ||
      (k_row_stride & 7) || (vt8_seg_stride & 15) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt8 & 15) || ((uintptr_t)out & 7)) return OSK_EINVAL;
  if (hd != 128 && hd != 72) return OSK_EUNSUPPORTED;
  osk_attn::AttnParams p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kss = k_seg_stride; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = nullptr; p.vt8 = (const unsigned char*)vt8; p.vtss = vt8_seg_stride; p.v_scale = v_scale;
  p.out = (unsigned short*)out; p.obs = o_batch_stride; p.ors = o_row_stride;
  p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.sc = q_prescaled ? 1.0f : scale * 1.4426950408889634f;
  p.q_prescaled = q_prescaled;
  if (kv_batches < 0 || kv_batches > B) return OSK_EINVAL;
  p.Bkv = kv_batches > 0 ? kv_batches : B;
  p.map = 1;   // XCD-contiguous work order (each XCD walks one head's K / V^T stream)
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  osk_attn::split_tail(p, ((Lq + 255) / 256) * B * H, hd, workspace, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int rc = hd == 128 ? osk_attn::launch_asm128p8(p, st) : osk_attn::launch_asm72p8(p, st);
  return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, hd, st);
}
