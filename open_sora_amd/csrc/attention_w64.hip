// Flash-attention forward (non-causal) for gfx950 — "w64" structure: ONE wave per SIMD, 64 query rows per wave.
//
// Workgroup = 4 waves = 256 query rows; wave w owns rows [64 w, 64 w + 64) as two 32-row query blocks that
// share every K / V^T fragment it reads from LDS (one ds_read_b128 feeds two MFMAs: half the LDS traffic per
// FLOP of the 8-wave x 32-row kernel in attention_fwd.hip).  The wave has the SIMD's whole 512-register file:
// O^T accumulators and the Q fragments live in AGPRs, two score tiles (current / next) in VGPRs.
//
// Same operand convention as attention_fwd.hip (v_mfma_f32_32x32x16_bf16, operands swapped so that a lane owns
// one query: S^T = K . Q^T, O^T += V^T . P^T; P stays in registers because osk_v_transpose_bf16 bakes the
// accumulator's key order into the V^T buffer), same software pipeline (iteration t: QK^T of tile t+1 beside
// exp2/pack of tile t, then P.V of tile t beside the row max of tile t+1) — but within ONE instruction stream:
// sched_group_barrier pipelines place ~1 LDS read and a slice of the softmax VALU work in every MFMA shadow.
//
// Staging: K and V^T tiles (64 keys) go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction), K two tiles ahead, V^T one tile ahead, 2-slot rings, one barrier per tile, no staging registers.
// LDS images are 128-byte rows with the SOURCE-side XOR swizzle chunk' = chunk ^ ((row >> 1) & 7) (conflict-free
// for the 32-row x 16-B ds_read_b128 fragment reads):
//   K tile   : HD/64 images [64 keys][64 dims] + (head_dim 72) one [64 keys][8 dims] column image
//   V^T tile : [HD rows + padding to 32][64 keys]; padding rows are written once (row HD = 1.0 when a padding row
//              exists, so the softmax denominator falls out of the P.V MFMA as accumulator row HD).
// head_dim 72: the k-step 4 fragment (dims 64..79) reads the 8-dim column image in BOTH half-waves; the upper
// half multiplies Q's zero padding.  No byte of the K / V^T stream is fetched twice, nothing is padded in HBM.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 4 * B * H * Lq * Lk * hd.
#include "attention_params.h"

namespace osk_attn {
namespace {

template <int HD>
struct W64 {
  static constexpr int NKS = (HD + 15) / 16;        // QK^T k-steps
  static constexpr int NDT = (HD + 31) / 32;        // O^T row tiles
  static constexpr int HDV = NDT * 32;
  static constexpr int NKI = HD / 64;               // full 64-dim K images
  static constexpr int KREM = (HD % 64) / 8;        // remainder chunks (0 or 1)
  static_assert(KREM <= 1 && HD % 8 == 0, "head_dim must be 64 j or 64 j + 8");
  static constexpr int KTILE = NKI * 8192 + KREM * 1024;
  static constexpr int VTILE = HDV * 128;
  static constexpr int NKD = NKI * 8 + KREM;        // LDS-DMA wave-instructions per K tile
  static constexpr int NVD = HD / 8;                // ... per V^T tile (8 rows each)
  static constexpr int KI = (NKD + 3) / 4;          // per wave
  static constexpr int VI = (NVD + 3) / 4;
  static constexpr int KOFF = 0;
  static constexpr int VOFF = 2 * KTILE;
  static constexpr int SMEM = 2 * KTILE + 2 * VTILE;
  static constexpr bool ONES_ROW = (HD % 32) != 0;
};

OSK_DEV void glds16(const unsigned short* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int HD, int HINTS>
__global__ void __launch_bounds__(256, 1) attn_w64_kernel(const AttnParams p) {
  using C = W64<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bh, qb;
  block_to_work(p, (p.Lq + 255) / 256, bh, qb);
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- V^T padding rows [HD, HDV) of both ring slots: written once, never touched by the DMA
  if constexpr (C::HDV > HD) {
    constexpr int PADW = (C::HDV - HD) * 128 / 4;  // dwords per slot
    for (int i = tid; i < 2 * PADW; i += 256) {
      const int slot = i / PADW, w = i - slot * PADW;
      const unsigned val = (C::ONES_ROW && w < 32) ? 0x3F803F80u : 0u;  // first padding row = 1.0
      reinterpret_cast<unsigned*>(smem + C::VOFF + slot * C::VTILE + HD * 128)[w] = val;
    }
  }

  // ---- Q fragments (B operand) of the two query blocks: Q[q][ks*16 + hi*8 .. +8)
  int qi[2];
  bf16x8_t qf[2][C::NKS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    qi[u] = qb * 256 + wave * 64 + u * 32 + l31;
    const int qc = qi[u] < p.Lq ? qi[u] : p.Lq - 1;
    const unsigned short* qrow = p.q + b * p.qbs + (int64_t)qc * p.qrs + h * HD;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e0 < HD) v = *reinterpret_cast<const uint4*>(qrow + e0);
      qf[u][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- LDS-DMA sources.  K instruction j -> wave j & 3 (slot i = j >> 2); V^T instruction j -> wave 3 - (j & 3).
  const int srow8 = lane >> 3, spos = lane & 7;
  const unsigned short* kp[C::KI];
  int kclamp_row[C::KI];  // key row (inside the tile) this lane fetches for slot i
  const int bkv = b % p.Bkv;
  const unsigned short* kseg = p.k + bkv * p.kbs + h * HD;
#pragma unroll
  for (int i = 0; i < C::KI; ++i) {
    const int j = wave + 4 * i;
    int row, eoff;
    if (j < C::NKI * 8) {
      const int m = j >> 3, blk = j & 7;
      row = blk * 8 + srow8;
      eoff = m * 64 + ((spos ^ ((row >> 1) & 7)) << 3);
    } else {  // remainder column image: lane = key row, dims [64 NKI, +8)
      row = lane;
      eoff = C::NKI * 64;
    }
    kclamp_row[i] = row;
    kp[i] = kseg + (int64_t)row * p.krs + eoff;
  }
  const unsigned short* vp[C::VI];
#pragma unroll
  for (int i = 0; i < C::VI; ++i) {
    const int j = (3 - wave) + 4 * i;
    const int d = (j < C::NVD ? j : 0) * 8 + srow8;
    vp[i] = p.vt + ((int64_t)(bkv * p.H + h) * HD + d) * p.seg_lp + ((spos ^ ((d >> 1) & 7)) << 3);
  }
  const int64_t k_tile_step = (int64_t)64 * p.krs;
  const int64_t k_seg_jump = p.kss - (int64_t)p.tps * 64 * p.krs;
  const int64_t v_seg_jump = p.vtss - (int64_t)p.tps * 64;
  const int last_valid = p.seg_len - (p.tps - 1) * 64;  // keys in the last tile of a segment (1..64)
  int ktt = 0, vtt = 0;                                 // tile-in-segment counters of the two loaders

  // DMA of the K loader's current tile into ring slot SLOT, then advance the loader (ragged last tile of a
  // segment: rows past the segment re-fetch its last key; their scores are masked)
#define K_DMA(SLOT)                                                                                     \
  {                                                                                                     \
    unsigned char* kdst_ = smem + C::KOFF + (SLOT) * C::KTILE;                                          \
    const bool ragged_ = (ktt == p.tps - 1) && (last_valid < 64);                                       \
    _Pragma("unroll") for (int i = 0; i < C::KI; ++i) {                                                 \
      const int j_ = wave + 4 * i;                                                                      \
      if (j_ < C::NKD) {                                                                                \
        const unsigned short* src_ = kp[i];                                                             \
        if (ragged_ && kclamp_row[i] >= last_valid) src_ += (int64_t)(last_valid - 1 - kclamp_row[i]) * p.krs; \
        glds16(src_, kdst_ + j_ * 1024);                                                                \
      }                                                                                                 \
      kp[i] += k_tile_step;                                                                             \
    }                                                                                                   \
    if (++ktt == p.tps) {                                                                               \
      ktt = 0;                                                                                          \
      _Pragma("unroll") for (int i = 0; i < C::KI; ++i) kp[i] += k_seg_jump;                            \
    }                                                                                                   \
  }
#define V_DMA(SLOT)                                                                                     \
  {                                                                                                     \
    unsigned char* vdst_ = smem + C::VOFF + (SLOT) * C::VTILE;                                          \
    _Pragma("unroll") for (int i = 0; i < C::VI; ++i) {                                                 \
      const int j_ = (3 - wave) + 4 * i;                                                                \
      if (j_ < C::NVD) glds16(vp[i], vdst_ + j_ * 1024);                                                \
      vp[i] += 64;                                                                                      \
    }                                                                                                   \
    if (++vtt == p.tps) {                                                                               \
      vtt = 0;                                                                                          \
      _Pragma("unroll") for (int i = 0; i < C::VI; ++i) vp[i] += v_seg_jump;                            \
    }                                                                                                   \
  }

  // ---- fragment read offsets: 128-byte rows, chunk (2 j + hi) lands at position (2 j + hi) ^ sw
  const int sw = (l31 >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fo[j] = l31 * 128 + (((2 * j + hi) ^ sw) << 4);
  const int fr = l31 * 16;  // remainder column image

#define K_FRAG(SLOT, T2, KS)                                                                            \
  (((KS) < C::NKI * 4)                                                                                  \
       ? *reinterpret_cast<const bf16x8_t*>(smem + C::KOFF + (SLOT) * C::KTILE + ((KS) >> 2) * 8192 +   \
                                            (T2) * 4096 + fo[(KS) & 3])                                 \
       : *reinterpret_cast<const bf16x8_t*>(smem + C::KOFF + (SLOT) * C::KTILE + C::NKI * 8192 +        \
                                            (T2) * 512 + fr))
#define V_FRAG(SLOT, D, G) \
  (*reinterpret_cast<const bf16x8_t*>(smem + C::VOFF + (SLOT) * C::VTILE + (D) * 4096 + fo[G]))

  // S^T tiles of key tile in ring slot SLOT for both query blocks: SDST[u][t2]
#define QK_TILE(SDST, SLOT)                                                                             \
  {                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) SDST[u][t2][r] = 0.f;                            \
    _Pragma("unroll") for (int ks = 0; ks < C::NKS; ++ks) {                                             \
      _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                                \
        const bf16x8_t kf_ = K_FRAG(SLOT, t2, ks);                                                      \
        SDST[0][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[0][ks], SDST[0][t2], 0, 0, 0);    \
        SDST[1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[1][ks], SDST[1][t2], 0, 0, 0);    \
      }                                                                                                 \
    }                                                                                                   \
  }
#define MASK_TILE(S, VALID)                                                                             \
  {                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
      const int kl_ = t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                        \
      if (kl_ >= (VALID)) S[u][t2][r] = -INFINITY;                                                      \
    }                                                                                                   \
  }
  // row max of a score tile per query block, the new running max, the accumulator rescale factor, and whether
  // any lane of the wave needs the rescale
#define ROW_MAX(S)                                                                                      \
  {                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
      float m0_ = fmaxf(fmaxf(S[u][0][0], S[u][0][1]), S[u][0][2]);                                     \
      float m1_ = fmaxf(fmaxf(S[u][1][0], S[u][1][1]), S[u][1][2]);                                     \
      _Pragma("unroll") for (int r = 3; r < 15; r += 2) {                                               \
        m0_ = fmaxf(fmaxf(m0_, S[u][0][r]), S[u][0][r + 1]);                                            \
        m1_ = fmaxf(fmaxf(m1_, S[u][1][r]), S[u][1][r + 1]);                                            \
      }                                                                                                 \
      float mt_ = fmaxf(fmaxf(m0_, m1_), fmaxf(S[u][0][15], S[u][1][15]));                              \
      const unsigned mu_ = __float_as_uint(mt_);                                                        \
      auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                              \
      mt_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                    \
      m_new[u] = fmaxf(m_run[u], mt_);                                                                  \
      alpha[u] = __builtin_amdgcn_exp2f((m_run[u] - m_new[u]) * p.sc);                                  \
    }                                                                                                   \
    resc = !__all(m_new[0] == m_run[0] && m_new[1] == m_run[1]);                                        \
  }

  f32x16_t o[2][C::NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < C::NDT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][d][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  float m_new[2], alpha[2];
  int resc;
  const int nt = p.n_seg * p.tps;

  // ---- prologue: K0, V0, K1 in flight together; scores and row max of tile 0
  K_DMA(0);
  V_DMA(0);
  if (nt > 1) K_DMA(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // also covers the padding-row writes

  f32x16_t sa_[2][2], sb_[2][2];  // score tiles [query block][32-key half]; roles alternate every iteration
  int ctt = 0;                    // tile-in-segment counter of the tile whose scores are "current"
  QK_TILE(sa_, 0);
  if (p.tps == 1 && last_valid < 64) MASK_TILE(sa_, last_valid);
  ROW_MAX(sa_);

  // one iteration: SC = scores of tile t (masked, row max known), SN receives tile t+1.  K_t+1 sits in ring slot
  // CUR ^ 1, V_t in slot CUR; K_t+2 is fetched into slot CUR (K_t was last read in iteration t-1), V_t+1 into
  // slot CUR ^ 1 (V_t-1 was last read in iteration t-1).
#define ITERATION(SC, SN, CUR)                                                                          \
  {                                                                                                     \
    const bool has1_ = t + 1 < nt, has2_ = t + 2 < nt;                                                  \
    if (has2_) K_DMA(CUR);                                                                              \
    if (has1_) V_DMA((CUR) ^ 1);                                                                        \
    if (resc) { /* rare after the first tiles: new running max somewhere in the wave */                 \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int d = 0; d < C::NDT; ++d)  \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) o[u][d][r] *= alpha[u];                        \
    }                                                                                                   \
    if constexpr (!C::ONES_ROW) { l_run[0] *= alpha[0]; l_run[1] *= alpha[1]; }                         \
    m_run[0] = m_new[0];                                                                                \
    m_run[1] = m_new[1];                                                                                \
    const float msc0_ = m_run[0] * p.sc, msc1_ = m_run[1] * p.sc;                                       \
    /* region A: QK^T of tile t+1 (MFMA) || exp2 + pack of tile t (VALU).  Unconditional: on the last tile it  \
       multiplies stale LDS data and the result is dropped. */                                          \
    QK_TILE(SN, (CUR) ^ 1);                                                                             \
    bf16x8_t pb_[2][4];                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
      const float msc_ = u ? msc1_ : msc0_;                                                             \
      _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) _Pragma("unroll") for (int r = 0; r < 16; ++r)   \
          SC[u][t2][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(SC[u][t2][r], p.sc, -msc_));            \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
        const int t2 = g >> 1, r0 = (g & 1) * 8;                                                        \
        uint4 w_;                                                                                       \
        w_.x = pack_bf16x2(SC[u][t2][r0 + 0], SC[u][t2][r0 + 1]);                                       \
        w_.y = pack_bf16x2(SC[u][t2][r0 + 2], SC[u][t2][r0 + 3]);                                       \
        w_.z = pack_bf16x2(SC[u][t2][r0 + 4], SC[u][t2][r0 + 5]);                                       \
        w_.w = pack_bf16x2(SC[u][t2][r0 + 6], SC[u][t2][r0 + 7]);                                       \
        pb_[u][g] = __builtin_bit_cast(bf16x8_t, w_);                                                   \
      }                                                                                                 \
    }                                                                                                   \
    { /* anchor: all packed P words exist HERE (keeps the exp2 / pack work inside region A) */          \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int g = 0; g < 4; ++g) {     \
        const uint4 a_ = __builtin_bit_cast(uint4, pb_[u][g]);                                          \
        asm volatile("" ::"v"(a_.x), "v"(a_.y), "v"(a_.z), "v"(a_.w));                                  \
      }                                                                                                 \
    }                                                                                                   \
    if constexpr (HINTS == 1) { /* 1 K-fragment read : 2 MFMAs, a slice of the VALU work behind every MFMA */ \
      constexpr int NM_ = 4 * C::NKS;                                                                   \
      constexpr int NV_ = (C::ONES_ROW ? 160 : 160) / NM_;                                              \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                                 \
        if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x002, NV_, 0);                                            \
      }                                                                                                 \
    }                                                                                                   \
    if (++ctt == p.tps) ctt = 0;                                                                        \
    if (has1_ && ctt == p.tps - 1 && last_valid < 64) MASK_TILE(SN, last_valid);                        \
    /* region B: O^T += V^T . P^T (MFMA) || row sums of tile t, row max of tile t+1 (VALU) */           \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                     \
      _Pragma("unroll") for (int d = 0; d < C::NDT; ++d) {                                              \
        const bf16x8_t vf_ = V_FRAG(CUR, d, g);                                                         \
        o[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_, pb_[0][g], o[0][d], 0, 0, 0);            \
        o[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_, pb_[1][g], o[1][d], 0, 0, 0);            \
      }                                                                                                 \
    }                                                                                                   \
    if constexpr (!C::ONES_ROW) {                                                                       \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                   \
        float rs0_ = 0.f, rs1_ = 0.f;                                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { rs0_ += SC[u][0][r]; rs1_ += SC[u][1][r]; }    \
        l_run[u] += rs0_ + rs1_;                                                                        \
      }                                                                                                 \
    }                                                                                                   \
    ROW_MAX(SN);                                                                                        \
    if constexpr (HINTS == 1) {                                                                         \
      constexpr int NM_ = 8 * C::NDT;                                                                   \
      _Pragma("unroll") for (int i = 0; i < NM_; ++i) {                                                 \
        if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);                                              \
        __builtin_amdgcn_sched_group_barrier(0x002, C::ONES_ROW ? 2 : 4, 1);                            \
      }                                                                                                 \
    }                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    __syncthreads();                                                                                    \
  }

  for (int t = 0; t < nt; t += 2) {
    ITERATION(sa_, sb_, 0);
    ++t;
    if (t < nt) ITERATION(sb_, sa_, 1);
    --t;
  }
#undef ITERATION
#undef ROW_MAX
#undef MASK_TILE
#undef QK_TILE
#undef K_FRAG
#undef V_FRAG
#undef K_DMA
#undef V_DMA

  // ---- epilogue
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float l_tot;
    if constexpr (C::ONES_ROW) {
      // row HD of O^T = sum_k P: lanes hi == 0, register of d_local = HD % 32 (d_local = (r&3) + 8 (r>>2) + 4 hi)
      constexpr int dloc = HD % 32;
      constexpr int rr = (dloc & 3) + 4 * (dloc >> 3);
      static_assert(((dloc >> 2) & 1) == 0, "ones row must sit in the hi == 0 half");
      const unsigned lu = __float_as_uint(o[u][HD / 32][rr]);
      auto sw2 = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
      l_tot = __uint_as_float(sw2[0]);  // lanes 0-31 keep their own, lanes 32-63 receive lanes 0-31
    } else {
      const unsigned lu = __float_as_uint(l_run[u]);
      auto sw2 = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
      l_tot = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
    }
    const float inv = 1.0f / l_tot;
    if (qi[u] < p.Lq) {
      unsigned short* orow = p.out + b * p.obs + (int64_t)qi[u] * p.ors + h * HD;
#pragma unroll
      for (int d = 0; d < C::NDT; ++d) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d0 = d * 32 + qd * 8 + hi * 4;
          if (d0 < HD) {
            uint2 w;
            w.x = pack_bf16x2(o[u][d][qd * 4 + 0] * inv, o[u][d][qd * 4 + 1] * inv);
            w.y = pack_bf16x2(o[u][d][qd * 4 + 2] * inv, o[u][d][qd * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(orow + d0) = w;
          }
        }
      }
      if (p.lse && hi == 0)
        p.lse[(int64_t)bh * p.Lq + qi[u]] =
            (m_run[u] * p.sc + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
  }
}

template <typename KernelT>
int launch_kernel(KernelT kernel, int smem, const AttnParams& p, hipStream_t st, bool* attr_set) {
  if (!*attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    *attr_set = true;
  }
  const int nqb = (p.Lq + 255) / 256;
  dim3 grid(nqb * p.B * p.H), block(256);
  hipLaunchKernelGGL(kernel, grid, block, smem, st, p);
  return (int)hipGetLastError();
}

template <int HD>
int launch_hd(const AttnParams& p, int hints, hipStream_t st) {
  static bool set0 = false, set1 = false;
  if (hints == 1) return launch_kernel(attn_w64_kernel<HD, 1>, W64<HD>::SMEM, p, st, &set1);
  return launch_kernel(attn_w64_kernel<HD, 0>, W64<HD>::SMEM, p, st, &set0);
}

}  // namespace

int launch_w64(const AttnParams& p, int hd, int hints, hipStream_t st) {
  switch (hd) {
    case 64: return launch_hd<64>(p, hints, st);
    case 72: return launch_hd<72>(p, hints, st);
    case 128: return launch_hd<128>(p, hints, st);
    default: return OSK_EUNSUPPORTED;
  }
}

}  // namespace osk_attn
