// Flash-attention forward for head_dim 128 (the 11B MMDiT geometry), gfx950: hand-scheduled main loop.
//
// Same dataflow and generator as attention_asm72.hip (see tools/gen_attn_asm.py), 4 waves x 64 query rows, one wave
// per SIMD; what differs with 128 real dims:
//  * K tile = two swizzled 64-dim LDS images (dims 0..63, 64..127), 16 LDS-DMA instructions, 4 per wave;
//  * the reference max M rides in a NINTH, pure-padding QK^T k-step whose K fragment is a constant register quad
//    {1.0, 0, ...} (no LDS read) and whose Q fragment holds -M in dim 128;
//  * V^T tile = 128 dim rows + the ones row 128 (softmax denominator = accumulator row 128, a fifth O^T row tile).
// Numerics are those of the head_dim-72 kernel: P = exp2(S') with S' = q.k - M straight out of the MFMA, M moves only
// when a row max exceeds it by more than 8 (log2 units).
#include "acc_quads.h"
#include "attention_params.h"
#include "attention_asm_regs.inc"

namespace osk_attn {
namespace {

constexpr int HD = 128, NKS = OSK128_NKS, NDT = OSK128_NDT, NU = 2, NW = 4, NSLOT = OSK128N2_NSLOT;
static_assert(NKS == 9 && NDT == 5 && NSLOT == 4, "generated geometry changed: update the wrapper");

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) {
  return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v);
}

template <bool FAST>   // FAST: a score bound was given (attention_params.h::attn_fast_path); any key count, any segment layout
__global__ void __launch_bounds__(256, 1) attn_asm128_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bh, qb, part, tail_unit;
  const bool tail = block_to_work_split(p, (p.Lq + 255) / 256, bh, qb, part, tail_unit);
  const int b = bh / p.H, h = bh - b * p.H;
  float bound;   // the caller's score bound, or -- auto-dispatched pairs -- the one this (batch, head)'s operands imply (attention_params.h)
  if (!attn_auto_bound(p, b, h, FAST, bound)) return;
  (void)bound;   // (this head_dim's FAST body exponentiates the raw scores: the bound only decides which body runs)

  // ---- LDS: zero (rows 129..159 of the V^T slots stay zero for good), ones row 128 of both V^T slots
  for (int i = tid; i < OSK128_SMEM / 16; i += 64 * NW) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 64) {
    const int slot = tid >> 5;
    reinterpret_cast<unsigned*>(smem + OSK128_VOFF0 + slot * OSK128_VTILE + HD * 128)[tid & 31] = 0x3F803F80u;
  }
  // ragged last key tile of a segment: see attention_asm72.hip (clamped K rows + validity-mask ones row)
  const int last_valid = p.seg_len - (p.tps - 1) * 64;
  const KeyPart kp = key_part(p, tail, part, last_valid < 64);   // the whole key axis, or one part of a split tail unit
  const bool ragged = kp.ragged;
  unsigned maskval = 0;
  if (lane < 32) {
    // row 128 is not swizzled ((128 >> 1) & 7 == 0): dword `lane` = columns 2 lane, 2 lane + 1 of the V^T tile,
    // whose baked key order is key = 16 (c / 16) + perm(c % 16)
    const int c0 = lane * 2, c1 = c0 + 1;
    auto key_of = [](int c) { const int j = c & 15; return (c & ~15) + ((j & 3) | ((j & 4) << 1) | ((j & 8) >> 1)); };
    maskval = (key_of(c0) < last_valid ? 0x3F80u : 0u) | (key_of(c1) < last_valid ? 0x3F800000u : 0u);
  }
  if (ragged && kp.tps == 1 && tid < 32)
    reinterpret_cast<unsigned*>(smem + OSK128_VOFF0 + HD * 128)[tid] = maskval;
  __syncthreads();

  // ---- Q fragments (pre-scaled by scale*log2(e)) -> AGPRs; k-step 8 = padding (zero; dim 128 receives -M in the asm)
  int qi[NU];
  osk_v4f qv[NU * 9];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    qi[u] = qb * 256 + wave * 64 + u * 32 + l31;
    const int qc = qi[u] < p.Lq ? qi[u] : p.Lq - 1;
    const unsigned short* qrow = p.q + b * p.qbs + (int64_t)qc * p.qrs + h * HD;
    unsigned w[NKS * 4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      uint4 s = make_uint4(0, 0, 0, 0);
      if (ks < HD / 16) {
        s = *reinterpret_cast<const uint4*>(qrow + ks * 16 + hi * 8);
        if (!p.q_prescaled) {
          float f[8];
          unpack8(s, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] *= p.sc;
          s = pack8(f);
        }
      }
      w[ks * 4 + 0] = s.x; w[ks * 4 + 1] = s.y; w[ks * 4 + 2] = s.z; w[ks * 4 + 3] = s.w;
    }
    // FAST: with a score bound B <= 56 no softmax reference is needed for range control: the body skips the padding k-step
    // (whose only job is to carry the reference through the MFMA) and exponentiates the raw scores, P = exp2(s) in
    // [2^-56, 2^56]; O and the row sum carry the same factor (attention_asm128_n2_f0.inc, tools/gen_attn_asm.py "nom")
    // Q fragments as VALUES: quad ks of block u; the loop statement takes them as inputs in their fixed AGPRs (acc_quads.h), so the
    // compiler writes them there itself and knows they are live until the loop has read them
#pragma unroll
    for (int ks = 0; ks < 9; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) qv[u * 9 + ks][i] = __uint_as_float(w[ks * 4 + i]);
  }

  // ---- per-lane LDS-DMA source offsets: K instruction j = wave + 4 i -> image j / 8, key rows 8 (j % 8) + lane / 8;
  //      V^T instruction j = (3 - wave) + 4 i -> dim rows 8 j + lane / 8
  const int srow8 = lane >> 3, spos = lane & 7;
  unsigned koff[NSLOT], koffL[NSLOT], voff[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int j = wave + NW * i;
    const int img = j >> 3, row = (j & 7) * 8 + srow8;
    const int rowL = row < last_valid ? row : last_valid - 1;
    const int ch = img * 64 + ((spos ^ ((row >> 1) & 7)) << 3);
    koff[i] = (unsigned)(((int64_t)row * p.krs + ch) * 2);
    koffL[i] = (unsigned)(((int64_t)rowL * p.krs + ch) * 2);
    const int d = ((NW - 1 - wave) + NW * i) * 8 + srow8;
    voff[i] = (unsigned)(((int64_t)d * p.seg_lp + ((spos ^ ((d >> 1) & 7)) << 3)) * 2);
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (l31 >> 1) & 7;
  unsigned fo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fo[j] = lds_base + l31 * 128 + (((2 * j + hi) ^ sw) << 4);
  const unsigned onesaddr = lds_base + OSK128_VOFF0 + HD * 128 + lane * 4;   // lanes 32..63: the zero row behind it

  const int bkv = b % p.Bkv;
  const uint64_t kbase = rfl64((uint64_t)(uintptr_t)(p.k + bkv * p.kbs + h * HD + kp.k_off));
  const uint64_t vbase = rfl64((uint64_t)(uintptr_t)(p.vt + (int64_t)(bkv * p.H + h) * HD * p.seg_lp + kp.v_off));
  const unsigned kstep = rfl((unsigned)(128 * p.krs));
  const uint64_t kjump = rfl64((uint64_t)((p.kss - (int64_t)p.tps * 64 * p.krs) * 2));
  const uint64_t vjump = rfl64((uint64_t)((p.vtss - (int64_t)p.tps * 64) * 2));
  const unsigned tps = rfl((unsigned)kp.tps), nt = rfl((unsigned)kp.nt);
  const unsigned kdst = rfl(lds_base + wave * 1024), vdst = rfl(lds_base + OSK128_VOFF0 + (NW - 1 - wave) * 1024);
  const unsigned nvw = rfl((unsigned)NSLOT | (ragged ? 0u : 1u << 8) | ((ragged && wave == 0) ? 1u << 9 : 0u));

  float m_ref[2];
#define OSK128_OPERANDS                                                                                             \
  : "=&v"(m_ref[0]), "=&v"(m_ref[1])                                                                                 \
  : "v"(koff[0]), "v"(koff[1]), "v"(koff[2]), "v"(koff[3]), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), \
    "v"(fo[0]), "v"(fo[1]), "v"(fo[2]), "v"(fo[3]), "v"(koffL[0]), "v"(koffL[1]), "v"(koffL[2]), "v"(koffL[3]),      \
    "v"(maskval), "v"(onesaddr), "s"(kbase), "s"(vbase),                                                             \
    "s"(kstep), "s"(kjump), "s"(vjump), "s"(tps), "s"(nt), "s"(kdst), "s"(vdst), "s"(nvw), \
    OSK_AQ_IN_40_5(qv), OSK_AQ_IN_45_4(qv + 5), OSK_AQ_IN_49_5(qv + 9), OSK_AQ_IN_54_4(qv + 14)
  if constexpr (FAST) {
    asm volatile(
#include "attention_asm128_n2_f0.inc"
        OSK128_OPERANDS : OSK128N2_CLOBBERS);
    m_ref[0] = m_ref[1] = 0.f;   // reference point of the exponentials: 0
  } else {
    asm volatile(
#include "attention_asm128_n2_v0.inc"
        OSK128_OPERANDS : OSK128N2_CLOBBERS);
  }

  // the O^T accumulators as values the compiler knows (acc_quads.h): outputs of an empty statement right behind the loop
  static_assert(OSK128N2_AQ0 == 160 && OSK128N2_AQ1 == 196 && OSK128N2_AO_REGS == 160,
                "the generated loops' register map: the operand lists above and below bind exactly these AGPRs");
  osk_v4f ov[40];
  asm volatile("" : OSK_AQ_OUT_0_40(ov));

  // ---- epilogue: O^T out of the AGPRs, normalise by accumulator row 128 (sum of P), store
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float o[NDT][16];
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
#pragma unroll
      for (int i = 0; i < 16; ++i)   // row tile (u, d) = registers 16 (u NDT + d) ..: in place, in program order
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o[d][i]) : "a"(ov[(u * NDT + d) * 4 + i / 4][i % 4]));
    }
    // row 128 of O^T = sum_k P: row 0 of row tile 4 = lanes hi == 0, register 0
    const unsigned lu = __float_as_uint(o[4][0]);
    auto sw2 = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    const float l_tot = __uint_as_float(sw2[0]);
    const float inv = 1.0f / l_tot;
    if (tail) {
      // part of a split tail unit: normalised partial O (f32) + log2-domain LSE -> workspace (attn_merge_kernel)
      if (qi[u] < p.Lq) {
        const int64_t slot = ((int64_t)tail_unit * p.tail_split + part) * 256 + (wave * 64 + u * 32 + l31);
        float* wo = p.ws_o + slot * HD;
#pragma unroll
        for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int d0 = d * 32 + qd * 8 + hi * 4;
          {
              *reinterpret_cast<float4*>(wo + d0) = make_float4(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv,
                                                                 o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
            }
          }
        }
        if (hi == 0) p.ws_lse[slot] = m_ref[u] + __builtin_amdgcn_logf(l_tot);
      }
    } else if (qi[u] < p.Lq) {
      unsigned short* orow = p.out + b * p.obs + (int64_t)qi[u] * p.ors + h * HD;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d0 = d * 32 + qd * 8 + hi * 4;
          uint2 w2;
          w2.x = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
          w2.y = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d0) = w2;
        }
      }
      if (p.lse && hi == 0)
        p.lse[(int64_t)bh * p.Lq + qi[u]] = (m_ref[u] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
  }
}

template <bool FAST>
int launch_one(const AttnParams& p, hipStream_t st) {
  auto kernel = attn_asm128_kernel<FAST>;
  OSK_ENSURE_MAX_SMEM(kernel, OSK128_SMEM);
  const int units = ((p.Lq + 255) / 256) * p.B * p.H;
  const int tail_units = p.tail_split > 1 ? units - p.tail_first : 0;
  dim3 grid(units + tail_units * (p.tail_split - 1)), block(64 * NW);
  hipLaunchKernelGGL(kernel, grid, block, OSK128_SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

int launch_asm128(const AttnParams& p, hipStream_t st) { return attn_fast_path(p) ? launch_one<true>(p, st) : launch_one<false>(p, st); }

}  // namespace osk_attn
