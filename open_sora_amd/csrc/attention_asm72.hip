// Flash-attention forward for head_dim 72 (DiT-XL geometry), gfx950: hand-scheduled main loop.
//
// Structure: 4 waves x 64 query rows, one wave per SIMD (the generator can also emit 8 waves x 32 rows, two per SIMD),
// LDS-DMA staged 64-key tiles, swapped operands: S^T = K . Q^T on v_mfma_f32_32x32x16_bf16 (80 padded dims), O^T += V^T . P^T
// on v_mfma_f32_16x16x32_bf16 (80 padded rows = 5 blocks of 16 instead of 3 x 32; P crosses from the 32-query score layout
// to the 16-query operand layout by one v_permlane16_swap per packed register pair), and the whole K/V loop is ONE asm
// statement emitted by tools/gen_attn_asm.py (attention_asm72_n{NU}_v{VAR}.inc): explicit register file (O^T and Q in AGPRs, two score tiles,
// P and two 4-slot fragment rings in VGPRs), every MFMA shadow filled by hand with ~5 issue slots of LDS reads /
// exp2 / pack / max work, counted lgkmcnt waits, one barrier per tile.  See the generator's header for the
// dataflow; this file is the wrapper: LDS init, Q pre-scale, the asm operands, the epilogue.
//
// Numerics vs the compiler-scheduled attention_fwd.hip (head_dim 64; rounds 1-2 also had a head_dim-72 twin, attention_w64.hip):
//  * Q carries scale*log2(e): folded into q's single rounding by osk_qknorm_rope_bf16 (q_prescaled, the model path),
//    or applied here with one extra bf16 rounding of q (stand-alone calls);
//  * the online-softmax reference max M is kept bf16-exact inside the contraction (Q padding dim 72 = -M, K
//    padding dim 72 = 1.0), so P = exp2(S') with S' straight out of the MFMA; M moves only when a row max exceeds
//    it by more than 8 (log2 units): P <= 2^8, O and the row sum (ones row of V^T) carry the same factor.
// Any key count: a ragged last tile of a key segment re-fetches the segment's last key for the missing rows and masks
// them out of the denominator through the ones row (see the wrapper).
#include "acc_quads.h"
#include "attention_params.h"
#include "attention_asm_regs.inc"

namespace osk_attn {
namespace {

constexpr int HDL = 72, NKS = 5, NDT = 3;   // HDL: the head_dim of the LDS images / register layout; the tensors' head_dim is the template parameter HD (72, or 64: see below)

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) {
  return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v);
}

// HD = 64 (round 4: the S-width models; the compiler-scheduled attn_fwd_kernel<64> is gone): the SAME loop on tensors with 64-wide heads --
// Q's dims 64..71 are zero, K's 8-dim column image is never fetched (its LDS chunk stays zero: the loader slot that fetches it does not
// exist), V^T has 64 rows (LDS rows 64..71 stay zero), the ones row stays LDS row 72.  80 padded dims for 64 real ones: issued / useful 1.25.
template <bool FAST, int HD>   // FAST: a score bound was given (attention_params.h::attn_fast_path); any key count, any segment layout
__global__ void __launch_bounds__(256, 1) attn_asm72_kernel(const AttnParams p) {
  constexpr int NU = 2;                                        // 32-row query blocks per wave (the generator's 8 waves x 32
  constexpr int NW = 8 / NU;                                   // rows layout tied this one in rounds 1-2 and is not shipped)
  constexpr int NSLOT = OSK72N2_NSLOT;                         // LDS-DMA slots per wave and tile
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

  // ---- LDS: zero (a tile slot that is never filled must hold finite data), ones row of both V^T slots,
  //      constant chunk {1.0, 0 x 7} = K's padding dims 72..79
  for (int i = tid; i < OSK72_SMEM / 16; i += 64 * NW) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 64) {
    const int slot = tid >> 5;
    reinterpret_cast<unsigned*>(smem + OSK72_VOFF0 + slot * OSK72_VTILE + HDL * 128)[tid & 31] = 0x3F803F80u;
  }
  if (tid == 64 || tid == 65)  // one copy per K ring slot, KTILE apart (the slot is an immediate offset in the asm)
    *reinterpret_cast<unsigned*>(smem + OSK72_CONST_OFF + (tid - 64) * OSK72_KTILE) = 0x00003F80u;
  // ragged last key tile of a segment (seg_len % 64 != 0): K rows past the segment re-fetch its last key (finite
  // scores), V^T is zero there (osk_v_transpose_bf16 pads), and the tile's ones row becomes a validity mask so the
  // duplicates do not count in the softmax denominator.  The mask is in the V^T tile's baked key order.
  const int last_valid = p.seg_len - (p.tps - 1) * 64;   // keys in the last tile of a segment (1..64)
  const KeyPart kp = key_part(p, tail, part, last_valid < 64);   // the whole key axis, or one part of a split tail unit
  const bool ragged = kp.ragged;
  unsigned maskval = 0;
  if (lane < 32) {
    // dword `lane` of LDS row 72: 16-byte position lane / 4 holds logical chunk c = (lane / 4) ^ ((72 >> 1) & 7) of the swizzled
    // 128-byte row image; chunk c of the V^T tile = 32-key half c / 4, lane row c % 4, whose 8 keys (baked by
    // osk_v_transpose_bf16 for this head_dim) are PV16_KEYS[c % 4] (tools/gen_attn_asm.py)
    const int c = (lane >> 2) ^ ((HDL >> 1) & 7), e0 = (lane & 3) * 2;
    auto key_of = [](int c_, int e_) {
      const int r_ = c_ & 3;   // lane row: keys {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31}
      return 32 * (c_ >> 2) + ((r_ & 1) << 4) + ((r_ >> 1) << 2) + ((e_ >> 2) << 3) + (e_ & 3);
    };
    maskval = (key_of(c, e0) < last_valid ? 0x3F80u : 0u) | (key_of(c, e0 + 1) < last_valid ? 0x3F800000u : 0u);
  }
  if (ragged && kp.tps == 1 && tid < 32)                 // tile 0 itself is ragged: no loop body precedes it
    reinterpret_cast<unsigned*>(smem + OSK72_VOFF0 + HDL * 128)[tid] = maskval;
  __syncthreads();

  // ---- Q fragments, pre-scaled by scale*log2(e), -> AGPRs (u-major, k-step, 4 words)
  int qi[NU];
  osk_v4f qv[NU * 5];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    qi[u] = qb * 256 + wave * (32 * NU) + u * 32 + l31;
    const int qc = qi[u] < p.Lq ? qi[u] : p.Lq - 1;
    const unsigned short* qrow = p.q + b * p.qbs + (int64_t)qc * p.qrs + h * HD;
    unsigned w[NKS * 4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int e0 = ks * 16 + hi * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e0 < HD) v = *reinterpret_cast<const uint4*>(qrow + e0);
      uint4 s = v;
      if (!p.q_prescaled) {  // fold scale*log2(e) in here (one extra bf16 rounding of q); see osk.h
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= p.sc;
        s = pack8(f);
      }
      w[ks * 4 + 0] = s.x; w[ks * 4 + 1] = s.y; w[ks * 4 + 2] = s.z; w[ks * 4 + 3] = s.w;
    }
    // FAST: the reference "max" is the caller's bound B, constant for the whole launch: Q's padding dim 72 (k-step 4, lanes
    // 32..63, word 0 low half) = -B against the 1.0 in K's padding dim -> the MFMA returns s - B directly, from tile 0 on
    if constexpr (FAST) {
      if (hi) w[16] = (__float_as_uint(-bound) >> 16) & 0xFFFFu;
    }
    // Q fragments as VALUES: quad ks of block u; the loop statement takes them as inputs in their fixed AGPRs (acc_quads.h), so the
    // compiler writes them there itself and knows they are live until the loop has read them
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) qv[u * 5 + ks][i] = __uint_as_float(w[ks * 4 + i]);
  }

  // ---- per-lane LDS-DMA source offsets (bytes from the loader's tile base) of this wave's instruction slots:
  //      K instruction j = wave + NW i (j = 8: the 8-dim column image), V^T instruction j = (NW - 1 - wave) + NW i
  const int srow8 = lane >> 3, spos = lane & 7;
  unsigned koff[3] = {0, 0, 0}, koffL[3] = {0, 0, 0}, voff[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int j = wave + NW * i;
    unsigned o = 0, oL = 0;
    if (j < 8) {
      const int row = j * 8 + srow8;
      const int rowL = row < last_valid ? row : last_valid - 1;
      const int ch = (spos ^ ((row >> 1) & 7)) << 3;
      o = (unsigned)(((int64_t)row * p.krs + ch) * 2);
      oL = (unsigned)(((int64_t)rowL * p.krs + ch) * 2);
    } else if (j == 8) {
      const int rowL = lane < last_valid ? lane : last_valid - 1;
      o = (unsigned)(((int64_t)lane * p.krs + 64) * 2);
      oL = (unsigned)(((int64_t)rowL * p.krs + 64) * 2);
    }
    koff[i] = o;
    koffL[i] = oL;
    const int jv = (NW - 1 - wave) + NW * i;
    unsigned ov = 0;
    if (jv < HD / 8) {
      const int d = jv * 8 + srow8;
      ov = (unsigned)(((int64_t)d * p.seg_lp + ((spos ^ ((d >> 1) & 7)) << 3)) * 2);
    }
    voff[i] = ov;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (l31 >> 1) & 7;
  unsigned fo[4], kc[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) fo[j] = lds_base + l31 * 128 + (((2 * j + hi) ^ sw) << 4);
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)   // ring slot 1 = + KTILE (immediate), for the column image and the constant chunk alike
    kc[t2] = hi ? lds_base + OSK72_CONST_OFF : lds_base + 8192 + t2 * 512 + l31 * 16;
  const unsigned onesaddr = lds_base + OSK72_VOFF0 + HDL * 128 + lane * 4;   // lanes 32..63: the zero row behind it
  // V^T fragments of the 16x16x32 P.V product: lane (row r4 = lane / 16, dim l15 = lane % 16 of a 16-row block) reads chunk
  // 4 t2 + r4 of its row (+ block and ring-slot immediates in the asm); same swizzled 128-byte-row image as K
  const int l15 = lane & 15, r4 = lane >> 4;
  unsigned vo[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) vo[t2] = lds_base + l15 * 128 + (((4 * t2 + r4) ^ ((l15 >> 1) & 7)) << 4);

  const int bkv = b % p.Bkv;   // key / value batch of this query batch
  const uint64_t kbase = rfl64((uint64_t)(uintptr_t)(p.k + bkv * p.kbs + h * HD + kp.k_off));
  const uint64_t vbase = rfl64((uint64_t)(uintptr_t)(p.vt + (int64_t)(bkv * p.H + h) * HD * p.seg_lp + kp.v_off));
  const unsigned kstep = rfl((unsigned)(128 * p.krs));
  const uint64_t kjump = rfl64((uint64_t)((p.kss - (int64_t)p.tps * 64 * p.krs) * 2));
  const uint64_t vjump = rfl64((uint64_t)((p.vtss - (int64_t)p.tps * 64) * 2));
  const unsigned tpsnt = rfl((unsigned)kp.tps | ((unsigned)kp.nt << 16));   // (two operand slots went to vo[])
  const unsigned kdst = rfl(lds_base + wave * 1024), vdst = rfl(lds_base + OSK72_VOFF0 + (NW - 1 - wave) * 1024);
  // valid loader slots of this wave: the last one only where its instruction index is < 9
  const unsigned nkw = rfl(wave + NW * (NSLOT - 1) < HD / 8 ? (unsigned)NSLOT : (unsigned)(NSLOT - 1));   // (head_dim 64: no column image)
  const unsigned nvw = rfl(((NW - 1 - wave) + NW * (NSLOT - 1) < HD / 8 ? (unsigned)NSLOT : (unsigned)(NSLOT - 1)) |
                           (ragged ? 0u : 1u << 8) | ((ragged && wave == 0) ? 1u << 9 : 0u));
  const unsigned nkvw = rfl(nvw | (nkw << 16));

  float m_ref[2];
#define OSK72_OPERANDS                                                                                              \
  : "=&v"(m_ref[0]), "=&v"(m_ref[1])                                                                                 \
  : "v"(koff[0]), "v"(koff[1]), "v"(koff[2]), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(fo[0]), "v"(fo[1]),     \
    "v"(fo[2]), "v"(fo[3]), "v"(kc[0]), "v"(kc[1]), "v"(koffL[0]), "v"(koffL[1]), "v"(koffL[2]), "v"(maskval),      \
    "v"(onesaddr), "v"(vo[0]), "v"(vo[1]), "s"(kbase), "s"(vbase),                                                   \
    "s"(kstep), "s"(kjump), "s"(vjump), "s"(tpsnt), "s"(kdst), "s"(vdst), "s"(nkvw), \
    OSK_AQ_IN_20_5(qv), OSK_AQ_IN_25_5(qv + 5)
  if constexpr (FAST) {
    asm volatile(
#include "attention_asm72_n2_f0.inc"
        OSK72_OPERANDS : OSK72N2_CLOBBERS);
    m_ref[0] = m_ref[1] = bound;
  } else {
    asm volatile(
#include "attention_asm72_n2_v0.inc"
        OSK72_OPERANDS : OSK72N2_CLOBBERS);
  }

  // the O^T accumulators as values the compiler knows (acc_quads.h): outputs of an empty statement right behind the loop
  static_assert(OSK72N2_AQ0 == 80 && OSK72N2_AQ1 == 100 && OSK72N2_AO_REGS == 80,
                "the generated loops' register map: the operand lists above and below bind exactly these AGPRs");
  osk_v4f ov[20];
  asm volatile("" : OSK_AQ_OUT_0_20(ov));

  // ---- epilogue: O^T out of the AGPRs -- per 16-query block (u, half): lane = query l15, dims 16 db + 4 r4 + i in register
  //      4 db + i -- normalise by accumulator row 72 (sum of P: block 4, lane row 2, register 0), store
  static_assert(OSK72_NDB == 5, "epilogue written for 5 row blocks of 16");
#pragma unroll
  for (int u = 0; u < NU; ++u) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float o[20];
#pragma unroll
      for (int i = 0; i < 20; ++i)   // block (u, half) = registers 20 (2 u + half) ..: in place, in program order
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o[i]) : "a"(ov[(u * 2 + half) * 5 + i / 4][i % 4]));
      const float l_tot = __shfl(o[16], 32 + l15, 64);
      const float inv = 1.0f / l_tot;
      // the reference max lives in the SCORE layout (lane = query lane % 32 of block u): fetch this lane's query's
      const float mq = __shfl(m_ref[u], half * 16 + l15, 64);
      const int wrow = wave * (32 * NU) + u * 32 + half * 16 + l15;   // row inside the workgroup's 256
      const int qrow = qb * 256 + wrow;
      if (qrow >= p.Lq) continue;
      if (tail) {
        // part of a split tail unit: normalised partial O (f32) + log2-domain LSE -> workspace (attn_merge_kernel)
        const int64_t slot = ((int64_t)tail_unit * p.tail_split + part) * 256 + wrow;
        float* wo = p.ws_o + slot * HD;
#pragma unroll
        for (int db = 0; db < 5; ++db) {
          const int d0 = db * 16 + r4 * 4;
          if (d0 < HD)
            *reinterpret_cast<float4*>(wo + d0) = make_float4(o[db * 4 + 0] * inv, o[db * 4 + 1] * inv, o[db * 4 + 2] * inv, o[db * 4 + 3] * inv);
        }
        if (r4 == 0) p.ws_lse[slot] = mq + __builtin_amdgcn_logf(l_tot);
      } else {
        unsigned short* orow = p.out + b * p.obs + (int64_t)qrow * p.ors + h * HD;
#pragma unroll
        for (int db = 0; db < 5; ++db) {
          const int d0 = db * 16 + r4 * 4;
          if (d0 < HD) {
            uint2 w2;
            w2.x = pack_bf16x2(o[db * 4 + 0] * inv, o[db * 4 + 1] * inv);
            w2.y = pack_bf16x2(o[db * 4 + 2] * inv, o[db * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(orow + d0) = w2;
          }
        }
        if (p.lse && r4 == 0)
          p.lse[(int64_t)bh * p.Lq + qrow] = (mq + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
      }
    }
  }
}

template <bool FAST, int HD>
int launch_one(const AttnParams& p, hipStream_t st) {
  auto kernel = attn_asm72_kernel<FAST, HD>;
  OSK_ENSURE_MAX_SMEM(kernel, OSK72_SMEM);
  const int units = ((p.Lq + 255) / 256) * p.B * p.H;
  const int tail_units = p.tail_split > 1 ? units - p.tail_first : 0;
  dim3 grid(units + tail_units * (p.tail_split - 1)), block(256);
  hipLaunchKernelGGL(kernel, grid, block, OSK72_SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

int launch_asm72(const AttnParams& p, hipStream_t st) { return attn_fast_path(p) ? launch_one<true, 72>(p, st) : launch_one<false, 72>(p, st); }
int launch_asm64(const AttnParams& p, hipStream_t st) { return attn_fast_path(p) ? launch_one<true, 64>(p, st) : launch_one<false, 64>(p, st); }

}  // namespace osk_attn
