// Flash-attention forward for head_dim 72, gfx950, with the P.V product on the fp8 MFMA (opt-in fp8 mode): the
// head_dim-72 twin of attention_asm128p8.hip (see there): 4 waves x 64 query rows, QK^T and the softmax bookkeeping of
// attention_asm72.hip, P and V^T as e4m3, one v_mfma_f32_32x32x64_f8f6f4 per O^T row tile (3 x 64 cycles instead of
// 12 x 32), V^T with RP = 80 rows per head from osk_v_transpose_fp8 (72 dims, the ones / key-validity row 72, zero rows).
#include "acc_quads.h"
#include "attention_params.h"
#include "attention_asm_regs.inc"

namespace osk_attn {
namespace {

constexpr int HD = 72, NKS = 5, NDT = 3;
constexpr int NSLOT_V = OSK72P8N2_NSLOT_V, RP = OSK72P8_RP, NVD = OSK72P8_NVD;
static_assert(NSLOT_V == 2 && RP == 80 && NVD == 5, "generated geometry changed: update the wrapper");

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) {
  return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v);
}

template <int NU>
__global__ void __launch_bounds__(NU == 2 ? 256 : 512, NU == 2 ? 1 : 2) attn_asm72p8_kernel(const AttnParams p) {
  constexpr int NW = 8 / NU;                                   // waves per workgroup
  static_assert(NU == 2, "the fp8 P.V variant exists in the 4 waves x 64 rows layout only");
  constexpr int NSLOT = OSK72P8N2_NSLOT;  // LDS-DMA slots per wave and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bh, qb, part, tail_unit;
  const bool tail = block_to_work_split(p, (p.Lq + 255) / 256, bh, qb, part, tail_unit);
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- LDS: zero (a tile slot that is never filled must hold finite data), constant chunk {1.0, 0 x 7} = K's padding
  //      dims 72..79 (one copy per K ring slot)
  for (int i = tid; i < OSK72P8_SMEM / 16; i += 64 * NW) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid == 64 || tid == 65)
    *reinterpret_cast<unsigned*>(smem + OSK72P8_CONST_OFF + (tid - 64) * OSK72P8_KTILE) = 0x00003F80u;
  // ragged last key tile of a segment: K rows past the segment re-fetch its last key (finite scores); the key-validity
  // row of V^T comes baked from osk_v_transpose_fp8
  const int last_valid = p.seg_len - (p.tps - 1) * 64;
  const KeyPart kp = key_part(p, tail, part, last_valid < 64);
  const bool ragged = kp.ragged;
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
  unsigned koff[3] = {0, 0, 0}, koffL[3] = {0, 0, 0};
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
  }
  // V^T (e4m3, 64-byte rows): instruction j = (NW - 1 - wave) + NW i moves rows [16 j, 16 j + 16); LDS position lane % 4
  // of a row holds the 16-byte chunk (lane % 4) ^ ((row >> 2) & 3) of it
  // (64-byte rows: rows r, r + 4, r + 8, r + 12 share a 16-bank group, so the swizzle must tell THOSE apart)
  unsigned voff8[NSLOT_V];
#pragma unroll
  for (int i = 0; i < NSLOT_V; ++i) {
    const int jv = (NW - 1 - wave) + NW * i;
    const int row = (jv < NVD ? jv : 0) * 16 + (lane >> 2);
    voff8[i] = (unsigned)((int64_t)row * p.seg_lp + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (l31 >> 1) & 7;
  unsigned fo[4], kc[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) fo[j] = lds_base + l31 * 128 + (((2 * j + hi) ^ sw) << 4);
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)   // ring slot 1 = + KTILE (immediate), for the column image and the constant chunk alike
    kc[t2] = hi ? lds_base + OSK72P8_CONST_OFF : lds_base + 8192 + t2 * 512 + l31 * 16;
  // V^T fragment of a row tile: row l31, the 32-byte half hi of its 64 keys = logical chunks 2 hi, 2 hi + 1
  const int sw4 = (l31 >> 2) & 3;
  const unsigned vf0 = lds_base + l31 * 64 + (((2 * hi) ^ sw4) << 4), vf1 = lds_base + l31 * 64 + (((2 * hi + 1) ^ sw4) << 4);

  const int bkv = b % p.Bkv;   // key / value batch of this query batch
  const uint64_t kbase = rfl64((uint64_t)(uintptr_t)(p.k + bkv * p.kbs + h * HD + kp.k_off));
  const uint64_t vbase = rfl64((uint64_t)(uintptr_t)(p.vt8 + (int64_t)(bkv * p.H + h) * RP * p.seg_lp + kp.v_off));
  const unsigned kstep = rfl((unsigned)(128 * p.krs));
  const uint64_t kjump = rfl64((uint64_t)((p.kss - (int64_t)p.tps * 64 * p.krs) * 2));
  const uint64_t vjump = rfl64((uint64_t)(p.vtss - (int64_t)p.tps * 64));   // V^T strides are bytes here
  const unsigned tps = rfl((unsigned)kp.tps), nt = rfl((unsigned)kp.nt);
  const unsigned kdst = rfl(lds_base + wave * 1024), vdst = rfl(lds_base + OSK72P8_VOFF0 + (NW - 1 - wave) * 1024);
  // valid loader slots of this wave: the last one only where its instruction index is < 9
  const unsigned nkw = rfl(wave + NW * (NSLOT - 1) < 9 ? (unsigned)NSLOT : (unsigned)(NSLOT - 1));
  const unsigned nvw = rfl(((NW - 1 - wave) + NW * (NSLOT_V - 1) < NVD ? (unsigned)NSLOT_V : (unsigned)(NSLOT_V - 1)) |
                           (ragged ? 0u : 1u << 8));

  float m_ref[2];
#define OSK72P8_OPERANDS                                                                                            \
  : "=&v"(m_ref[0]), "=&v"(m_ref[1])                                                                                 \
  : "v"(koff[0]), "v"(koff[1]), "v"(koff[2]), "v"(voff8[0]), "v"(voff8[1]), "v"(fo[0]), "v"(fo[1]),                 \
    "v"(fo[2]), "v"(fo[3]), "v"(kc[0]), "v"(kc[1]), "v"(koffL[0]), "v"(koffL[1]), "v"(koffL[2]), "v"(vf0), "v"(vf1), \
    "s"(kbase), "s"(vbase),                                                                                          \
    "s"(kstep), "s"(kjump), "s"(vjump), "s"(tps), "s"(nt), "s"(kdst), "s"(vdst), "s"(nkw), "s"(nvw), \
    OSK_AQ_IN_24_5(qv), OSK_AQ_IN_29_5(qv + 5)
  asm volatile(
#include "attention_asm72p8_n2_v0.inc"
      OSK72P8_OPERANDS : OSK72P8N2_CLOBBERS);

  // the O^T accumulators as values the compiler knows (acc_quads.h): outputs of an empty statement right behind the loop
  static_assert(OSK72P8N2_AQ0 == 96 && OSK72P8N2_AQ1 == 116 && OSK72P8N2_AO_REGS == 96,
                "the generated loop's register map: the operand lists above and below bind exactly these AGPRs");
  osk_v4f ov[24];
  asm volatile("" : OSK_AQ_OUT_0_24(ov));

  // ---- epilogue: O^T out of the AGPRs, normalise by accumulator row 72 (sum of P), store
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float o[NDT][16];
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
#pragma unroll
      for (int i = 0; i < 16; ++i)   // row tile (u, d) = registers 16 (u NDT + d) ..: in place, in program order
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o[d][i]) : "a"(ov[(u * NDT + d) * 4 + i / 4][i % 4]));
    }
    // row 72 of O^T = sum_k P: lanes hi == 0, register (8 & 3) + 4 (8 >> 3) = 4 of row tile 2
    const unsigned lu = __float_as_uint(o[2][4]);
    auto sw2 = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    const float l_tot = __uint_as_float(sw2[0]);
    const float inv = p.v_scale[bkv * p.H + h] / l_tot;   // 1 / sum(P) and the e4m3 scale of V in one factor
    if (tail) {
      // part of a split tail unit: normalised partial O (f32) + log2-domain LSE -> workspace (attn_merge_kernel)
      if (qi[u] < p.Lq) {
        const int64_t slot = ((int64_t)tail_unit * p.tail_split + part) * 256 + (wave * (32 * NU) + u * 32 + l31);
        float* wo = p.ws_o + slot * HD;
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int d0 = d * 32 + qd * 8 + hi * 4;
          if (d0 < HD) {
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
      for (int d = 0; d < NDT; ++d) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int d0 = d * 32 + qd * 8 + hi * 4;
          if (d0 < HD) {
            uint2 w2;
            w2.x = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
            w2.y = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(orow + d0) = w2;
          }
        }
      }
      if (p.lse && hi == 0)
        p.lse[(int64_t)bh * p.Lq + qi[u]] = (m_ref[u] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
  }
}

int launch_one(const AttnParams& p, hipStream_t st) {
  auto kernel = attn_asm72p8_kernel<2>;
  OSK_ENSURE_MAX_SMEM(kernel, OSK72P8_SMEM);
  const int units = ((p.Lq + 255) / 256) * p.B * p.H;
  const int tail_units = p.tail_split > 1 ? units - p.tail_first : 0;
  dim3 grid(units + tail_units * (p.tail_split - 1)), block(64 * 4);
  hipLaunchKernelGGL(kernel, grid, block, OSK72P8_SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

int launch_asm72p8(const AttnParams& p, hipStream_t st) { return launch_one(p, st); }

}  // namespace osk_attn
