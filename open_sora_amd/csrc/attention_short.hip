// Short-sequence attention (Lq, Lk <= 64) with an optional ALiBi bias, gfx950 -- SURVEY.md section 8(f) rank 4 (round 4).
//
// BASELINE.json's north_star names the STDiT-generation block: "temporal self-attn over T" (B * H * W independent sequences of
// T <= 64 frames) with "RoPE/ALiBi".  The mounted v2.0 reference has neither call site (every flash-attn call passes
// alibi_slopes=None), so this kernel is PARITY-UNPINNED: semantics = flash-attn's documented `alibi_slopes`
// (a bias of -slope[h] * |i + Lk - Lq - j| added to the scaled score of query i and key j), checked against an fp64 softmax.
//
// The flash kernels of this library tile 256 or 512 query rows per workgroup: a 16-frame sequence would use 6 % of a tile.  This
// kernel is the other regime: one WAVE per (sequence, head), everything in registers, HBM-bound (q, k, v read once, o written
// once: 4 * L * hd * 2 bytes per unit).
//   * S^T = K . Q^T on v_mfma_f32_16x16x32_bf16 (A = 16 keys x 32 dims, B = 32 dims x 16 queries): both operands are 16-byte row
//     chunks straight from global memory -- no LDS; head_dim 72 pads the contraction to 96 with zero chunks.
//   * A lane of the 16 x 16 result owns ONE query (lane % 16) and keys 4 (lane / 16) .. + 3 of the tile: softmax statistics are two
//     xor-shuffles (16, 32) away, and two key tiles' accumulators ARE the B operand of the P.V product (8 keys per lane; which 8 is
//     a fixed permutation applied to V^T's read addresses) -- the register-only hand-over DESIGN.md section 7 describes for a
//     16 x 16 x 32 QK^T.
//   * O^T = V^T . P^T: V^T comes from a per-wave LDS transpose of the sequence's V (2-byte scatter: 4.6 k elements per unit).
#include "../../include/osk.h"
#include "osk_common.h"

namespace {

struct ShortParams {
  const unsigned short* q; int64_t qbs, qrs;
  const unsigned short* k; int64_t kbs, krs;
  const unsigned short* v; int64_t vbs, vrs;
  unsigned short* out; int64_t obs, ors;
  const float* slopes;   // [H] or nullptr
  int B, H, Lq, Lk;
  float sc;              // softmax scale * log2(e)
};

constexpr int VT_STRIDE = 72;   // bf16 elements per V^T row in LDS (64 keys + 8: rows 144 bytes apart, conflict-free 8-byte reads)

template <int HD, int NW>   // NW waves per workgroup (each wave owns an LDS slice for its V^T: 4 x 11.5 KB at head_dim 72, 2 x 18 KB at 128)
__global__ void __launch_bounds__(64 * NW) attn_short_kernel(const ShortParams p) {
  constexpr int NKS = (HD + 31) / 32;          // QK^T k-steps of 32 dims
  constexpr int NDB = (HD + 15) / 16;          // O^T row blocks of 16 dims
  constexpr int CPR = HD / 8;                  // 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) unsigned short vt_all[NW][NDB * 16 * VT_STRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  unsigned short* vt = vt_all[wave];
  for (int i = lane; i < NDB * 16 * VT_STRIDE / 8; i += 64) reinterpret_cast<uint4*>(vt)[i] = make_uint4(0, 0, 0, 0);
  const int nkt = (p.Lk + 15) >> 4, nqb = (p.Lq + 15) >> 4;   // 16-key tiles (1..4), 16-query blocks (1..4)
  const int nkp = (nkt + 1) >> 1;                              // 32-key pairs of tiles for the P.V product
  const int units = p.B * p.H;
  for (int unit = blockIdx.x * NW + wave; unit < units; unit += gridDim.x * NW) {
    const int b = unit / p.H, h = unit - b * p.H;
    const unsigned short* qg = p.q + b * p.qbs + h * HD;
    const unsigned short* kg = p.k + b * p.kbs + h * HD;
    const unsigned short* vg = p.v + b * p.vbs + h * HD;
    const float slope = p.slopes ? p.slopes[h] * 1.4426950408889634f : 0.f;   // log2 units like the scores
    // ---- V -> V^T in this wave's LDS slice (keys >= Lk and dims >= HD stay zero from the initial fill)
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < p.Lk * CPR; i += 64) {
      const int key = i / CPR, c = i - key * CPR;
      const uint4 u = *reinterpret_cast<const uint4*>(vg + (int64_t)key * p.vrs + c * 8);
      unsigned short* col = vt + (c * 8) * VT_STRIDE + key;
      col[0 * VT_STRIDE] = (unsigned short)(u.x & 0xFFFF); col[1 * VT_STRIDE] = (unsigned short)(u.x >> 16);
      col[2 * VT_STRIDE] = (unsigned short)(u.y & 0xFFFF); col[3 * VT_STRIDE] = (unsigned short)(u.y >> 16);
      col[4 * VT_STRIDE] = (unsigned short)(u.z & 0xFFFF); col[5 * VT_STRIDE] = (unsigned short)(u.z >> 16);
      col[6 * VT_STRIDE] = (unsigned short)(u.w & 0xFFFF); col[7 * VT_STRIDE] = (unsigned short)(u.w >> 16);
    }
    // ---- K fragments of every key tile: lane = (key l15 of the tile, dims 32 ks + 8 g .. + 7)
    bf16x8_t kf[4][NKS];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      int key = kt * 16 + l15;
      key = key < p.Lk ? key : p.Lk - 1;                       // (rows past the sequence: duplicates, masked below)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int c = 4 * ks + g;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (c < CPR && kt < nkt) u = *reinterpret_cast<const uint4*>(kg + (int64_t)key * p.krs + c * 8);
        kf[kt][ks] = __builtin_bit_cast(bf16x8_t, u);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int qb = 0; qb < nqb; ++qb) {
      int qi = qb * 16 + l15;
      const bool qok = qi < p.Lq;
      qi = qok ? qi : p.Lq - 1;
      bf16x8_t qf[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int c = 4 * ks + g;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (c < CPR) u = *reinterpret_cast<const uint4*>(qg + (int64_t)qi * p.qrs + c * 8);
        qf[ks] = __builtin_bit_cast(bf16x8_t, u);
      }
      // ---- scores (log2 units) of this lane's query against keys 16 kt + 4 g + i
      f32x4_t s[4];
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][ks], qf[ks], s[kt], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = kt * 16 + 4 * g + i;
          float t = s[kt][i] * p.sc - slope * fabsf((float)(qi + p.Lk - p.Lq - j));
          t = (j < p.Lk && kt < nkt) ? t : -INFINITY;
          s[kt][i] = t;
          m = fmaxf(m, t);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
      unsigned pk[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = __builtin_amdgcn_exp2f(s[kt][i] - m);
          sum += e[i];
        }
        pk[kt][0] = pack_bf16x2(e[0], e[1]);
        pk[kt][1] = pack_bf16x2(e[2], e[3]);
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      // ---- O^T block db (16 dims x 16 queries) = sum over key pairs: A = V^T rows (dim 16 db + l15; keys 32 kp + 4 g .. + 3 and
      //      32 kp + 16 + 4 g .. + 3: the order the two score tiles hold them in), B = the packed P of tiles 2 kp, 2 kp + 1
      unsigned short* orow = p.out + b * p.obs + (int64_t)qi * p.ors + h * HD;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        f32x4_t o = {0.f, 0.f, 0.f, 0.f};
        const unsigned short* vrow = vt + (db * 16 + l15) * VT_STRIDE + 4 * g;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          if (kp < nkp) {
            const uint2 a0 = *reinterpret_cast<const uint2*>(vrow + 32 * kp), a1 = *reinterpret_cast<const uint2*>(vrow + 32 * kp + 16);
            const uint4 au = make_uint4(a0.x, a0.y, a1.x, a1.y), bu = make_uint4(pk[2 * kp][0], pk[2 * kp][1], pk[2 * kp + 1][0], pk[2 * kp + 1][1]);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, au), __builtin_bit_cast(bf16x8_t, bu), o, 0, 0, 0);
          }
        }
        const int d0 = db * 16 + 4 * g;                           // this lane: dims d0 .. d0 + 3 of query qi
        if (qok && d0 < HD) {
          uint2 w;
          w.x = pack_bf16x2(o[0] * inv, o[1] * inv);
          w.y = pack_bf16x2(o[2] * inv, o[3] * inv);
          *reinterpret_cast<uint2*>(orow + d0) = w;
        }
      }
    }
  }
}

}  // namespace

extern "C" int osk_attention_short_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k,
                                        int64_t k_batch_stride, int64_t k_row_stride, const void* v, int64_t v_batch_stride,
                                        int64_t v_row_stride, void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                        const float* alibi_slopes, int B, int H, int Lq, int Lk, int hd, float scale, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return OSK_EINVAL;
  if (Lq > 64 || Lk > 64) return OSK_EUNSUPPORTED;      // longer sequences: osk_attention_fwd_bf16
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_batch_stride & 7) || (k_row_stride & 7) || (v_batch_stride & 7) ||
      (v_row_stride & 7) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 7)) return OSK_EINVAL;
  ShortParams p{(const unsigned short*)q, q_batch_stride, q_row_stride, (const unsigned short*)k, k_batch_stride, k_row_stride,
                (const unsigned short*)v, v_batch_stride, v_row_stride, (unsigned short*)out, o_batch_stride, o_row_stride,
                alibi_slopes, B, H, Lq, Lk, scale * 1.4426950408889634f};
  const int64_t units = (int64_t)B * H;
  auto blocks = [&](int nw) { return (int)((units + nw - 1) / nw < 8192 ? (units + nw - 1) / nw : 8192); };
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 64: hipLaunchKernelGGL((attn_short_kernel<64, 4>), dim3(blocks(4)), dim3(256), 0, st, p); break;
    case 72: hipLaunchKernelGGL((attn_short_kernel<72, 4>), dim3(blocks(4)), dim3(256), 0, st, p); break;
    case 128: hipLaunchKernelGGL((attn_short_kernel<128, 2>), dim3(blocks(2)), dim3(128), 0, st, p); break;
    default: return OSK_EUNSUPPORTED;
  }
  return (int)hipGetLastError();
}
