// CausalConv3d for gfx950, large tile + hand-scheduled K loop (Cin % 128 == 0, Cout >= 128, >= 256 output voxels).
//
// Same implicit GEMM as conv3d.hip (NDHWC activations, K = tap-major / channel-minor, replicate / causal padding as a
// CLAMP and the decoder's nearest upsample as a SHIFT of the gathered coordinate, weights pre-laid as [Cout][27 Cin],
// swapped-operand v_mfma_f32_32x32x16_bf16 so a lane owns one output voxel) on the tile and pipeline of gemm256.hip:
// 256 voxels x 256 (or 128) output channels x 64, 8 waves, LDS-DMA double buffer, accumulators in AGPRs.
// The K axis is walked one FILTER TAP at a time: within a tap the A rows are fixed gathered voxels and the K steps only
// advance the channel block, i.e. exactly a GEMM K loop with per-lane row offsets.  Each tap is one call of the asm
// segment emitted by tools/gen_gemm_asm.py (conv256_segment_n*.inc); the accumulators persist in the AGPRs between the
// calls, the last K step of a segment already fetches the next tap's first step (its offsets are passed in), and the
// per-tap voxel offsets (3 clamps per row slot) are ordinary compiler code between the calls.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2 * Cin * Cout * k^3 * B*To*Ho*Wo.
#include "conv_params.h"
#include "gemm256_regs_n256.inc"
#include "gemm256_regs_n128.inc"
#include "gemm256w_regs.inc"
#include "gemm256x_regs.inc"
#include <type_traits>

namespace osk_conv {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

#define OSKC_OUT16                                                                                               \
  "=v"(v16[0]), "=v"(v16[1]), "=v"(v16[2]), "=v"(v16[3]), "=v"(v16[4]), "=v"(v16[5]), "=v"(v16[6]), "=v"(v16[7]),    \
      "=v"(v16[8]), "=v"(v16[9]), "=v"(v16[10]), "=v"(v16[11]), "=v"(v16[12]), "=v"(v16[13]), "=v"(v16[14]),          \
      "=v"(v16[15])

// accumulator layouts: TM x TN MFMA tiles per wave, tile T = tn * TM + tm in AGPRs [16 T, 16 T + 16)
struct Lay256 {   // 8 waves, 256 x 256 tile, wave tile 128 x 64
  static constexpr int TM = OSKG256_TM, TN = OSKG256_TN;
  template <int T>
  OSK_DEV void read(float* v16) {
    if constexpr (T == 0) asm volatile(OSKG256_AR0 : OSKC_OUT16);
    else if constexpr (T == 1) asm volatile(OSKG256_AR1 : OSKC_OUT16);
    else if constexpr (T == 2) asm volatile(OSKG256_AR2 : OSKC_OUT16);
    else if constexpr (T == 3) asm volatile(OSKG256_AR3 : OSKC_OUT16);
    else if constexpr (T == 4) asm volatile(OSKG256_AR4 : OSKC_OUT16);
    else if constexpr (T == 5) asm volatile(OSKG256_AR5 : OSKC_OUT16);
    else if constexpr (T == 6) asm volatile(OSKG256_AR6 : OSKC_OUT16);
    else asm volatile(OSKG256_AR7 : OSKC_OUT16);
  }
};
struct Lay128 {   // 8 waves, 256 x 128 tile, wave tile 64 x 64
  static constexpr int TM = OSKG128_TM, TN = OSKG128_TN;
  template <int T>
  OSK_DEV void read(float* v16) {
    if constexpr (T == 0) asm volatile(OSKG128_AR0 : OSKC_OUT16);
    else if constexpr (T == 1) asm volatile(OSKG128_AR1 : OSKC_OUT16);
    else if constexpr (T == 2) asm volatile(OSKG128_AR2 : OSKC_OUT16);
    else asm volatile(OSKG128_AR3 : OSKC_OUT16);
  }
};
struct LayW {     // 4 waves, 256 x 256 tile, wave tile 128 x 128
  static constexpr int TM = OSKW_TM, TN = OSKW_TN;
  template <int T>
  OSK_DEV void read(float* v16) {
    if constexpr (T == 0) asm volatile(OSKW_AR0 : OSKC_OUT16);
    else if constexpr (T == 1) asm volatile(OSKW_AR1 : OSKC_OUT16);
    else if constexpr (T == 2) asm volatile(OSKW_AR2 : OSKC_OUT16);
    else if constexpr (T == 3) asm volatile(OSKW_AR3 : OSKC_OUT16);
    else if constexpr (T == 4) asm volatile(OSKW_AR4 : OSKC_OUT16);
    else if constexpr (T == 5) asm volatile(OSKW_AR5 : OSKC_OUT16);
    else if constexpr (T == 6) asm volatile(OSKW_AR6 : OSKC_OUT16);
    else if constexpr (T == 7) asm volatile(OSKW_AR7 : OSKC_OUT16);
    else if constexpr (T == 8) asm volatile(OSKW_AR8 : OSKC_OUT16);
    else if constexpr (T == 9) asm volatile(OSKW_AR9 : OSKC_OUT16);
    else if constexpr (T == 10) asm volatile(OSKW_AR10 : OSKC_OUT16);
    else if constexpr (T == 11) asm volatile(OSKW_AR11 : OSKC_OUT16);
    else if constexpr (T == 12) asm volatile(OSKW_AR12 : OSKC_OUT16);
    else if constexpr (T == 13) asm volatile(OSKW_AR13 : OSKC_OUT16);
    else if constexpr (T == 14) asm volatile(OSKW_AR14 : OSKC_OUT16);
    else asm volatile(OSKW_AR15 : OSKC_OUT16);
  }
};
template <int BN>
using LayOf = std::conditional_t<BN == 256, Lay256, Lay128>;

// Tile row -> output voxel (linear index over [B, To, Ho, Wo]).
//   linear (brick = 0): row r of M-tile bm is voxel 256 bm + r: a tile is a run of 256 voxels along W.
//   brick  (brick = 1, Ho % 16 == 0 and Wo % 16 == 0): M-tile bm = (spatial brick, frame) with the FRAME index fastest;
//     a tile is the 16 x 16 spatial brick of ONE frame.  The tiles an XCD runs at a time (consecutive list positions)
//     are then the successive frames of one spatial brick: the three time taps of a causal 3 x 3 x 3 filter re-read
//     frames that a neighbouring tile is reading right now (private-L2 hits) instead of planes fetched a whole frame of
//     tiles ago -- with linear tiles every input plane came over the fabric three times (round 1: 3.75 x the
//     algorithmic bytes); the in-plane halo of a 16 x 16 brick is 27 % instead of 200 % of a one-row tile's.
OSK_DEV int tile_row_to_voxel(const ConvParams& p, int bm, int r) {
  if (!p.brick) return bm * 256 + r;
  const int t = bm % p.To, sb = bm / p.To;
  const int nwb = p.Wo >> 4, nhb = p.Ho >> 4;
  const int wb = sb % nwb, q = sb / nwb;
  const int hb = q % nhb, b = q / nhb;
  return ((b * p.To + t) * p.Ho + hb * 16 + (r >> 4)) * p.Wo + wb * 16 + (r & 15);
}

// LDS offset table [tap][256 tile rows]: byte offset of the voxel that filter tap reads for that output row
// (thread -> row tid % 256, taps tid / 256, + tap_stride, ...)
OSK_DEV void build_tap_table(const ConvParams& p, int bm, int tid, int tap_stride, unsigned* table) {
  const int r = tid & 255;
  int m = tile_row_to_voxel(p, bm, r);
  m = m < p.M ? m : p.M - 1;
  const int wo = m % p.Wo;
  int q = m / p.Wo;
  const int ho = q % p.Ho;
  q /= p.Ho;
  const int to = q % p.To;
  const int b = q / p.To;
  const int HW = p.H * p.W;
  const unsigned cin_bytes = (unsigned)p.Cin * 2;
  for (int tap = tid >> 8; tap < p.ntaps; tap += tap_stride) {
    int dt = 0, dh = 0, dw = 0;
    if (p.ks == 3) {
      dt = tap / 9;
      const int r9 = tap - dt * 9;
      dh = r9 / 3;
      dw = r9 - dh * 3;
    }
    // clamp = replicate / causal padding, shift = nearest upsample (frame 0 is spatial-only)
    int tu = to * p.st + dt - (p.ks - 1);
    tu = tu < 0 ? 0 : (tu > p.Tu - 1 ? p.Tu - 1 : tu);
    const int ts = p.up_t ? (tu == 0 ? 0 : 1 + ((tu - 1) >> 1)) : tu;
    int hu = ho * p.sh + dh - (p.ks >> 1);
    hu = hu < 0 ? 0 : (hu > p.Hu - 1 ? p.Hu - 1 : hu);
    const int hs = p.up_hw ? (hu >> 1) : hu;
    int wu = wo * p.sw + dw - (p.ks >> 1);
    wu = wu < 0 ? 0 : (wu > p.Wu - 1 ? p.Wu - 1 : wu);
    const int ws = p.up_hw ? (wu >> 1) : wu;
    table[tap * 256 + r] = (unsigned)((b * p.T + ts) * HW + hs * p.W + ws) * cin_bytes;
  }
}

// bias + residual + bf16 store of one 32 x 32 accumulator tile T = tn * TM + tm (lane: voxel m, 4 channels per quad);
// r0w = first tile row of this wave
// fused GroupNorm statistics: a lane's partial (sum, sum of squares) of its voxel rows for the four 4-channel quads
// (channels nstrip + 8 qd + 4 hi + j) of the 32-column strip being walked, accumulated over the strip's TM row tiles
struct GnAcc {
  float s[4], q[4];
};

typedef __bf16 gn_bf16x2_t __attribute__((ext_vector_type(2)));

// two packed bf16 pairs = 4 rounded outputs: v_dot2c_f32_bf16 accumulates a pair's sum (against {1, 1}) or sum of squares
// in one instruction (bf16 x bf16 products are exact in f32)
OSK_DEV void gn_add(GnAcc& a, int qd, unsigned lo_hi0, unsigned lo_hi1) {
  const gn_bf16x2_t p0 = __builtin_bit_cast(gn_bf16x2_t, lo_hi0), p1 = __builtin_bit_cast(gn_bf16x2_t, lo_hi1);
  const gn_bf16x2_t one = __builtin_bit_cast(gn_bf16x2_t, 0x3f803f80u);
  a.s[qd] = __builtin_amdgcn_fdot2_f32_bf16(p0, one, a.s[qd], false);
  a.s[qd] = __builtin_amdgcn_fdot2_f32_bf16(p1, one, a.s[qd], false);
  a.q[qd] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, a.q[qd], false);
  a.q[qd] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, a.q[qd], false);
}

// sum over the 32 lanes of a half-wave, in every lane: four DPP adds inside a row of 16 (quad swaps, half mirror, mirror:
// one VALU instruction each, no LDS crossbar) and one cross-row exchange
OSK_DEV float half_wave_sum(float v) {
#define OSKC_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  OSKC_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  OSKC_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  OSKC_DPP_ADD(0x141);   // row_half_mirror
  OSKC_DPP_ADD(0x140);   // row_mirror
#undef OSKC_DPP_ADD
  return v + __shfl_xor(v, 16, 64);
}

// end of a strip: reduce over the 32 voxel lanes of each half-wave, then LDS float atomics into the tile's per-group slots
// ls[2 * (group - first group of the tile) + {0, 1}]
OSK_DEV void gn_flush(const ConvParams& p, GnAcc& a, int nstrip, int n0, int l31, int hi, float* ls) {
  const int cpg = p.Cout / p.gn_G;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const float s = half_wave_sum(a.s[qd]), q = half_wave_sum(a.q[qd]);
    if (l31 == 0) {
      const int n = nstrip + qd * 8 + hi * 4;
      if (n < p.Cout) {
        const int gl = n / cpg - n0 / cpg;
        atomicAdd(ls + 2 * gl, s);
        atomicAdd(ls + 2 * gl + 1, q);
      }
    }
    a.s[qd] = 0.f;
    a.q[qd] = 0.f;
  }
}

// bias + residual + bf16 store of one 32 x 32 accumulator tile T = tn * TM + tm (lane: voxel m, 4 channels per quad);
// r0w = first tile row of this wave; GN: accumulate the fused GroupNorm statistics (wave-uniform template switch)
template <class Lay, int T, bool GN>
OSK_DEV void epilogue_tile(const ConvParams& p, int bm, int r0w, int n0, int n0w, int l31, int hi, GnAcc& ga, float* ls) {
  constexpr int TM = Lay::TM;
  constexpr int tn = T / TM, tm = T % TM;
  float acc[16];
  Lay::template read<T>(acc);
  const int m = tile_row_to_voxel(p, bm, r0w + tm * 32 + l31);
  const bool valid = m < p.M;
  const int64_t roff = (int64_t)(valid ? m : 0) * p.Cout;
  const bool vec_ok = (p.Cout & 3) == 0;
  // whole 32-channel strip inside Cout and the row 16-byte aligned (Cout % 8 == 0): pair the half-waves and store 16 B
  // (v_permlane32_swap per dword: the lower half-wave takes the whole 8-channel block qd, the upper one block qd + 1)
  const int nstrip = n0w + tn * 32;
  if (vec_ok && (p.Cout & 7) == 0 && nstrip + 32 <= p.Cout && (((uintptr_t)p.out) & 15) == 0) {
    uint2 packed[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = nstrip + qd * 8 + hi * 4;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[qd * 4 + j];
      if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (p.res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
        v[0] += bf16_lo(rv.x); v[1] += bf16_hi(rv.x); v[2] += bf16_lo(rv.y); v[3] += bf16_hi(rv.y);
      }
      packed[qd].x = pack_bf16x2(v[0], v[1]);
      packed[qd].y = pack_bf16x2(v[2], v[3]);
      if constexpr (GN) {
        if (valid) gn_add(ga, qd, packed[qd].x, packed[qd].y);
      }
    }
#pragma unroll
    for (int qd = 0; qd < 4; qd += 2) {
      auto sx = __builtin_amdgcn_permlane32_swap(packed[qd].x, packed[qd + 1].x, false, false);
      auto sy = __builtin_amdgcn_permlane32_swap(packed[qd].y, packed[qd + 1].y, false, false);
      if (valid) *reinterpret_cast<uint4*>(p.out + roff + nstrip + (qd + hi) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
  } else {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = n0w + tn * 32 + qd * 8 + hi * 4;
      if (n >= p.Cout || !valid) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[qd * 4 + j];
      if (vec_ok && n + 3 < p.Cout) {
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (p.res) {
          const uint2 rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
          v[0] += bf16_lo(rv.x); v[1] += bf16_hi(rv.x); v[2] += bf16_lo(rv.y); v[3] += bf16_hi(rv.y);
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p.out + roff + n) = o;
        if constexpr (GN) gn_add(ga, qd, o.x, o.y);
      } else {
        for (int j = 0; j < 4 && n + j < p.Cout; ++j) {
          float t = v[j] + (p.bias ? p.bias[n + j] : 0.f);
          if (p.res) t += bf16_bits_to_f32(p.res[roff + n + j]);
          p.out[roff + n + j] = f32_to_bf16_bits(t);
        }
      }
    }
  }
  if constexpr (GN && tm == TM - 1) gn_flush(p, ga, nstrip, n0, l31, hi, ls);
}

template <class Lay, bool GN, int... Ts>
OSK_DEV void epilogue_tiles(const ConvParams& p, int bm, int r0w, int n0, int n0w, int l31, int hi, float* ls,
                            std::integer_sequence<int, Ts...>) {
  GnAcc ga;
#pragma unroll
  for (int i = 0; i < 4; ++i) ga.s[i] = ga.q[i] = 0.f;
  (epilogue_tile<Lay, Ts, GN>(p, bm, r0w, n0, n0w, l31, hi, ga, ls), ...);
}

// The whole workgroup calls this after its K loop (LDS is quiescent: every stage read and every LDS-DMA write was waited
// for before the loop's last barrier).  BN = channels per workgroup tile, n0 = its first channel.
// With p.gn_sums: the tile's per-group (sum, sum of squares) are collected in LDS floats (<= 256 voxels x 16 channels per
// slot), then ONE f64 atomic per (group, statistic) and tile goes to sums[b][g] -- the tile lies inside one batch item
// (conv256_gn_supported).
template <class Lay, int BN>
OSK_DEV void epilogue_all(const ConvParams& p, int bm, int r0w, int n0, int n0w, int l31, int hi, unsigned char* smem) {
  constexpr auto seq = std::make_integer_sequence<int, Lay::TM * Lay::TN>{};
  if (!p.gn_sums) {
    epilogue_tiles<Lay, false>(p, bm, r0w, n0, n0w, l31, hi, nullptr, seq);
    return;
  }
  float* ls = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  const int cpg = p.Cout / p.gn_G;
  int nch = p.Cout - n0;
  nch = nch < BN ? nch : BN;
  const int nslots = 2 * (nch / cpg);              // <= 2 * 256 / 4 = 128
  if (tid < nslots) ls[tid] = 0.f;
  __syncthreads();
  epilogue_tiles<Lay, true>(p, bm, r0w, n0, n0w, l31, hi, ls, seq);
  __syncthreads();
  if (tid < nslots) {
    const int m = tile_row_to_voxel(p, bm, 0);
    const int b = m / (p.To * p.Ho * p.Wo);
    atomicAdd(p.gn_sums + ((int64_t)b * p.gn_G + n0 / cpg) * 2 + tid, (double)ls[tid]);
  }
}

template <int BN>
__global__ void __launch_bounds__(512, 2) conv256_kernel(const ConvParams p) {
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int TN = BN == 256 ? OSKG256_TN : OSKG128_TN;
  constexpr int WN = BN / (TN * 32);
  constexpr int W_BASE = BN == 256 ? OSKG256_W_BASE : OSKG128_W_BASE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + 255) / 256, nbn = (p.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, nbm * nbn);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * 256, n0 = bn * BN;

  // ---- LDS-DMA row slots of this lane: instruction j = wave + 8 i covers tile rows [8 j, 8 j + 8)
  const int srow8 = lane >> 3, spos = lane & 7;
  unsigned woff[4], chunk16[4];
  int cb[4], cto[4], cho[4], cwo[4];   // batch, output coordinates of the slot's voxel
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 8 * i) * 8 + srow8;
    const int c = spos ^ ((r >> 1) & 7);
    chunk16[i] = (unsigned)(c * 16);
    int n = n0 + (r < BN ? r : 0);
    n = n < p.Cout ? n : p.Cout - 1;
    woff[i] = (unsigned)(((int64_t)n * p.wrs + c * 8) * 2);
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    cwo[i] = m % p.Wo;
    int q = m / p.Wo;
    cho[i] = q % p.Ho;
    q /= p.Ho;
    cto[i] = q % p.To;
    cb[i] = q / p.To;
  }
  const int HW = p.H * p.W;
  const unsigned cin_bytes = (unsigned)p.Cin * 2;
  // byte offset of the slot's gathered voxel for filter tap (dt, dh, dw): clamp = replicate / causal padding,
  // shift = nearest upsample (frame 0 is spatial-only)   [conv3d.hip / unet_causal_3d_blocks.py:82-96,136-150]
  auto tap_offset = [&](int i, int dt, int dh, int dw) -> unsigned {
    int tu = cto[i] * p.st + dt - (p.ks - 1);
    tu = tu < 0 ? 0 : (tu > p.Tu - 1 ? p.Tu - 1 : tu);
    const int ts = p.up_t ? (tu == 0 ? 0 : 1 + ((tu - 1) >> 1)) : tu;
    int hu = cho[i] * p.sh + dh - (p.ks >> 1);
    hu = hu < 0 ? 0 : (hu > p.Hu - 1 ? p.Hu - 1 : hu);
    const int hs = p.up_hw ? (hu >> 1) : hu;
    int wu = cwo[i] * p.sw + dw - (p.ks >> 1);
    wu = wu < 0 ? 0 : (wu > p.Wu - 1 ? p.Wu - 1 : wu);
    const int ws = p.up_hw ? (wu >> 1) : wu;
    const unsigned pos = (unsigned)((cb[i] * p.T + ts) * HW + hs * p.W + ws);
    return pos * cin_bytes + chunk16[i];
  };

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (l31 >> 1) & 7;
  unsigned faA[4], faW[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned sz = (unsigned)((((ks << 1) | hi) ^ sw) << 4);
    faA[ks] = lds_base + (wm * TM * 32 + l31) * 128 + sz;
    faW[ks] = lds_base + W_BASE + (wn * TN * 32 + l31) * 128 + sz;
  }
  const uint64_t xbase = rfl64((uint64_t)(uintptr_t)p.x);
  const unsigned nk = rfl((unsigned)(p.Cin / 64));   // K steps per tap (even: Cin % 128 == 0)
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + W_BASE + wave * 1024);

  unsigned aoffc[4], aoffn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoffc[i] = tap_offset(i, 0, 0, 0);
  for (int tap = 0; tap < p.ntaps; ++tap) {
    const int tn_ = tap + 1 < p.ntaps ? tap + 1 : tap;   // last segment: "next" = itself (harmless re-fetch)
    int dt = 0, dh = 0, dw = 0;
    if (p.ks == 3) {
      dt = tn_ / 9;
      const int r9 = tn_ - dt * 9;
      dh = r9 / 3;
      dw = r9 - dh * 3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) aoffn[i] = tap_offset(i, dt, dh, dw);
    const uint64_t wbase = rfl64((uint64_t)(uintptr_t)(p.w + (int64_t)tap * p.Cin));
    const unsigned flags = rfl((tap == 0 ? 1u : 0u) | (tap + 1 == p.ntaps ? 2u : 0u));
#define OSKC_OPERANDS                                                                                              \
  ::"v"(faA[0]), "v"(faA[1]), "v"(faA[2]), "v"(faA[3]), "v"(faW[0]), "v"(faW[1]), "v"(faW[2]), "v"(faW[3]),         \
      "v"(aoffc[0]), "v"(aoffc[1]), "v"(aoffc[2]), "v"(aoffc[3]), "v"(aoffn[0]), "v"(aoffn[1]), "v"(aoffn[2]),      \
      "v"(aoffn[3]), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "s"(xbase), "s"(wbase), "s"(nk),       \
      "s"(adst), "s"(wdst), "s"(flags)
    if constexpr (BN == 256) {
      asm volatile(
#include "conv256_segment_n256.inc"
          OSKC_OPERANDS : OSKG256_SEG_CLOBBERS);
    } else {
      asm volatile(
#include "conv256_segment_n128.inc"
          OSKC_OPERANDS : OSKG128_SEG_CLOBBERS);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) aoffc[i] = aoffn[i];
  }

  epilogue_all<LayOf<BN>, BN>(p, bm, wm * TM * 32, n0, n0 + wn * TN * 32, l31, hi, smem);   // (linear tiles: the launcher clears p.brick)
}

// ---------------------------------------------------------------------------------------------------------------
// One asm call for the whole K axis (conv256_body_n*.inc): the per-tap voxel offsets of the tile's 256 rows come from
// an LDS table [tap][row] built here, so the full gemm256 pipeline (LDS-DMA one step ahead, last k-sub-step issued
// across the barrier) runs uninterrupted over all 27 taps.
template <int BN>
__global__ void __launch_bounds__(512, 2) conv256t_kernel(const ConvParams p) {
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int TN = BN == 256 ? OSKG256_TN : OSKG128_TN;
  constexpr int WN = BN / (TN * 32);
  constexpr int W_BASE = BN == 256 ? OSKG256_W_BASE : OSKG128_W_BASE;
  constexpr int TABLE = BN == 256 ? OSKG256_SMEM : OSKG128_SMEM;   // the table sits behind the two stages
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + 255) / 256, nbn = (p.Cout + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {   // persistent (see conv256w_kernel)
  const int tile = xcd_remap(it, ntiles);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int n0 = bn * BN;

  build_tap_table(p, bm, tid, 2, reinterpret_cast<unsigned*>(smem + TABLE));
  __syncthreads();

  // ---- LDS-DMA row slots of this lane: instruction j = wave + 8 i covers tile rows [8 j, 8 j + 8)
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int srow8 = lane >> 3, spos = lane & 7;
  unsigned woff[4], chk[4], arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 8 * i) * 8 + srow8;
    const int c = spos ^ ((r >> 1) & 7);
    chk[i] = (unsigned)(c * 16);
    arow[i] = lds_base + TABLE + r * 4;
    int n = n0 + (r < BN ? r : 0);
    n = n < p.Cout ? n : p.Cout - 1;
    woff[i] = (unsigned)(((int64_t)n * p.wrs + c * 8) * 2);
  }
  const int sw = (l31 >> 1) & 7;
  unsigned faA[4], faW[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned sz = (unsigned)((((ks << 1) | hi) ^ sw) << 4);
    faA[ks] = lds_base + (wm * TM * 32 + l31) * 128 + sz;
    faW[ks] = lds_base + W_BASE + (wn * TN * 32 + l31) * 128 + sz;
  }
  const uint64_t xbase = rfl64((uint64_t)(uintptr_t)p.x), wbase = rfl64((uint64_t)(uintptr_t)p.w);
  const unsigned nkt = rfl((unsigned)(p.Cin / 64)), nk = rfl((unsigned)(p.ntaps * (p.Cin / 64)));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + W_BASE + wave * 1024);
#define OSKCT_OPERANDS                                                                                             \
  ::"v"(faA[0]), "v"(faA[1]), "v"(faA[2]), "v"(faA[3]), "v"(faW[0]), "v"(faW[1]), "v"(faW[2]), "v"(faW[3]),         \
      "v"(arow[0]), "v"(arow[1]), "v"(arow[2]), "v"(arow[3]), "v"(chk[0]), "v"(chk[1]), "v"(chk[2]), "v"(chk[3]),   \
      "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "s"(xbase), "s"(wbase), "s"(nk), "s"(nkt), "s"(adst), \
      "s"(wdst)
  if constexpr (BN == 256) {
    asm volatile(
#include "conv256_body_n256.inc"
        OSKCT_OPERANDS : OSKG256_CONV_CLOBBERS);
  } else {
    asm volatile(
#include "conv256_body_n128.inc"
        OSKCT_OPERANDS : OSKG128_CONV_CLOBBERS);
  }
  epilogue_all<LayOf<BN>, BN>(p, bm, wm * TM * 32, n0, n0 + wn * TN * 32, l31, hi, smem);
  }   // tile loop
}

// ---------------------------------------------------------------------------------------------------------------
// 4-wave form of conv256t_kernel<256> (Cout >= 256): one wave per SIMD with the whole register file, wave tile 128 x 128
// (gemm256w.hip's layout: a third less LDS read traffic per flop), K loop conv256w_body.inc = the table-driven loop above
// with the LDS-DMA instructions one per two MFMA shadows (what bounded the GEMM: profiles/r02_gemm_experiments.md).
__global__ void __launch_bounds__(256, 1) conv256w_kernel(const ConvParams p) {
  constexpr int TM = OSKW_TM, TN = OSKW_TN, BN = 256;
  constexpr int TABLE = OSKW_SMEM;   // the table sits behind the two stages
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  // persistent: one workgroup per CU (155 KB of LDS) walks the tile list with stride gridDim.x -- a one-tile workgroup's
  // successor cannot be dispatched before it retires, so every tile paid a dispatch gap on top of its serial prologue
  const int nbm = (p.M + 255) / 256, nbn = (p.Cout + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
  const int tile = xcd_remap(it, ntiles);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int n0 = bn * BN;

  build_tap_table(p, bm, tid, 1, reinterpret_cast<unsigned*>(smem + TABLE));
  __syncthreads();

  // ---- LDS-DMA row slots of this lane: instruction j = wave + 4 i (i = 0..7) covers tile rows [8 j, 8 j + 8); the swizzle
  // key (r >> 1) & 7 of row r = 8 (wave + 4 i) + lane / 8 does not depend on i
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int srow8 = lane >> 3, spos = lane & 7;
  const int r0 = wave * 8 + srow8;
  const int c = spos ^ ((r0 >> 1) & 7);
  const unsigned chk = (unsigned)(c * 16);
  const unsigned arow0 = lds_base + TABLE + r0 * 4;
  unsigned woff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int n = n0 + r0 + 32 * i;
    n = n < p.Cout ? n : p.Cout - 1;
    woff[i] = (unsigned)(((int64_t)n * p.wrs + c * 8) * 2);
  }
  const unsigned sz0 = (unsigned)((hi ^ ((l31 >> 1) & 7)) << 4);
  const unsigned faA0 = lds_base + (wm * TM * 32 + l31) * 128 + sz0;
  const unsigned faW0 = lds_base + OSKW_W_BASE + (wn * TN * 32 + l31) * 128 + sz0;
  const uint64_t xbase = rfl64((uint64_t)(uintptr_t)p.x), wbase = rfl64((uint64_t)(uintptr_t)p.w);
  const unsigned nkt = rfl((unsigned)(p.Cin / 64)), nk = rfl((unsigned)(p.ntaps * (p.Cin / 64)));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + OSKW_W_BASE + wave * 1024);
  asm volatile(
#include "conv256w_body.inc"
      ::"v"(faA0), "v"(faW0), "v"(arow0), "v"(chk), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(woff[4]),
      "v"(woff[5]), "v"(woff[6]), "v"(woff[7]), "s"(xbase), "s"(wbase), "s"(nk), "s"(nkt), "s"(adst), "s"(wdst)
      : OSKW_CONV_CLOBBERS);
  epilogue_all<LayW, BN>(p, bm, wm * TM * 32, n0, n0 + wn * TN * 32, l31, hi, smem);
  }   // tile loop (the next tile's table build ends in a barrier: nobody refills stage 0 while the statistics slots are read)
}

// ---------------------------------------------------------------------------------------------------------------
// conv256w_kernel on v_mfma_f32_16x16x32_bf16 (gemm256x.hip's compute side: at the board's power cap the 16x16x32 stream is the
// cheaper one per flop, profiles/r02_gemm_experiments.md): 8 x 8 accumulator tiles of 16 x 16 per wave, the same LDS image read
// through a 16-row x 32-k lane map, K loop conv256x_body.inc (tools/gen_gemm_asm.py::gen_conv_x4).  Of every 16 x 16 tile
// (J = channel block, I = voxel block) a lane owns voxel row 16 I + l15 and channels 16 J + 4 q4 .. + 3.
#define OSKCX_OUT4 "=v"(v4[0]), "=v"(v4[1]), "=v"(v4[2]), "=v"(v4[3])
template <int T>
OSK_DEV void read_x(float* v4) {
#define OSKCX_CASE(t) else if constexpr (T == t) asm volatile(OSKX_AR##t : OSKCX_OUT4)
  if constexpr (T < 0) {}
  OSKCX_CASE(0); OSKCX_CASE(1); OSKCX_CASE(2); OSKCX_CASE(3); OSKCX_CASE(4); OSKCX_CASE(5); OSKCX_CASE(6); OSKCX_CASE(7);
  OSKCX_CASE(8); OSKCX_CASE(9); OSKCX_CASE(10); OSKCX_CASE(11); OSKCX_CASE(12); OSKCX_CASE(13); OSKCX_CASE(14); OSKCX_CASE(15);
  OSKCX_CASE(16); OSKCX_CASE(17); OSKCX_CASE(18); OSKCX_CASE(19); OSKCX_CASE(20); OSKCX_CASE(21); OSKCX_CASE(22); OSKCX_CASE(23);
  OSKCX_CASE(24); OSKCX_CASE(25); OSKCX_CASE(26); OSKCX_CASE(27); OSKCX_CASE(28); OSKCX_CASE(29); OSKCX_CASE(30); OSKCX_CASE(31);
  OSKCX_CASE(32); OSKCX_CASE(33); OSKCX_CASE(34); OSKCX_CASE(35); OSKCX_CASE(36); OSKCX_CASE(37); OSKCX_CASE(38); OSKCX_CASE(39);
  OSKCX_CASE(40); OSKCX_CASE(41); OSKCX_CASE(42); OSKCX_CASE(43); OSKCX_CASE(44); OSKCX_CASE(45); OSKCX_CASE(46); OSKCX_CASE(47);
  OSKCX_CASE(48); OSKCX_CASE(49); OSKCX_CASE(50); OSKCX_CASE(51); OSKCX_CASE(52); OSKCX_CASE(53); OSKCX_CASE(54); OSKCX_CASE(55);
  OSKCX_CASE(56); OSKCX_CASE(57); OSKCX_CASE(58); OSKCX_CASE(59); OSKCX_CASE(60); OSKCX_CASE(61); OSKCX_CASE(62); OSKCX_CASE(63);
#undef OSKCX_CASE
}

// sum over the 16 lanes of a lane row, in every lane (the DPP half of half_wave_sum)
OSK_DEV float row16_sum(float v) {
#define OSKC_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  OSKC_DPP_ADD(0xB1);
  OSKC_DPP_ADD(0x4E);
  OSKC_DPP_ADD(0x141);
  OSKC_DPP_ADD(0x140);
#undef OSKC_DPP_ADD
  return v;
}

// fast path of one pair of voxel blocks (I, I + 1) of channel block J: whole 16-channel block inside Cout, rows 16-byte
// addressable.  RES / GN are compile-time, bias arrives as a register quad: 64 tiles per wave make per-tile branches count.
template <bool RES, bool GN, int J, int I>
OSK_DEV void pair_x(const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid,
                    int n, int ncol, const float4& bq, float& gs, float& gq) {
  float a0[4], a1[4];
  uint2 r0 = make_uint2(0, 0), r1 = r0;
  if constexpr (RES) {
    r0 = *reinterpret_cast<const uint2*>(p.res + rowoff[I] + n);        // rows beyond M read row 0 (clamped offsets)
    r1 = *reinterpret_cast<const uint2*>(p.res + rowoff[I + 1] + n);
  }
  read_x<J * OSKX_NB + I>(a0);
  read_x<J * OSKX_NB + I + 1>(a1);
  a0[0] += bq.x; a0[1] += bq.y; a0[2] += bq.z; a0[3] += bq.w;
  a1[0] += bq.x; a1[1] += bq.y; a1[2] += bq.z; a1[3] += bq.w;
  if constexpr (RES) {
    a0[0] += bf16_lo(r0.x); a0[1] += bf16_hi(r0.x); a0[2] += bf16_lo(r0.y); a0[3] += bf16_hi(r0.y);
    a1[0] += bf16_lo(r1.x); a1[1] += bf16_hi(r1.x); a1[2] += bf16_lo(r1.y); a1[3] += bf16_hi(r1.y);
  }
  const unsigned x0 = pack_bf16x2(a0[0], a0[1]), y0 = pack_bf16x2(a0[2], a0[3]);
  const unsigned x1 = pack_bf16x2(a1[0], a1[1]), y1 = pack_bf16x2(a1[2], a1[3]);
  if constexpr (GN) {
    const gn_bf16x2_t one = __builtin_bit_cast(gn_bf16x2_t, 0x3f803f80u);
    if (valid[I]) {
      const gn_bf16x2_t u = __builtin_bit_cast(gn_bf16x2_t, x0), v = __builtin_bit_cast(gn_bf16x2_t, y0);
      gs = __builtin_amdgcn_fdot2_f32_bf16(u, one, gs, false); gs = __builtin_amdgcn_fdot2_f32_bf16(v, one, gs, false);
      gq = __builtin_amdgcn_fdot2_f32_bf16(u, u, gq, false);   gq = __builtin_amdgcn_fdot2_f32_bf16(v, v, gq, false);
    }
    if (valid[I + 1]) {
      const gn_bf16x2_t u = __builtin_bit_cast(gn_bf16x2_t, x1), v = __builtin_bit_cast(gn_bf16x2_t, y1);
      gs = __builtin_amdgcn_fdot2_f32_bf16(u, one, gs, false); gs = __builtin_amdgcn_fdot2_f32_bf16(v, one, gs, false);
      gq = __builtin_amdgcn_fdot2_f32_bf16(u, u, gq, false);   gq = __builtin_amdgcn_fdot2_f32_bf16(v, v, gq, false);
    }
  }
  // v_permlane16_swap(tile I, tile I + 1): even lane rows end up with tile I's 8 channels 8 (q4 / 2) .., odd rows with tile I + 1's
  auto sx = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
  auto sy = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
  if (svalid[I / 2]) *reinterpret_cast<uint4*>(p.out + storeoff[I / 2] + ncol) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

// all channel blocks J of one pair of voxel blocks, back to back: consecutive stores fill a voxel row's 32-byte pieces left to
// right (with the channel block outermost the pieces of one 64-byte sector left four stores apart: +30 % fabric-side writes).
// FULL: every channel block of the wave tile lies inside Cout (no per-block test)
template <bool RES, bool GN, bool FULL, int I, int... Js>
OSK_DEV void rowpair_x(const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid,
                       int n0w, int q4, const float4* bq, float* gs, float* gq, std::integer_sequence<int, Js...>) {
  ((FULL || n0w + Js * 16 < p.Cout
        ? pair_x<RES, GN, Js, I>(p, rowoff, valid, storeoff, svalid, n0w + Js * 16 + q4 * 4, n0w + Js * 16, bq[Js], gs[Js], gq[Js])
        : (void)0), ...);
}

template <bool RES, bool GN, bool FULL, int NBJ, int... Is>
OSK_DEV void tile_x(const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid, int n0,
                    int n0w, int l15, int q4, float* ls, std::integer_sequence<int, Is...>) {
  float4 bq[NBJ];
  float gs[NBJ], gq[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    gs[j] = gq[j] = 0.f;
    const int n = n0w + j * 16 + q4 * 4;
    if (p.bias && (FULL || n < p.Cout)) bq[j] = *reinterpret_cast<const float4*>(p.bias + n);
  }
  (rowpair_x<RES, GN, FULL, 2 * Is>(p, rowoff, valid, storeoff, svalid, n0w, q4, bq, gs, gq, std::make_integer_sequence<int, NBJ>{}), ...);
  if constexpr (GN) {
    const int cpg = p.Cout / p.gn_G;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int n = n0w + j * 16 + q4 * 4;
      const float s = row16_sum(gs[j]), q = row16_sum(gq[j]);
      if (l15 == 0 && (FULL || n < p.Cout)) {
        const int gl = n / cpg - n0 / cpg;
        atomicAdd(ls + 2 * gl, s);
        atomicAdd(ls + 2 * gl + 1, q);
      }
    }
  }
}

template <bool RES, bool GN, int NBJ>
OSK_DEV void cols_x(const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid, int n0,
                    int n0w, int l15, int q4, float* ls) {
  constexpr auto seq = std::make_integer_sequence<int, OSKX_NB / 2>{};
  if (n0w + NBJ * 16 <= p.Cout) tile_x<RES, GN, true, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls, seq);
  else tile_x<RES, GN, false, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls, seq);   // ragged last tile column
}

// generic path of one tile (any Cout, any alignment): per-element bounds checks, no statistics
template <int T>
OSK_DEV void tile_x_generic(const ConvParams& p, int bm, int r0w, int n0w, int l15, int q4) {
  constexpr int J = T / OSKX_NB, I = T % OSKX_NB;
  float acc[4];
  read_x<T>(acc);
  const int m = tile_row_to_voxel(p, bm, r0w + I * 16 + l15);
  if (m >= p.M) return;
  const int64_t roff = (int64_t)m * p.Cout;
  const int n = n0w + J * 16 + q4 * 4;
  for (int j = 0; j < 4 && n + j < p.Cout; ++j) {
    float t = acc[j] + (p.bias ? p.bias[n + j] : 0.f);
    if (p.res) t += bf16_bits_to_f32(p.res[roff + n + j]);
    p.out[roff + n + j] = f32_to_bf16_bits(t);
  }
}
template <int... Ts>
OSK_DEV void tiles_x_generic(const ConvParams& p, int bm, int r0w, int n0w, int l15, int q4, std::integer_sequence<int, Ts...>) {
  (tile_x_generic<Ts>(p, bm, r0w, n0w, l15, q4), ...);
}

// the whole workgroup calls this after its K loop (see epilogue_all): NBJ = 16-channel blocks per wave (8: 256 channels per
// workgroup tile from n0, 4: 128)
template <int NBJ>
OSK_DEV void epilogue_all_x(const ConvParams& p, int bm, int r0w, int n0, int n0w, int l15, int q4, unsigned char* smem) {
  constexpr int NB = OSKX_NB, BN = 32 * NBJ;
  const bool fast = (p.Cout & 15) == 0 && ((((uintptr_t)p.out) & 15) == 0) && (!p.res || (((uintptr_t)p.res) & 7) == 0) &&
                    (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
  float* ls = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  int nslots = 0;
  if (p.gn_sums) {   // conv256_gn_supported(): Cout % 32 == 0 -> the fast path
    const int cpg = p.Cout / p.gn_G;
    int nch = p.Cout - n0;
    nch = nch < BN ? nch : BN;
    nslots = 2 * (nch / cpg);
    if (tid < nslots) ls[tid] = 0.f;
    __syncthreads();
  }
  if (n0w < p.Cout) {                      // wave tiles entirely beyond Cout have nothing to store
    if (!fast) {
      tiles_x_generic(p, bm, r0w, n0w, l15, q4, std::make_integer_sequence<int, NB * NBJ>{});
    } else {
      // element offsets of this lane's NB voxel rows (rows beyond M: clamped to row 0 for loads, masked for stores and
      // statistics) and of the NB / 2 rows it STORES after the lane-row exchange (+ its 8-channel half)
      int64_t rowoff[NB], storeoff[NB / 2];
      bool valid[NB], svalid[NB / 2];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int m = tile_row_to_voxel(p, bm, r0w + i * 16 + l15);
        valid[i] = m < p.M;
        rowoff[i] = (int64_t)(valid[i] ? m : 0) * p.Cout;
      }
#pragma unroll
      for (int i = 0; i < NB / 2; ++i) {
        storeoff[i] = ((q4 & 1) ? rowoff[2 * i + 1] : rowoff[2 * i]) + (q4 >> 1) * 8;
        svalid[i] = (q4 & 1) ? valid[2 * i + 1] : valid[2 * i];
      }
      if (p.gn_sums) {
        if (p.res) cols_x<true, true, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
        else cols_x<false, true, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
      } else {
        if (p.res) cols_x<true, false, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
        else cols_x<false, false, NBJ>(p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
      }
    }
  }
  if (p.gn_sums) {
    __syncthreads();
    if (tid < nslots) {
      const int m = tile_row_to_voxel(p, bm, 0);
      const int b = m / (p.To * p.Ho * p.Wo);
      atomicAdd(p.gn_sums + ((int64_t)b * p.gn_G + n0 / (p.Cout / p.gn_G)) * 2 + tid, (double)ls[tid]);
    }
  }
}

template <int NBJ>
__global__ void __launch_bounds__(256, 1) conv256x_kernel(const ConvParams p) {
  constexpr int WT = OSKX_NB * 16, WTN = NBJ * 16, BN = 32 * NBJ;        // wave tile WT voxels x WTN channels, 2 x 2 waves
  constexpr int TABLE = NBJ == 8 ? OSKX_SMEM : OSKX128_SMEM;             // the table sits behind the two stages
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q4 = lane >> 4, l15 = lane & 15;

  const int nbm = (p.M + 255) / 256, nbn = (p.Cout + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {   // persistent (see conv256w_kernel)
  const int tile = xcd_remap(it, ntiles);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int n0 = bn * BN;

  build_tap_table(p, bm, tid, 1, reinterpret_cast<unsigned*>(smem + TABLE));
  __syncthreads();

  // ---- LDS-DMA side: exactly conv256w_kernel's (same LDS image)
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int srow8 = lane >> 3, spos = lane & 7;
  const int r0 = wave * 8 + srow8;
  const int c = spos ^ ((r0 >> 1) & 7);
  const unsigned chk = (unsigned)(c * 16);
  const unsigned arow0 = lds_base + TABLE + r0 * 4;
  unsigned woff[8];                        // NBJ LDS-DMA pieces per wave cover the BN weight rows (slots beyond NBJ: unused copies)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int n = n0 + r0 + 32 * (i & (NBJ - 1));
    n = n < p.Cout ? n : p.Cout - 1;
    woff[i] = (unsigned)(((int64_t)n * p.wrs + c * 8) * 2);
  }
  // ---- fragment side: row l15 of a 16-row block, 16-byte chunk q4 (k 8 q4 .. + 7 of the sub-step's 32) under the row's swizzle key
  const unsigned sz0 = (unsigned)((q4 ^ ((l15 >> 1) & 7)) << 4);
  const unsigned faA0 = lds_base + (wm * WT + l15) * 128 + sz0;
  const unsigned faW0 = lds_base + OSKX_W_BASE + (wn * WTN + l15) * 128 + sz0;
  const uint64_t xbase = rfl64((uint64_t)(uintptr_t)p.x), wbase = rfl64((uint64_t)(uintptr_t)p.w);
  const unsigned nkt = rfl((unsigned)(p.Cin / 64)), nk = rfl((unsigned)(p.ntaps * (p.Cin / 64)));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + OSKX_W_BASE + wave * 1024);
#define OSKCX_OPERANDS                                                                                               \
  ::"v"(faA0), "v"(faW0), "v"(arow0), "v"(chk), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(woff[4]), \
      "v"(woff[5]), "v"(woff[6]), "v"(woff[7]), "s"(xbase), "s"(wbase), "s"(nk), "s"(nkt), "s"(adst), "s"(wdst)
  if constexpr (NBJ == 8) {
    asm volatile(
#include "conv256x_body.inc"
        OSKCX_OPERANDS : OSKX_CONV_CLOBBERS);
  } else {
    asm volatile(
#include "conv256x_body_n128.inc"
        OSKCX_OPERANDS : OSKX128_CONV_CLOBBERS);
  }
  epilogue_all_x<NBJ>(p, bm, wm * WT, n0, n0 + wn * WTN, l15, q4, smem);
  }   // tile loop
}

// grid of the persistent kernels: one workgroup per CU (a multiple of 8, so that the XCD remap of the tile list keeps a
// workgroup inside one XCD's range); OSK_CONV_PERSIST=0: one workgroup per tile (A/B runs)
int persistent_grid(int ntiles) {
  static const bool on = [] { const char* e = getenv("OSK_CONV_PERSIST"); return !e || atoi(e) != 0; }();
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    n -= n % 8;
    return n < 8 ? 8 : n;
  }();
  return on && ntiles > n_cu ? n_cu : ntiles;
}

template <int NBJ>
int launch_x(const ConvParams& p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int BN = 32 * NBJ, SMEM = (NBJ == 8 ? OSKX_SMEM : OSKX128_SMEM) + 27 * 1024;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv256x_kernel<NBJ>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nblk = ((p.M + 255) / 256) * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(conv256x_kernel<NBJ>, dim3(persistent_grid(nblk)), dim3(256), SMEM, st, p);
  return (int)hipGetLastError();
}

int launch_w(const ConvParams& p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int SMEM = OSKW_SMEM + 27 * 1024;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv256w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nblk = ((p.M + 255) / 256) * ((p.Cout + 255) / 256);
  hipLaunchKernelGGL(conv256w_kernel, dim3(persistent_grid(nblk)), dim3(256), SMEM, st, p);
  return (int)hipGetLastError();
}

template <int BN, bool TABLE_VERSION>
int launch_one(const ConvParams& p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int SMEM = (BN == 256 ? OSKG256_SMEM : OSKG128_SMEM) + (TABLE_VERSION ? 27 * 1024 : 0);
  auto kernel = TABLE_VERSION ? conv256t_kernel<BN> : conv256_kernel<BN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nblk = ((p.M + 255) / 256) * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kernel, dim3(TABLE_VERSION ? persistent_grid(nblk) : nblk), dim3(512), SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

// 32-bit per-lane byte offsets: both tensors must span < 4 GiB; whole K steps per tap in pairs: Cin % 128 == 0
bool conv256_supported(const ConvParams& p, int64_t x_bytes, int64_t w_bytes) {
  return p.Cin % 128 == 0 && p.Cout >= 128 && p.M >= 256 && x_bytes < (int64_t)0xFFFFFFFF && w_bytes < (int64_t)0xFFFFFFFF;
}

// fused GroupNorm statistics: whole groups inside a 32-column strip quad structure (4, 8 or 16 channels per group), every
// workgroup tile inside one batch item (a tile is 256 consecutive voxels)
bool conv256_gn_supported(const ConvParams& p) {
  if (p.gn_G <= 0 || p.Cout % p.gn_G || p.Cout % 32) return false;
  const int cpg = p.Cout / p.gn_G;
  if (cpg != 4 && cpg != 8 && cpg != 16) return false;
  const int64_t per_b = (int64_t)p.To * p.Ho * p.Wo;
  return p.B == 1 || per_b % 256 == 0;
}

// variant 1: one asm segment per filter tap (conv256_kernel); otherwise the single-call table version
int launch_conv256(const ConvParams& p0, int variant, hipStream_t st) {
  ConvParams p = p0;
  p.brick = 0;
  if (variant == 1) return p.Cout >= 256 ? launch_one<256, false>(p, st) : launch_one<128, false>(p, st);
  // OSK_CONV_BRICK=1: 16 x 16 spatial bricks in frame-fastest order instead of linear 256-voxel runs.  Measured (round 2,
  // profiles/r02_pmc_gemm_conv.txt): parity-green, same time (69.4 vs 69.0 ms per VAE encode+decode) and 12 % MORE fabric-side
  // reads (49.3 vs 43.8 GB) -- the short 16-voxel row segments cost more than the time-tap reuse saves -- so it stays off.
  static const bool brick = [] { const char* e = getenv("OSK_CONV_BRICK"); return e && atoi(e) != 0; }();
  p.brick = brick && (p.Ho % 16 == 0) && (p.Wo % 16 == 0) ? 1 : 0;
  // OSK_CONV_W4=0: the 8-wave kernel for Cout >= 256 too (A/B runs)
  static const bool w4 = [] { const char* e = getenv("OSK_CONV_W4"); return !e || atoi(e) != 0; }();
  // the 4-wave kernel on v_mfma_f32_16x16x32_bf16 (conv256x_kernel: VAE encode + decode 64.5 -> 62.2 ms); OSK_CONV_X=0 = the
  // 32x32x16 form (conv256w_kernel) for A/B runs
  static const bool x16 = [] { const char* e = getenv("OSK_CONV_X"); return !e || atoi(e) != 0; }();
  if (p.Cout >= 256 && w4 && x16) return launch_x<8>(p, st);
  // Cout < 256: the 4-wave 256 x 128 tile of the same kernel (VAE encode + decode 62.8 -> 61.6 ms); OSK_CONV_X128=0 = the 8-wave
  // conv256t_kernel<128> for A/B runs
  static const bool x128 = [] { const char* e = getenv("OSK_CONV_X128"); return !e || atoi(e) != 0; }();
  if (p.Cout < 256 && x128) return launch_x<4>(p, st);
  if (p.Cout >= 256) return w4 ? launch_w(p, st) : launch_one<256, true>(p, st);
  return launch_one<128, true>(p, st);
}

}  // namespace osk_conv
