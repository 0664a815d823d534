// Fused epilogue of the large-tile GEMM kernels (gemm256.hip [fp8], gemm256p.hip; the 16 x 16 accumulator layout of gemm256x.hip has its own: gemm_epilogue16.h): bias / GELU-tanh / gate * x + residual /
// bf16 or f32 store, on the accumulator layout of v_mfma_f32_32x32x16_bf16 with swapped operands (a lane owns ONE output
// row and 4 consecutive columns per 8-column block).  Geo supplies the wave tile: TM x TN MFMA tiles and
// read<T>(aq, float[16]) = the 16 accumulator registers of tile T = tn * TM + tm out of aq = the wave's accumulator quads as
// compiler-visible values (acc_quads.h: outputs of an empty asm statement behind the K loop; tile T = quads 4 T .. 4 T + 3).
#pragma once
#include "acc_quads.h"
#include "gemm_params.h"

namespace osk_gemm {
namespace epi {

enum { GELU_NONE = 0, GELU_ALL = 1, GELU_MIXED = 2 };

// Interior 32 x 32 accumulator tile T = tn * TM + tm of a wave whose whole tile lies inside C and inside one batch:
// no bounds checks.  FOLDED: the bias is already in the accumulator.  A lane owns row m0w + tm*32 + l31 and columns
// tn*32 + qd*8 + hi*4 + {0..3}, qd = 0..3.
template <class Geo, bool OUT_F32, int T>
OSK_DEV void tile_interior(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool folded, int gelu, const float4* bq,
                           const float4* gq) {
  constexpr int TM = Geo::TM;
  constexpr int tn = T / TM, tm = T % TM;
  const int m = m0w + tm * 32 + l31;
  const int b = m / p.crpb, l = m - b * p.crpb;
  const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
  uint2 rv[4];
  if (p.gate) {
    const unsigned short* rrow = p.res + roff + n0w + tn * 32;
#ifndef OSK_GEMM_NARROW_RES   // (A/B builds of tools/: the residual in 8-byte pieces)
    if ((((uintptr_t)rrow) & 15) == 0) {
      // round 5: the residual read the way the bf16 result is stored below -- 16 bytes per lane, the whole 8-column block qd + hi of
      // the lane's row -- and swapped back (v_permlane32_swap is an involution on its register pair) into the 8-byte pieces of blocks
      // qd and qd + 1 in the accumulator layout: 2 loads of 32 rows x 32 bytes per tile instead of 4 of 32 rows x 16 bytes (the
      // address path charges per (instruction, line), gemm_epilogue16.h).  In place (res == C): same bytes as this tile's stores,
      // all read before the first of them.
#pragma unroll
      for (int qd = 0; qd < 4; qd += 2) {
        const uint4 rc = *reinterpret_cast<const uint4*>(rrow + (qd + hi) * 8);
        auto ux = __builtin_amdgcn_permlane32_swap(rc.x, rc.z, false, false);
        auto uy = __builtin_amdgcn_permlane32_swap(rc.y, rc.w, false, false);
        rv[qd] = make_uint2(ux[0], uy[0]);
        rv[qd + 1] = make_uint2(ux[1], uy[1]);
      }
    } else
#endif
    {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) rv[qd] = *reinterpret_cast<const uint2*>(rrow + qd * 8 + hi * 4);
    }
  }
  float acc[16];
  Geo::template read<T>(aq, acc);
  if (!folded && p.bias) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      acc[qd * 4 + 0] += bq[qd].x; acc[qd * 4 + 1] += bq[qd].y; acc[qd * 4 + 2] += bq[qd].z; acc[qd * 4 + 3] += bq[qd].w;
    }
  }
  if (gelu == GELU_ALL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = gelu_tanh(acc[i]);
  } else if (gelu == GELU_MIXED) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0w + tn * 32 + (i >> 2) * 8 + hi * 4 + (i & 3);
      const float g = gelu_tanh(acc[i]);
      acc[i] = n >= p.gelu_from ? g : acc[i];
    }
  }
  if (p.gate) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      acc[qd * 4 + 0] = bf16_lo(rv[qd].x) + gq[qd].x * acc[qd * 4 + 0];
      acc[qd * 4 + 1] = bf16_hi(rv[qd].x) + gq[qd].y * acc[qd * 4 + 1];
      acc[qd * 4 + 2] = bf16_lo(rv[qd].y) + gq[qd].z * acc[qd * 4 + 2];
      acc[qd * 4 + 3] = bf16_hi(rv[qd].y) + gq[qd].w * acc[qd * 4 + 3];
    }
  }
  if constexpr (OUT_F32) {
    float* crow = reinterpret_cast<float*>(p.C) + roff + n0w + tn * 32 + hi * 4;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      *reinterpret_cast<float4*>(crow + qd * 8) = make_float4(acc[qd * 4], acc[qd * 4 + 1], acc[qd * 4 + 2], acc[qd * 4 + 3]);
  } else {
    uint2 packed[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      packed[qd].x = pack_bf16x2(acc[qd * 4 + 0], acc[qd * 4 + 1]);
      packed[qd].y = pack_bf16x2(acc[qd * 4 + 2], acc[qd * 4 + 3]);
    }
    // the partner lane (other half-wave, same row) holds the other 4 columns of every 8-column block: one
    // v_permlane32_swap per dword gives the lower half-wave the whole block qd and the upper one the whole block qd + 1
    unsigned short* crow = reinterpret_cast<unsigned short*>(p.C) + roff + n0w + tn * 32;
    const bool wide = (((uintptr_t)crow) & 15) == 0;
#pragma unroll
    for (int qd = 0; qd < 4; qd += 2) {
      if (wide) {
        auto sx = __builtin_amdgcn_permlane32_swap(packed[qd].x, packed[qd + 1].x, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(packed[qd].y, packed[qd + 1].y, false, false);
        *reinterpret_cast<uint4*>(crow + (qd + hi) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      } else {
        *reinterpret_cast<uint2*>(crow + qd * 8 + hi * 4) = packed[qd];
        *reinterpret_cast<uint2*>(crow + (qd + 1) * 8 + hi * 4) = packed[qd + 1];
      }
    }
  }
}

// edge tiles: per-element bounds checks (rows >= M were computed on clamped copies of row M-1 and are dropped)
template <class Geo, bool OUT_F32, int T>
OSK_DEV void tile_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool folded) {
  constexpr int TM = Geo::TM;
  constexpr int tn = T / TM, tm = T % TM;
  float acc[16];
  Geo::template read<T>(aq, acc);
  const int m = m0w + tm * 32 + l31;
  if (m >= p.M) return;
  const int b = m / p.crpb, l = m - b * p.crpb;
  const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
  const float* grow = p.gate ? p.gate + b * p.gbs : nullptr;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int n = n0w + tn * 32 + qd * 8 + hi * 4;
    for (int j = 0; j < 4 && n + j < p.N; ++j) {
      float t = acc[qd * 4 + j];
      if (!folded && p.bias) t += p.bias[n + j];
      if (n + j >= p.gelu_from) t = gelu_tanh(t);
      if (grow) t = bf16_bits_to_f32(p.res[roff + n + j]) + grow[n + j] * t;
      if constexpr (OUT_F32) reinterpret_cast<float*>(p.C)[roff + n + j] = t;
      else reinterpret_cast<unsigned short*>(p.C)[roff + n + j] = f32_to_bf16_bits(t);
    }
  }
}

template <class Geo, bool OUT_F32, int... Ts>
OSK_DEV void epilogue_tn(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool interior, bool folded,
                         std::integer_sequence<int, Ts...>) {
  constexpr int TM = Geo::TM;
  constexpr int tn = ((Ts, ...)) / TM;   // all Ts share tn
  if (interior) {
    const int nf = n0w + tn * 32;        // wave-uniform: GELU for none / all / some of this tile's 32 columns
    const int gelu = nf >= p.gelu_from ? GELU_ALL : (nf + 32 <= p.gelu_from ? GELU_NONE : GELU_MIXED);
    float4 bq[4], gq[4];
    const int b = m0w / p.crpb;          // an interior wave tile lies inside one batch
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = nf + qd * 8 + hi * 4;
      if (!folded && p.bias) bq[qd] = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.gate) gq[qd] = *reinterpret_cast<const float4*>(p.gate + b * p.gbs + n);
    }
    (tile_interior<Geo, OUT_F32, Ts>(aq, p, m0w, n0w, l31, hi, folded, gelu, bq, gq), ...);
  } else {
    (tile_edge<Geo, OUT_F32, Ts>(aq, p, m0w, n0w, l31, hi, folded), ...);
  }
}

template <class Geo, bool OUT_F32, int TN_, int... Is>
OSK_DEV void epilogue_rows(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool interior, bool folded,
                           std::integer_sequence<int, Is...>) {
  // Is = 0 .. TM-1: the tiles of column block TN_
  epilogue_tn<Geo, OUT_F32>(aq, p, m0w, n0w, l31, hi, interior, folded, std::integer_sequence<int, (TN_ * Geo::TM + Is)...>{});
}

template <class Geo, bool OUT_F32, int... TNs>
OSK_DEV void epilogue_cols(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool interior, bool folded,
                           std::integer_sequence<int, TNs...>) {
  (epilogue_rows<Geo, OUT_F32, TNs>(aq, p, m0w, n0w, l31, hi, interior, folded, std::make_integer_sequence<int, Geo::TM>{}), ...);
}

// the whole wave tile: column block by column block (column vectors -- bias, gate -- are loaded once per block)
template <class Geo, bool OUT_F32>
OSK_DEV void epilogue_all(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, bool interior, bool folded) {
  if (m0w >= p.M || n0w >= p.N) return;   // the whole wave tile lies outside C (ragged last tile row / column): wave-uniform
  epilogue_cols<Geo, OUT_F32>(aq, p, m0w, n0w, l31, hi, interior, folded, std::make_integer_sequence<int, Geo::TN>{});
}

}  // namespace epi
}  // namespace osk_gemm
