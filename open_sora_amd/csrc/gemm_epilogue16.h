// Fused epilogue of gemm256x.hip: gemm_epilogue.h's bias / GELU-tanh / gate * x + residual / bf16 or f32 store on the
// accumulator layout of v_mfma_f32_16x16x32_bf16 with swapped operands: of every 16 x 16 tile (J = column block, I = row
// block) a lane owns output row 16 I + l15 (l15 = lane % 16) and the 4 consecutive columns 16 J + 4 q4 .. + 3 (q4 = lane / 16).
// Geo supplies NB (16-blocks per wave tile side) and read<T>(acc, float[4]) = the 4 accumulator registers of tile T = J * NB + I out of
// acc = the wave's accumulator quads as compiler-visible values (acc_quads.h: outputs of an empty asm statement behind the K loop).
#pragma once
#include "gemm_epilogue.h"

namespace osk_gemm {
namespace epi16 {

using epi::GELU_ALL;
using epi::GELU_MIXED;
using epi::GELU_NONE;

// one interior tile up to (not including) the store: acc[4] -> final values.  Interior tiles never add the bias here: the
// accumulators started from it (an interior wave tile has all its columns inside N, which is the kernel's "folded" condition).
// GATE and the tile's GELU class are compile-time / hoisted: 64 tiles per wave make every per-tile branch count.
template <class Geo, int T, bool GATE, int GELU>
OSK_DEV void tile_values_rv(const osk_v4f* aq, const GemmParams& p, int n, const float4& gq, const uint2& rv, float* acc) {
  Geo::template read<T>(aq, acc);
  if constexpr (GELU == GELU_ALL) {
    // (round 6 tried the five full-rate operations of gelu_tanh on hand-written packed-FP32 pairs, 18 instructions per quad instead
    //  of 28: no measurable gain once the wait states a transcendental's result needs before a packed read were in -- withdrawn,
    //  profiles/r06g_gelu_packed_ab.jsonl)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = gelu_tanh(acc[i]);
  } else if constexpr (GELU == GELU_MIXED) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = gelu_tanh(acc[i]);
      acc[i] = n + i >= p.gelu_from ? g : acc[i];
    }
  }
  if constexpr (GATE) {
    acc[0] = bf16_lo(rv.x) + gq.x * acc[0];
    acc[1] = bf16_hi(rv.x) + gq.y * acc[1];
    acc[2] = bf16_lo(rv.y) + gq.z * acc[2];
    acc[3] = bf16_hi(rv.y) + gq.w * acc[3];
  }
}
// (the residual piece of the tile read in the accumulator layout: 8 bytes of the lane's row -- 16 rows x 32 bytes per instruction)
template <class Geo, int T, bool GATE, int GELU>
OSK_DEV void tile_values(const osk_v4f* aq, const GemmParams& p, int64_t roff, int n, const float4& gq, float* acc) {
  uint2 rv = make_uint2(0, 0);
  if constexpr (GATE) rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
  tile_values_rv<Geo, T, GATE, GELU>(aq, p, n, gq, rv, acc);
}

// Interior: the pair of row blocks (I, I + 1) of column block J.  bf16: v_permlane16_swap turns the two tiles' 8-byte pieces
// into 16-byte stores -- the lanes of an even 16-lane row keep tile I and take their right neighbour row's 4 columns, the
// odd rows take tile I + 1: lane (q4, l15) stores 8 columns (16 J + 8 (q4 / 2) ..) of output row 16 (I + (q4 & 1)) + l15.
// (The caller routes outputs that are not 16-byte addressable to the edge path.)
template <class Geo, bool OUT_F32, bool GATE, int GELU, int J, int I>
OSK_DEV void pair_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, const int64_t* storeoff, int n0w, int q4, const float4& gq) {
  constexpr int NB = Geo::NB;
  const int n = n0w + J * 16 + q4 * 4;
  float a0[4], a1[4];
  tile_values<Geo, J * NB + I, GATE, GELU>(aq, p, rowoff[I], n, gq, a0);
  tile_values<Geo, J * NB + I + 1, GATE, GELU>(aq, p, rowoff[I + 1], n, gq, a1);
  if constexpr (OUT_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + rowoff[I] + n) = make_float4(a0[0], a0[1], a0[2], a0[3]);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + rowoff[I + 1] + n) = make_float4(a1[0], a1[1], a1[2], a1[3]);
  } else {
    // swap(vdst = tile I, src = tile I + 1): [0] = {I.row0, (I+1).row0, I.row2, (I+1).row2}, [1] = {I.row1, (I+1).row1, I.row3, (I+1).row3}
    auto sx = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a1[0], a1[1]), false, false);
    auto sy = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[2], a1[3]), false, false);
    // storeoff[I / 2] = element offset of this lane's output row 16 (I + (q4 & 1)) + l15, + its 8-column half (q4 / 2)
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.C) + storeoff[I / 2] + n0w + J * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  }
}

// ---- bf16 interior, round 4: stores that cover whole 128-byte row pieces.  After pair_interior's lane-row exchange a store
// instruction of column block J writes 32 rows x 32 bytes: 32 half-line accesses per KiB, and a workgroup tile's 128 KiB took
// ~11.4 k cycles (19-20 % of a K = 1152 tile; tools/gemm_tile_timing.py) at ~2.8 cycles per (instruction, line) -- the store path
// is issue-bound per touched line, not per byte.  Here the 16-byte chunks of four column blocks are transposed across the four
// lanes of a quad (rows l15 = 4 a .. 4 a + 3: two DPP butterfly steps) so that register r of lane j holds row 4 a + r's chunk of
// column block 4 b + j: one store instruction then writes 8 rows x 128 contiguous bytes (the two halves q4 >> 1 complete a block).
// (quad_transpose: osk_common.h)

// the 16-byte chunk pair_interior stores for (J, I): columns 16 J + 8 (q4 >> 1) .. + 7 of output row 16 (I + (q4 & 1)) + l15
template <class Geo, bool GATE, int GELU, int J, int I>
OSK_DEV uint4 chunk_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, int n0w, int q4, const float4& gq) {
  constexpr int NB = Geo::NB;
  const int n = n0w + J * 16 + q4 * 4;
  float a0[4], a1[4];
  tile_values<Geo, J * NB + I, GATE, GELU>(aq, p, rowoff[I], n, gq, a0);
  tile_values<Geo, J * NB + I + 1, GATE, GELU>(aq, p, rowoff[I + 1], n, gq, a1);
  auto sx = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a1[0], a1[1]), false, false);
  auto sy = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[2], a1[3]), false, false);
  return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

// the same chunk with the residual handed in as the 16-byte chunk of the SAME (row, columns) in the store layout (row_pair_wide's
// wide residual loads): v_permlane16_swap is an involution on its register pair, so swapping the chunk's (x, z) and (y, w) gives
// back the two tiles' 8-byte pieces in the accumulator layout
template <class Geo, int GELU, int J, int I>
OSK_DEV uint4 chunk_interior_res(const osk_v4f* aq, const GemmParams& p, int n0w, int q4, const float4& gq, const uint4& rc) {
  constexpr int NB = Geo::NB;
  const int n = n0w + J * 16 + q4 * 4;
  auto ux = __builtin_amdgcn_permlane16_swap(rc.x, rc.z, false, false);
  auto uy = __builtin_amdgcn_permlane16_swap(rc.y, rc.w, false, false);
  float a0[4], a1[4];
  tile_values_rv<Geo, J * NB + I, true, GELU>(aq, p, n, gq, make_uint2(ux[0], uy[0]), a0);
  tile_values_rv<Geo, J * NB + I + 1, true, GELU>(aq, p, n, gq, make_uint2(ux[1], uy[1]), a1);
  auto sx = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a1[0], a1[1]), false, false);
  auto sy = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[2], a1[3]), false, false);
  return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

// one pair of row blocks (I, I + 1), all NB column blocks: NB chunks per lane, transposed in groups of four, NB stores of
// 8 rows x 128 bytes.  In-place residual (res == C, the blocks' x = x + gate * proj(..)): the stores of this call cover exactly the 32
// rows x 16 NB columns whose residual pieces this call read, and every one of those loads is issued before the
// first store (the transposes need all NB chunks), as in pair_interior.  own = element offset of this lane's own store row (storeoff[I / 2]: + its 8-column half), crs = row stride:
// the rows of an interior wave tile lie in one batch item, so row 4 a + r is (r - j) rows from the lane's own row 4 a + j.
// RESW (round 5): the residual is read the way the result is stored -- NB loads of 8 rows x 128 bytes -- and brought back to the
// accumulator layout by the inverse lane exchanges (quad_transpose and the swap are involutions): the 8-byte pieces of the
// accumulator layout cost 2 NB loads of 16 rows x 32 bytes, ~2.8 cycles per (instruction, line) on the CU's one address path --
// 13 k of the gate class's 21 k epilogue cycles per workgroup tile (profiles/r05b_gemm_gate_epilogue_ab.jsonl).
template <class Geo, bool GATE, int GELU, bool RESW, int I, int... Js>
OSK_DEV void row_pair_wide(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, int64_t own, int n0w, int q4, int lane, const float4* gq,
                           std::integer_sequence<int, Js...>) {
  constexpr int NB = Geo::NB;
  static_assert(NB % 4 == 0, "column blocks are transposed in groups of four");
  uint4 d[NB];
  const int j = lane & 3;
  const bool odd = lane & 1, hi = lane & 2;
  const int64_t base_off = own + n0w + 16 * j - (int64_t)j * p.crs;
  if constexpr (GATE && RESW) {
    uint4 rw[NB];
    const unsigned short* rbase = p.res + base_off;
#pragma unroll
    for (int b = 0; b < NB / 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[4 * b + r] = *reinterpret_cast<const uint4*>(rbase + (int64_t)r * p.crs + 64 * b);
#pragma unroll
    for (int b = 0; b < NB / 4; ++b) {
      quad_transpose(rw[4 * b].x, rw[4 * b + 1].x, rw[4 * b + 2].x, rw[4 * b + 3].x, odd, hi);
      quad_transpose(rw[4 * b].y, rw[4 * b + 1].y, rw[4 * b + 2].y, rw[4 * b + 3].y, odd, hi);
      quad_transpose(rw[4 * b].z, rw[4 * b + 1].z, rw[4 * b + 2].z, rw[4 * b + 3].z, odd, hi);
      quad_transpose(rw[4 * b].w, rw[4 * b + 1].w, rw[4 * b + 2].w, rw[4 * b + 3].w, odd, hi);
    }
    ((d[Js] = chunk_interior_res<Geo, GELU, Js, I>(aq, p, n0w, q4, gq[Js], rw[Js])), ...);
  } else {
    ((d[Js] = chunk_interior<Geo, GATE, GELU, Js, I>(aq, p, rowoff, n0w, q4, gq[Js])), ...);
  }
#pragma unroll
  for (int b = 0; b < NB / 4; ++b) {
    quad_transpose(d[4 * b].x, d[4 * b + 1].x, d[4 * b + 2].x, d[4 * b + 3].x, odd, hi);
    quad_transpose(d[4 * b].y, d[4 * b + 1].y, d[4 * b + 2].y, d[4 * b + 3].y, odd, hi);
    quad_transpose(d[4 * b].z, d[4 * b + 1].z, d[4 * b + 2].z, d[4 * b + 3].z, odd, hi);
    quad_transpose(d[4 * b].w, d[4 * b + 1].w, d[4 * b + 2].w, d[4 * b + 3].w, odd, hi);
  }
  unsigned short* base = reinterpret_cast<unsigned short*>(p.C) + base_off;
#pragma unroll
  for (int b = 0; b < NB / 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<uint4*>(base + (int64_t)r * p.crs + 64 * b) = d[4 * b + r];
}

template <class Geo, bool GATE, int GELU, int... Is>
OSK_DEV void tile_interior_wide(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, const int64_t* storeoff, int n0w, int q4, int lane,
                                const float4* gq, std::integer_sequence<int, Is...>) {
  if constexpr (GATE) {
#ifndef OSK_GEMM_NARROW_RES   // (A/B builds of tools/: the residual in 8-byte pieces)
    if ((((uintptr_t)p.res) & 15) == 0) {   // (kernel-argument uniform; strides are C's, checked by the caller)
      (row_pair_wide<Geo, GATE, GELU, true, 2 * Is>(aq, p, rowoff, storeoff[Is], n0w, q4, lane, gq, std::make_integer_sequence<int, Geo::NB>{}), ...);
      return;
    }
#endif
  }
  (row_pair_wide<Geo, GATE, GELU, false, 2 * Is>(aq, p, rowoff, storeoff[Is], n0w, q4, lane, gq, std::make_integer_sequence<int, Geo::NB>{}), ...);
}

// edge tiles: per-element bounds checks (rows >= M were computed on clamped copies of row M-1 and are dropped)
template <class Geo, bool OUT_F32, int T>
OSK_DEV void tile_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool folded) {
  constexpr int NB = Geo::NB;
  constexpr int J = T / NB, I = T % NB;
  float acc[4];
  Geo::template read<T>(aq, acc);
  const int m = m0w + I * 16 + l15;
  if (m >= p.M) return;
  const int b = m / p.crpb, l = m - b * p.crpb;
  const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
  const float* grow = p.gate ? p.gate + b * p.gbs : nullptr;
  const int n = n0w + J * 16 + q4 * 4;
  for (int j = 0; j < 4 && n + j < p.N; ++j) {
    float t = acc[j];
    if (!folded && p.bias) t += p.bias[n + j];
    if (n + j >= p.gelu_from) t = gelu_tanh(t);
    if (grow) t = bf16_bits_to_f32(p.res[roff + n + j]) + grow[n + j] * t;
    if constexpr (OUT_F32) reinterpret_cast<float*>(p.C)[roff + n + j] = t;
    else reinterpret_cast<unsigned short*>(p.C)[roff + n + j] = f32_to_bf16_bits(t);
  }
}

// all column blocks J of one pair of row blocks, back to back: consecutive stores fill a row's 32-byte pieces left to right
// (with the column block outermost, the pieces of one 64-byte sector left four stores apart: +20 % fabric-side write traffic)
template <class Geo, bool OUT_F32, bool GATE, int GELU, int I, int... Js>
OSK_DEV void row_pair(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, const int64_t* storeoff, int n0w, int q4, const float4* gq,
                      std::integer_sequence<int, Js...>) {
  (pair_interior<Geo, OUT_F32, GATE, GELU, Js, I>(aq, p, rowoff, storeoff, n0w, q4, gq[Js]), ...);
}

template <class Geo, bool OUT_F32, bool GATE, int GELU, int... Is>
OSK_DEV void tile_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, const int64_t* storeoff, int n0w, int q4, const float4* gq,
                           std::integer_sequence<int, Is...>) {
  (row_pair<Geo, OUT_F32, GATE, GELU, 2 * Is>(aq, p, rowoff, storeoff, n0w, q4, gq, std::make_integer_sequence<int, Geo::NB>{}), ...);
}

template <class Geo, bool OUT_F32, bool GATE>
OSK_DEV void cols_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* rowoff, const int64_t* storeoff, int m0w, int n0w, int q4) {
  constexpr int NB = Geo::NB;
  constexpr auto seq = std::make_integer_sequence<int, NB / 2>{};
  float4 gq[NB];                          // gate of this lane's 4 channels of every column block (one batch per interior wave tile)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    gq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (GATE) gq[j] = *reinterpret_cast<const float4*>(p.gate + (m0w / p.crpb) * p.gbs + n0w + j * 16 + q4 * 4);
  }
  // GELU class of the whole wave tile (wave-uniform): none / all / per element where the boundary cuts through it
  if constexpr (!OUT_F32) {
#ifndef OSK_GEMM_NARROW_STORES   // (A/B builds of tools/: the 32-byte row pieces of round 3)
    const int lane = q4 * 16 + (int)(threadIdx.x & 15);
    if (n0w + NB * 16 <= p.gelu_from) tile_interior_wide<Geo, GATE, GELU_NONE>(aq, p, rowoff, storeoff, n0w, q4, lane, gq, seq);
    else if (n0w >= p.gelu_from) tile_interior_wide<Geo, GATE, GELU_ALL>(aq, p, rowoff, storeoff, n0w, q4, lane, gq, seq);
    else tile_interior_wide<Geo, GATE, GELU_MIXED>(aq, p, rowoff, storeoff, n0w, q4, lane, gq, seq);
    return;
#endif
  }
  if (n0w + NB * 16 <= p.gelu_from) tile_interior<Geo, OUT_F32, GATE, GELU_NONE>(aq, p, rowoff, storeoff, n0w, q4, gq, seq);
  else if (n0w >= p.gelu_from) tile_interior<Geo, OUT_F32, GATE, GELU_ALL>(aq, p, rowoff, storeoff, n0w, q4, gq, seq);
  else tile_interior<Geo, OUT_F32, GATE, GELU_MIXED>(aq, p, rowoff, storeoff, n0w, q4, gq, seq);
}

template <class Geo, bool OUT_F32, int... Ts>
OSK_DEV void tiles_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool folded, std::integer_sequence<int, Ts...>) {
  (tile_edge<Geo, OUT_F32, Ts>(aq, p, m0w, n0w, l15, q4, folded), ...);
}

// ---- GEGLU class (f4; north_star's "GEGLU MLP"): column blocks come in (value, gate) pairs 2 J2, 2 J2 + 1 -- the same lane holds
// the same output row and the same 4 relative channels of both -- and the wave tile's 128 GEMM columns become 64 output columns at
// n0w / 2.  bf16 output only; bias folded into the accumulators (interior) or added here (edge), in the packed column order.
template <class Geo, int J2, int I>
OSK_DEV void geglu_pair_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* storeoff, int n0o) {
  constexpr int NB = Geo::NB;
  float v0[4], g0[4], v1[4], g1[4];
  Geo::template read<(2 * J2) * NB + I>(aq, v0);
  Geo::template read<(2 * J2 + 1) * NB + I>(aq, g0);
  Geo::template read<(2 * J2) * NB + I + 1>(aq, v1);
  Geo::template read<(2 * J2 + 1) * NB + I + 1>(aq, g1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v0[i] *= gelu_tanh(g0[i]);
    v1[i] *= gelu_tanh(g1[i]);
  }
  auto sx = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
  auto sy = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
  *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.C) + storeoff[I / 2] + n0o + J2 * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

template <class Geo, int I, int... J2s>
OSK_DEV void geglu_row_pair(const osk_v4f* aq, const GemmParams& p, const int64_t* storeoff, int n0o, std::integer_sequence<int, J2s...>) {
  (geglu_pair_interior<Geo, J2s, I>(aq, p, storeoff, n0o), ...);
}

template <class Geo, int... Is>
OSK_DEV void geglu_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* storeoff, int n0o, std::integer_sequence<int, Is...>) {
  (geglu_row_pair<Geo, 2 * Is>(aq, p, storeoff, n0o, std::make_integer_sequence<int, Geo::NB / 2>{}), ...);
}

template <class Geo, int T2>   // T2 = J2 * NB + I
OSK_DEV void geglu_tile_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool folded) {
  constexpr int NB = Geo::NB;
  constexpr int J2 = T2 / NB, I = T2 % NB;
  float v[4], g[4];
  Geo::template read<(2 * J2) * NB + I>(aq, v);
  Geo::template read<(2 * J2 + 1) * NB + I>(aq, g);
  const int m = m0w + I * 16 + l15;
  if (m >= p.M) return;
  const int b = m / p.crpb, l = m - b * p.crpb;
  const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
  const int nv = n0w + (2 * J2) * 16 + q4 * 4;          // packed GEMM column of the value; its gate sits 16 columns on
  const int no = n0w / 2 + J2 * 16 + q4 * 4;            // output column
  for (int j = 0; j < 4 && no + j < p.N / 2; ++j) {
    float tv = v[j], tg = g[j];
    if (!folded && p.bias) { tv += p.bias[nv + j]; tg += p.bias[nv + 16 + j]; }
    reinterpret_cast<unsigned short*>(p.C)[roff + no + j] = f32_to_bf16_bits(tv * gelu_tanh(tg));
  }
}

template <class Geo, int... Ts>
OSK_DEV void geglu_tiles_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool folded, std::integer_sequence<int, Ts...>) {
  (geglu_tile_edge<Geo, Ts>(aq, p, m0w, n0w, l15, q4, folded), ...);
}

template <class Geo>
OSK_DEV void geglu_all(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool interior, bool folded) {
  constexpr int NB = Geo::NB;
  if (m0w >= p.M || n0w >= p.N) return;
  const bool wide = ((((uintptr_t)p.C) & 15) == 0) && ((p.crs & 7) == 0) && ((p.cbs & 7) == 0);
  if (!interior || !wide || (p.bias && !folded)) {
    geglu_tiles_edge<Geo>(aq, p, m0w, n0w, l15, q4, folded, std::make_integer_sequence<int, (NB / 2) * NB>{});
    return;
  }
  int64_t storeoff[NB / 2];
  const int b = m0w / p.crpb, l0 = m0w - b * p.crpb + l15;
#pragma unroll
  for (int i = 0; i < NB / 2; ++i) storeoff[i] = b * p.cbs + (int64_t)(l0 + 16 * (2 * i + (q4 & 1))) * p.crs + (q4 >> 1) * 8;
  geglu_interior<Geo>(aq, p, storeoff, n0w / 2, std::make_integer_sequence<int, NB / 2>{});
}

// ---- V^T class (round 6, osk_gemm_group_bf16's V^T task): the tile is a piece of V^T = W_v X^T -- rows m = (head, dim), columns =
// (batch, position on the key axis) -- stored at C[b * ccbs + m * crs + position] with the per-ROW bias b_v[m] added here (the K loop's
// folded bias is per column).  Interior wave tiles (all rows inside M, all 128 positions inside one batch and all their keys valid:
// the permutation stays inside 64-key groups) use row_pair_wide's 8 rows x 128 byte stores; everything else goes element by element,
// positions whose key lies behind the sequence end are stored as ZERO (the attention kernels rely on a zero V^T there).
template <class Geo, int J, int I>
OSK_DEV uint4 vt_chunk(const osk_v4f* aq, float rb0, float rb1) {
  constexpr int NB = Geo::NB;
  float a0[4], a1[4];
  Geo::template read<J * NB + I>(aq, a0);
  Geo::template read<J * NB + I + 1>(aq, a1);
#pragma unroll
  for (int i = 0; i < 4; ++i) { a0[i] += rb0; a1[i] += rb1; }
  auto sx = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a1[0], a1[1]), false, false);
  auto sy = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[2], a1[3]), false, false);
  return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}

template <class Geo, int I, int... Js>
OSK_DEV void vt_row_pair(const osk_v4f* aq, const GemmParams& p, int64_t own, const float* rb, int lane, std::integer_sequence<int, Js...>) {
  constexpr int NB = Geo::NB;
  uint4 d[NB];
  const int j = lane & 3;
  const bool odd = lane & 1, hi = lane & 2;
  ((d[Js] = vt_chunk<Geo, Js, I>(aq, rb[I], rb[I + 1])), ...);
#pragma unroll
  for (int b = 0; b < NB / 4; ++b) {
    quad_transpose(d[4 * b].x, d[4 * b + 1].x, d[4 * b + 2].x, d[4 * b + 3].x, odd, hi);
    quad_transpose(d[4 * b].y, d[4 * b + 1].y, d[4 * b + 2].y, d[4 * b + 3].y, odd, hi);
    quad_transpose(d[4 * b].z, d[4 * b + 1].z, d[4 * b + 2].z, d[4 * b + 3].z, odd, hi);
    quad_transpose(d[4 * b].w, d[4 * b + 1].w, d[4 * b + 2].w, d[4 * b + 3].w, odd, hi);
  }
  unsigned short* base = reinterpret_cast<unsigned short*>(p.C) + own + 16 * j - (int64_t)j * p.crs;
#pragma unroll
  for (int b = 0; b < NB / 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<uint4*>(base + (int64_t)r * p.crs + 64 * b) = d[4 * b + r];
}

template <class Geo, int... Is>
OSK_DEV void vt_interior(const osk_v4f* aq, const GemmParams& p, const int64_t* storeoff, const float* rb, int lane, std::integer_sequence<int, Is...>) {
  (vt_row_pair<Geo, 2 * Is>(aq, p, storeoff[Is], rb, lane, std::make_integer_sequence<int, Geo::NB>{}), ...);
}

template <class Geo, int T>
OSK_DEV void vt_tile_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4) {
  constexpr int NB = Geo::NB;
  constexpr int J = T / NB, I = T % NB;
  float acc[4];
  Geo::template read<T>(aq, acc);
  const int m = m0w + I * 16 + l15;
  if (m >= p.M) return;
  const float rb = p.rowbias ? p.rowbias[m] : 0.f;
  const int n = n0w + J * 16 + q4 * 4;
  for (int j = 0; j < 4 && n + j < p.N; ++j) {
    const int b = (n + j) / p.wrpb, pos = (n + j) - b * p.wrpb;
    const float t = vt_perm64(pos, p.vt) < p.wvalid ? acc[j] + rb : 0.f;
    reinterpret_cast<unsigned short*>(p.C)[b * p.ccbs + (int64_t)m * p.crs + pos] = f32_to_bf16_bits(t);
  }
}

template <class Geo, int... Ts>
OSK_DEV void vt_tiles_edge(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, std::integer_sequence<int, Ts...>) {
  (vt_tile_edge<Geo, Ts>(aq, p, m0w, n0w, l15, q4), ...);
}

template <class Geo>
OSK_DEV void vt_all(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4) {
  constexpr int NB = Geo::NB, WT = NB * 16;
  if (m0w >= p.M || n0w >= p.N) return;
  const int bcol = n0w / p.wrpb, pos0 = n0w - bcol * p.wrpb;                       // wave-uniform
  const bool interior = m0w + WT <= p.M && n0w + WT <= p.N && pos0 + WT <= p.wrpb && pos0 + WT <= p.wvalid &&
                        ((((uintptr_t)p.C) & 15) == 0) && ((p.crs & 7) == 0) && ((p.ccbs & 7) == 0);
  if (!interior) {
    vt_tiles_edge<Geo>(aq, p, m0w, n0w, l15, q4, std::make_integer_sequence<int, NB * NB>{});
    return;
  }
  float rb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) rb[i] = p.rowbias ? p.rowbias[m0w + 16 * i + l15] : 0.f;
  // element offset of the row this lane STORES after the lane-row exchange (row_pair_wide's `own`), + the tile's column origin
  int64_t storeoff[NB / 2];
  const int64_t col0 = bcol * p.ccbs + pos0 + (q4 >> 1) * 8;
#pragma unroll
  for (int i = 0; i < NB / 2; ++i) storeoff[i] = (int64_t)(m0w + l15 + 16 * (2 * i + (q4 & 1))) * p.crs + col0;
  const int lane = q4 * 16 + l15;
  vt_interior<Geo>(aq, p, storeoff, rb, lane, std::make_integer_sequence<int, NB / 2>{});
}

// the whole 128 x 128 wave tile
template <class Geo, bool OUT_F32>
OSK_DEV void epilogue_all(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l15, int q4, bool interior, bool folded) {
  constexpr int NB = Geo::NB;
  if constexpr (!OUT_F32) {
    if (p.geglu) {   // (kernel-argument uniform)
      geglu_all<Geo>(aq, p, m0w, n0w, l15, q4, interior, folded);
      return;
    }
  }
  if (m0w >= p.M || n0w >= p.N) return;   // the whole wave tile lies outside C (ragged last tile row / column): wave-uniform
  // the fast path stores 16 bytes per lane (bf16) / reads 8-byte residual pieces: C, its strides and the tile origin must allow it
  const bool wide = OUT_F32 || (((((uintptr_t)p.C) & 15) == 0) && ((p.crs & 7) == 0) && ((p.cbs & 7) == 0));
  if (!interior || !wide || (p.bias && !folded)) {
    tiles_edge<Geo, OUT_F32>(aq, p, m0w, n0w, l15, q4, folded, std::make_integer_sequence<int, NB * NB>{});
    return;
  }
  // element offsets of this lane's NB output rows (an interior wave tile lies inside one batch: one division for all of them),
  // and of the NB / 2 rows it STORES after the lane-row exchange
  int64_t rowoff[NB], storeoff[NB / 2];
  const int b = m0w / p.crpb, l0 = m0w - b * p.crpb + l15;
#pragma unroll
  for (int i = 0; i < NB; ++i) rowoff[i] = b * p.cbs + (int64_t)(l0 + 16 * i) * p.crs;
#pragma unroll
  for (int i = 0; i < NB / 2; ++i) storeoff[i] = ((q4 & 1) ? rowoff[2 * i + 1] : rowoff[2 * i]) + (q4 >> 1) * 8;
  if (p.gate) cols_interior<Geo, OUT_F32, true>(aq, p, rowoff, storeoff, m0w, n0w, q4);
  else cols_interior<Geo, OUT_F32, false>(aq, p, rowoff, storeoff, m0w, n0w, q4);
}

}  // namespace epi16
}  // namespace osk_gemm
