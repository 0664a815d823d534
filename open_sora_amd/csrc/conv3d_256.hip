// CausalConv3d for gfx950, large tile + hand-scheduled K loop (Cin % 128 == 0, Cout >= 128, >= 256 output voxels).
//
// Same implicit GEMM as conv3d.hip (NDHWC activations, K = tap-major / channel-minor, replicate / causal padding as a
// CLAMP and the decoder's nearest upsample as a SHIFT of the gathered coordinate, weights pre-laid as [Cout][27 Cin]) on the
// tile, LDS image and pipeline of gemm256x.hip: 256 voxels x 256 (NBJ = 8) or 128 (NBJ = 4) output channels x 64, 4 waves (one
// per SIMD), v_mfma_f32_16x16x32_bf16, LDS-DMA double buffer, accumulators in AGPRs, persistent workgroups.  The K axis is all
// 27 taps in ONE asm call (conv256x_body*.inc, tools/gen_gemm_asm.py::gen_conv_x4): inside a tap a K step only advances the
// channel block (exactly a GEMM K step with per-lane row offsets); at a tap boundary every row slot takes a new voxel offset
// from an LDS table [tap][256 rows] built per tile by build_tap_table().
// (Rounds 1-2 also shipped 8-wave and 32x32x16 forms of this kernel and a per-tap-segment variant: they lost their A/B runs --
// profiles/r02_* -- and are generator options / git history now.)
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2 * Cin * Cout * k^3 * B*To*Ho*Wo.
#include "acc_quads.h"
#include "conv_params.h"
#include "gemm256x_regs.inc"
#include "convsw_regs.inc"
#include <type_traits>

namespace osk_conv {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

#ifdef OSK_CONV_TILE_TIMING   // tools/make_conv_timing_lib.sh: where a sliding-window tile's time goes (s_memtime sums of wave 0 of every workgroup)
__device__ unsigned long long osk_conv_tile_ticks[4];   // address set-up, asm statement (prologue + K loop), epilogue, tiles
#define OSK_CT(i, t0) if (threadIdx.x == 0) atomicAdd(&osk_conv_tile_ticks[i], __builtin_amdgcn_s_memtime() - (t0))
#define OSK_CT_STAMP(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#else
#define OSK_CT(i, t0)
#define OSK_CT_STAMP(name)
#endif

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
  if (p.brick == 2) {   // 512-row tiles (convsw2_kernel): the 16 x 16 brick of the frame PAIR (2 tp, 2 tp + 1), pair index fastest;
    const int ntp = (p.To + 1) >> 1;                 // rows 256.. are the second frame (beyond To for the last pair of an odd To:
    const int tp = bm % ntp, sb = bm / ntp;          // "row >= M", which every caller masks)
    const int t = 2 * tp + (r >> 8);
    if (t >= p.To) return p.M;
    const int nwb = p.Wo >> 4, nhb = p.Ho >> 4;
    const int wb = sb % nwb, q = sb / nwb;
    const int hb = q % nhb, b = q / nhb;
    return ((b * p.To + t) * p.Ho + hb * 16 + ((r >> 4) & 15)) * p.Wo + wb * 16 + (r & 15);
  }
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

typedef __bf16 gn_bf16x2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// conv256w_kernel on v_mfma_f32_16x16x32_bf16 (gemm256x.hip's compute side: at the board's power cap the 16x16x32 stream is the
// cheaper one per flop, profiles/r02_gemm_experiments.md): 8 x 8 accumulator tiles of 16 x 16 per wave, the same LDS image read
// through a 16-row x 32-k lane map, K loop conv256x_body.inc (tools/gen_gemm_asm.py::gen_conv_x4).  Of every 16 x 16 tile
// (J = channel block, I = voxel block) a lane owns voxel row 16 I + l15 and channels 16 J + 4 q4 .. + 3.
// tile T's 4 accumulators = quad T of aq: the wave's accumulators as compiler-visible values (acc_quads.h: outputs of an empty asm
// statement behind the K-loop statement), read in place and in program order
template <int T>
OSK_DEV void read_x(const osk_v4f* aq, float* v4) {
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v4[i]) : "a"(aq[T][i]));
}
static_assert(OSKX_ACC_QUADS == 64 && OSKX128_ACC_QUADS == 32, "the generated loops' accumulator map: quad T = tile T");
#define OSKCX_ACC(NBJ_, aq)                                     \
  osk_v4f aq[(NBJ_) * 8];                                       \
  if constexpr ((NBJ_) == 8) asm volatile("" : OSK_AQ_OUT_0_64(aq)); \
  else asm volatile("" : OSK_AQ_OUT_0_32(aq))

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
OSK_DEV void pair_x(const osk_v4f* aq, const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid,
                    int n, int ncol, const float4& bq, float& gs, float& gq) {
  float a0[4], a1[4];
  uint2 r0 = make_uint2(0, 0), r1 = r0;
  if constexpr (RES) {
#ifdef OSK_CONV_NARROW_RES   // (A/B builds of tools/: the residual in the accumulator layout's 8-byte pieces, 2 loads of 16 rows x 32 bytes)
    r0 = *reinterpret_cast<const uint2*>(p.res + rowoff[I] + n);        // rows beyond M read row 0 (clamped offsets)
    r1 = *reinterpret_cast<const uint2*>(p.res + rowoff[I + 1] + n);
#else
    // round 6: the residual is read the way the result is STORED -- the 16-byte chunk of this lane's store row (ONE load of 32 rows x
    // 32 bytes instead of two of 16 x 32: the CU's address path charges per (instruction, line), gemm_epilogue.h's round-5 finding)
    // -- and brought back to the accumulator layout by the inverse lane-row exchange (v_permlane16_swap is an involution on its
    // register pair).  Rows beyond M: storeoff is clamped to row 0 like rowoff.
    const uint4 rc = *reinterpret_cast<const uint4*>(p.res + storeoff[I / 2] + ncol);
    auto ux = __builtin_amdgcn_permlane16_swap(rc.x, rc.z, false, false);
    auto uy = __builtin_amdgcn_permlane16_swap(rc.y, rc.w, false, false);
    r0 = make_uint2(ux[0], uy[0]);
    r1 = make_uint2(ux[1], uy[1]);
#endif
  }
  read_x<J * OSKX_NB + I>(aq, a0);
  read_x<J * OSKX_NB + I + 1>(aq, a1);
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
OSK_DEV void rowpair_x(const osk_v4f* aq, const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid,
                       int n0w, int q4, const float4* bq, float* gs, float* gq, std::integer_sequence<int, Js...>) {
  ((FULL || n0w + Js * 16 < p.Cout
        ? pair_x<RES, GN, Js, I>(aq, p, rowoff, valid, storeoff, svalid, n0w + Js * 16 + q4 * 4, n0w + Js * 16, bq[Js], gs[Js], gq[Js])
        : (void)0), ...);
}

template <bool RES, bool GN, bool FULL, int NBJ, int... Is>
OSK_DEV void tile_x(const osk_v4f* aq, const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid, int n0,
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
  (rowpair_x<RES, GN, FULL, 2 * Is>(aq, p, rowoff, valid, storeoff, svalid, n0w, q4, bq, gs, gq, std::make_integer_sequence<int, NBJ>{}), ...);
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
OSK_DEV void cols_x(const osk_v4f* aq, const ConvParams& p, const int64_t* rowoff, const bool* valid, const int64_t* storeoff, const bool* svalid, int n0,
                    int n0w, int l15, int q4, float* ls) {
  constexpr auto seq = std::make_integer_sequence<int, OSKX_NB / 2>{};
  if (n0w + NBJ * 16 <= p.Cout) tile_x<RES, GN, true, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls, seq);
  else tile_x<RES, GN, false, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls, seq);   // ragged last tile column
}

// generic path of one tile (any Cout, any alignment): per-element bounds checks, no statistics
template <int T>
OSK_DEV void tile_x_generic(const osk_v4f* aq, const ConvParams& p, int bm, int r0w, int n0w, int l15, int q4) {
  constexpr int J = T / OSKX_NB, I = T % OSKX_NB;
  float acc[4];
  read_x<T>(aq, acc);
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
OSK_DEV void tiles_x_generic(const osk_v4f* aq, const ConvParams& p, int bm, int r0w, int n0w, int l15, int q4, std::integer_sequence<int, Ts...>) {
  (tile_x_generic<Ts>(aq, p, bm, r0w, n0w, l15, q4), ...);
}

// the whole workgroup calls this after its K loop (see epilogue_all): NBJ = 16-channel blocks per wave (8: 256 channels per
// workgroup tile from n0, 4: 128)
template <int NBJ>
OSK_DEV void epilogue_all_x(const osk_v4f* aq, const ConvParams& p, int bm, int r0w, int n0, int n0w, int l15, int q4, unsigned char* smem) {
  constexpr int NB = OSKX_NB, BN = 32 * NBJ;
  const bool fast = (p.Cout & 15) == 0 && ((((uintptr_t)p.out) & 15) == 0) && (!p.res || (((uintptr_t)p.res) & 15) == 0) &&
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
      tiles_x_generic(aq, p, bm, r0w, n0w, l15, q4, std::make_integer_sequence<int, NB * NBJ>{});
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
        if (p.res) cols_x<true, true, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
        else cols_x<false, true, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
      } else {
        if (p.res) cols_x<true, false, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
        else cols_x<false, false, NBJ>(aq, p, rowoff, valid, storeoff, svalid, n0, n0w, l15, q4, ls);
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
  OSKCX_ACC(NBJ, aq);
  epilogue_all_x<NBJ>(aq, p, bm, wm * WT, n0, n0 + wn * WTN, l15, q4, smem);
  }   // tile loop
}

// ---------------------------------------------------------------------------------------------------------------
// Sliding-window form (3 x 3 x 3, stride 1, plain or with the nearest 2x upsample folded in (UP), Ho % 16 == 0 and Wo % 16 == 0, Cin % 64 == 0): a tile is the
// 16 x 16 spatial brick of one output frame; the 3-frame x 18 x 18 halo brick of one 32-channel block sits in LDS and the 27
// taps differ only in the immediate offset of their fragment reads -- the activations of a tile cross the fabric once per
// channel block instead of 27 times.  K loop, LDS plan and the lane formulas below: tools/gen_conv_sw_asm.py (its header is the
// design note; tests/test_conv_sw_model.py executes the generated stream symbolically and lane by lane in numpy with THESE
// formulas).  Accumulator layout and epilogue are conv256x_kernel's.
// UP: the decoder's nearest 2x upsample in H and W (and, with p.up_t, in T) folded in: a 16 x 16 OUTPUT brick reads a 10 x 10
// SOURCE patch per frame slot (output row o reads source row o >> 1), pairs of lanes share a source voxel.
OSK_DEV int swu_key(int ww) { return ww >= 6 ? 2 : 0; }   // swizzle key of source halo column ww (tests/conv_sw_emulator.py::up_key)

// GN (NBJ == 8, plain geometry): p.gn_in != null -- the input is silu(GroupNorm(x)); the halo pieces travel through registers
// (global_load_dwordx4, a lane always holds chunk lane % 4 of its voxel), are transformed in the MFMA shadows and written to the
// swizzled LDS position with ds_write_b128 (generator: the GN form in tools/gen_conv_sw_asm.py's header)
template <int NBJ, bool UP, bool GN = false>
__global__ void __launch_bounds__(256, 1) convsw_kernel(const ConvParams p) {
  static_assert(!GN || (NBJ == 8 && !UP), "the GN form exists for the 256-channel plain geometry");
  constexpr int WT = OSKX_NB * 16, WTN = NBJ * 16, BN = 32 * NBJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int nwb = p.Wo >> 4, nhb = p.Ho >> 4;
  const int nbm = p.B * p.To * nhb * nwb, nbn = (p.Cout + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // ---- fragment side (tile-independent): activation row l15 of a 16-voxel block = brick column l15; 16-byte chunk q4 of the
  // voxel at position q4 ^ key(halo column), key(ww) = (ww >> 1) & 3, halo column = l15 + dw
  unsigned xa[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) {
    if constexpr (UP) {
      const int ww = ((l15 + dw - 1) >> 1) + 1;                    // source halo column of brick column l15 under tap shift dw
      xa[dw] = lds_base + (unsigned)((40 * wm + ww) * 64 + ((q4 ^ swu_key(ww)) << 4));
    } else {
      xa[dw] = lds_base + (unsigned)((144 * wm + l15) * 64 + ((q4 ^ (((l15 + dw) >> 1) & 3)) << 4));
    }
  }
  const unsigned yb = lds_base + (unsigned)((wn * WTN + l15) * 64 + ((q4 ^ ((l15 >> 1) & 3)) << 4));
  // the last halo piece of a slot: block 20 of 21 (every wave) / block min(4 + wave, 6) of 7
  const unsigned dst = rfl(lds_base + wave * 1024), dst5 = rfl(lds_base + (UP ? (4 + wave < 6 ? 4 + wave : 6) : 20) * 1024);
  const unsigned cin2 = rfl((unsigned)p.Cin * 2), nbody = rfl((unsigned)p.Cin / 64);
  const int sub = lane >> 2, pos = lane & 3;                     // an LDS-DMA piece = 16 rows of 64 bytes: lane -> (row, position)

  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
  OSK_CT_STAMP(ct0);
  const int tile = xcd_remap(it, ntiles);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int n0 = bn * BN;
  const int t = bm % p.To, sb = bm / p.To;                         // tile_row_to_voxel()'s brick order: frame fastest
  const int wb = sb % nwb, qq = sb / nwb;
  const int hb = qq % nhb, b = qq / nhb;
  __syncthreads();                                                 // every wave has left the previous tile's K loop and epilogue

  // ---- weight pieces: piece k of this wave = rows 16 (4 k + wave) .. + 15 of the BN-row stage
  unsigned woff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int nl = 16 * (4 * (k & (NBJ / 2 - 1)) + wave) + sub;
    int n = n0 + nl;
    n = n < p.Cout ? n : p.Cout - 1;
    woff[k] = (unsigned)(((int64_t)n * p.wrs + (pos ^ ((nl >> 1) & 3)) * 8) * 2);
  }
  // ---- halo pieces: piece k of this wave = halo voxels 16 q .. + 15 of a frame slot, q = min(4 k + wave, 20); voxel v = 18 hh + ww
  // reads input (hb 16 - 1 + hh, wb 16 - 1 + ww) clamped into the frame (replicate padding)
  // (UP: two pieces per wave, q = min(4 k + wave, 6), of the 10 x 10 source patch from (hb 8 - 1, wb 8 - 1))
  unsigned hoff[6], hdw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int hs, ws, key;
    hdw[k] = 0;
    if constexpr (UP) {
      int q = 4 * (k & 1) + wave;
      q = q < 6 ? q : 6;
      int v = 16 * q + sub;
      v = v < 99 ? v : 99;
      const int hh = v / 10, ww = v - hh * 10;
      hs = hb * 8 - 1 + hh; ws = wb * 8 - 1 + ww; key = swu_key(ww);
    } else {
      int q = 4 * k + wave;
      q = q < 20 ? q : 20;
      int v = 16 * q + sub;
      v = v < 323 ? v : 323;
      const int hh = v / 18, ww = v - hh * 18;
      hs = hb * 16 - 1 + hh; ws = wb * 16 - 1 + ww; key = (ww >> 1) & 3;
      if constexpr (GN) {          // the swizzle moves from the global offset to the LDS write address
        hdw[k] = lds_base + (unsigned)(v * 64 + ((pos ^ key) << 4));
        key = 0;
      }
    }
    hs = hs < 0 ? 0 : (hs > p.H - 1 ? p.H - 1 : hs);
    ws = ws < 0 ? 0 : (ws > p.W - 1 ? p.W - 1 : ws);
    hoff[k] = (unsigned)((((int64_t)hs * p.W + ws) * p.Cin + (pos ^ key) * 8) * 2);
  }
  // ---- frame slot dt holds the source frame of conv-input frame max(t + dt - 2, 0) (causal padding = replicate the first frame;
  // under the time upsample conv-input frame tu > 0 is source frame 1 + (tu - 1) / 2, frame 0 stays single)
  uint64_t xb[4];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
    int tu = t + dt - 2;
    tu = tu < 0 ? 0 : tu;
    const int fs = p.up_t ? (tu == 0 ? 0 : 1 + ((tu - 1) >> 1)) : tu;
    xb[dt] = rfl64((uint64_t)(uintptr_t)(p.x + ((int64_t)b * p.T + fs) * p.H * p.W * p.Cin));
  }
  xb[3] = xb[2];                                                   // (the two-frame form's fourth slot: unused here)
  const uint64_t wbase = rfl64((uint64_t)(uintptr_t)p.w);
#define OSKSW_OPERANDS                                                                                                       \
  ::"v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(yb), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(hoff[0]),       \
      "v"(hoff[1]), "v"(hoff[2]), "v"(hoff[3]), "v"(hoff[4]), "v"(hoff[5]), "s"(wbase), "s"(xb[0]), "s"(xb[1]), "s"(xb[2]),  \
      "s"(xb[3]), "s"(cin2), "s"(nbody), "s"(dst), "s"(dst), "s"(dst5)
  OSK_CT(0, ct0);
  OSK_CT_STAMP(ct1);
  // GN form: + the LDS write address of each halo piece, this lane's 64 bytes inside a channel block's table rows, the table of batch b
#define OSKSWG_OPERANDS                                                                                                      \
  ::"v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(yb), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(hoff[0]),       \
      "v"(hoff[1]), "v"(hoff[2]), "v"(hoff[3]), "v"(hoff[4]), "v"(hoff[5]), "v"(hdw[0]), "v"(hdw[1]), "v"(hdw[2]),           \
      "v"(hdw[3]), "v"(hdw[4]), "v"(hdw[5]), "v"(goff), "s"(wbase), "s"(xb[0]), "s"(xb[1]), "s"(xb[2]), "s"(xb[3]),          \
      "s"(cin2), "s"(nbody), "s"(dst), "s"(dst), "s"(dst5), "s"(gbase)
  if constexpr (GN) {
    const unsigned goff = (unsigned)pos * 64;
    const uint64_t gbase = rfl64((uint64_t)(uintptr_t)(p.gn_in + (int64_t)b * p.Cin * 2));
    asm volatile(
#include "convswg_body_n256.inc"
        OSKSWG_OPERANDS : OSKSWG256_CLOBBERS);
  } else if constexpr (NBJ == 8 && !UP) {
    asm volatile(
#include "convsw_body_n256.inc"
        OSKSW_OPERANDS : OSKSW256_CLOBBERS);
  } else if constexpr (NBJ == 4 && !UP) {
    asm volatile(
#include "convsw_body_n128.inc"
        OSKSW_OPERANDS : OSKSW128_CLOBBERS);
  } else if constexpr (NBJ == 8) {
    asm volatile(
#include "convswu_body_n256.inc"
        OSKSW_OPERANDS : OSKSW256_CLOBBERS);
  } else {
    asm volatile(
#include "convswu_body_n128.inc"
        OSKSW_OPERANDS : OSKSW128_CLOBBERS);
  }
  OSKCX_ACC(NBJ, aq);
  OSK_CT(1, ct1);
  OSK_CT_STAMP(ct2);
  epilogue_all_x<NBJ>(aq, p, bm, wm * WT, n0, n0 + wn * WTN, l15, q4, smem);
  OSK_CT(2, ct2);
#ifdef OSK_CONV_TILE_TIMING
  if (threadIdx.x == 0) atomicAdd(&osk_conv_tile_ticks[3], 1ull);
#endif
  }   // tile loop
}

// Two-frame form for Cout == 128 (plain geometry): the tile is the 16 x 16 brick of TWO consecutive output frames x all 128
// channels = 512 voxels; wave = (frame f = wave >> 1, brick half = wave & 1) with a 128 x 128 wave tile (the 256-wide form's 64
// accumulator tiles), four frame slots in LDS (frame f's taps read slots f .. f + 2).  Per 512 voxels ONE prologue / epilogue
// ramp and one pass of the weights instead of two, 16 fragment reads per 64 MFMAs instead of 12 per 32: the 128-channel layers at
// full resolution (a third of the VAE's conv FLOPs) lost 37 % of their time to per-tile fixed costs in the one-frame form.
template <bool GN>
__global__ void __launch_bounds__(256, 1) convsw2_kernel(const ConvParams p) {
  constexpr int NBJ = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = wave >> 1, half = wave & 1;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int nwb = p.Wo >> 4, nhb = p.Ho >> 4, ntp = (p.To + 1) >> 1;
  const int ntiles = p.B * ntp * nhb * nwb;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned xa[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw)
    xa[dw] = lds_base + (unsigned)(fr * OSKSWF128_SLOT + (144 * half + l15) * 64 + ((q4 ^ (((l15 + dw) >> 1) & 3)) << 4));
  const unsigned yb = lds_base + (unsigned)(l15 * 64 + ((q4 ^ ((l15 >> 1) & 3)) << 4));
  const unsigned dst = rfl(lds_base + wave * 1024), dst5 = rfl(lds_base + 20 * 1024);
  const unsigned cin2 = rfl((unsigned)p.Cin * 2), nbody = rfl((unsigned)p.Cin / 64);
  const int sub = lane >> 2, pos = lane & 3;

  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
  OSK_CT_STAMP(ct0);
  const int bm = xcd_remap(it, ntiles);
  const int tp = bm % ntp, sb = bm / ntp;                          // tile_row_to_voxel()'s brick = 2 order
  const int wb = sb % nwb, qq = sb / nwb;
  const int hb = qq % nhb, b = qq / nhb;
  __syncthreads();
  unsigned woff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {                                    // 8 pieces cover the 128 weight rows: two per wave
    const int nl = 16 * (4 * (k & 1) + wave) + sub;
    const int n = nl < p.Cout ? nl : p.Cout - 1;
    woff[k] = (unsigned)(((int64_t)n * p.wrs + (pos ^ ((nl >> 1) & 3)) * 8) * 2);
  }
  unsigned hoff[6], hdw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int q = 4 * k + wave;
    q = q < 20 ? q : 20;
    int v = 16 * q + sub;
    v = v < 323 ? v : 323;
    const int hh = v / 18, ww = v - hh * 18;
    int hs = hb * 16 - 1 + hh, ws = wb * 16 - 1 + ww;
    hs = hs < 0 ? 0 : (hs > p.H - 1 ? p.H - 1 : hs);
    ws = ws < 0 ? 0 : (ws > p.W - 1 ? p.W - 1 : ws);
    const int key = (ww >> 1) & 3;
    hdw[k] = lds_base + (unsigned)(v * 64 + ((pos ^ key) << 4));           // (GN form only)
    hoff[k] = (unsigned)((((int64_t)hs * p.W + ws) * p.Cin + (GN ? pos : pos ^ key) * 8) * 2);
  }
  uint64_t xb[4];                                                  // slot d = input frame clamp(2 tp + d - 2) (the last one exists
#pragma unroll                                                     // only if the pair's second frame does: clamped to T - 1 otherwise)
  for (int d = 0; d < 4; ++d) {
    int fs = 2 * tp + d - 2;
    fs = fs < 0 ? 0 : (fs > p.T - 1 ? p.T - 1 : fs);
    xb[d] = rfl64((uint64_t)(uintptr_t)(p.x + ((int64_t)b * p.T + fs) * p.H * p.W * p.Cin));
  }
  const uint64_t wbase = rfl64((uint64_t)(uintptr_t)p.w);
  OSK_CT(0, ct0);
  OSK_CT_STAMP(ct1);
  if constexpr (GN) {
    const unsigned goff = (unsigned)pos * 64;
    const uint64_t gbase = rfl64((uint64_t)(uintptr_t)(p.gn_in + (int64_t)b * p.Cin * 2));
    asm volatile(
#include "convswgf_body_n128.inc"
        OSKSWG_OPERANDS : OSKSWGF128_CLOBBERS);
  } else {
    asm volatile(
#include "convswf_body_n128.inc"
        OSKSW_OPERANDS : OSKSW256_CLOBBERS);
  }
  OSKCX_ACC(NBJ, aq);
  OSK_CT(1, ct1);
  OSK_CT_STAMP(ct2);
  epilogue_all_x<NBJ>(aq, p, bm, wave * 128, 0, 0, l15, q4, smem);
  OSK_CT(2, ct2);
#ifdef OSK_CONV_TILE_TIMING
  if (threadIdx.x == 0) atomicAdd(&osk_conv_tile_ticks[3], 1ull);
#endif
  }   // tile loop
}

// one workgroup per CU (a multiple of 8, so that the XCD remap of the tile list keeps a workgroup inside one XCD's range)
int persistent_grid(int ntiles) {
  int n_cu = osk_device_cus();
  n_cu -= n_cu % 8;
  if (n_cu < 8) n_cu = 8;
  return ntiles > n_cu ? n_cu : ntiles;
}

template <int NBJ>
int launch_x(const ConvParams& p, hipStream_t st) {
  constexpr int BN = 32 * NBJ, SMEM = (NBJ == 8 ? OSKX_SMEM : OSKX128_SMEM) + 27 * 1024;
  OSK_ENSURE_MAX_SMEM(conv256x_kernel<NBJ>, SMEM);
  const int nblk = ((p.M + 255) / 256) * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(conv256x_kernel<NBJ>, dim3(persistent_grid(nblk)), dim3(256), SMEM, st, p);
  return (int)hipGetLastError();
}

template <int NBJ, bool UP, bool GN = false>
int launch_sw(const ConvParams& p0, hipStream_t st) {
  constexpr int BN = 32 * NBJ;
  constexpr int SMEM = UP ? (NBJ == 8 ? OSKSWU256_SMEM : OSKSWU128_SMEM) : (NBJ == 8 ? OSKSW256_SMEM : OSKSW128_SMEM);
  ConvParams p = p0;
  p.brick = 1;
  OSK_ENSURE_MAX_SMEM((convsw_kernel<NBJ, UP, GN>), SMEM);
  const int nblk = (p.M / 256) * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL((convsw_kernel<NBJ, UP, GN>), dim3(persistent_grid(nblk)), dim3(256), SMEM, st, p);
  return (int)hipGetLastError();
}

template <bool GN>
int launch_sw2(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  p.brick = 2;
  OSK_ENSURE_MAX_SMEM(convsw2_kernel<GN>, OSKSWF128_SMEM);
  const int nblk = p.B * ((p.To + 1) / 2) * (p.Ho / 16) * (p.Wo / 16);
  hipLaunchKernelGGL(convsw2_kernel<GN>, dim3(persistent_grid(nblk)), dim3(256), OSKSWF128_SMEM, st, p);
  return (int)hipGetLastError();
}

// sliding-window form: 3 x 3 x 3, stride 1, whole 16 x 16 output bricks, whole pairs of 32-channel blocks; the tensor as stored or
// under the decoder's upsample (H and W together, T optionally); per-lane byte offsets inside one frame / the weight tensor
// are 32-bit (conv256_supported)
bool convsw_supported(const ConvParams& p) {
  return p.ks == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && (p.up_hw || !p.up_t) && (p.Ho & 15) == 0 && (p.Wo & 15) == 0 &&
         p.Cin % 64 == 0 && p.Cout >= 128;
}

}  // namespace

// input GroupNorm + SiLU folded in: the sliding-window kernels in plain geometry, 256-channel tiles or the two-frame form
bool conv256_gn_in_supported(const ConvParams& p) {
#ifdef OSK_CONV_NO_SW
  return false;
#else
  return convsw_supported(p) && !p.up_hw && (p.Cout >= 256 || (p.Cout == 128 && p.To >= 2));
#endif
}

#ifdef OSK_CONV_TILE_TIMING
// read and clear the tick sums (tools/conv_tile_timing.py)
extern "C" int osk_conv_tile_timing_read(unsigned long long* out4) {
  unsigned long long zero[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(osk_conv_tile_ticks), sizeof(zero)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(osk_conv_tile_ticks), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

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

// linear 256-voxel tiles (the 16 x 16-brick tile order measured in round 2 -- same time, 12 % more fabric-side reads,
// profiles/r02_pmc_gemm_conv.txt -- stays a ConvParams field the kernel honours, never set)
int launch_conv256(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
#ifndef OSK_CONV_NO_SW   // (A/B builds of tools/: -DOSK_CONV_NO_SW keeps every layer on the implicit-GEMM kernel)
  if (p.gn_in) {
    if (!conv256_gn_in_supported(p)) return OSK_EUNSUPPORTED;
    return p.Cout >= 256 ? launch_sw<8, false, true>(p, st) : launch_sw2<true>(p, st);
  }
  if (convsw_supported(p)) {
    if (p.up_hw) return p.Cout >= 256 ? launch_sw<8, true>(p, st) : launch_sw<4, true>(p, st);
#ifndef OSK_CONV_NO_SW2
    if (p.Cout == 128 && p.To >= 2) return launch_sw2<false>(p, st);
#endif
    return p.Cout >= 256 ? launch_sw<8, false>(p, st) : launch_sw<4, false>(p, st);
  }
#endif
  p.brick = 0;
  return p.Cout >= 256 ? launch_x<8>(p, st) : launch_x<4>(p, st);
}

}  // namespace osk_conv
