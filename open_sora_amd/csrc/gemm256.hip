// Large-tile GEMM frame of round 1 (one output tile per workgroup), now instantiated for FP8 operands only -- the bf16
// Linears run on gemm256x.hip (256-wide tiles) / gemm256p.hip (128-wide) / gemm_bf16.hip (small shapes):
//     C = epi(A[M,K] @ W[N,K]^T + bias)
//
// Tile 256 (M) x BN (N, 256 or 128) x 64 (K), 512 threads = 8 waves (two per SIMD).  The K loop is ONE asm
// statement emitted by tools/gen_gemm_asm.py (gemm256_body_n*.inc): two LDS stages filled by LDS-DMA one K step
// ahead, fragment sets double-buffered in registers, the last k-sub-step of a K step issued AFTER the tile barrier
// so that the first LDS reads of the next stage and its DMA sit in MFMA shadows, accumulators in AGPRs.
// Operand convention, LDS image (128-byte rows, source-side XOR swizzle) and the fused epilogue (bias / GELU-tanh /
// gate*x+residual / f32 out; a lane owns one output row and 4 consecutive columns per quad) are those of
// gemm_bf16.hip, which remains the kernel for small or odd shapes (the dispatcher is in gemm_bf16.hip).
//
// FP8 instantiation (osk_gemm_fp8): A and W are OCP e4m3 bytes with one f32 scale per row (activation row m, weight
// row n); the same tile, LDS image (a 128-byte row now holds 128 K elements) and loaders, K loop from
// gemm256_fp8_body_n*.inc on v_mfma_f32_32x32x64_f8f6f4 (2x the bf16 MAC rate), epilogue v = acc * sa[m] * sw[n] first.
//
// Roofline: MFMA bf16 (fp8 instantiation: MFMA fp8).  Algorithmic FLOPs = 2*M*N*K.
#include "acc_quads.h"
#include "gemm_params.h"
#include "gemm256_regs_n256.inc"
#include "gemm256_regs_n128.inc"
#include "gemm256_fp8_regs_n256.inc"
#include "gemm256_fp8_regs_n128.inc"

namespace osk_gemm {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

// tile T's 16 accumulators = quads 4 T .. 4 T + 3 of aq (acc_quads.h: compiler-visible values, outputs of an empty asm statement
// behind the K loop): read in place and in program order
template <int T>
OSK_DEV void read_acc(const osk_v4f* aq, float* v16) {
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v16[i]) : "a"(aq[4 * T + i / 4][i % 4]));
}

// one 32 x 32 accumulator tile T = tn * TM + tm.  INTERIOR: the wave's whole tile lies inside C (wave-uniform), so
// there is no per-element bounds check and the column vectors (bias, gate) were loaded once per tn by the caller.
template <int BN, bool OUT_F32, bool FP8, bool INTERIOR, int T>
OSK_DEV void epilogue_tile(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, const float4* bq, const float4* gq,
                           const float4* sq) {
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int tn = T / TM, tm = T % TM;
  const int m = m0w + tm * 32 + l31;
  const int mc = m < p.M ? m : p.M - 1;
  const int b = mc / p.crpb, l = mc - b * p.crpb;
  const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
  if constexpr (INTERIOR) {
    uint2 rv[4];
    if (p.gate) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        rv[qd] = *reinterpret_cast<const uint2*>(p.res + roff + n0w + tn * 32 + qd * 8 + hi * 4);
    }
    float acc[16];
    read_acc<T>(aq, acc);
    float sa = 1.f;
    if constexpr (FP8) sa = p.sa[mc];
    uint2 packed[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = n0w + tn * 32 + qd * 8 + hi * 4;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[qd * 4 + j];
      if constexpr (FP8) { v[0] *= sa * sq[qd].x; v[1] *= sa * sq[qd].y; v[2] *= sa * sq[qd].z; v[3] *= sa * sq[qd].w; }
      if (p.bias) { v[0] += bq[qd].x; v[1] += bq[qd].y; v[2] += bq[qd].z; v[3] += bq[qd].w; }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j >= p.gelu_from) v[j] = gelu_tanh(v[j]);
      if (p.gate) {
        v[0] = bf16_lo(rv[qd].x) + gq[qd].x * v[0];
        v[1] = bf16_hi(rv[qd].x) + gq[qd].y * v[1];
        v[2] = bf16_lo(rv[qd].y) + gq[qd].z * v[2];
        v[3] = bf16_hi(rv[qd].y) + gq[qd].w * v[3];
      }
      if constexpr (OUT_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + roff + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        packed[qd].x = pack_bf16x2(v[0], v[1]);
        packed[qd].y = pack_bf16x2(v[2], v[3]);
      }
    }
    if constexpr (!OUT_F32) {
      // a lane holds columns [8 qd + 4 hi, +4) of its row; its partner lane (other half-wave, same row) the other 4 of
      // every 8-column block.  One v_permlane32_swap per dword hands the lower half-wave the whole block qd and the
      // upper one the whole block qd + 1: 16-byte stores instead of two 8-byte ones (guide T21).
      unsigned short* crow = reinterpret_cast<unsigned short*>(p.C) + roff + n0w + tn * 32;
      const bool wide = (((uintptr_t)crow) & 15) == 0;   // row base 16-byte aligned (c strides are multiples of 4 only)
#pragma unroll
      for (int qd = 0; qd < 4; qd += 2) {
        if (wide) {
          auto sx = __builtin_amdgcn_permlane32_swap(packed[qd].x, packed[qd + 1].x, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(packed[qd].y, packed[qd + 1].y, false, false);
          // lower half-wave: sx[0], sy[0] = own block-qd columns 0..3, sx[1], sy[1] = partner's columns 4..7;
          // upper half-wave: sx[0], sy[0] = partner's block-(qd+1) columns 0..3, sx[1], sy[1] = own columns 4..7
          *reinterpret_cast<uint4*>(crow + (qd + hi) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        } else {
          *reinterpret_cast<uint2*>(crow + qd * 8 + hi * 4) = packed[qd];
          *reinterpret_cast<uint2*>(crow + (qd + 1) * 8 + hi * 4) = packed[qd + 1];
        }
      }
    }
  } else {
    float acc[16];
    read_acc<T>(aq, acc);
    if (m >= p.M) return;
    const float* grow = p.gate ? p.gate + b * p.gbs : nullptr;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = n0w + tn * 32 + qd * 8 + hi * 4;
      for (int j = 0; j < 4 && n + j < p.N; ++j) {
        float t = acc[qd * 4 + j];
        if constexpr (FP8) t *= p.sa[m] * p.sw[n + j];
        t += p.bias ? p.bias[n + j] : 0.f;
        if (n + j >= p.gelu_from) t = gelu_tanh(t);
        if (grow) t = bf16_bits_to_f32(p.res[roff + n + j]) + grow[n + j] * t;
        if constexpr (OUT_F32) reinterpret_cast<float*>(p.C)[roff + n + j] = t;
        else reinterpret_cast<unsigned short*>(p.C)[roff + n + j] = f32_to_bf16_bits(t);
      }
    }
  }
}

template <int BN, bool OUT_F32, bool FP8, bool INTERIOR, int... Ts>
OSK_DEV void epilogue_tn(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi, std::integer_sequence<int, Ts...>) {
  // Ts = the TM tiles of one tn: column vectors once, then the row tiles
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int tn = ((Ts, ...)) / TM;  // all Ts share tn
  float4 bq[4], gq[4], sq[4];
  if constexpr (INTERIOR) {
    // the gate vector belongs to the batch of the tile's rows; a 256-row tile may straddle two batches only when
    // c_rows_per_batch is not a multiple of 256 -- then INTERIOR is refused by the caller
    const int b = m0w / p.crpb;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int n = n0w + tn * 32 + qd * 8 + hi * 4;
      if (p.bias) bq[qd] = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.gate) gq[qd] = *reinterpret_cast<const float4*>(p.gate + b * p.gbs + n);
      if constexpr (FP8) sq[qd] = *reinterpret_cast<const float4*>(p.sw + n);
    }
  }
  (epilogue_tile<BN, OUT_F32, FP8, INTERIOR, Ts>(aq, p, m0w, n0w, l31, hi, bq, gq, sq), ...);
}

template <int BN, bool OUT_F32, bool FP8, bool INTERIOR>
OSK_DEV void epilogue_all(const osk_v4f* aq, const GemmParams& p, int m0w, int n0w, int l31, int hi) {
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  if constexpr (TM == 4) {
    epilogue_tn<BN, OUT_F32, FP8, INTERIOR>(aq, p, m0w, n0w, l31, hi, std::integer_sequence<int, 0, 1, 2, 3>{});
    epilogue_tn<BN, OUT_F32, FP8, INTERIOR>(aq, p, m0w, n0w, l31, hi, std::integer_sequence<int, 4, 5, 6, 7>{});
  } else {
    epilogue_tn<BN, OUT_F32, FP8, INTERIOR>(aq, p, m0w, n0w, l31, hi, std::integer_sequence<int, 0, 1>{});
    epilogue_tn<BN, OUT_F32, FP8, INTERIOR>(aq, p, m0w, n0w, l31, hi, std::integer_sequence<int, 2, 3>{});
  }
}

template <int BN, bool OUT_F32, bool FP8>
__global__ void __launch_bounds__(512, 2) gemm256_kernel(const GemmParams p) {
  constexpr int ES = FP8 ? 1 : 2;              // bytes per operand element; a 128-byte LDS row = 128 / ES K elements
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int TN = BN == 256 ? OSKG256_TN : OSKG128_TN;
  constexpr int WN = BN / (TN * 32);           // waves along N (4 or 2); waves along M = 8 / WN
  constexpr int W_BASE = BN == 256 ? OSKG256_W_BASE : OSKG128_W_BASE;
  constexpr int NWD = BN / 64;                 // weight LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + 255) / 256, nbn = (p.N + BN - 1) / BN;
  // tile order: every XCD (private 4 MiB L2) owns a contiguous range of the list below; inside it the tiles run in
  // groups of GRP row bands, N-major within a group: the ~32 tiles resident on an XCD cover GRP bands x a few weight
  // tiles, so a weight tile is streamed once per GRP bands instead of once per ~2 (OSK_GEMM_GROUP, default 8)
  const int tile = xcd_remap(blockIdx.x, nbm * nbn);
  const int grp = p.group > 0 ? p.group : 1;
  const int per_group = grp * nbn;
  const int g = tile / per_group, r = tile - g * per_group;
  const int rows_here = nbm - g * grp < grp ? nbm - g * grp : grp;   // last group may be short
  const int bn = r / rows_here, bm = g * grp + (r - bn * rows_here);
  const int m0 = bm * 256, n0 = bn * BN;

  // ---- LDS-DMA sources: instruction j = wave + 8 i covers tile rows [8 j, 8 j + 8); byte offsets from the tensor base
  const int srow8 = lane >> 3, spos = lane & 7;
  unsigned aoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 8 * i) * 8 + srow8;
    const int c = spos ^ ((r >> 1) & 7);
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    const int b = m / p.arpb, l = m - b * p.arpb;
    aoff[i] = (unsigned)((b * p.abs_ + (int64_t)l * p.ars) * ES + c * 16);
    int n = n0 + (r < BN ? r : 0);
    n = n < p.N ? n : p.N - 1;
    woff[i] = (unsigned)((int64_t)n * p.wrs * ES + c * 16);
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int sw = (l31 >> 1) & 7;
  unsigned faA[4], faW[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned sz = (unsigned)((((ks << 1) | hi) ^ sw) << 4);
    faA[ks] = lds_base + (wm * TM * 32 + l31) * 128 + sz;
    faW[ks] = lds_base + W_BASE + (wn * TN * 32 + l31) * 128 + sz;
  }
  const uint64_t abase = rfl64((uint64_t)(uintptr_t)p.A), wbase = rfl64((uint64_t)(uintptr_t)p.W);
  const unsigned nk = rfl((unsigned)(p.K / (128 / ES)));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + W_BASE + wave * 1024);

#define OSKG_OPERANDS                                                                                              \
  ::"v"(faA[0]), "v"(faA[1]), "v"(faA[2]), "v"(faA[3]), "v"(faW[0]), "v"(faW[1]), "v"(faW[2]), "v"(faW[3]),         \
      "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), \
      "s"(abase), "s"(wbase), "s"(nk), "s"(adst), "s"(wdst)
  static_assert(FP8, "the bf16 instantiations of this one-tile-per-workgroup kernel were replaced by gemm256x.hip / gemm256p.hip");
  if constexpr (BN == 256) {
    asm volatile(
#include "gemm256_fp8_body_n256.inc"
        OSKG_OPERANDS : OSKQ256_CLOBBERS);
  } else {
    asm volatile(
#include "gemm256_fp8_body_n128.inc"
        OSKG_OPERANDS : OSKQ128_CLOBBERS);
  }
  (void)NWD;
  static_assert(OSKG256_ACC_QUADS == 32 && OSKG128_ACC_QUADS == 16, "the generated loops' accumulator map: tile t = quads 4 t .. 4 t + 3");
  osk_v4f aq[TM * TN * 4];
  if constexpr (BN == 256) asm volatile("" : OSK_AQ_OUT_0_32(aq));
  else asm volatile("" : OSK_AQ_OUT_0_16(aq));

  // ---- epilogue: lane owns row m = ... + l31, columns n = quad*8 + hi*4 + {0..3} of every 32 x 32 tile
  const int m0w = m0 + wm * TM * 32, n0w = n0 + wn * TN * 32;
  const int b_first = m0w / p.crpb, b_last = (m0w + TM * 32 - 1) / p.crpb;
  const bool interior = m0w + TM * 32 <= p.M && n0w + TN * 32 <= p.N && b_first == b_last;  // wave-uniform
  if (interior) epilogue_all<BN, OUT_F32, FP8, true>(aq, p, m0w, n0w, l31, hi);
  else epilogue_all<BN, OUT_F32, FP8, false>(aq, p, m0w, n0w, l31, hi);
}

template <int BN, bool OUT_F32, bool FP8>
int launch_one(const GemmParams& p, hipStream_t st) {
  constexpr int SMEM = BN == 256 ? OSKG256_SMEM : OSKG128_SMEM;
  auto kernel = gemm256_kernel<BN, OUT_F32, FP8>;
  OSK_ENSURE_MAX_SMEM(kernel, SMEM);
  const int nblk = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kernel, dim3(nblk), dim3(512), SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

// the 32-bit per-lane source offsets require both operand tensors to span < 4 GiB from their base pointers
bool gemm256_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems) {
  return p.K % 64 == 0 && p.M >= 256 && p.N >= 128 && a_span_elems * 2 < (int64_t)0xFFFFFFFF &&
         w_span_elems * 2 < (int64_t)0xFFFFFFFF;
}

// fp8 operands (1 byte per element: spans in bytes), K % 128 == 0, per-row scales p.sa / p.sw
bool gemm256_fp8_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems) {
  return p.K % 128 == 0 && p.M >= 256 && p.N >= 128 && a_span_elems < (int64_t)0xFFFFFFFF &&
         w_span_elems < (int64_t)0xFFFFFFFF;
}

int launch_gemm256_fp8(const GemmParams& p, int bn, int out_f32, hipStream_t st) {
  if (bn == 256) return out_f32 ? launch_one<256, true, true>(p, st) : launch_one<256, false, true>(p, st);
  return out_f32 ? launch_one<128, true, true>(p, st) : launch_one<128, false, true>(p, st);
}

}  // namespace osk_gemm
