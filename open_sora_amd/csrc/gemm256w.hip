// 4-wave form of the persistent large-tile bf16 GEMM:  C = epi(A[M,K] @ W[N,K]^T + bias)
//
// The same 256 x 256 x 64 tile, LDS image, LDS-DMA loaders, tile walk and fused epilogue as gemm256p.hip, but the
// workgroup is 256 threads = ONE wave per SIMD, each owning the whole 512-entry register file: wave tile 128 x 128 =
// 4 x 4 MFMA tiles in 256 accumulator AGPRs (2 x 2 waves).  Why: on this board a long MFMA kernel runs at the power cap
// (profiles/r02_ab_vendor.json: this library's 8-wave kernel and hipBLASLt's 4-wave MT256x256x64 kernel both sit at
// 1.37 GHz on 8192^3, the vendor kernel doing 25 % more flops at that clock), so time ~ ENERGY per flop, not stall
// cycles -- removing idle time (gemm256p.hip's cross-tile prefetch) barely moves the denoise step.  What the 8-wave
// layout spends beyond the MFMAs is LDS traffic: a 128 x 64 wave tile reads 6 fragments per 8 MFMAs, a 128 x 128 one
// 8 per 16 -- a third less LDS read energy per flop, half the waves per barrier, half the scalar bookkeeping.
// K loop: tools/gen_gemm_asm.py::gen_w4.  What round 2's counters say bounds it (profiles/r02_gemm_experiments.md): an LDS-DMA
// instruction holds the wave's issue port far longer than an MFMA shadow (the guide: 60-185 cycles), so 16 of them in 16
// consecutive shadows idle the matrix pipe behind each one; ONE PIECE PER TWO SHADOWS over the trailing sub-step and
// sub-step 0 (gemm256w_body_spread2.inc) gives +4..10 % -- the vendor kernel gets the same spacing from a
// deeper pipeline (three barriers per K step).
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2*M*N*K.
#include "gemm_epilogue.h"
#include "gemm256w_regs.inc"

namespace osk_gemm {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

#define OSKW_OUT16                                                                                               \
  "=v"(v16[0]), "=v"(v16[1]), "=v"(v16[2]), "=v"(v16[3]), "=v"(v16[4]), "=v"(v16[5]), "=v"(v16[6]), "=v"(v16[7]),    \
      "=v"(v16[8]), "=v"(v16[9]), "=v"(v16[10]), "=v"(v16[11]), "=v"(v16[12]), "=v"(v16[13]), "=v"(v16[14]),          \
      "=v"(v16[15])

struct GeoW {
  static constexpr int TM = OSKW_TM, TN = OSKW_TN;
  template <int T>
  OSK_DEV void read(float* v16) {
    if constexpr (T == 0) asm volatile(OSKW_AR0 : OSKW_OUT16);
    else if constexpr (T == 1) asm volatile(OSKW_AR1 : OSKW_OUT16);
    else if constexpr (T == 2) asm volatile(OSKW_AR2 : OSKW_OUT16);
    else if constexpr (T == 3) asm volatile(OSKW_AR3 : OSKW_OUT16);
    else if constexpr (T == 4) asm volatile(OSKW_AR4 : OSKW_OUT16);
    else if constexpr (T == 5) asm volatile(OSKW_AR5 : OSKW_OUT16);
    else if constexpr (T == 6) asm volatile(OSKW_AR6 : OSKW_OUT16);
    else if constexpr (T == 7) asm volatile(OSKW_AR7 : OSKW_OUT16);
    else if constexpr (T == 8) asm volatile(OSKW_AR8 : OSKW_OUT16);
    else if constexpr (T == 9) asm volatile(OSKW_AR9 : OSKW_OUT16);
    else if constexpr (T == 10) asm volatile(OSKW_AR10 : OSKW_OUT16);
    else if constexpr (T == 11) asm volatile(OSKW_AR11 : OSKW_OUT16);
    else if constexpr (T == 12) asm volatile(OSKW_AR12 : OSKW_OUT16);
    else if constexpr (T == 13) asm volatile(OSKW_AR13 : OSKW_OUT16);
    else if constexpr (T == 14) asm volatile(OSKW_AR14 : OSKW_OUT16);
    else asm volatile(OSKW_AR15 : OSKW_OUT16);
  }
};

template <bool OUT_F32>
__global__ void __launch_bounds__(256, 1) gemm256w_kernel(const GemmParams p) {
  constexpr int TM = OSKW_TM, TN = OSKW_TN, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + 255) / 256, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int grp = p.group > 0 ? p.group : 1;
  const int per_group = grp * nbn;
  auto tile_of = [&](int it, int& m0, int& n0) {      // tile order of gemm256.hip / gemm256p.hip
    const int tile = xcd_remap(it, ntiles);
    const int g = tile / per_group, r = tile - g * per_group;
    const int rows_here = nbm - g * grp < grp ? nbm - g * grp : grp;
    const int bn = r / rows_here, bm = g * grp + (r - bn * rows_here);
    m0 = bm * 256;
    n0 = bn * BN;
  };
  // LDS-DMA sources: instruction j = wave + 4 i (i = 0..7) covers tile rows [8 j, 8 j + 8); byte offsets from the bases
  const int srow8 = lane >> 3, spos = lane & 7;
  auto offsets = [&](int m0, int n0, unsigned* aoff, unsigned* woff) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (wave + 4 * i) * 8 + srow8;
      const int c = spos ^ ((r >> 1) & 7);
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const int b = m / p.arpb, l = m - b * p.arpb;
      aoff[i] = (unsigned)((b * p.abs_ + (int64_t)l * p.ars) * 2 + c * 16);
      int n = n0 + r;
      n = n < p.N ? n : p.N - 1;
      woff[i] = (unsigned)((int64_t)n * p.wrs * 2 + c * 16);
    }
  };
  // a tile whose 256 A rows lie inside M and inside one batch, and whose 256 W rows lie inside N: its per-lane source
  // offsets are an affine function of (m0, n0), so the next tile's are this tile's plus a wave-uniform delta
  auto affine = [&](int m0, int n0) {
    return m0 + 256 <= p.M && n0 + 256 <= p.N && m0 / p.arpb == (m0 + 255) / p.arpb;
  };
  auto a_origin = [&](int m0) -> int64_t {
    const int b = m0 / p.arpb, l = m0 - b * p.arpb;
    return (b * p.abs_ + (int64_t)l * p.ars) * 2;
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned sz0 = (unsigned)((hi ^ ((l31 >> 1) & 7)) << 4);
  const unsigned faA0 = lds_base + (wm * TM * 32 + l31) * 128 + sz0;
  const unsigned faW0 = lds_base + OSKW_W_BASE + (wn * TN * 32 + l31) * 128 + sz0;
  const uint64_t abase = rfl64((uint64_t)(uintptr_t)p.A), wbase = rfl64((uint64_t)(uintptr_t)p.W);
  const uint64_t bbase = rfl64((uint64_t)(uintptr_t)p.bias);
  const unsigned nk = rfl((unsigned)(p.K / 64));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + OSKW_W_BASE + wave * 1024);

  unsigned prefetched = 0;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
    const int itn = it + (int)gridDim.x;
    int m0, n0, m0n = 0, n0n = 0;
    tile_of(it, m0, n0);
    bool has_next = itn < ntiles;
    unsigned dA = 0, dW = 0;
    if (has_next) {
      tile_of(itn, m0n, n0n);
      // cross-tile prefetch only between two affine tiles (edge tiles start with their own cold fetch)
      has_next = affine(m0, n0) && affine(m0n, n0n);
      dA = (unsigned)(a_origin(m0n) - a_origin(m0));
      dW = (unsigned)(((int64_t)n0n - n0) * p.wrs * 2);
    }
    unsigned aoff[8], woff[8];
    offsets(m0, n0, aoff, woff);
    const int m0w = m0 + wm * TM * 32, n0w = n0 + wn * TN * 32;
    const bool folded = p.bias != nullptr && n0w + TN * 32 <= p.N;                 // wave-uniform
    const unsigned boff = (unsigned)((n0w + hi * 4) * 4);
    const unsigned flags = rfl(prefetched | (has_next ? 2u : 0u) | (folded ? 4u : 0u));
    const unsigned dAs = rfl(dA), dWs = rfl(dW);

    // prefetch lanes: lane l of wave w touches row 64 w + l of the A tile and of the W tile (one dword per 128-byte line)
    unsigned aoffp, woffp;
    {
      int m = m0 + wave * 64 + lane;
      m = m < p.M ? m : p.M - 1;
      const int b = m / p.arpb, l = m - b * p.arpb;
      aoffp = (unsigned)((b * p.abs_ + (int64_t)l * p.ars) * 2);
      int n = n0 + wave * 64 + lane;
      n = n < p.N ? n : p.N - 1;
      woffp = (unsigned)((int64_t)n * p.wrs * 2);
    }
#define OSKW_OPERANDS                                                                                               \
  ::"v"(faA0), "v"(faW0), "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(aoff[4]), "v"(aoff[5]),          \
      "v"(aoff[6]), "v"(aoff[7]), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(woff[4]), "v"(woff[5]),  \
      "v"(woff[6]), "v"(woff[7]), "v"(boff), "s"(abase), "s"(wbase), "s"(bbase), "s"(nk), "s"(adst), "s"(wdst),        \
      "s"(flags), "s"(dAs), "s"(dWs), "v"(aoffp), "v"(woffp)
    asm volatile(
#include "gemm256w_body_spread2.inc"
        OSKW_OPERANDS : OSKW_CLOBBERS);

    const int b_first = m0w / p.crpb, b_last = (m0w + TM * 32 - 1) / p.crpb;
    const bool interior = m0w + TM * 32 <= p.M && n0w + TN * 32 <= p.N && b_first == b_last;  // wave-uniform
    epi::epilogue_all<GeoW, OUT_F32>(p, m0w, n0w, l31, hi, interior, folded);
    prefetched = has_next ? 1u : 0u;
  }
}

template <bool OUT_F32>
int launch_one(const GemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  static int n_cu = 0;
  auto kernel = gemm256w_kernel<OUT_F32>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, OSKW_SMEM);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
    if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
    n_cu -= n_cu % 8;    // the tile walk keeps a workgroup inside one XCD's range only for a grid that is a multiple of 8
    if (n_cu < 8) n_cu = 8;
    attr_set = true;
  }
  const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int grid = ntiles < n_cu ? ntiles : n_cu;   // one workgroup per CU (LDS: 128 KiB of 160)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), OSKW_SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

int launch_gemm256w(const GemmParams& p, int out_f32, hipStream_t st) {
  // one generated K loop: LDS-DMA one piece per 2 MFMA shadows.  The other schedules of round 2 (one per shadow -4..10 %, one per 3
  // shadows, buffer_load ... lds, L2 software prefetch -6 %) are in profiles/r02_gemm_experiments.md and tools/gen_gemm_asm.py.
  return out_f32 ? launch_one<true>(p, st) : launch_one<false>(p, st);
}

}  // namespace osk_gemm
