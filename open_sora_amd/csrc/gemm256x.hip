// The 4-wave persistent bf16 GEMM (round 2's gemm256w.hip, removed in round 3) on v_mfma_f32_16x16x32_bf16:  C = epi(A[M,K] @ W[N,K]^T + bias)
//
// Same 256 x 256 x 64 workgroup tile, LDS image, LDS-DMA loaders, tile walk, cross-tile prefetch and bias-initialised
// accumulators; the wave tile 128 x 128 is 8 x 8 accumulator tiles of 16 x 16 (4 AGPRs each) and a K step is two sub-steps of
// K = 32.  Why a second MFMA shape (profiles/r02_gemm_experiments.md, "what the K loop's time is made of"): every kernel of
// this library runs at the board's 1.4 kW cap, so time ~ energy per flop; with that kernel's loop otherwise unchanged, issuing the
// same flops as 16x16x32 MFMAs (4 accumulator registers written per 16 matrix cycles instead of 16 per 32) measured +5-7 %.
// A fragment is 16 rows x 32 k: lane l reads row l % 16, 16-byte chunk (l / 16) + 4 s of the swizzled 128-byte LDS row --
// conflict-free for ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ...: the 8 chunk ^ key values of a group are distinct).
// Operands are swapped (first = weight fragment): a lane owns output row l % 16 and channels 4 (l / 16) .. + 3 of each tile.
// K loop: tools/gen_gemm_asm.py::gen_x4; epilogue: gemm_epilogue16.h.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2*M*N*K.
#include "acc_quads.h"
#include "gemm_epilogue16.h"
#include "gemm256x_regs.inc"

namespace osk_gemm {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

// The wave's 256 accumulators as 64 quads (tile T = J * NB + I is quad T): made compiler-visible values by an empty asm
// statement behind the K-loop statement (acc_quads.h) -- the epilogue reads aq[T][i], the compiler emits the v_accvgpr_read.
struct GeoX {
  static constexpr int NB = OSKX_NB;
  template <int T>
  OSK_DEV void read(const osk_v4f* aq, float* v4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v4[i]) : "a"(aq[T][i]));   // (in place, in program order)
  }
};

// NP > 1: several problems that share K walked as ONE tile list -- problem 0's tiles, then problem 1's, ... -- so that a small
// problem (the text-stream Linear of a double block: 6 row tiles, a third of the chip for one round) fills the last round of a
// large one instead of launching alone.  Round 6: the problems may differ in N, M, epilogue class, skip range and operand roles
// (V^T tasks: gemm_params.h) -- everything per-tile is read from the tile's own GemmParams.
template <int NP>
struct GemmPack {
  GemmParams p[NP];
};

#ifdef OSK_GEMM_TILE_TIMING   // tools/make_gemm_timing_lib.sh: where a tile's time goes (s_memtime sums of wave 0 of every workgroup)
__device__ unsigned long long osk_gemm_tile_ticks[4];   // address set-up, asm statement (cold start + K loop), epilogue, tiles
#define OSK_TT(i, t0) if (threadIdx.x == 0) atomicAdd(&osk_gemm_tile_ticks[i], __builtin_amdgcn_s_memtime() - (t0))
#else
#define OSK_TT(i, t0)
#endif

template <bool OUT_F32, int NP>
__global__ void __launch_bounds__(256, 1) gemm256x_kernel(const GemmPack<NP> pk) {
  constexpr int WT = OSKX_NB * 16, BN = 256;   // wave tile side
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q4 = lane >> 4, l15 = lane & 15;

  const int nbn = (pk.p[0].N - pk.p[0].skip_len + BN - 1) / BN;     // N, K, group, skip range are the pack's (equal in every problem)
  const int nt0 = ((pk.p[0].M + 255) / 256) * nbn;
  const int ntiles = NP == 1 ? nt0 : nt0 + ((pk.p[NP - 1].M + 255) / 256) * nbn;
  const int grp = pk.p[0].group > 0 ? pk.p[0].group : 1;
  const int per_group = grp * nbn;
  // position in the tile list -> (problem, tile origin): tile order of gemm256.hip / gemm256p.hip inside each problem
  auto tile_of = [&](int it, int& sel, int& m0, int& n0) {
    int tile = xcd_remap(it, ntiles);
    sel = (NP > 1 && tile >= nt0) ? 1 : 0;
    tile -= sel ? nt0 : 0;
    const int nbm = (pk.p[sel].M + 255) / 256;
    const int g = tile / per_group, r = tile - g * per_group;
    const int rows_here = nbm - g * grp < grp ? nbm - g * grp : grp;
    const int bn = r / rows_here, bm = g * grp + (r - bn * rows_here);
    m0 = bm * 256;
    n0 = bn * BN;
    n0 += n0 >= pk.p[0].skip_from ? pk.p[0].skip_len : 0;      // PHYSICAL column origin (round 6: a skipped column range, gemm_params.h)
  };
  // LDS-DMA sources: instruction j = wave + 4 i (i = 0..7) covers tile rows [8 j, 8 j + 8); byte offsets from the bases
  const int srow8 = lane >> 3, spos = lane & 7;
  auto offsets = [&](const GemmParams& p, int m0, int n0, unsigned* aoff, unsigned* woff) {
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
  auto affine = [&](const GemmParams& p, int m0, int n0) {
    return m0 + 256 <= p.M && n0 + 256 <= p.N && m0 / p.arpb == (m0 + 255) / p.arpb;
  };
  auto a_origin = [&](const GemmParams& p, int m0) -> int64_t {
    const int b = m0 / p.arpb, l = m0 - b * p.arpb;
    return (b * p.abs_ + (int64_t)l * p.ars) * 2;
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // fragment row l15 of a 16-row block, 16-byte chunk q4 (k 8 q4 .. + 7 of the sub-step's 32) under the row's swizzle key
  const unsigned sz0 = (unsigned)((q4 ^ ((l15 >> 1) & 7)) << 4);
  const unsigned faA0 = lds_base + (wm * WT + l15) * 128 + sz0;
  const unsigned faW0 = lds_base + OSKX_W_BASE + (wn * WT + l15) * 128 + sz0;
  const unsigned nk = rfl((unsigned)(pk.p[0].K / 64));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + OSKX_W_BASE + wave * 1024);

  unsigned prefetched = 0;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
    const int itn = it + (int)gridDim.x;
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
#endif
    int sel, m0, n0, seln = 0, m0n = 0, n0n = 0;
    tile_of(it, sel, m0, n0);
    const GemmParams& p = pk.p[NP == 1 ? 0 : sel];                 // wave-uniform: kernel-argument loads at a scalar offset
    bool has_next = itn < ntiles;
    unsigned dA = 0, dW = 0;
    if (has_next) {
      tile_of(itn, seln, m0n, n0n);
      // cross-tile prefetch only between two affine tiles of the SAME problem (the deltas are relative to its bases; edge
      // tiles and the first tile of the second problem start with their own cold fetch)
      has_next = seln == sel && affine(p, m0, n0) && affine(p, m0n, n0n);
      dA = (unsigned)(a_origin(p, m0n) - a_origin(p, m0));
      dW = (unsigned)(((int64_t)n0n - n0) * p.wrs * 2);
    }
    const uint64_t abase = rfl64((uint64_t)(uintptr_t)p.A), wbase = rfl64((uint64_t)(uintptr_t)p.W);
    const uint64_t bbase = rfl64((uint64_t)(uintptr_t)p.bias);
    unsigned aoff[8], woff[8];
    offsets(p, m0, n0, aoff, woff);
    const int m0w = m0 + wm * WT, n0w = n0 + wn * WT;
    const bool folded = p.bias != nullptr && n0w + WT <= p.N;                 // wave-uniform
    const unsigned boff = (unsigned)((n0w + q4 * 4) * 4);
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
    OSK_TT(0, tt0);
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile(
#include "gemm256x_body.inc"
        OSKW_OPERANDS : OSKX_CLOBBERS);
    static_assert(OSKX_ACC_QUADS == 64, "the generated loop's accumulator map: quad T = tile T, a0 .. a255");
    osk_v4f aq[64];
    asm volatile("" : OSK_AQ_OUT_0_64(aq));
    OSK_TT(1, tt1);
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
#endif

    const int b_first = m0w / p.crpb, b_last = (m0w + WT - 1) / p.crpb;
    const bool interior = m0w + WT <= p.M && n0w + WT <= p.N && b_first == b_last;  // wave-uniform
    epi16::epilogue_all<GeoX, OUT_F32>(aq, p, m0w, n0w, l15, q4, interior, folded);
    OSK_TT(2, tt2);
#ifdef OSK_GEMM_TILE_TIMING
    if (threadIdx.x == 0) atomicAdd(&osk_gemm_tile_ticks[3], 1ull);
#endif
    prefetched = has_next ? 1u : 0u;
  }
}

#undef OSKW_OPERANDS

// ---- the same tile loop for the V^T tasks of osk_gemm_group_bf16 (round 6): one or two problems (img + txt stream of a double
// block) whose A operand is the V weight and whose W operand are the activations -- V^T = W_v X^T written directly in the attention
// kernels' key-major operand layout (gemm_params.h: vt, wrpb, wvalid, ccbs, rowbias; epilogue class vt_all).  A separate kernel so
// that the Linear launches above keep their register allocation (one wave per SIMD, 256 accumulators + ~250 VGPRs: a handful more
// live values spill to scratch -- measured in this round: +3 k cycles of set-up per tile) and because this one carries ONE epilogue
// class instead of nine.
template <int NP>
__global__ void __launch_bounds__(256, 1) gemm256x_vt_kernel(const GemmPack<NP> pk) {
  static_assert(NP <= 2, "a run-time index into a wider pack makes hipcc copy the pack to scratch");
  constexpr int WT = OSKX_NB * 16, BN = 256;   // wave tile side
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int q4 = lane >> 4, l15 = lane & 15;

  // the two problems differ in N (positions on the key axis) and may differ in M
  const int nbn0 = (pk.p[0].N + BN - 1) / BN, nbn1 = (pk.p[NP - 1].N + BN - 1) / BN;
  const int nt0 = ((pk.p[0].M + 255) / 256) * nbn0;
  const int ntiles = NP == 1 ? nt0 : nt0 + ((pk.p[NP - 1].M + 255) / 256) * nbn1;
  const int grp = pk.p[0].group > 0 ? pk.p[0].group : 1;
  // position in the tile list -> (problem, tile origin): tile order of gemm256.hip / gemm256p.hip inside each problem
  auto tile_of = [&](int it, int& sel, int& m0, int& n0) {
    int tile = xcd_remap(it, ntiles);
    sel = (NP > 1 && tile >= nt0) ? 1 : 0;
    tile -= sel ? nt0 : 0;
    const int nbm = (pk.p[sel].M + 255) / 256;
    const int per_group = grp * (sel ? nbn1 : nbn0);
    const int g = tile / per_group, r = tile - g * per_group;
    const int rows_here = nbm - g * grp < grp ? nbm - g * grp : grp;
    const int bn = r / rows_here, bm = g * grp + (r - bn * rows_here);
    m0 = bm * 256;
    n0 = bn * BN;
  };
  // LDS-DMA sources: instruction j = wave + 4 i (i = 0..7) covers tile rows [8 j, 8 j + 8); byte offsets from the bases
  const int srow8 = lane >> 3, spos = lane & 7;
  // A = the V weight (one "batch"); W = the activations: column n of the product = (batch n / wrpb, position n % wrpb) of the key axis,
  // fed from the activation row key = vt_perm64(position) (osk_v_transpose_bf16's order inside every 64-key group); keys behind the
  // sequence end read the last key (finite values) and are stored as zero by the epilogue
  auto w_src = [&](const GemmParams& p, int n, bool permute) -> int64_t {
    n = n < p.N ? n : p.N - 1;
    const int wb = n / p.wrpb;
    int pos = n - wb * p.wrpb;
    if (permute) pos = vt_perm64(pos, p.vt);
    pos = pos < p.wvalid ? pos : p.wvalid - 1;
    return wb * p.wbs + (int64_t)pos * p.wrs;
  };
  auto offsets = [&](const GemmParams& p, int m0, int n0, unsigned* aoff, unsigned* woff) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (wave + 4 * i) * 8 + srow8;
      const int c = spos ^ ((r >> 1) & 7);
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      aoff[i] = (unsigned)((int64_t)m * p.ars * 2 + c * 16);
      woff[i] = (unsigned)(w_src(p, n0 + r, true) * 2 + c * 16);
    }
  };
  // a tile whose 256 weight rows lie inside M and whose 256 positions lie inside one batch with every key valid (the key order stays
  // inside 64-key groups): its per-lane source offsets are an affine function of (m0, n0)
  auto affine = [&](const GemmParams& p, int m0, int n0) {
    const int wb = n0 / p.wrpb, pos0 = n0 - wb * p.wrpb;
    return m0 + 256 <= p.M && n0 + 256 <= p.N && pos0 + 256 <= p.wrpb && pos0 + 256 <= p.wvalid;
  };
  auto a_origin = [&](const GemmParams& p, int m0) -> int64_t { return (int64_t)m0 * p.ars * 2; };
  auto w_origin = [&](const GemmParams& p, int n0) -> int64_t {
    const int wb = n0 / p.wrpb, pos0 = n0 - wb * p.wrpb;
    return (wb * p.wbs + (int64_t)pos0 * p.wrs) * 2;
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // fragment row l15 of a 16-row block, 16-byte chunk q4 (k 8 q4 .. + 7 of the sub-step's 32) under the row's swizzle key
  const unsigned sz0 = (unsigned)((q4 ^ ((l15 >> 1) & 7)) << 4);
  const unsigned faA0 = lds_base + (wm * WT + l15) * 128 + sz0;
  const unsigned faW0 = lds_base + OSKX_W_BASE + (wn * WT + l15) * 128 + sz0;
  const unsigned nk = rfl((unsigned)(pk.p[0].K / 64));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + OSKX_W_BASE + wave * 1024);

  unsigned prefetched = 0;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
    const int itn = it + (int)gridDim.x;
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
#endif
    int sel, m0, n0, seln = 0, m0n = 0, n0n = 0;
    tile_of(it, sel, m0, n0);
    const GemmParams& p = pk.p[NP == 1 ? 0 : sel];                 // wave-uniform: kernel-argument loads at a scalar offset
    bool has_next = itn < ntiles;
    unsigned dA = 0, dW = 0;
    if (has_next) {
      tile_of(itn, seln, m0n, n0n);
      // cross-tile prefetch only between two affine tiles of the SAME problem (the deltas are relative to its bases; edge
      // tiles and the first tile of the second problem start with their own cold fetch)
      has_next = seln == sel && affine(p, m0, n0) && affine(p, m0n, n0n);
      dA = (unsigned)(a_origin(p, m0n) - a_origin(p, m0));
      dW = (unsigned)(w_origin(p, n0n) - w_origin(p, n0));
    }
    const uint64_t abase = rfl64((uint64_t)(uintptr_t)p.A), wbase = rfl64((uint64_t)(uintptr_t)p.W);
    const uint64_t bbase = rfl64((uint64_t)(uintptr_t)p.bias);
    unsigned aoff[8], woff[8];
    offsets(p, m0, n0, aoff, woff);
    const int m0w = m0 + wm * WT, n0w = n0 + wn * WT;
    const bool folded = false;                                                // (the per-ROW bias b_v is added by the epilogue)
    const unsigned boff = (unsigned)((n0w + q4 * 4) * 4);
    const unsigned flags = rfl(prefetched | (has_next ? 2u : 0u) | (folded ? 4u : 0u));
    const unsigned dAs = rfl(dA), dWs = rfl(dW);

    // prefetch lanes: lane l of wave w touches row 64 w + l of the A tile and of the W tile (one dword per 128-byte line)
    unsigned aoffp, woffp;
    {
      int m = m0 + wave * 64 + lane;
      m = m < p.M ? m : p.M - 1;
      aoffp = (unsigned)((int64_t)m * p.ars * 2);
      woffp = (unsigned)(w_src(p, n0 + wave * 64 + lane, false) * 2);   // (one dword per line of the tile's rows: the order inside a group does not matter)
    }
#define OSKW_OPERANDS                                                                                               \
  ::"v"(faA0), "v"(faW0), "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(aoff[4]), "v"(aoff[5]),          \
      "v"(aoff[6]), "v"(aoff[7]), "v"(woff[0]), "v"(woff[1]), "v"(woff[2]), "v"(woff[3]), "v"(woff[4]), "v"(woff[5]),  \
      "v"(woff[6]), "v"(woff[7]), "v"(boff), "s"(abase), "s"(wbase), "s"(bbase), "s"(nk), "s"(adst), "s"(wdst),        \
      "s"(flags), "s"(dAs), "s"(dWs), "v"(aoffp), "v"(woffp)
    OSK_TT(0, tt0);
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile(
#include "gemm256x_body.inc"
        OSKW_OPERANDS : OSKX_CLOBBERS);
    static_assert(OSKX_ACC_QUADS == 64, "the generated loop's accumulator map: quad T = tile T, a0 .. a255");
    osk_v4f aq[64];
    asm volatile("" : OSK_AQ_OUT_0_64(aq));
    OSK_TT(1, tt1);
#ifdef OSK_GEMM_TILE_TIMING
    const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
#endif

    epi16::vt_all<GeoX>(aq, p, m0w, n0w, l15, q4);
    OSK_TT(2, tt2);
#ifdef OSK_GEMM_TILE_TIMING
    if (threadIdx.x == 0) atomicAdd(&osk_gemm_tile_ticks[3], 1ull);
#endif
    prefetched = has_next ? 1u : 0u;
  }
}

#undef OSKW_OPERANDS

int grid_for(int ntiles) {
  int n_cu = osk_device_cus();
  n_cu -= n_cu % 8;    // the tile walk keeps a workgroup inside one XCD's range only for a grid that is a multiple of 8
  if (n_cu < 8) n_cu = 8;
  return ntiles < n_cu ? ntiles : n_cu;   // one workgroup per CU (LDS: 128 KiB of 160)
}

template <bool OUT_F32>
int launch_one(const GemmParams& p, hipStream_t st) {
  auto kernel = gemm256x_kernel<OUT_F32, 1>;
  OSK_ENSURE_MAX_SMEM(kernel, OSKX_SMEM);
  GemmPack<1> pk;
  pk.p[0] = p;
  const int ntiles = ((p.M + 255) / 256) * ((p.N - p.skip_len + 255) / 256);
  hipLaunchKernelGGL(kernel, dim3(grid_for(ntiles)), dim3(256), OSKX_SMEM, st, pk);
  return (int)hipGetLastError();
}

}  // namespace

#ifdef OSK_GEMM_TILE_TIMING
// read and clear the tick sums (tools/gemm_tile_timing.py)
extern "C" int osk_gemm_tile_timing_read(unsigned long long* out4) {
  unsigned long long zero[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(osk_gemm_tile_ticks), sizeof(zero)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(osk_gemm_tile_ticks), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

int launch_gemm256x(const GemmParams& p, int out_f32, hipStream_t st) {
  return out_f32 ? launch_one<true>(p, st) : launch_one<false>(p, st);
}

static int tiles_of(const GemmParams& p) { return ((p.M + 255) / 256) * ((p.N - p.skip_len + 255) / 256); }

// two problems with equal K in one launch (bf16 output)
int launch_gemm256x_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t st) {
  auto kernel = gemm256x_kernel<false, 2>;
  OSK_ENSURE_MAX_SMEM(kernel, OSKX_SMEM);
  GemmPack<2> pk;
  pk.p[0] = p0;
  pk.p[1] = p1;
  hipLaunchKernelGGL(kernel, dim3(grid_for(tiles_of(p0) + tiles_of(p1))), dim3(256), OSKX_SMEM, st, pk);
  return (int)hipGetLastError();
}

// one or two V^T problems with equal K in one launch
int launch_gemm256x_vt(const GemmParams* ps, int n, hipStream_t st) {
  if (n < 1 || n > 2) return OSK_EINVAL;
  auto kernel = gemm256x_vt_kernel<2>;
  OSK_ENSURE_MAX_SMEM(kernel, OSKX_SMEM);
  GemmPack<2> pk;
  int ntiles = 0;
  for (int i = 0; i < 2; ++i) {
    pk.p[i] = ps[i < n ? i : 0];
    if (i >= n) pk.p[i].M = 0;          // (an empty second slot: no tiles)
    ntiles += tiles_of(pk.p[i]);
  }
  hipLaunchKernelGGL(kernel, dim3(grid_for(ntiles)), dim3(256), OSKX_SMEM, st, pk);
  return (int)hipGetLastError();
}

}  // namespace osk_gemm
