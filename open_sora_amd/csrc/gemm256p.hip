// Persistent-workgroup form of the large-tile bf16 GEMM (gemm256.hip):  C = epi(A[M,K] @ W[N,K]^T + bias)
//
// One 512-thread workgroup per CU walks a list of 256 x BN output tiles (tile i, i + grid, i + 2 grid, ... of the
// XCD-aware grouped order of gemm256.hip).  Same tile, LDS image, LDS-DMA loaders, fragment layout and K-step schedule;
// what changes is everything AROUND the K loop -- measured in round 1 at ~11 us per round of tiles against 1.7 us per
// K step, i.e. 27 % of a K = 1152 GEMM (every CU issuing its 64 KiB prologue fetch and its 128 KiB store burst at the
// same moment, with the matrix pipe idle):
//   * the next tile's first two K steps are fetched by LDS-DMA at the END of this tile's K loop (both LDS stages are
//     free once the last fragments sit in registers) and land while the epilogue runs: a tile never starts with a
//     cold fetch except the first one of the workgroup (tools/gen_gemm_asm.py::gen_pers);
//   * the accumulators start from the bias, so the epilogue of an un-gated Linear has no load to wait for and no
//     add; GELU is decided per 32-column tile (wave-uniform), not per element, and uses v_exp_f32 + v_rcp_f32
//     (the old epilogue spent ~30 instructions per element on exec-mask branches and an IEEE division sequence);
//   * CUs drift apart instead of marching in lock-step, so the store bursts spread out.
// The vendor kernel this was measured against (profiles/r02_ab_vendor.json): hipBLASLt MT256x256x64, +7..16 % over the
// round-1 kernel at the K = 1152 shapes.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2*M*N*K.
#include "gemm_epilogue.h"
#include "gemm256_regs_n256.inc"
#include "gemm256_regs_n128.inc"
#include "gemm256p_regs_n128.inc"

namespace osk_gemm {
namespace {

OSK_DEV unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
OSK_DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((unsigned)(v >> 32)) << 32) | rfl((unsigned)v); }

// tile T's 16 accumulators = quads 4 T .. 4 T + 3 of aq (acc_quads.h): read in place and in program order
template <int T>
OSK_DEV void read_acc(const osk_v4f* aq, float* v16) {
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v16[i]) : "a"(aq[4 * T + i / 4][i % 4]));
}

template <int BN>
struct Geo {
  static constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  static constexpr int TN = BN == 256 ? OSKG256_TN : OSKG128_TN;
  template <int T>
  OSK_DEV void read(const osk_v4f* aq, float* v16) { read_acc<T>(aq, v16); }
};

// SCHED: K-step schedule of the generated body (tools/gen_gemm_asm.py::gen_pers): 0 = the round-1 order (fragment reads
// of a new stage issued in a block right behind the barrier, one prefetch read per MFMA shadow), 1 = matrix pipe first
// (the trailing sub-step starts behind the barrier, reads two per shadow at the head of every sub-step)
template <int BN, bool OUT_F32, int SCHED>
__global__ void __launch_bounds__(512, 2) gemm256p_kernel(const GemmParams p) {
  constexpr int TM = BN == 256 ? OSKG256_TM : OSKG128_TM;
  constexpr int TN = BN == 256 ? OSKG256_TN : OSKG128_TN;
  constexpr int WN = BN / (TN * 32);           // waves along N (4 or 2); waves along M = 8 / WN
  constexpr int W_BASE = BN == 256 ? OSKG256_W_BASE : OSKG128_W_BASE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + 255) / 256, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int grp = p.group > 0 ? p.group : 1;
  const int per_group = grp * nbn;
  // tile order of gemm256.hip: every XCD owns a contiguous range of the list; inside it groups of `grp` row bands,
  // N-major within a group.  Iteration i of this workgroup is list position blockIdx.x + i * gridDim.x (gridDim.x is a
  // multiple of 8 whenever a workgroup has more than one tile, so a workgroup stays inside its XCD's range).
  auto tile_of = [&](int it, int& m0, int& n0) {
    const int tile = xcd_remap(it, ntiles);
    const int g = tile / per_group, r = tile - g * per_group;
    const int rows_here = nbm - g * grp < grp ? nbm - g * grp : grp;
    const int bn = r / rows_here, bm = g * grp + (r - bn * rows_here);
    m0 = bm * 256;
    n0 = bn * BN;
  };
  // LDS-DMA sources: instruction j = wave + 8 i covers tile rows [8 j, 8 j + 8); byte offsets from the tensor bases
  const int srow8 = lane >> 3, spos = lane & 7;
  auto offsets = [&](int m0, int n0, unsigned* aoff, unsigned* woff) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave + 8 * i) * 8 + srow8;
      const int c = spos ^ ((r >> 1) & 7);
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const int b = m / p.arpb, l = m - b * p.arpb;
      aoff[i] = (unsigned)((b * p.abs_ + (int64_t)l * p.ars) * 2 + c * 16);
      int n = n0 + (r < BN ? r : 0);
      n = n < p.N ? n : p.N - 1;
      woff[i] = (unsigned)((int64_t)n * p.wrs * 2 + c * 16);
    }
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned sz0 = (unsigned)((hi ^ ((l31 >> 1) & 7)) << 4);   // k-sub-step ks: ^ (ks << 5) inside the asm
  const unsigned faA0 = lds_base + (wm * TM * 32 + l31) * 128 + sz0;
  const unsigned faW0 = lds_base + W_BASE + (wn * TN * 32 + l31) * 128 + sz0;
  const uint64_t abase = rfl64((uint64_t)(uintptr_t)p.A), wbase = rfl64((uint64_t)(uintptr_t)p.W);
  const uint64_t bbase = rfl64((uint64_t)(uintptr_t)p.bias);
  const unsigned nk = rfl((unsigned)(p.K / 64));
  const unsigned adst = rfl(lds_base + wave * 1024), wdst = rfl(lds_base + W_BASE + wave * 1024);

  // Per tile the per-lane source offsets of this tile and of the next one are recomputed (a few hundred VALU
  // instructions against ~40k cycles of K loop) instead of being carried across the epilogue: nothing but wave-uniform
  // scalars stays live there.
  unsigned prefetched = 0;
  for (int it = blockIdx.x; it < ntiles; it += (int)gridDim.x) {
    const int itn = it + (int)gridDim.x;
    const bool has_next = itn < ntiles;
    int m0, n0, m0n, n0n;
    tile_of(it, m0, n0);
    tile_of(has_next ? itn : it, m0n, n0n);
    unsigned aoff[4], woff[4], aoffn[4], woffn[4];
    offsets(m0, n0, aoff, woff);
    offsets(m0n, n0n, aoffn, woffn);
    const int m0w = m0 + wm * TM * 32, n0w = n0 + wn * TN * 32;
    const bool folded = p.bias != nullptr && n0w + TN * 32 <= p.N;                 // wave-uniform
    const unsigned boff = (unsigned)((n0w + hi * 4) * 4);
    const unsigned flags = rfl(prefetched | (has_next ? 2u : 0u) | (folded ? 4u : 0u));

#define OSKP_OPERANDS                                                                                              \
  ::"v"(faA0), "v"(faW0), "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "v"(aoff[3]), "v"(woff[0]), "v"(woff[1]),       \
      "v"(woff[2]), "v"(woff[3]), "v"(aoffn[0]), "v"(aoffn[1]), "v"(aoffn[2]), "v"(aoffn[3]), "v"(woffn[0]),         \
      "v"(woffn[1]), "v"(woffn[2]), "v"(woffn[3]), "v"(boff), "s"(abase), "s"(wbase), "s"(bbase), "s"(nk), "s"(adst), \
      "s"(wdst), "s"(flags)
    static_assert(BN == 128 && SCHED == 0, "shipped: the 128-wide tile, schedule 0 (256-wide tiles run on gemm256x.hip)");
    asm volatile(
#include "gemm256p_body_n128_s0.inc"
        OSKP_OPERANDS : OSKP128_CLOBBERS);
    static_assert(BN == 128, "accumulator quads below: 64 registers");
    static_assert(OSKG128_ACC_QUADS == 16, "the generated loop's accumulator map: tile t = quads 4 t .. 4 t + 3");
    osk_v4f aq[16];
    asm volatile("" : OSK_AQ_OUT_0_16(aq));

    const int b_first = m0w / p.crpb, b_last = (m0w + TM * 32 - 1) / p.crpb;
    const bool interior = m0w + TM * 32 <= p.M && n0w + TN * 32 <= p.N && b_first == b_last;  // wave-uniform
    epi::epilogue_all<Geo<BN>, OUT_F32>(aq, p, m0w, n0w, l31, hi, interior, folded);
    prefetched = 1;
  }
}

template <int BN, bool OUT_F32, int SCHED>
int launch_one(const GemmParams& p, hipStream_t st) {
  constexpr int SMEM = BN == 256 ? OSKG256_SMEM : OSKG128_SMEM;
  auto kernel = gemm256p_kernel<BN, OUT_F32, SCHED>;
  OSK_ENSURE_MAX_SMEM(kernel, SMEM);
  int n_cu = osk_device_cus();
  n_cu -= n_cu % 8;    // the tile walk keeps a workgroup inside one XCD's range only for a grid that is a multiple of 8
  if (n_cu < 8) n_cu = 8;
  const int ntiles = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
  const int grid = ntiles < n_cu ? ntiles : n_cu;   // one workgroup per CU (LDS: 96 KiB of 160)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), SMEM, st, p);
  return (int)hipGetLastError();
}

}  // namespace

// 256 x 128 tiles, 8 waves, persistent workgroups (the generator's other K-step schedules and the 256-wide instantiation of
// this frame lost their A/B runs in round 2 -- profiles/r02_gemm_experiments.md -- and are generator options only)
int launch_gemm256p(const GemmParams& p, int out_f32, hipStream_t st) {
  return out_f32 ? launch_one<128, true, 0>(p, st) : launch_one<128, false, 0>(p, st);
}

}  // namespace osk_gemm
