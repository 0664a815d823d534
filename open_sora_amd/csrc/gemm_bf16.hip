// bf16 GEMM with fused epilogues for gfx950:  C = epi(A[M,K] @ W[N,K]^T + bias)
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA
// v_mfma_f32_32x32x16_bf16 tiles.  Operands are SWAPPED in the MFMA (A-operand = W fragment, B-operand
// = activation fragment) so that a lane of the accumulator owns one output ROW m and 4 consecutive
// columns n per register quad: the epilogue (bias / GELU / gate*x+residual) is lane-local in m and
// stores 8 B (4 bf16) contiguous pieces.
//
// Staging: HBM -> LDS with global_load_lds_dwordx4 (16 B per lane, LDS image lane-linear), double
// buffered, one barrier per K tile.  The 128-B LDS rows are XOR-swizzled on the SOURCE side
// (chunk' = chunk ^ ((row >> 1) & 7)) and un-swizzled on the ds_read_b128 side: conflict-free for the
// 32-row x 16-B fragment reads of the 32x32x16 MFMA (see DESIGN.md "LDS layouts").
//
// Roofline: MFMA bf16 (2.5 PFLOP/s dense).  Algorithmic FLOPs = 2*M*N*K.
#include <atomic>
#include "gemm_params.h"
#include "../../include/osk.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile
constexpr int SMEM_BYTES = 2 * 2 * TILE_BYTES;

using osk_gemm::GemmParams;

OSK_DEV void glds16(const unsigned short* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool OUT_F32>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, nbm * nbn);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- staging addresses: 4 row-blocks of 8 rows per wave per operand
  const unsigned short* ga[4];
  const unsigned short* gw[4];
  int lds_off[4];  // byte offset of this wave's 1-KiB row block inside a tile (wave-uniform)
  const int srow8 = lane >> 3, spos = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = i * 4 + wave;
    const int r = rb * 8 + srow8;
    const int c = spos ^ ((r >> 1) & 7);  // source chunk that must land at LDS position spos
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    const int b = m / p.arpb, l = m - b * p.arpb;
    ga[i] = p.A + b * p.abs_ + (int64_t)l * p.ars + c * 8;
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    gw[i] = p.W + (int64_t)n * p.wrs + c * 8;
    lds_off[i] = rb * 1024;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  const int sw = (l31 >> 1) & 7;
  // fragment read offsets (bytes) inside a tile for ks = 0; other ks: chunk = (ks*2+hi) ^ sw
  const int a_row_off = (wm * 64 + l31) * 128;
  const int w_row_off = (wn * 64 + l31) * 128;

  // (macros rather than lambdas: by-reference captured arrays were placed in scratch by hipcc)
#define STAGE_ISSUE(BUFI, KT)                                                                  \
  {                                                                                            \
    const int k0_ = (KT) * BK;                                                                 \
    unsigned char* ta_ = smem + (BUFI) * 2 * TILE_BYTES;                                       \
    unsigned char* tw_ = ta_ + TILE_BYTES;                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(ga[i] + k0_, ta_ + lds_off[i]);       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(gw[i] + k0_, tw_ + lds_off[i]);       \
  }

  STAGE_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) STAGE_ISSUE(cur ^ 1, kt + 1);
    const unsigned char* ta = smem + cur * 2 * TILE_BYTES;
    const unsigned char* tw = ta + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = (((ks << 1) | hi) ^ sw) << 4;
      bf16x8_t af[2], wf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *reinterpret_cast<const bf16x8_t*>(ta + a_row_off + t * 32 * 128 + coff);
        wf[t] = *reinterpret_cast<const bf16x8_t*>(tw + w_row_off + t * 32 * 128 + coff);
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane owns row m = ... + l31, columns n = quad*8 + hi*4 + {0..3}
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 64 + tm * 32 + l31;
    if (m >= p.M) continue;
    const int b = m / p.crpb, l = m - b * p.crpb;
    const int64_t roff = b * p.cbs + (int64_t)l * p.crs;
    const float* grow = p.gate ? p.gate + b * p.gbs : nullptr;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int n = n0 + wn * 64 + tn * 32 + qd * 8 + hi * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][qd * 4 + j];
        if (n + 3 < p.N) {
          if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j >= p.gelu_from) v[j] = gelu_tanh(v[j]);
          if (grow) {
            const float4 gv = *reinterpret_cast<const float4*>(grow + n);
            const uint2 rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
            v[0] = bf16_lo(rv.x) + gv.x * v[0];
            v[1] = bf16_hi(rv.x) + gv.y * v[1];
            v[2] = bf16_lo(rv.y) + gv.z * v[2];
            v[3] = bf16_hi(rv.y) + gv.w * v[3];
          }
          if constexpr (OUT_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + roff + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.C) + roff + n) = o;
          }
        } else {
          for (int j = 0; j < 4 && n + j < p.N; ++j) {
            float t = v[j] + (p.bias ? p.bias[n + j] : 0.f);
            if (n + j >= p.gelu_from) t = gelu_tanh(t);
            if (grow) t = bf16_bits_to_f32(p.res[roff + n + j]) + grow[n + j] * t;
            if constexpr (OUT_F32) reinterpret_cast<float*>(p.C)[roff + n + j] = t;
            else reinterpret_cast<unsigned short*>(p.C)[roff + n + j] = f32_to_bf16_bits(t);
          }
        }
      }
    }
  }
}

}  // namespace

// argument checks of osk_gemm_bf16 + the launch parameters
static int fill_params(GemmParams& p, const void* A, int64_t a_batch_stride, int64_t a_row_stride, int a_rows_per_batch,
                       const void* W, int64_t w_row_stride, const float* bias, void* C, int64_t c_batch_stride,
                       int64_t c_row_stride, int c_rows_per_batch, const void* res, const float* gate,
                       int64_t gate_batch_stride, int M, int N, int K, int gelu_from, int out_f32) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return OSK_EINVAL;
  if (K % BK) return OSK_EINVAL;
  if (a_rows_per_batch <= 0 || c_rows_per_batch <= 0) return OSK_EINVAL;
  if ((a_batch_stride & 7) || (a_row_stride & 7) || (w_row_stride & 7)) return OSK_EINVAL;
  if ((c_batch_stride & 3) || (c_row_stride & 3)) return OSK_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) || ((uintptr_t)bias & 15)) return OSK_EINVAL;
  if (gate && (!res || ((uintptr_t)gate & 15) || (gate_batch_stride & 3) || ((uintptr_t)res & 7))) return OSK_EINVAL;
  if (out_f32 && ((uintptr_t)C & 15)) return OSK_EINVAL;   // the f32 epilogue stores float4
  p.A = (const unsigned short*)A; p.abs_ = a_batch_stride; p.ars = a_row_stride; p.arpb = a_rows_per_batch;
  p.W = (const unsigned short*)W; p.wrs = w_row_stride; p.bias = bias;
  p.C = C; p.cbs = c_batch_stride; p.crs = c_row_stride; p.crpb = c_rows_per_batch;
  p.res = (const unsigned short*)res; p.gate = gate; p.gbs = gate_batch_stride;
  p.M = M; p.N = N; p.K = K; p.gelu_from = gelu_from;
  // row bands per tile group (L2 blocking of the large tiles' order; measured at the XL shapes: 8 for wide N, 4 when there
  // are only a few weight tiles)
  p.group = (N + 255) / 256 <= 6 ? 4 : 8;
  return OSK_OK;
}

// tile choice by estimated time = rounds of the grid over the chip x time of one tile.  Per-tile throughput relative to the 256 x 256
// kernel, measured on MI355X with random data at whole rounds (round 5, tools/gemm_tile_ab.py, profiles/r05i_gemm_tile_ab.jsonl):
// 256 x 128 (gemm256p) 0.74 at K = 1152 and 0.60 at K >= 4608 (round 2 assumed 0.78 whatever K: at CFG batch 1 that sent the three
// N = 1152 Linears of every block to the 256 x 128 kernel, 9-13 % slower there than two rounds of 256 x 256 tiles); 128 x 128 (this file)
// ~0.55 with two co-resident 256-thread workgroups per CU (round 2: 0.72).  -> 2: 256 x 256, 1: 256 x 128, 0: 128 x 128
static std::atomic<int> g_tile_override{-1};   // tools / tests only (osk_gemm_tile_override): -1 = by estimate
static int tile_choice(int M, int N, int K, double* cost = nullptr) {
  auto rounds = [](int64_t tiles, int64_t slots) { return (double)((tiles + slots - 1) / slots); };
  const int forced = g_tile_override.load(std::memory_order_relaxed);
  if (forced >= 0 && !cost) return forced;
  const int64_t m256 = (M + 255) / 256, m128 = (M + 127) / 128;
  const double r128 = K >= 2304 ? 0.60 : 0.74;
  const double c256 = rounds(m256 * ((N + 255) / 256), 256) * 4.0 / 1.00;
  const double c128 = rounds(m256 * ((N + 127) / 128), 256) * 2.0 / r128;
  const double cold = rounds(m128 * ((N + 127) / 128), 512) * 1.0 / (0.5 * 0.55);  // 2 co-resident tiles share a CU
  const bool wide = N >= 256 && c256 <= c128;
  const double best = wide ? c256 : c128;
  if (cost) *cost = cold < best ? cold : best;
  if (cold < best) return 0;
  return wide ? 2 : 1;
}

static bool large_tiles_ok(const GemmParams& p) {
  const int nb = (p.M + p.arpb - 1) / p.arpb;
  const int64_t a_span = (int64_t)(nb - 1) * p.abs_ + (int64_t)(p.arpb - 1) * p.ars + p.K;
  const int64_t w_span = (int64_t)(p.N - 1) * p.wrs + p.K;
  return osk_gemm::gemm256_supported(p, a_span, w_span);
}

// ---- GEGLU (f4): the fused path is gemm256x's epilogue class; shapes the 256 x 256 tiles do not take run the plain GEMM into the
// caller's workspace ([M, 2 N_out] bf16, packed column order) and this row kernel
namespace {
__global__ void __launch_bounds__(256) geglu_rows_kernel(const unsigned short* __restrict__ t, unsigned short* __restrict__ c,
                                                         int64_t cbs, int64_t crs, int crpb, int M, int n_out) {
  const int per_row = n_out / 4;
  const int64_t total = (int64_t)M * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / per_row), c4 = (int)(i - (int64_t)m * per_row) * 4;
    const int j2 = c4 >> 4, r = c4 & 15;
    const unsigned short* row = t + (int64_t)m * 2 * n_out + j2 * 32 + r;
    const uint2 v = *reinterpret_cast<const uint2*>(row), g = *reinterpret_cast<const uint2*>(row + 16);
    uint2 o;
    o.x = pack_bf16x2(bf16_lo(v.x) * gelu_tanh(bf16_lo(g.x)), bf16_hi(v.x) * gelu_tanh(bf16_hi(g.x)));
    o.y = pack_bf16x2(bf16_lo(v.y) * gelu_tanh(bf16_lo(g.y)), bf16_hi(v.y) * gelu_tanh(bf16_hi(g.y)));
    const int b = m / crpb, l = m - b * crpb;
    *reinterpret_cast<uint2*>(c + b * cbs + (int64_t)l * crs + c4) = o;
  }
}
}  // namespace

extern "C" int osk_gemm_geglu_bf16(const void* A, int64_t a_batch_stride, int64_t a_row_stride, int a_rows_per_batch,
                                   const void* W_packed, int64_t w_row_stride, const float* bias_packed, void* C,
                                   int64_t c_batch_stride, int64_t c_row_stride, int c_rows_per_batch, int M, int N_out, int K,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (N_out <= 0 || (N_out & 15)) return OSK_EINVAL;
  GemmParams p;
  const int N = 2 * N_out;
  const int rc = fill_params(p, A, a_batch_stride, a_row_stride, a_rows_per_batch, W_packed, w_row_stride, bias_packed, C,
                             c_batch_stride, c_row_stride, c_rows_per_batch, nullptr, nullptr, 0, M, N, K, N, 0);
  if (rc != OSK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  p.geglu = 1;
  if (large_tiles_ok(p) && tile_choice(M, N, K) == 2) return osk_gemm::launch_gemm256x(p, 0, st);
  // small / odd shapes: plain GEMM (bias added, no activation) into the workspace, then the row kernel
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < (int64_t)M * N * 2) return OSK_EUNSUPPORTED;
  const int rc2 = osk_gemm_bf16(A, a_batch_stride, a_row_stride, a_rows_per_batch, W_packed, w_row_stride, bias_packed, workspace,
                                (int64_t)M * N, N, M, nullptr, nullptr, 0, M, N, K, N, 0, stream);
  if (rc2 != OSK_OK) return rc2;
  const int64_t total = (int64_t)M * (N_out / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(geglu_rows_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned short*)workspace, (unsigned short*)C,
                     c_batch_stride, c_row_stride, c_rows_per_batch, M, N_out);
  return (int)hipGetLastError();
}

// reporting / tools: which tile kernel osk_gemm_bf16 launches for an [M, N] problem the large tiles support (2: 256 x 256
// gemm256x_kernel, 1: 256 x 128 gemm256p_kernel, 0: 128 x 128 gemm_bf16_kernel), and an override for same-process A/B timing of the
// three (tile_kind -1 restores the estimate; process-wide, not for production use)
extern "C" int osk_gemm_tile_choice(int M, int N, int K) { return tile_choice(M, N, K); }
extern "C" int osk_gemm_tile_override(int tile_kind) {
  if (tile_kind < -1 || tile_kind > 2) return OSK_EINVAL;
  g_tile_override.store(tile_kind, std::memory_order_relaxed);
  return OSK_OK;
}

extern "C" int osk_gemm_bf16_pair(const OskGemmOperands* a, const OskGemmOperands* b, int N, int K, int gelu_from, void* stream) {
  if (!a || !b) return OSK_EINVAL;
  GemmParams p[2];
  const OskGemmOperands* o[2] = {a, b};
  for (int i = 0; i < 2; ++i) {
    const int rc = fill_params(p[i], o[i]->A, o[i]->a_batch_stride, o[i]->a_row_stride, o[i]->a_rows_per_batch, o[i]->W,
                               o[i]->w_row_stride, o[i]->bias, o[i]->C, o[i]->c_batch_stride, o[i]->c_row_stride,
                               o[i]->c_rows_per_batch, o[i]->res, o[i]->gate, o[i]->gate_batch_stride, o[i]->M, N, K, gelu_from, 0);
    if (rc != OSK_OK) return rc;
  }
  // one tile list where BOTH problems take the 256 x 256 tile kernel; the larger problem first (its tiles fill whole rounds, the
  // smaller one's the tail); otherwise exactly the two single calls
  const int big = p[0].M >= p[1].M ? 0 : 1;
  // one tile list when the larger problem alone would take the 256 x 256 kernel AND the estimate says so: the smaller problem's
  // tiles ride in the larger one's last round (XL: 2688 + 84 tiles = 11 rounds either way) -- but not when they would open a new,
  // nearly empty round (11B geometry: 2304 tiles are exactly 9 rounds; + 72 tiles would make it 10)
  double c_big = 0.0, c_small = 0.0;
  if (large_tiles_ok(p[0]) && large_tiles_ok(p[1]) && tile_choice(p[big].M, N, K, &c_big) == 2) {
    tile_choice(p[big ^ 1].M, N, K, &c_small);
    const int64_t nbn = (N + 255) / 256, tiles = ((p[0].M + 255) / 256 + (p[1].M + 255) / 256) * nbn;
    const double c_pair = (double)((tiles + 255) / 256) * 4.0;
    if (c_pair < c_big + c_small) return osk_gemm::launch_gemm256x_pair(p[big], p[big ^ 1], (hipStream_t)stream);
  }
  (void)big;
  for (int i = 0; i < 2; ++i) {
    const int rc = osk_gemm_bf16(o[i]->A, o[i]->a_batch_stride, o[i]->a_row_stride, o[i]->a_rows_per_batch, o[i]->W, o[i]->w_row_stride,
                                 o[i]->bias, o[i]->C, o[i]->c_batch_stride, o[i]->c_row_stride, o[i]->c_rows_per_batch, o[i]->res,
                                 o[i]->gate, o[i]->gate_batch_stride, o[i]->M, N, K, gelu_from, 0, stream);
    if (rc != OSK_OK) return rc;
  }
  return OSK_OK;
}

extern "C" int osk_gemm_bf16(const void* A, int64_t a_batch_stride, int64_t a_row_stride,
                             int a_rows_per_batch, const void* W, int64_t w_row_stride,
                             const float* bias, void* C, int64_t c_batch_stride, int64_t c_row_stride,
                             int c_rows_per_batch, const void* res, const float* gate,
                             int64_t gate_batch_stride, int M, int N, int K, int gelu_from,
                             int out_f32, void* stream) {
  GemmParams p;
  const int rc = fill_params(p, A, a_batch_stride, a_row_stride, a_rows_per_batch, W, w_row_stride, bias, C, c_batch_stride,
                             c_row_stride, c_rows_per_batch, res, gate, gate_batch_stride, M, N, K, gelu_from, out_f32);
  if (rc != OSK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  {
    const int nb = (M + a_rows_per_batch - 1) / a_rows_per_batch;
    const int64_t a_span = (int64_t)(nb - 1) * a_batch_stride + (int64_t)(a_rows_per_batch - 1) * a_row_stride + K;
    const int64_t w_span = (int64_t)(N - 1) * w_row_stride + K;
    if (osk_gemm::gemm256_supported(p, a_span, w_span)) {
      const int kind = tile_choice(M, N, K);
      if (kind == 2) return osk_gemm::launch_gemm256x(p, out_f32, st);    // 256 x 256 tiles, 4 waves, v_mfma_f32_16x16x32_bf16
      if (kind == 1) return osk_gemm::launch_gemm256p(p, out_f32, st);    // 256 x 128 tiles, 8 waves, v_mfma_f32_32x32x16_bf16
    }
  }
  const int nblk = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  dim3 grid(nblk), block(256);
  if (out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<true>), grid, block, SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false>), grid, block, SMEM_BYTES, st, p);
  return (int)hipGetLastError();
}

// ---- a group of Linear problems sharing K (include/osk.h: OskGemmTask): the plain tasks go out through the 256 x 256 tile kernel's
// single / pair launches (a skipped column range included), the V^T tasks as ONE launch of its V^T instantiation -- two launches
// for a block's projection instead of GEMM + osk_v_transpose_bf16, with V never written token-major
extern "C" int osk_gemm_group_bf16(const OskGemmTask* tasks, int n_tasks, int K, void* stream) {
  if (!tasks || n_tasks < 1 || n_tasks > 4 || K <= 0 || (K % BK)) return OSK_EINVAL;
  GemmParams plain[4], vts[4];
  int n_plain = 0, n_vt = 0;
  for (int i = 0; i < n_tasks; ++i) {
    const OskGemmTask& t = tasks[i];
    const OskGemmOperands& o = t.op;
    if (t.vt_head_dim == 0) {
      GemmParams& p = plain[n_plain++];
      const int rc = fill_params(p, o.A, o.a_batch_stride, o.a_row_stride, o.a_rows_per_batch, o.W, o.w_row_stride, o.bias, o.C,
                                 o.c_batch_stride, o.c_row_stride, o.c_rows_per_batch, o.res, o.gate, o.gate_batch_stride, o.M, t.N, K,
                                 t.gelu_from, 0);
      if (rc != OSK_OK) return rc;
      if (t.skip_len < 0 || t.skip_from < 0) return OSK_EINVAL;
      if (t.skip_len > 0) {
        if (o.gate || (t.skip_from % 256) || (t.skip_len % 8) || t.skip_from + t.skip_len > t.N) return OSK_EINVAL;
        p.skip_from = t.skip_from;
        p.skip_len = t.skip_len;
      }
      if (!large_tiles_ok(p) || t.N - t.skip_len < 128) return OSK_EUNSUPPORTED;
      continue;
    }
    // V^T task: the kernel's A operand is the weight, its W operand the activations (gemm_params.h)
    GemmParams& p = vts[n_vt++];
    const int hd = t.vt_head_dim, L = o.a_rows_per_batch;
    if (hd != 64 && hd != 72 && hd != 128) return OSK_EUNSUPPORTED;
    if (!o.A || !o.W || !o.C || o.res || o.gate || L <= 0 || o.M <= 0 || (o.M % L) || t.N <= 0 || (t.N % hd)) return OSK_EINVAL;
    if ((o.a_batch_stride & 7) || (o.a_row_stride & 7) || (o.w_row_stride & 7) || ((uintptr_t)o.A & 15) || ((uintptr_t)o.W & 15) ||
        ((uintptr_t)o.C & 1) || ((uintptr_t)o.bias & 3))
      return OSK_EINVAL;
    const int B = o.M / L, Lp = (L + 63) / 64 * 64;
    if ((int64_t)B * Lp > 0x7fffff00) return OSK_EUNSUPPORTED;
    p = GemmParams{};
    p.A = (const unsigned short*)o.W; p.abs_ = 0; p.ars = o.w_row_stride; p.arpb = 0x7fffffff;
    p.W = (const unsigned short*)o.A; p.wrs = o.a_row_stride; p.wbs = o.a_batch_stride; p.wrpb = Lp; p.wvalid = L;
    p.bias = nullptr; p.rowbias = o.bias;
    p.C = o.C; p.cbs = 0; p.crs = o.c_row_stride; p.crpb = t.N; p.ccbs = o.c_batch_stride;
    p.res = nullptr; p.gate = nullptr; p.gbs = 0;
    p.M = t.N; p.N = B * Lp; p.K = K; p.gelu_from = p.N;
    p.group = 8;
    p.vt = hd == 128 ? 2 : 1;
    const int64_t a_span = (int64_t)(t.N - 1) * o.w_row_stride + K;
    const int64_t w_span = (int64_t)(B - 1) * o.a_batch_stride + (int64_t)(L - 1) * o.a_row_stride + K;
    if (!osk_gemm::gemm256_supported(p, a_span, w_span)) return OSK_EUNSUPPORTED;
  }
  if (n_vt > 2 || n_plain > 2) return OSK_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // every task qualified: launch.  Two plain tasks of equal N / gelu_from / no skip range share a tile list (the pair kernel).
  if (n_plain == 2 && plain[0].N == plain[1].N && plain[0].gelu_from == plain[1].gelu_from && !plain[0].skip_len && !plain[1].skip_len) {
    const int big = plain[0].M >= plain[1].M ? 0 : 1;
    plain[big ^ 1].group = plain[big].group;
    const int rc = osk_gemm::launch_gemm256x_pair(plain[big], plain[big ^ 1], st);
    if (rc != OSK_OK) return rc;
  } else {
    for (int i = 0; i < n_plain; ++i) {
      const int rc = osk_gemm::launch_gemm256x(plain[i], 0, st);
      if (rc != OSK_OK) return rc;
    }
  }
  if (n_vt) {
    if (n_vt == 2 && vts[1].M * (int64_t)vts[1].N > vts[0].M * (int64_t)vts[0].N) { const GemmParams t_ = vts[0]; vts[0] = vts[1]; vts[1] = t_; }
    return osk_gemm::launch_gemm256x_vt(vts, n_vt, st);
  }
  return OSK_OK;
}
