// Entry points of the flash-attention family (bf16 in / out, f32 softmax, non-causal) + the tail-split merge kernel + the fp8
// V^T preparation.  The kernels themselves are the hand-scheduled, generated loops of attention_asm72.hip (head_dim 72 and, on the
// same loop with zero dims 64..71, head_dim 64), attention_asm72w.hip (their wide layout for bounded calls with >= 1024 query
// rows), attention_asm128.hip and the fp8 P.V variants.  Round 4 removed the last compiler-scheduled flash kernel
// (attn_fwd_kernel<64>, 400 lines: 8 waves x 32 rows, register-staged tiles) -- head_dim 64 now runs the head_dim-72 loop.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 4 * B * H * Lq * Lk * hd (QK^T + PV, not halved).
#include <atomic>
#include "attention_params.h"
#include "../../include/osk.h"

namespace {
using osk_attn::AttnParams;

// ONE kernel family per head_dim (tail units split along the keys + merge when the caller handed over a workspace)
template <int HD>
int launch(const AttnParams& p, hipStream_t st) {
  int rc;
  if constexpr (HD == 72) rc = p.rows == 512 ? osk_attn::launch_asm72w(p, st) : osk_attn::launch_asm72(p, st);
  else if constexpr (HD == 64) rc = p.rows == 512 ? osk_attn::launch_asm64w(p, st) : osk_attn::launch_asm64(p, st);
  else rc = osk_attn::launch_asm128(p, st);
  return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, HD, st);
}

}  // namespace

namespace osk_attn {
namespace {

// combine the key parts of the split tail units: out = sum_p 2^(lse_p - lse) O_p, lse = log2 sum_p 2^lse_p.
// grid (tail units, 4): a block owns 64 query rows; thread = (row, 16-byte chunk of the head dim), so the partial rows
// -- contiguous [row][hd] f32 in the workspace -- are read as one linear stream per part.  A few tens of MB in all.
template <int HD>
__global__ void __launch_bounds__(256) attn_merge_kernel(const AttnParams p) {
  constexpr int CPR = HD / 4;                    // float4 chunks per row
  const int nqb = (p.Lq + p.rows - 1) / p.rows;
  int bh, qb;
  unit_to_work(p, nqb, p.tail_first + (int)blockIdx.x, bh, qb);
  const int b = bh / p.H, h = bh - b * p.H;
  float bound_;
  if (!attn_auto_bound(p, b, h, true, bound_)) return;   // auto-dispatched pair: this unit was the general twin's (it writes final rows itself)
  const int S = p.tail_split;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int r = blockIdx.y * 64 + i / CPR, c = i % CPR;
    const int row = qb * p.rows + r;
    if (row >= p.Lq) continue;
    const int64_t slot0 = (int64_t)blockIdx.x * S * p.rows + r;
    float l[8], m = -INFINITY;
    for (int s = 0; s < S; ++s) { l[s] = p.ws_lse[slot0 + s * p.rows]; m = fmaxf(m, l[s]); }
    float tot = 0.f;
    for (int s = 0; s < S; ++s) { l[s] = exp2f(l[s] - m); tot += l[s]; }
    const float inv = 1.0f / tot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(p.ws_o + (slot0 + (int64_t)s * p.rows) * HD + c * 4);
      acc.x += l[s] * v.x; acc.y += l[s] * v.y; acc.z += l[s] * v.z; acc.w += l[s] * v.w;
    }
    uint2 w2;
    w2.x = pack_bf16x2(acc.x * inv, acc.y * inv);
    w2.y = pack_bf16x2(acc.z * inv, acc.w * inv);
    *reinterpret_cast<uint2*>(p.out + b * p.obs + (int64_t)row * p.ors + h * HD + c * 4) = w2;
    if (p.lse && c == 0) p.lse[(int64_t)bh * p.Lq + row] = (m + log2f(tot)) * 0.6931471805599453f;
  }
}

}  // namespace

// A workgroup owns a CU for its whole key loop, so a launch takes ceil(units / CUs) rounds and the last round may run
// a fraction of the chip (3168 units on 256 CUs: 12.4 -> 13; sequence-parallel ranks: 1.7 -> 2, 3.2 -> 4).  With a
// workspace the units of that last round are cut into `s` key parts (whole key segments when there are several,
// else runs of 64-key tiles) so that the round costs ceil(R s / CUs) / s instead of 1; only those units pay the
// partial-result traffic.  s <= 8, chosen to minimise that cost; no split when it saves < 15 % of a round.
void split_tail(AttnParams& p, int units, int hd, void* workspace, int64_t workspace_bytes) {
  p.tail_split = 1;
  p.tail_first = 0x7fffffff;
  if (!workspace) return;
  const int cus = osk_device_cus();
  const int R = units % cus;
  if (R == 0) return;
  auto cost = [&](int s) { return (double)((R * s + cus - 1) / cus) / s; };
  int best = 1;
  for (int s = 2; s <= 8; ++s) {
    if (p.n_seg > 1 ? (p.n_seg % s != 0) : (s > p.tps)) continue;
    const int64_t need = (int64_t)R * s * p.rows * (hd + 1) * 4;
    if (need > workspace_bytes) continue;
    if (cost(s) < cost(best) - 1e-9) best = s;
  }
  if (cost(1) - cost(best) < 0.15) return;
  p.tail_split = best;
  p.tail_first = units - R;
  p.ws_o = (float*)workspace;
  p.ws_lse = p.ws_o + (int64_t)R * best * p.rows * hd;
}

int launch_merge(const AttnParams& p, int hd, hipStream_t st) {
  const int units = ((p.Lq + p.rows - 1) / p.rows) * p.B * p.H;
  dim3 grid(units - p.tail_first, p.rows / 64), block(256);
  if (hd == 72) hipLaunchKernelGGL(attn_merge_kernel<72>, grid, block, 0, st, p);
  else if (hd == 64) hipLaunchKernelGGL(attn_merge_kernel<64>, grid, block, 0, st, p);
  else if (hd == 128) hipLaunchKernelGGL(attn_merge_kernel<128>, grid, block, 0, st, p);
  else return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

}  // namespace osk_attn

// reporting: (key parts, rows per work unit) the launch of a call with these arguments uses -- the SAME selection code as
// osk_attention_fwd_bounded_bf16 below (round 4 modelled 256-row units whatever the call: ADVICE r4)
static std::atomic<int> g_rows_override{0};   // tools only (osk_attention_rows_override): 0 = by estimate, 256 / 512 = forced where legal
static void launch_shape(osk_attn::AttnParams& p, int hd, void* workspace, int64_t workspace_bytes) {
  p.rows = 256;
  const int forced = g_rows_override.load(std::memory_order_relaxed);
  const bool wide_legal = (hd == 72 || hd == 64) && osk_attn::attn_fast_path(p) && p.Lq >= 1024;
  if (forced ? (forced == 512 && wide_legal) : osk_attn::attn_wide_path(p, hd, osk_device_cus(), workspace != nullptr)) p.rows = 512;
  if (hd == 64 || hd == 72 || hd == 128)
    osk_attn::split_tail(p, ((p.Lq + p.rows - 1) / p.rows) * p.B * p.H, hd, workspace, workspace_bytes);
}

static float bound_to_bf16_up(float score_bound) {
  // the kernels keep the bound in a bf16 field of Q's padding dim: round it UP to the next bf16 value (it stays a bound)
  const unsigned bits = __builtin_bit_cast(unsigned, score_bound);
  return __builtin_bit_cast(float, (bits + 0xFFFFu) & 0xFFFF0000u);
}

extern "C" int osk_attention_launch_shape(int B, int H, int Lq, int n_seg, int seg_len, int hd, float score_bound,
                                          int64_t workspace_bytes, int* rows_per_unit) {
  if (rows_per_unit) *rows_per_unit = 256;
  if (B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0 || (hd != 64 && hd != 72 && hd != 128) || !(score_bound >= 0.f)) return 1;
  osk_attn::AttnParams p{};
  p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.bound = bound_to_bf16_up(score_bound);
  static char dummy[16] __attribute__((aligned(16)));
  launch_shape(p, hd, workspace_bytes > 0 ? dummy : nullptr, workspace_bytes);
  if (rows_per_unit) *rows_per_unit = p.rows;
  return p.tail_split;
}

extern "C" int osk_attention_rows_override(int rows_per_unit) {
  if (rows_per_unit != 0 && rows_per_unit != 256 && rows_per_unit != 512) return OSK_EINVAL;
  g_rows_override.store(rows_per_unit, std::memory_order_relaxed);
  return OSK_OK;
}

extern "C" int osk_attention_tail_split_factor(int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                               int64_t workspace_bytes) {
  return osk_attention_launch_shape(B, H, Lq, n_seg, seg_len, hd, 0.0f, workspace_bytes, nullptr);
}

extern "C" int64_t osk_attention_workspace_bytes(void) {
  // enough for 512 key parts (two rounds of a 256-CU chip) of 256 rows at head_dim 128: partial O + LSE in f32
  return (int64_t)512 * 256 * (128 + 1) * 4;
}

extern "C" int osk_attention_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                      const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                      int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                      void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                      float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                      float scale, int q_prescaled, int kv_batches, void* stream) {
  return osk_attention_fwd_ws_bf16(q, q_batch_stride, q_row_stride, k, k_seg_stride, k_batch_stride, k_row_stride, vt,
                                   vt_seg_stride, out, o_batch_stride, o_row_stride, lse, B, H, Lq, n_seg, seg_len, hd,
                                   scale, q_prescaled, kv_batches, nullptr, 0, stream);
}

extern "C" int osk_attention_fwd_bounded_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                         const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                         int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                         void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                         float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                         float scale, int q_prescaled, int kv_batches, float score_bound,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
  if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0) return OSK_EINVAL;
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_seg_stride & 7) || (k_batch_stride & 7) ||
      (k_row_stride & 7) || (vt_seg_stride & 7) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7)) return OSK_EINVAL;
  AttnParams p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kss = k_seg_stride; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = (const unsigned short*)vt; p.vtss = vt_seg_stride;
  p.out = (unsigned short*)out; p.obs = o_batch_stride; p.ors = o_row_stride;
  p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.sc = q_prescaled ? 1.0f : scale * 1.4426950408889634f;  // log2 units; 1: q already carries it
  p.q_prescaled = q_prescaled;
  if (!(score_bound >= 0.f)) return OSK_EINVAL;   // (also rejects NaN)
  p.bound = bound_to_bf16_up(score_bound);
  if (kv_batches < 0 || kv_batches > B) return OSK_EINVAL;
  p.Bkv = kv_batches > 0 ? kv_batches : B;
  p.map = 1;   // XCD-contiguous work order (each XCD walks one head's K / V^T stream)
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  launch_shape(p, hd, workspace, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  switch (hd) {
    case 64: return launch<64>(p, st);
    case 72: return launch<72>(p, st);
    case 128: return launch<128>(p, st);
    default: return OSK_EUNSUPPORTED;
  }
}

// ---- auto-dispatched pair (round 6): the bound comes from the operands themselves, per (batch, head), on the device
extern "C" int osk_attention_fwd_auto_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                           const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                           int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                           void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                           float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                           float scale, int q_prescaled, int kv_batches, const float* q_norm2_max,
                                           const float* k_norm2_max, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!q || !k || !vt || !out || !q_norm2_max || !k_norm2_max || B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0) return OSK_EINVAL;
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_seg_stride & 7) || (k_batch_stride & 7) ||
      (k_row_stride & 7) || (vt_seg_stride & 7) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7) ||
      ((uintptr_t)q_norm2_max & 3) || ((uintptr_t)k_norm2_max & 3)) return OSK_EINVAL;
  if (hd != 64 && hd != 72 && hd != 128) return OSK_EUNSUPPORTED;
  if (!q_prescaled && scale * 1.4426950408889634f != 1.0f) return OSK_EUNSUPPORTED;   // the norms must be those of the operands the MFMA sees
  AttnParams p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kss = k_seg_stride; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = (const unsigned short*)vt; p.vtss = vt_seg_stride;
  p.out = (unsigned short*)out; p.obs = o_batch_stride; p.ors = o_row_stride;
  p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.sc = 1.0f;
  p.q_prescaled = 1;
  if (kv_batches < 0 || kv_batches > B) return OSK_EINVAL;
  p.Bkv = kv_batches > 0 ? kv_batches : B;
  p.map = 1;
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  p.bound = 1.0f;                                  // "a bound exists": the FAST body's layout rules (attn_fast_path)
  const bool fast_possible = osk_attn::attn_fast_path(p);
  auto run = [&](const AttnParams& a) {
    switch (hd) {
      case 64: return a.rows == 512 ? osk_attn::launch_asm64w(a, st) : osk_attn::launch_asm64(a, st);
      case 72: return a.rows == 512 ? osk_attn::launch_asm72w(a, st) : osk_attn::launch_asm72(a, st);
      default: return osk_attn::launch_asm128(a, st);
    }
  };
  if (!fast_possible) {                            // several key segments of fewer than 3 tiles: only the general body takes them
    p.bound = 0.f;
    launch_shape(p, hd, workspace, workspace_bytes);
    const int rc = run(p);
    return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, hd, st);
  }
  // the FAST twin: launch shape as a bounded call's (wide layout, tail split), every workgroup checks its own (batch, head)
  p.qn2 = q_norm2_max; p.kn2 = k_norm2_max;
  launch_shape(p, hd, workspace, workspace_bytes);
  int rc = run(p);
  if (rc != 0) return rc;
  // the general twin: 256-row units, no tail split (its units write their final rows; the merge kernel skips them)
  AttnParams g = p;
  g.bound = 0.f; g.rows = 256; g.tail_split = 1; g.tail_first = 0x7fffffff; g.ws_o = nullptr; g.ws_lse = nullptr;
  rc = run(g);
  if (rc != 0) return rc;
  return p.tail_split == 1 ? OSK_OK : osk_attn::launch_merge(p, hd, st);
}

// squared row norms of a bf16 [B, L, H * hd] view, maximum per (batch, head): out[b * H + h] = max_l sum_d x[b, l, h, d]^2  (f32).
// HBM-bound: one read of the tensor.  accumulate != 0 keeps the values already in `out` (several key segments / sequence-parallel ranks
// fold into one maximum); else out is zeroed first (stream-ordered).  Non-negative floats order like their bit patterns: integer atomics.
namespace {
__global__ void zero_u32_kernel(unsigned* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

template <int HD>
__global__ void __launch_bounds__(256) rownorm2_max_kernel(const unsigned short* __restrict__ x, int64_t bs, int64_t rs, int L, int H,
                                                           int rows_per_block, unsigned* __restrict__ out) {
  __shared__ unsigned smax[256];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < H; i += 256) smax[i] = 0;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = r0 + rows_per_block < L ? r0 + rows_per_block : L;
  const int pairs = (r1 - r0) * H;                       // (row, head) pairs of this block's slab, head fastest
  constexpr int CPR = HD / 8;
  for (int i = threadIdx.x; i < pairs; i += 256) {
    const int r = r0 + i / H, h = i - (i / H) * H;
    const unsigned short* px = x + b * bs + (int64_t)r * rs + h * HD;
    uint4 u[CPR];
#pragma unroll
    for (int c = 0; c < CPR; ++c) u[c] = *reinterpret_cast<const uint4*>(px + c * 8);   // all loads first
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
      float f[8];
      unpack8(u[c], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    atomicMax(&smax[h], __float_as_uint(ss));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256)
    if (smax[i]) atomicMax(&out[b * H + i], smax[i]);
}
}  // namespace

extern "C" int osk_rownorm2_max_bf16(const void* x, int64_t batch_stride, int64_t row_stride, int B, int L, int H, int hd,
                                     float* out, int accumulate, void* stream) {
  if (!x || !out || B <= 0 || L <= 0 || H <= 0 || H > 256 || (batch_stride & 7) || (row_stride & 7) || ((uintptr_t)x & 15) || ((uintptr_t)out & 3))
    return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // (zeroed by a KERNEL, not hipMemsetAsync: under hipGraph capture the memset of these few bytes was not replayed -- a replayed step
  //  then folded its maxima into the previous replay's, found by tests/test_gpu_mmdit.py::test_denoise_step_is_hipgraph_capturable[device_bound])
  if (!accumulate) hipLaunchKernelGGL(zero_u32_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, reinterpret_cast<unsigned*>(out), B * H);
  int rows = (L + 255) / 256;             // about 256 blocks per batch item: few, fat blocks (same-address atomics are slow)
  if (rows < 16) rows = 16;
  dim3 grid((L + rows - 1) / rows, B), block(256);
  unsigned* o = reinterpret_cast<unsigned*>(out);
  if (hd == 64) hipLaunchKernelGGL(rownorm2_max_kernel<64>, grid, block, 0, st, (const unsigned short*)x, batch_stride, row_stride, L, H, rows, o);
  else if (hd == 72) hipLaunchKernelGGL(rownorm2_max_kernel<72>, grid, block, 0, st, (const unsigned short*)x, batch_stride, row_stride, L, H, rows, o);
  else if (hd == 128) hipLaunchKernelGGL(rownorm2_max_kernel<128>, grid, block, 0, st, (const unsigned short*)x, batch_stride, row_stride, L, H, rows, o);
  else return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

extern "C" int osk_attention_fwd_ws_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                         const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                         int64_t k_row_stride, const void* vt, int64_t vt_seg_stride,
                                         void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                         float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                         float scale, int q_prescaled, int kv_batches, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  return osk_attention_fwd_bounded_bf16(q, q_batch_stride, q_row_stride, k, k_seg_stride, k_batch_stride, k_row_stride, vt,
                                        vt_seg_stride, out, o_batch_stride, o_row_stride, lse, B, H, Lq, n_seg, seg_len, hd,
                                        scale, q_prescaled, kv_batches, 0.0f, workspace, workspace_bytes, stream);
}

extern "C" const char* osk_attention_kernel_name(int hd, int seg_len) {
  (void)seg_len;
  return hd == 72 ? "attn_asm72_kernel" : hd == 128 ? "attn_asm128_kernel" : hd == 64 ? "attn_asm72_kernel (head_dim 64 instantiation)" : "unsupported";
}

extern "C" const char* osk_attention_body_name(int hd, int n_seg, int seg_len, float score_bound) {
  if (hd != 64 && hd != 72 && hd != 128) return "unsupported";
  AttnParams p{};
  p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  if (score_bound > 0.f) {   // rounded up to bf16 as osk_attention_fwd_bounded_bf16 does
    const unsigned bits = __builtin_bit_cast(unsigned, score_bound);
    p.bound = __builtin_bit_cast(float, (bits + 0xFFFFu) & 0xFFFF0000u);
  }
  const bool fast = osk_attn::attn_fast_path(p);
  if (hd == 72 || hd == 64) return fast ? "attn_asm72_kernel<FAST>" : "attn_asm72_kernel<general>";   // (Lq >= 1024: the FAST body's wide layout, attn_asm72w_kernel)
  return fast ? "attn_asm128_kernel<FAST>" : "attn_asm128_kernel<general>";
}


// =============================================================================================
// fp8 P.V variant: V -> e4m3 V^T with the key order, ones row and zero rows the kernels expect; entry point
// =============================================================================================
namespace {

// V [B, L, H, hd] bf16 (strided) -> vt8 [B, H, RP, Lp] e4m3 bytes, RP = (hd + 1) rounded up to 16.
// One block = one 64-key tile of one head.  Row d < hd: byte hi*32 + t2*16 + j*4 + i of the tile holds key
// 32 t2 + 8 j + 4 hi + i (the order in which a lane of the score accumulators holds its 32 P values), quantised
// with the (batch, head) scale; row hd: 1.0 for keys < L, 0 behind them (softmax denominator / key validity);
// rows hd+1 .. RP-1: zero.
template <int HD>
__global__ void __launch_bounds__(256) v_transpose_fp8_kernel(const unsigned short* __restrict__ v, int64_t bs,
                                                              int64_t rs, const float* __restrict__ scales,
                                                              unsigned char* __restrict__ vt8, int L, int Lp, int H) {
  constexpr int RP = (HD + 1 + 15) / 16 * 16;
  __shared__ unsigned short tile[64][HD + 2];
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int key0 = kt * 64;
  constexpr int CPR = HD / 8;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int r = i / CPR, c = i % CPR;
    const int key = key0 + r;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (key < L) u = *reinterpret_cast<const uint4*>(v + b * bs + (int64_t)key * rs + h * HD + c * 8);
    unsigned* dst = reinterpret_cast<unsigned*>(&tile[r][c * 8]);
    dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
  }
  __syncthreads();
  const float inv = 1.0f / scales[b * H + h];
  unsigned char* obase = vt8 + ((int64_t)(b * H + h) * RP) * Lp + key0;
  // thread = (row, 16-byte chunk of the 64-byte row)
  for (int i = threadIdx.x; i < RP * 4; i += 256) {
    const int d = i >> 2, c = i & 3;
    unsigned w[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {          // dword q4 of the chunk: bytes pos = c*16 + q4*4 + 0..3
      const int pos = c * 16 + q4 * 4;
      const int hi = pos >> 5, s5 = pos & 31, t2 = s5 >> 4, j = (s5 & 15) >> 2;
      const int k0 = 32 * t2 + 8 * j + 4 * hi;   // keys k0 .. k0 + 3
      if (d < HD) {
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          f[e] = fminf(fmaxf(bf16_bits_to_f32(tile[k0 + e][d]) * inv, -448.0f), 448.0f);
        int r = 0;
        r = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r, true);
        w[q4] = (unsigned)r;
      } else if (d == HD) {
        unsigned r = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) r |= (key0 + k0 + e < L ? 0x38u : 0u) << (8 * e);
        w[q4] = r;
      } else {
        w[q4] = 0;
      }
    }
    *reinterpret_cast<uint4*>(obase + (int64_t)d * Lp + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace

namespace {

// absmax of V per (batch, head) -> e4m3 scale: bit patterns of |bf16| order like the values, so the reduction runs on
// integers.  grid (row slabs, B); a thread owns 16-byte chunks of the [H*hd] row (one head each), walks the slab's rows,
// then one LDS atomic per chunk and one global atomic per (block, head): few, fat blocks (same-address atomics are slow).
__global__ void __launch_bounds__(256) v_absmax_kernel(const unsigned short* __restrict__ v, int64_t bs, int64_t rs,
                                                       int L, int H, int hd, int rows_per_block,
                                                       unsigned* __restrict__ amax_bits) {
  __shared__ unsigned smax[256];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < H; i += 256) smax[i] = 0;
  __syncthreads();
  const int cpr = (H * hd) >> 3;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = r0 + rows_per_block < L ? r0 + rows_per_block : L;
  for (int c = threadIdx.x; c < cpr; c += 256) {
    const unsigned short* p = v + b * bs + c * 8;
    unsigned m = 0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      uint4 u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const uint4*>(p + (int64_t)(r + i) * rs);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned w[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned a = w[j] & 0x7FFF7FFFu;
          m = max(m, max(a & 0xFFFFu, a >> 16));
        }
      }
    }
    for (; r < r1; ++r) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + (int64_t)r * rs);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned a = w[j] & 0x7FFF7FFFu;
        m = max(m, max(a & 0xFFFFu, a >> 16));
      }
    }
    atomicMax(&smax[(c * 8) / hd], m);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256)
    if (smax[i]) atomicMax(&amax_bits[b * H + i], smax[i]);
}

__global__ void v_scale_zero_kernel(unsigned* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

__global__ void v_scale_finalize_kernel(unsigned* __restrict__ amax_bits, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float amax = __uint_as_float(amax_bits[i] << 16);
  reinterpret_cast<float*>(amax_bits)[i] = amax > 0.f ? amax / 448.0f : 1.0f;
}

}  // namespace

extern "C" int osk_v_scale_fp8(const void* v, int64_t bs, int64_t rs, float* scales, int B, int L, int H, int hd,
                               void* stream) {
  if (!v || !scales || B <= 0 || L <= 0 || H <= 0 || H > 256 || (hd & 7) || (bs & 7) || (rs & 7) || ((uintptr_t)v & 15))
    return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(v_scale_zero_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, reinterpret_cast<unsigned*>(scales), B * H);   // (a kernel, not hipMemsetAsync: see osk_rownorm2_max_bf16)
  int rows = (L + 127) / 128;            // about 128 blocks per batch item
  if (rows < 16) rows = 16;
  dim3 grid((L + rows - 1) / rows, B), block(256);
  hipLaunchKernelGGL(v_absmax_kernel, grid, block, 0, st, (const unsigned short*)v, bs, rs, L, H, hd, rows,
                     reinterpret_cast<unsigned*>(scales));
  hipLaunchKernelGGL(v_scale_finalize_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<unsigned*>(scales), B * H);
  return (int)hipGetLastError();
}

extern "C" int osk_v_transpose_fp8(const void* v, int64_t bs, int64_t rs, const float* scales, void* vt8, int B, int L,
                                   int H, int hd, void* stream) {
  if (!v || !vt8 || !scales || B <= 0 || L <= 0 || H <= 0 || (bs & 7) || (rs & 7) || ((uintptr_t)vt8 & 15)) return OSK_EINVAL;
  const int Lp = (L + 63) / 64 * 64;
  dim3 grid(Lp / 64, H, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (hd == 72)
    hipLaunchKernelGGL(v_transpose_fp8_kernel<72>, grid, block, 0, st, (const unsigned short*)v, bs, rs, scales, (unsigned char*)vt8, L, Lp, H);
  else if (hd == 128)
    hipLaunchKernelGGL(v_transpose_fp8_kernel<128>, grid, block, 0, st, (const unsigned short*)v, bs, rs, scales, (unsigned char*)vt8, L, Lp, H);
  else
    return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

extern "C" int osk_attention_fwd_pv8_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride,
                                          const void* k, int64_t k_seg_stride, int64_t k_batch_stride,
                                          int64_t k_row_stride, const void* vt8, int64_t vt8_seg_stride,
                                          const float* v_scale, void* out, int64_t o_batch_stride, int64_t o_row_stride,
                                          float* lse, int B, int H, int Lq, int n_seg, int seg_len, int hd,
                                          float scale, int q_prescaled, int kv_batches, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (!q || !k || !vt8 || !v_scale || !out || B <= 0 || H <= 0 || Lq <= 0 || n_seg <= 0 || seg_len <= 0) return OSK_EINVAL;
  if ((q_batch_stride & 7) || (q_row_stride & 7) || (k_seg_stride & 7) || (k_batch_stride & 7) ||
      (k_row_stride & 7) || (vt8_seg_stride & 15) || (o_batch_stride & 3) || (o_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt8 & 15) || ((uintptr_t)out & 7)) return OSK_EINVAL;
  if (hd != 128 && hd != 72) return OSK_EUNSUPPORTED;
  osk_attn::AttnParams p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kss = k_seg_stride; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = nullptr; p.vt8 = (const unsigned char*)vt8; p.vtss = vt8_seg_stride; p.v_scale = v_scale;
  p.out = (unsigned short*)out; p.obs = o_batch_stride; p.ors = o_row_stride;
  p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.n_seg = n_seg; p.seg_len = seg_len;
  p.seg_lp = (seg_len + 63) / 64 * 64;
  p.tps = p.seg_lp / 64;
  p.sc = q_prescaled ? 1.0f : scale * 1.4426950408889634f;
  p.q_prescaled = q_prescaled;
  if (kv_batches < 0 || kv_batches > B) return OSK_EINVAL;
  p.Bkv = kv_batches > 0 ? kv_batches : B;
  p.map = 1;   // XCD-contiguous work order (each XCD walks one head's K / V^T stream)
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  osk_attn::split_tail(p, ((Lq + 255) / 256) * B * H, hd, workspace, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int rc = hd == 128 ? osk_attn::launch_asm128p8(p, st) : osk_attn::launch_asm72p8(p, st);
  return rc != 0 || p.tail_split == 1 ? rc : osk_attn::launch_merge(p, hd, st);
}
