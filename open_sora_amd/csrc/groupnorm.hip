// HBM-bound kernels of the causal 3-D VAE for gfx950, channels-last (NDHWC) bf16 activations:
//   GroupNorm statistics (one streaming read), GroupNorm apply + SiLU (one read, one write),
//   frame-causal masked row softmax of the mid-block attention scores.
// All accesses are 16 B per lane (8 bf16 channels of one voxel).
#include "osk_common.h"
#include "../../include/osk.h"

namespace {

// ---------------------------------------------------------------------------------------------
// statistics: sums[b][g] = (sum x, sum x^2) over the S voxels x (C/G) channels of group g, in f64.
// A 1024-thread block walks a contiguous slab of voxels; thread = (row-in-pass, 16-B channel chunk); per-thread f32
// partials, fixed-order reduction (wave shuffles, then across the 16 waves through LDS), one f64 atomic pair per
// (block, group).  FEW, FAT blocks on purpose: all blocks add into the same 2 G addresses, and same-address f64
// atomics retire at ~45 ns each -- with 2048 blocks per tensor that tail (~90 us) was longer than the streaming
// itself for every tensor below ~300 MB.
// Algorithmic bytes: 2 * S * C per batch item (read once).
// ---------------------------------------------------------------------------------------------
constexpr int GN_NT = 1024;

__global__ void __launch_bounds__(GN_NT) gn_stats_kernel(const unsigned short* __restrict__ x, int64_t S, int C, int G,
                                                        int rows_per_block, double* __restrict__ sums) {
  __shared__ float red[GN_NT / 64][512];
  const int b = blockIdx.y;
  const int cpr = C >> 3;            // 16-B chunks per voxel row (4..64, power of two)
  const int rpp = GN_NT / cpr;       // voxel rows per pass
  const int c = threadIdx.x % cpr, r = threadIdx.x / cpr;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  row1 = row1 < S ? row1 : S;
  const unsigned short* xb = x + (int64_t)b * S * C;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  // 4 independent 16-byte loads in flight per lane
  int64_t row = row0 + r;
  for (; row + 3 * rpp < row1; row += 4 * rpp) {
    uint4 u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const uint4*>(xb + (row + i * rpp) * C + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[8];
      unpack8(u[i], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
  }
  for (; row < row1; row += rpp) {
    const uint4 u = *reinterpret_cast<const uint4*>(xb + row * C + c * 8);
    float v[8];
    unpack8(u, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
  }
  // lanes of a wave that hold the same chunk are cpr apart
  for (int o = cpr; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float tot[2] = {0.f, 0.f};         // thread ch < C: the block's sums of channel ch
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (lane < cpr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = pass ? q[j] : s[j];
    }
    __syncthreads();
    if (threadIdx.x < C) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < GN_NT / 64; ++w) a += red[w][threadIdx.x];
      tot[pass] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x < C) {
    const int ch = threadIdx.x;
    float a = tot[0], a2 = tot[1];
    // channels of one group are adjacent lanes: reduce cpg = C/G lanes (power of two <= 16) by shuffles
    const int cpg = C / G;
    for (int o = cpg >> 1; o >= 1; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      a2 += __shfl_xor(a2, o, 64);
    }
    if ((ch & (cpg - 1)) == 0) {
      double* dst = sums + ((int64_t)b * G + ch / cpg) * 2;
      atomicAdd(dst, (double)a);
      atomicAdd(dst + 1, (double)a2);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// apply: y = bf16((x - mean_g) * rstd_g * gamma_c + beta_c);  out = silu ? bf16(y * sigmoid(y)) : y
// (two roundings, as torch's bf16 GroupNorm followed by bf16 SiLU: unet_causal_3d_blocks.py:250-254).
// Algorithmic bytes: 4 * S * C per batch item.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_apply_kernel(const unsigned short* __restrict__ x, const double* __restrict__ sums,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      unsigned short* __restrict__ out, int64_t S, int C, int G,
                                                      float eps, int do_silu, int rows_per_block) {
  __shared__ float sc[512], sh[512];
  const int b = blockIdx.y;
  const int cpg = C / G;
  const double cnt = (double)S * cpg;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    const double* sg = sums + ((int64_t)b * G + ch / cpg) * 2;
    const double mean = sg[0] / cnt;
    double var = sg[1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = rsqrtf((float)var + eps);
    const float a = rstd * gamma[ch];
    sc[ch] = a;
    sh[ch] = beta[ch] - (float)mean * a;
  }
  __syncthreads();
  const int cpr = C >> 3, rpp = 256 / cpr;
  const int c = threadIdx.x % cpr, r = threadIdx.x / cpr;
  float a[8], d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = sc[c * 8 + j]; d[j] = sh[c * 8 + j]; }
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  row1 = row1 < S ? row1 : S;
  const unsigned short* xb = x + (int64_t)b * S * C;
  unsigned short* ob = out + (int64_t)b * S * C;
  auto one = [&](const uint4& u) {
    float v[8];
    unpack8(u, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = bf16_bits_to_f32(f32_to_bf16_bits(v[j] * a[j] + d[j]));
      if (do_silu) y = silu(y);
      v[j] = y;
    }
    return pack8(v);
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto ld = [&](int64_t rw) -> uint4 {
    const u32x4* p = reinterpret_cast<const u32x4*>(xb + rw * C + c * 8);
    const u32x4 v = __builtin_nontemporal_load(p);   // streamed once: +7 % stand-alone on the 0.5 - 1 GB tensors, +0.15 % on the VAE
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  auto st = [&](int64_t rw, const uint4& v) {
    u32x4* p = reinterpret_cast<u32x4*>(ob + rw * C + c * 8);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, p);
  };
  int64_t row = row0 + r;
  for (; row + 3 * rpp < row1; row += 4 * rpp) {   // 4 independent loads in flight per lane, then 4 stores (8: no gain)
    uint4 u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = ld(row + i * rpp);
#pragma unroll
    for (int i = 0; i < 4; ++i) st(row + i * rpp, one(u[i]));
  }
  for (; row < row1; row += rpp) st(row, one(ld(row)));
}

// ---------------------------------------------------------------------------------------------
// P[i, j] = softmax_j(scale * s[i, j] + (frame(j) <= frame(i) ? 0 : -inf)), bf16 out, columns >= S_k zero-filled
// up to ldp.  One wave per row, two passes over the f32 scores (max, then exp-sum-write).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) masked_softmax_kernel(const float* __restrict__ s, int64_t lds_,
                                                            unsigned short* __restrict__ pr, int64_t ldp, int Sq,
                                                            int Sk, int n_hw, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Sq) return;
  const int kmax = n_hw > 0 ? ((row / n_hw + 1) * n_hw < Sk ? (row / n_hw + 1) * n_hw : Sk) : Sk;  // keys allowed
  const float* sr = s + (int64_t)row * lds_;
  unsigned short* po = pr + (int64_t)row * ldp;
  float mx = -INFINITY;
  for (int j = lane; j < kmax; j += 64) mx = fmaxf(mx, sr[j]);
  mx = wave_max(mx) * scale;
  float sum = 0.f;
  for (int j = lane; j < kmax; j += 64) sum += __expf(sr[j] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < (int)ldp; j += 64) {
    const float e = j < kmax ? __expf(sr[j] * scale - mx) * inv : 0.f;
    po[j] = f32_to_bf16_bits(e);
  }
}

}  // namespace

namespace {
__global__ void gn_zero_kernel(double* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0;
}
}  // namespace

extern "C" int osk_groupnorm_stats_ndhwc_bf16(const void* x, int B, int64_t S, int C, int G, double* sums,
                                              void* stream) {
  if (!x || !sums || B <= 0 || S <= 0 || C < 32 || C > 512 || G <= 0) return OSK_EINVAL;
  if ((C & (C - 1)) || C % G || ((C / G) & (C / G - 1)) || C / G > 16 || ((uintptr_t)x & 15)) return OSK_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // (zeroed by a kernel, not hipMemsetAsync: a memset of a few bytes was not replayed from a captured hipGraph -- round 6, attention_fwd.hip)
  hipLaunchKernelGGL(gn_zero_kernel, dim3((2 * B * G + 255) / 256), dim3(256), 0, st, sums, 2 * B * G);
  const int rpp = GN_NT / (C >> 3);
  // 256 blocks per batch item (one per CU) for small tensors, 512 above 64 MB; at least one pass of the 4-deep
  // unrolled loop per block
  const int64_t target = S * C * 2 < (int64_t)64 << 20 ? 256 : 512;
  int64_t rows = (S + target - 1) / target;
  if (rows < 4 * rpp) rows = 4 * rpp;
  rows = (rows + rpp - 1) / rpp * rpp;
  const int64_t nblk = (S + rows - 1) / rows;
  dim3 grid((unsigned)nblk, B), block(GN_NT);
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, st, (const unsigned short*)x, S, C, G, (int)rows, sums);
  return (int)hipGetLastError();
}

extern "C" int osk_groupnorm_apply_ndhwc_bf16(const void* x, const double* sums, const float* gamma,
                                              const float* beta, void* out, int B, int64_t S, int C, int G,
                                              float eps, int silu, void* stream) {
  if (!x || !sums || !gamma || !beta || !out || B <= 0 || S <= 0 || C < 32 || C > 512 || G <= 0) return OSK_EINVAL;
  if ((C & (C - 1)) || C % G || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return OSK_EUNSUPPORTED;
  const int rpp = 256 / (C >> 3);
  int64_t rows = (S + 4095) / 4096;           // about 4096 blocks per batch item, >= one 4-deep unrolled pass each
  if (rows < 4 * rpp) rows = 4 * rpp;
  rows = (rows + rpp - 1) / rpp * rpp;
  int64_t nblk = (S + rows - 1) / rows;
  dim3 grid((unsigned)nblk, B), block(256);
  hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, (hipStream_t)stream, (const unsigned short*)x, sums, gamma, beta,
                     (unsigned short*)out, S, C, G, eps, silu, (int)rows);
  return (int)hipGetLastError();
}

// (scale, shift) rows for the conv kernels that fold GroupNorm + SiLU into their input path (conv3d_256.hip, GN form): the same
// a = rstd gamma, d = beta - mean a as gn_apply_kernel's prologue, laid out [B][C / 8][8 a | 8 d]
namespace {
__global__ void __launch_bounds__(256) gn_table_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ table, int64_t S,
                                                      int C, int G, float eps) {
  const int b = blockIdx.x;
  const int cpg = C / G;
  const double cnt = (double)S * cpg;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    const double* sg = sums + ((int64_t)b * G + ch / cpg) * 2;
    const double mean = sg[0] / cnt;
    double var = sg[1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = rsqrtf((float)var + eps);
    const float a = rstd * gamma[ch];
    float* row = table + ((int64_t)b * (C >> 3) + (ch >> 3)) * 16 + (ch & 7);
    row[0] = a;
    row[8] = beta[ch] - (float)mean * a;
  }
}
}  // namespace

extern "C" int osk_groupnorm_table_f32(const double* sums, const float* gamma, const float* beta, float* table, int B,
                                       int64_t S, int C, int G, float eps, void* stream) {
  if (!sums || !gamma || !beta || !table || B <= 0 || S <= 0 || C <= 0 || G <= 0 || C % G || (C & 7)) return OSK_EINVAL;
  if (((uintptr_t)table & 15) || ((uintptr_t)sums & 7)) return OSK_EINVAL;
  hipLaunchKernelGGL(gn_table_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, table, S, C, G, eps);
  return (int)hipGetLastError();
}

extern "C" int osk_masked_softmax_f32_bf16(const float* scores, int64_t ld_scores, void* probs, int64_t ld_probs,
                                           int Sq, int Sk, int keys_per_frame, float scale, void* stream) {
  if (!scores || !probs || Sq <= 0 || Sk <= 0 || ld_scores < Sk || ld_probs < Sk || keys_per_frame < 0) return OSK_EINVAL;
  dim3 grid((Sq + 3) / 4), block(256);
  hipLaunchKernelGGL(masked_softmax_kernel, grid, block, 0, (hipStream_t)stream, scores, ld_scores,
                     (unsigned short*)probs, ld_probs, Sq, Sk, keys_per_frame, scale);
  return (int)hipGetLastError();
}

// =============================================================================================
// Tile cross-fade of the tiled VAE encode / decode (blend_v / blend_h / blend_t,
// /root/reference/opensora/models/hunyuan_vae/autoencoder_kl_causal_3d.py:360-382), in place in b:
//   b[o, e, i] = a[o, Da - extent + e, i] * (1 - e / extent) + b[o, e, i] * (e / extent),   e < extent
// a, b contiguous bf16 viewed as [outer, Da | Db, inner] along the blended axis; f32 math, one rounding.
// HBM-bound (3 * 2 bytes per blended element), one launch per seam instead of `extent` Python-level slice ops.
// =============================================================================================
namespace {
template <int VEC>
__global__ void __launch_bounds__(256) blend_kernel(const unsigned short* __restrict__ a, unsigned short* __restrict__ b,
                                                    int64_t outer, int Da, int Db, int extent, int64_t inner) {
  const int64_t per_o = (int64_t)extent * inner / VEC;
  const int64_t total = outer * per_o;
  const float inv = 1.0f / (float)extent;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t o = idx / per_o, r = idx - o * per_o;
    const int64_t e = r * VEC / inner, i = r * VEC - e * inner;
    const float wb = (float)e * inv, wa = 1.0f - wb;
    const unsigned short* pa = a + (o * Da + (Da - extent + e)) * inner + i;
    unsigned short* pb = b + (o * Db + e) * inner + i;
    if constexpr (VEC == 8) {
      const uint4 ua = *reinterpret_cast<const uint4*>(pa);
      const uint4 ub = *reinterpret_cast<const uint4*>(pb);
      float fa[8], fb[8];
      unpack8(ua, fa);
      unpack8(ub, fb);
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = fa[j] * wa + fb[j] * wb;
      *reinterpret_cast<uint4*>(pb) = pack8(fb);
    } else {
      *pb = f32_to_bf16_bits(bf16_bits_to_f32(*pa) * wa + bf16_bits_to_f32(*pb) * wb);
    }
  }
}
}  // namespace

extern "C" int osk_blend_bf16(const void* a, void* b, int64_t outer, int Da, int Db, int extent, int64_t inner,
                              void* stream) {
  if (!a || !b || outer <= 0 || inner <= 0 || extent < 0 || extent > Da || extent > Db) return OSK_EINVAL;
  if (extent == 0) return OSK_OK;
  const bool vec = inner % 8 == 0 && !((uintptr_t)a & 15) && !((uintptr_t)b & 15);
  const int64_t total = outer * extent * inner / (vec ? 8 : 1);
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (vec)
    hipLaunchKernelGGL(blend_kernel<8>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (unsigned short*)b, outer, Da, Db, extent, inner);
  else
    hipLaunchKernelGGL(blend_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (unsigned short*)b, outer, Da, Db, extent, inner);
  return (int)hipGetLastError();
}
