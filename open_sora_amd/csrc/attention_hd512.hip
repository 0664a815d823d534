// Flash attention for the causal VAE's mid-block: ONE head of dimension 512, frame-causal mask, bf16 in / out.
//
// Replaces diffusers Attention + prepare_causal_attention_mask inside UNetMidBlockCausal3D
// (/root/reference/opensora/models/hunyuan_vae/unet_causal_3d_blocks.py:52-60, 312-351): the reference materialises an
// S x S additive mask and lets SDPA materialise the scores; round 1 of this library ran QK^T and P.V as GEMMs around a
// masked-softmax kernel with an S x S f32 score matrix (340 MB at 33 x 256 x 256) in HBM.  Here scores never leave
// registers and the mask is a predicate: key j is visible to query i iff  j / keys_per_frame <= i / keys_per_frame.
//
// Layout (the conventions of attention_fwd.hip): both products are issued "swapped" on v_mfma_f32_32x32x16_bf16 so a
// lane owns ONE query column:  S^T[key][q] = K[key][:] . Q^T  and  O^T[d][q] += V^T[d][key] . P^T.
// Workgroup = 4 waves (one per SIMD) x 32 queries; key tile = 32 keys.  The 512 output dims are produced in TWO halves
// of 256 (O^T = 8 row tiles x 16 = 128 accumulators per half; with Q's 32 k-steps x 4 = 128 registers a one-pass
// kernel needs 256 + 128 + working registers and spills): each half recomputes QK^T and the softmax statistics
// bit-identically -- 1.5x the MFMAs of a kernel that is 0.3 % of the VAE's work, for no scratch traffic.  The halves are
// separate workgroups (blockIdx.z): at 33 x 256 x 256 the launch is 72 query blocks -- a quarter of the chip -- and its
// time is the chain of 288 key tiles of the last frame's query block, so the second half costs nothing when it runs
// beside the first (1.52 -> 0.8 ms); query blocks are launched last-frame-first (longest chain first).
//   LDS: K tile [32 keys][1 KiB] + V^T half tile [256 dims][64 B], double buffered (96 KiB), filled by LDS-DMA
//        (global_load_lds_dwordx4) with source-side XOR swizzles: K chunk ^ (key & 15), V^T chunk ^ ((dim >> 2) & 3)
//        -> conflict-free ds_read_b128 fragment reads for both row strides.
// V^T comes in natural key order ([512][ld] from the V projection GEMM, V^T = W_v x^T): the accumulator order of a
// lane's 16 scores (keys 8g + 4hi + j) is turned into the B-operand order (8 consecutive keys per half-wave) by one
// v_permlane32_swap pair per 16-key step.  Online softmax in f32 (base 2); O is rescaled only when some lane's
// running maximum moved (wave-uniform test).  The V bias is added after normalisation (softmax rows sum to one).
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 4 * B * S_q * S_k(visible) * 512 -- 0.3 % of the VAE's work; the point of
// this kernel is the memory it does not touch.
#include "osk_common.h"
#include "../../include/osk.h"

namespace {

constexpr int HD = 512, QW = 32, NW = 4, QB = QW * NW, KT = 32;
constexpr int NKS = HD / 16;          // QK^T k-steps
constexpr int NPASS = 2;              // passes over the head dim (output dims [256 pass, 256 pass + 256))
constexpr int NDT = HD / 32 / NPASS;  // O^T row tiles per pass
constexpr int KTILE = KT * HD * 2;    // 32 KiB
constexpr int VTILE = HD / NPASS * KT * 2;    // 16 KiB
constexpr int BUF = KTILE + VTILE;
constexpr int SMEM = 2 * BUF;

struct P512 {
  const unsigned short* q; int64_t qbs, qrs;
  const unsigned short* k; int64_t kbs, krs;
  const unsigned short* vt; int64_t vbs, vrs;     // [B][512][vrs], natural key order, zero beyond S
  const float* bias_v;
  unsigned short* out; int64_t obs, ors;
  int S, kpf;
  float scale_log2;
  // key split (round 4): every (query block, dim pass) is cut into `parts` runs of key tiles that run as separate workgroups and
  // write normalised partial rows (f32) + log2-domain LSE into the workspace; attn_hd512_merge_kernel combines them
  int parts = 1;
  float* ws_o = nullptr;     // [B][query blocks][parts][QB rows][512]
  float* ws_lse = nullptr;   // [B][query blocks][parts][QB rows]
};

OSK_DEV void glds16(const unsigned short* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ void __launch_bounds__(256, 1) attn_hd512_kernel(const P512 p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.y;
  const int q0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * QB;   // longest key range first
  const int pass = (int)blockIdx.z % NPASS, part = (int)blockIdx.z / NPASS;
  const int qi = q0 + wave * QW + l31;
  const int qc = qi < p.S ? qi : p.S - 1;

  // ---- Q fragments (B operand of QK^T): dims 16 ks + 8 hi .. + 8 of this lane's query
  bf16x8_t qf[NKS];
  {
    const unsigned short* qrow = p.q + b * p.qbs + (int64_t)qc * p.qrs;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      qf[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qrow + ks * 16 + hi * 8));
  }
  // visible keys: j < limit(query); the workgroup walks key tiles up to the limit of its last query
  const int kpf = p.kpf > 0 ? p.kpf : p.S;
  int my_limit = (qc / kpf + 1) * kpf;
  my_limit = my_limit < p.S ? my_limit : p.S;
  int q_last = q0 + QB - 1;
  q_last = q_last < p.S ? q_last : p.S - 1;
  int wg_limit = (q_last / kpf + 1) * kpf;
  wg_limit = wg_limit < p.S ? wg_limit : p.S;
  const int ntiles_all = (wg_limit + KT - 1) / KT;
  // this workgroup's run of key tiles [t_first, t_first + ntiles)
  const int t_first = (int)((int64_t)part * ntiles_all / p.parts);
  const int ntiles = (int)((int64_t)(part + 1) * ntiles_all / p.parts) - t_first;

  const unsigned short* kb = p.k + b * p.kbs;
  const unsigned short* vb = p.vt + b * p.vbs;
  // LDS-DMA issue of key tile t into buffer bi: wave w loads K rows w, w+4, .. and V^T row blocks w, w+4, ..
  auto issue = [&](int t, int bi, int pass) {
    unsigned char* kbuf = smem + bi * BUF;
    unsigned char* vbuf = kbuf + KTILE;
    const int key0 = (t_first + t) * KT;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave + 4 * i;                                  // key row of the tile: one 1 KiB row per instruction
      int key = key0 + row;
      key = key < p.S ? key : p.S - 1;
      glds16(kb + (int64_t)key * p.krs + ((lane ^ (row & 15)) << 3), kbuf + row * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave + 4 * i;                                    // 16 V^T rows (dims) x 64 B per instruction
      const int row = j * 16 + (lane >> 2);                          // row inside the pass's 256-dim half
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      glds16(vb + (int64_t)(pass * (HD / NPASS) + row) * p.vrs + key0 + c * 8, vbuf + j * 1024);
    }
  };

  const unsigned ksw = (unsigned)(l31 & 15), vsw = (unsigned)((l31 >> 2) & 3);
  f32x16_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;   // running max (shared by the two half-waves of a query), this lane's partial sum

  if (ntiles > 0) issue(0, 0, pass);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) issue(t + 1, cur ^ 1, pass);
    const unsigned char* kbuf = smem + cur * BUF;
    const unsigned char* vbuf = kbuf + KTILE;
    // ---- S^T = K . Q^T  (32 keys x 32 queries)
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kbuf + l31 * 1024 + (((unsigned)(2 * ks + hi) ^ ksw) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
    }
    // ---- mask + online softmax: register r = 4 g + j of this lane is key 8 g + 4 hi + j of the tile
    const int key0 = (t_first + t) * KT;
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + 8 * (r >> 2) + 4 * hi + (r & 3);
      s[r] = key < my_limit ? s[r] * p.scale_log2 : -1e30f;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {     // some query's maximum moved: rescale this wave's O
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m_run = m_new;
    }
    unsigned pk[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float e0 = __builtin_amdgcn_exp2f(s[r] - m_run), e1 = __builtin_amdgcn_exp2f(s[r + 1] - m_run);
      l_run += e0 + e1;
      pk[r >> 1] = pack_bf16x2(e0, e1);
    }
    // ---- O^T += V^T . P^T: per 16-key step the half-waves trade one 4-key group so that each holds 8 consecutive keys
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      // a lane holds two 4-key groups of the step: G_a = keys 4 hi + j, G_b = keys 8 + 4 hi + j.  v_permlane32_swap(G_a, G_b)
      // returns [0]: lower lanes own G_a, upper lanes the lower partner's G_b;  [1]: lower lanes the upper partner's G_a,
      // upper lanes own G_b -- i.e. ([0], [1]) = keys 8 hi + 0..3, 8 hi + 4..7: the B-operand order, for both half-waves
      auto x0 = __builtin_amdgcn_permlane32_swap(pk[4 * st + 0], pk[4 * st + 2], false, false);
      auto x1 = __builtin_amdgcn_permlane32_swap(pk[4 * st + 1], pk[4 * st + 3], false, false);
      const uint4 pu = make_uint4(x0[0], x1[0], x0[1], x1[1]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vbuf + (d * 32 + l31) * 64 + (((unsigned)(2 * st + hi) ^ vsw) << 4));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- normalise, add the V bias, store: register r = 4 g + j of row tile d is dim 32 d + 8 g + 4 hi + j of this query
  // a key part that lies entirely behind this row's visible keys (only possible when a query block spans two frames): every score
  // was the mask value AND so was the running maximum, i.e. P = exp2(0) = 1 for masked keys -- the part must weigh nothing
  const bool dead = ntiles == 0 || (int64_t)t_first * KT >= my_limit;
  const float l_sum = l_run + __shfl_xor(l_run, 32, 64);
  const float l_tot = dead ? 0.f : l_sum;
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (p.parts > 1) {
    // partial result of this key part: normalised rows in f32 + LSE (log2 units) -> workspace
    const int64_t slot = (((int64_t)b * gridDim.x + blockIdx.x) * p.parts + part) * QB + wave * QW + l31;
    float* wo = p.ws_o + slot * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dim = pass * (HD / NPASS) + 32 * d + 8 * g + 4 * hi;
        *reinterpret_cast<float4*>(wo + dim) = make_float4(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
      }
    if (pass == 0 && hi == 0) p.ws_lse[slot] = l_tot > 0.f ? m_run + __builtin_amdgcn_logf(l_tot) : -1e30f;
    return;
  }
  if (qi < p.S) {
    unsigned short* orow = p.out + b * p.obs + (int64_t)qi * p.ors;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dim = pass * (HD / NPASS) + 32 * d + 8 * g + 4 * hi;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias_v) bv = *reinterpret_cast<const float4*>(p.bias_v + dim);
        uint2 u;
        u.x = pack_bf16x2(o[d][4 * g + 0] * inv + bv.x, o[d][4 * g + 1] * inv + bv.y);
        u.y = pack_bf16x2(o[d][4 * g + 2] * inv + bv.z, o[d][4 * g + 3] * inv + bv.w);
        *reinterpret_cast<uint2*>(orow + dim) = u;
      }
  }
}

// out = sum_p 2^(lse_p - lse) O_p + bias_v over the key parts of a query block (lse = log2 sum_p 2^lse_p); grid (query blocks, B,
// MSPLIT row groups), thread = (row, float4 chunk of the 512 dims): the partial rows are contiguous f32 [row][512] per part.
// HBM-bound: parts x 2 KiB read + 1 KiB written per row; the row groups are there to put > 256 workgroups on the chip.
constexpr int MSPLIT = 8, MROWS = QB / MSPLIT;
__global__ void __launch_bounds__(256) attn_hd512_merge_kernel(const P512 p) {
  const int b = blockIdx.y;
  const int q0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * QB;
  const int64_t base = ((int64_t)b * gridDim.x + blockIdx.x) * p.parts * QB;
  for (int i = (int)blockIdx.z * MROWS * (HD / 4) + threadIdx.x; i < ((int)blockIdx.z + 1) * MROWS * (HD / 4); i += 256) {
    const int r = i / (HD / 4), c = i - r * (HD / 4);
    const int qi = q0 + r;
    if (qi >= p.S) continue;
    float w[8], m = -1e30f;
    for (int s = 0; s < p.parts; ++s) { w[s] = p.ws_lse[base + (int64_t)s * QB + r]; m = fmaxf(m, w[s]); }
    float tot = 0.f;
    for (int s = 0; s < p.parts; ++s) { w[s] = __builtin_amdgcn_exp2f(w[s] - m); tot += w[s]; }
    const float inv = 1.0f / tot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.parts; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(p.ws_o + (base + (int64_t)s * QB + r) * HD + c * 4);
      acc.x += w[s] * v.x; acc.y += w[s] * v.y; acc.z += w[s] * v.z; acc.w += w[s] * v.w;
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias_v) bv = *reinterpret_cast<const float4*>(p.bias_v + c * 4);
    uint2 u;
    u.x = pack_bf16x2(acc.x * inv + bv.x, acc.y * inv + bv.y);
    u.y = pack_bf16x2(acc.z * inv + bv.z, acc.w * inv + bv.w);
    *reinterpret_cast<uint2*>(p.out + b * p.obs + (int64_t)qi * p.ors + c * 4) = u;
  }
}

}  // namespace

extern "C" int osk_attention_hd512_fwd_ws_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k,
                                               int64_t k_batch_stride, int64_t k_row_stride, const void* vt,
                                               int64_t vt_batch_stride, int64_t vt_row_stride, const float* bias_v, void* out,
                                               int64_t out_batch_stride, int64_t out_row_stride, int B, int S,
                                               int keys_per_frame, float scale, void* workspace, int64_t workspace_bytes,
                                               void* stream) {
  if (!q || !k || !vt || !out || B <= 0 || S <= 0 || keys_per_frame < 0) return OSK_EINVAL;
  if ((q_row_stride & 7) || (k_row_stride & 7) || (vt_row_stride & 7) || (out_row_stride & 3) || (q_batch_stride & 7) ||
      (k_batch_stride & 7) || (vt_batch_stride & 7) || (out_batch_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7) || ((uintptr_t)bias_v & 15))
    return OSK_EINVAL;
  if (vt_row_stride < (int64_t)((S + KT - 1) / KT) * KT) return OSK_EINVAL;   // whole 32-key tiles are fetched
  if (workspace && (((uintptr_t)workspace & 15) || workspace_bytes < 0)) return OSK_EINVAL;
  OSK_ENSURE_MAX_SMEM(attn_hd512_kernel, SMEM);
  P512 p;
  p.q = (const unsigned short*)q; p.qbs = q_batch_stride; p.qrs = q_row_stride;
  p.k = (const unsigned short*)k; p.kbs = k_batch_stride; p.krs = k_row_stride;
  p.vt = (const unsigned short*)vt; p.vbs = vt_batch_stride; p.vrs = vt_row_stride;
  p.bias_v = bias_v;
  p.out = (unsigned short*)out; p.obs = out_batch_stride; p.ors = out_row_stride;
  p.S = S; p.kpf = keys_per_frame;
  p.scale_log2 = scale * 1.4426950408889634f;
  const int nqb = (S + QB - 1) / QB;
  // The launch is nqb x B x 2 workgroups and its time is the key-tile chain of the longest one (at 33 x 256 x 256: 144 workgroups
  // on 256 CUs, 288 tiles).  With a workspace the chains are cut into up to 4 parts of >= 64 tiles that run side by side.
  const int longest = (S + KT - 1) / KT;
  int parts = longest / 64;
  parts = parts < 1 ? 1 : (parts > 4 ? 4 : parts);
  while (parts > 1 && (!workspace || (int64_t)B * nqb * parts * QB * (HD + 1) * 4 > workspace_bytes)) --parts;
  p.parts = parts;
  if (parts > 1) {
    p.ws_o = (float*)workspace;
    p.ws_lse = p.ws_o + (int64_t)B * nqb * parts * QB * HD;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nqb, B, NPASS * parts), block(256);
  hipLaunchKernelGGL(attn_hd512_kernel, grid, block, SMEM, st, p);
  if (parts > 1) hipLaunchKernelGGL(attn_hd512_merge_kernel, dim3(nqb, B, MSPLIT), dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

extern "C" int64_t osk_attention_hd512_workspace_bytes(int B, int S) {
  if (B <= 0 || S <= 0) return 0;
  return (int64_t)B * ((S + QB - 1) / QB) * 4 * QB * (HD + 1) * 4;
}

extern "C" int osk_attention_hd512_fwd_bf16(const void* q, int64_t q_batch_stride, int64_t q_row_stride, const void* k,
                                            int64_t k_batch_stride, int64_t k_row_stride, const void* vt,
                                            int64_t vt_batch_stride, int64_t vt_row_stride, const float* bias_v, void* out,
                                            int64_t out_batch_stride, int64_t out_row_stride, int B, int S,
                                            int keys_per_frame, float scale, void* stream) {
  return osk_attention_hd512_fwd_ws_bf16(q, q_batch_stride, q_row_stride, k, k_batch_stride, k_row_stride, vt, vt_batch_stride,
                                         vt_row_stride, bias_v, out, out_batch_stride, out_row_stride, B, S, keys_per_frame, scale,
                                         nullptr, 0, stream);
}
