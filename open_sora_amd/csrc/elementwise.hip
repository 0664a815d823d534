// HBM-bound kernels of the MMDiT denoise step for gfx950: LayerNorm+modulate, QK-RMSNorm+RoPE,
// V transpose, skinny GEMV task list (adaLN modulation / embedders), timestep embedding, RoPE tables,
// CFG + Euler update.  All loads/stores are 16 B per lane (8 bf16) where the layout allows (guide G13).
#include "osk_common.h"
#include "../../include/osk.h"

// =============================================================================================
// LayerNorm (no affine) + modulate.  One wave per row, row kept in registers (<= 8 chunks of 8 per
// lane -> D <= 4096), two-pass mean/variance in f32.  Algorithmic bytes: 4*D per row (2 in, 2 out).
// =============================================================================================
// FP8 instantiation (osk_ln_modulate_fp8, opt-in fp8 mode): the modulated row is rounded to bf16 exactly as above, then
// quantised like osk_quantize_rows_fp8 (absmax / 448 per row, e4m3) without leaving the registers: out = e4m3 bytes
// [M, D] contiguous, row scales to scales8[M].  Bytes: 3*D per row instead of 4*D + 3*D for the two kernels.
template <int MAXC, bool FP8 = false>
__global__ void __launch_bounds__(256) ln_modulate_kernel(
    const unsigned short* __restrict__ x, int64_t xbs, int64_t xrs, unsigned short* __restrict__ out,
    int64_t obs, int64_t ors, const float* __restrict__ shift, const float* __restrict__ scale,
    int64_t mbs, int M, int L, int D, float eps, float* __restrict__ scales8 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = row / L, l = row - b * L;
  const unsigned short* xr = x + b * xbs + l * xrs;
  unsigned short* orow = out + b * obs + l * ors;
  const int nchunk = D >> 3;
  // Every load of the row is unconditional with a clamped chunk index (lanes past the row re-read its last chunk and are masked
  // out of the sums and the stores): a load predicated together with its use waits inside the branch, so the row's 2-8 chunks
  // and the modulation vectors became dependent round trips and a wave never had more than 1 KB in flight (round 5).
  float v[MAXC][8];
  uint4 u[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + i * 64;
    u[i] = *reinterpret_cast<const uint4*>(xr + (c < nchunk ? c : nchunk - 1) * 8);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const bool in = lane + i * 64 < nchunk;
    unpack8(u[i], v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += in ? v[i][j] : 0.f;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const bool in = lane + i * 64 < nchunk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[i][j] - mean;
      q += in ? d * d : 0.f;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  const float* sh = shift + b * mbs;
  const float* sc = scale + b * mbs;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + i * 64;
    const int cc = c < nchunk ? c : nchunk - 1;
    const float4 s0 = *reinterpret_cast<const float4*>(sc + cc * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(sc + cc * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(sh + cc * 8);
    const float4 h1 = *reinterpret_cast<const float4*>(sh + cc * 8 + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (1.0f + scv[j]) * ((v[i][j] - mean) * rstd) + shv[j];
    if constexpr (FP8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = bf16_bits_to_f32(f32_to_bf16_bits(o[j]));   // the bf16 value the GEMM would read
    } else {
      if (c < nchunk) *reinterpret_cast<uint4*>(orow + c * 8) = pack8(o);
    }
  }
  if constexpr (FP8) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (lane + i * 64 < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
      }
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.0f / amax : 0.f;
    if (lane == 0) scales8[row] = amax > 0.f ? amax / 448.0f : 1.0f;
    unsigned char* o8 = reinterpret_cast<unsigned char*>(out) + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[i][j] * inv, -448.0f), 448.0f);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        *reinterpret_cast<uint2*>(o8 + c * 8) = make_uint2((unsigned)w0, (unsigned)w1);
      }
    }
  }
}

extern "C" int osk_ln_modulate_bf16(const void* x, int64_t xbs, int64_t xrs, void* out, int64_t obs,
                                    int64_t ors, const float* shift, const float* scale, int64_t mbs,
                                    int B, int L, int D, float eps, void* stream) {
  if (!x || !out || !shift || !scale || B <= 0 || L <= 0 || D <= 0) return OSK_EINVAL;
  if ((D & 7) || D > 8 * 64 * 8 || (xrs & 7) || (ors & 7) || (xbs & 7) || (obs & 7) || (mbs & 3)) return OSK_EINVAL;
  const int M = B * L;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nch = (D / 8 + 63) / 64;
#define LAUNCH(MC)                                                                              \
  hipLaunchKernelGGL((ln_modulate_kernel<MC, false>), grid, block, 0, st, (const unsigned short*)x, xbs, xrs, \
                     (unsigned short*)out, obs, ors, shift, scale, mbs, M, L, D, eps, (float*)nullptr)
  if (nch <= 1) LAUNCH(1);
  else if (nch <= 2) LAUNCH(2);
  else if (nch <= 3) LAUNCH(3);
  else if (nch <= 4) LAUNCH(4);
  else if (nch <= 6) LAUNCH(6);
  else LAUNCH(8);
#undef LAUNCH
  return (int)hipGetLastError();
}

extern "C" int osk_ln_modulate_fp8(const void* x, int64_t xbs, int64_t xrs, void* out8, float* scales,
                                   const float* shift, const float* scale, int64_t mbs, int B, int L, int D, float eps,
                                   void* stream) {
  if (!x || !out8 || !scales || !shift || !scale || B <= 0 || L <= 0 || D <= 0) return OSK_EINVAL;
  if ((D & 7) || D > 8 * 64 * 8 || (xrs & 7) || (xbs & 7) || (mbs & 3) || ((uintptr_t)out8 & 7)) return OSK_EINVAL;
  const int M = B * L;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nch = (D / 8 + 63) / 64;
#define LAUNCH8(MC)                                                                                     \
  hipLaunchKernelGGL((ln_modulate_kernel<MC, true>), grid, block, 0, st, (const unsigned short*)x, xbs, xrs, \
                     (unsigned short*)out8, (int64_t)0, (int64_t)0, shift, scale, mbs, M, L, D, eps, scales)
  if (nch <= 1) LAUNCH8(1);
  else if (nch <= 2) LAUNCH8(2);
  else if (nch <= 3) LAUNCH8(3);
  else if (nch <= 4) LAUNCH8(4);
  else if (nch <= 6) LAUNCH8(6);
  else LAUNCH8(8);
#undef LAUNCH8
  return (int)hipGetLastError();
}

// =============================================================================================
// QK RMSNorm + RoPE, in place.  A group of LPR lanes owns one (token, head) row of hd elements,
// CW elements per lane.  rope_mode 0: pairs (2j, 2j+1) are lane-local.  rope_mode 1: pair (j, j+hd/2)
// lives NCH/2 lanes away inside the group -> one shuffle per element.
// Algorithmic bytes: 2 tensors * (read + write) * 2 B = 8*hd per (token, head) + the cos/sin rows.
// =============================================================================================
template <int HD, int CW, int MODE>
__global__ void __launch_bounds__(256) qknorm_rope_kernel(
    unsigned short* __restrict__ q, unsigned short* __restrict__ k, int64_t bs, int64_t rs,
    const unsigned short* __restrict__ qs0, const unsigned short* __restrict__ ks0,
    const unsigned short* __restrict__ qs1, const unsigned short* __restrict__ ks1, int l_split,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, int64_t csb, int B, int L, int H,
    float eps, float q_mult) {
  constexpr int NCH = HD / CW;  // active lanes per row
  constexpr int LPR = NCH <= 8 ? 8 : (NCH <= 16 ? 16 : 32);
  constexpr int RPB = 256 / LPR;  // rows per block
  const int g = threadIdx.x / LPR, c = threadIdx.x % LPR;
  const int64_t ridx = (int64_t)blockIdx.x * RPB + g;  // over B*L*H
  const int64_t total = (int64_t)B * L * H;
  const bool rvalid = ridx < total;
  const int64_t rr = rvalid ? ridx : total - 1;
  const int h = (int)(rr % H);
  const int64_t tok = rr / H;
  const int l = (int)(tok % L), b = (int)(tok / L);
  const bool act = c < NCH;
  const int cc = act ? c : 0;
  const float* cr = cos_t + b * csb + (int64_t)l * (HD / 2);
  const float* sr = sin_t + b * csb + (int64_t)l * (HD / 2);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    unsigned short* tb = which ? k : q;
    if (!tb) continue;  // one-sided call (sequence-parallel path norms K before Q); uniform over the grid
    unsigned short* p = tb + b * bs + (int64_t)l * rs + h * HD + cc * CW;
    const unsigned short* sc = which ? (l < l_split ? ks0 : ks1) : (l < l_split ? qs0 : qs1);
    float v[CW];
    if constexpr (CW == 8) {
      uint4 u = *reinterpret_cast<const uint4*>(p);
      unpack8(u, v);
    } else {
      uint2 u = *reinterpret_cast<const uint2*>(p);
      v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CW; ++j) ss += act ? v[j] * v[j] : 0.f;
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rrms = rsqrtf(ss / (float)HD + eps);
    float y[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) {
      // reference rounding points: (x*rrms).to(bf16) * scale(bf16) -> bf16   (layers.py:107-111)
      const float t = bf16_bits_to_f32(f32_to_bf16_bits(v[j] * rrms));
      const float w = bf16_bits_to_f32(sc[cc * CW + j]);
      y[j] = bf16_bits_to_f32(f32_to_bf16_bits(t * w));
    }
    float o[CW];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < CW; j += 2) {
        const int pj = (cc * CW + j) >> 1;
        const float cs = cr[pj], sn = sr[pj];
        o[j] = cs * y[j] - sn * y[j + 1];
        o[j + 1] = sn * y[j] + cs * y[j + 1];
      }
    } else {
      // partner chunk NCH/2 lanes further (mod NCH) inside this row's lane group
      const int pc = (cc + NCH / 2) % NCH;
      const int src_lane = (threadIdx.x & 63) - c + pc;
      const bool first = cc < NCH / 2;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const float other = __shfl(y[j], src_lane, 64);
        const int pj = (first ? cc : pc) * CW + j;  // index in [0, hd/2)
        const float cs = cr[pj], sn = sr[pj];
        o[j] = first ? (y[j] * cs - other * sn) : (y[j] * cs + other * sn);
      }
    }
    if (which == 0) {  // softmax scale * log2(e) folded into q BEFORE its one rounding to bf16 (see osk.h)
#pragma unroll
      for (int j = 0; j < CW; ++j) o[j] *= q_mult;
    }
    if (act && rvalid) {
      if constexpr (CW == 8) {
        *reinterpret_cast<uint4*>(p) = pack8(o);
      } else {
        uint2 u;
        u.x = pack_bf16x2(o[0], o[1]);
        u.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p) = u;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-per-thread variant (default).  A block owns TPB = NT / H consecutive tokens = TPB * H (token, head) rows.
// Global traffic is fully coalesced both ways: the TPB contiguous [H * hd] spans are copied to LDS with 16-byte loads
// by all lanes, every thread then norms + rotates ONE row out of LDS entirely in registers (no cross-lane reduction,
// both RoPE conventions lane-local), writes it back to LDS, and the spans leave with 16-byte stores.  cos / sin of
// the block's tokens and this tensor's two scale vectors (as f32) are staged in LDS once.  Same rounding points as above.
// LDS row stride hd*2 (+16 when hd % 32 == 0) bytes keeps the per-thread 16-byte row reads off each other's banks.
// ONE global round trip per block (round 5): the scale, cos / sin and row loads are all issued into registers before the first
// of them is waited for -- round 4's three dependent trips (scales -> barrier -> cos / sin -> barrier -> rows) left the
// 7 resident blocks of a CU waiting on memory for most of their life (VALU ~35 % busy at 0.42 of the HBM roofline).
// ---------------------------------------------------------------------------------------------
template <int HD, int MODE, int NT>   // NT threads = NT (token, head) rows per block
__global__ void __launch_bounds__(NT) qknorm_rope_rows_kernel(
    unsigned short* __restrict__ q, unsigned short* __restrict__ k, int64_t bs, int64_t rs,
    const unsigned short* __restrict__ qs0, const unsigned short* __restrict__ ks0,
    const unsigned short* __restrict__ qs1, const unsigned short* __restrict__ ks1, int l_split,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, int64_t csb, int B, int L, int H,
    float eps, float q_mult) {
  constexpr int RS = HD * 2 + ((HD % 32) == 0 ? 16 : 0);  // LDS row stride, bytes
  constexpr int CPR = HD / 8;                              // 16-byte chunks per row
  constexpr int CSMAX = (HD / 4 + 3) / 4;                  // 16-byte cos / sin pieces per thread: (HD / 4) / TPT, TPT >= H >= 4
  constexpr int SCMAX = (2 * HD + NT - 1) / NT;            // scale elements per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tpb = NT / H, rows = tpb * H;
  unsigned char* s_rows = sm;                                            // [rows][RS]
  float* s_cs = reinterpret_cast<float*>(sm + NT * RS);                 // [tpb][2][HD/2]
  float* s_sc = s_cs + tpb * HD;                                         // [2][HD]: this tensor's scales, tokens < l_split | the rest
  int* s_bl = reinterpret_cast<int*>(s_sc + 2 * HD);                     // [tpb][2]: (batch, position) of the block's tokens
  const int tid = threadIdx.x;
  // blockIdx.y picks the tensor: q and k are independent passes, two blocks instead of two serial phases
  const int which = q && k ? (int)blockIdx.y : (k ? 1 : 0);
  unsigned short* tb = which ? k : q;
  // token arithmetic in 32 bits (the entry point checks B * L < 2^31): a 64-bit division is ~10x the instructions of a 32-bit one
  const int tok0 = (int)blockIdx.x * tpb;                                // first token (over B * L)
  const int ntok = B * L;
  if (tid < tpb) {
    int tok = tok0 + tid;
    tok = tok < ntok ? tok : ntok - 1;
    const int b = (int)((unsigned)tok / (unsigned)L);
    s_bl[2 * tid] = b;
    s_bl[2 * tid + 1] = tok - b * L;
  }
  unsigned short scv[SCMAX];               // (these loads do not need the token table: in flight across the barrier; index clamped,
#pragma unroll                             //  not predicated: no branch, and the conversion waits until the LDS write)
  for (int i = 0; i < SCMAX; ++i) {
    int e = tid + i * NT;
    e = e < 2 * HD ? e : 2 * HD - 1;
    const unsigned short* src = e < HD ? (which ? ks0 : qs0) : (which ? ks1 : qs1);
    scv[i] = src[e < HD ? e : e - HD];
  }
  __syncthreads();                         // token table visible (no global value has been used yet)
  // ---- every remaining global read of the block: the token spans (TPT = NT / tpb consecutive threads walk ONE token's span,
  //      TPT x 16 contiguous bytes per step) and the cos / sin rows of the block's tokens
  const int tpt = NT / tpb;
  const int ct = (int)((unsigned)tid / (unsigned)tpt), cj = tid - ct * tpt;
  const bool cvalid = ct < tpb && tok0 + ct < ntok;
  const int cti = ct < tpb ? ct : tpb - 1;
  unsigned short* cspan = tb + s_bl[2 * cti] * bs + (int64_t)s_bl[2 * cti + 1] * rs;
  const int span_chunks = H * CPR;         // 16-byte chunks of one token's [H * hd] span
  // Loads AND LDS writes below are unconditional with clamped indices (a step past the end repeats the last piece: same bytes
  // to the same LDS address): a predicated write lets the compiler sink its load into the branch, one round trip per piece.
  uint4 pre[CPR];                          // span_chunks / TPT <= CPR steps (TPT >= H)
#pragma unroll
  for (int i = 0; i < CPR; ++i) {
    int w = cj + i * tpt;
    w = w < span_chunks ? w : span_chunks - 1;
    pre[i] = *reinterpret_cast<const uint4*>(cspan + w * 8);
  }
  // cos | sin row of token ct: HD / 4 16-byte pieces (the entry point checked the alignment), walked by the same TPT threads
  const int64_t csoff = s_bl[2 * cti] * csb + (int64_t)s_bl[2 * cti + 1] * (HD / 2);
  float4 cs_pre[CSMAX];
#pragma unroll
  for (int i = 0; i < CSMAX; ++i) {
    int c = cj + i * tpt;
    c = c < HD / 4 ? c : HD / 4 - 1;
    const float* src = c < HD / 8 ? cos_t + csoff + c * 4 : sin_t + csoff + (c - HD / 8) * 4;
    cs_pre[i] = *reinterpret_cast<const float4*>(src);
  }
  // ---- into LDS
  // (the q pass: softmax scale * log2(e) rides on the staged rotation, once per token instead of once per head; the two
  //  orders of the products differ by f32 rounding only, far below the bf16 rounding that follows)
  const float stage_mult = which ? 1.0f : q_mult;
#pragma unroll
  for (int i = 0; i < SCMAX; ++i) {
    int e = tid + i * NT;
    e = e < 2 * HD ? e : 2 * HD - 1;
    s_sc[e] = bf16_bits_to_f32(scv[i]);
  }
#pragma unroll
  for (int i = 0; i < CSMAX; ++i) {
    int c = cj + i * tpt;
    c = c < HD / 4 ? c : HD / 4 - 1;
    float4 u = cs_pre[i];
    u.x *= stage_mult; u.y *= stage_mult; u.z *= stage_mult; u.w *= stage_mult;
    reinterpret_cast<float4*>(s_cs)[cti * (HD / 4) + c] = u;
  }
#pragma unroll
  for (int i = 0; i < CPR; ++i) {
    int w = cj + i * tpt;
    w = w < span_chunks ? w : span_chunks - 1;
    *reinterpret_cast<uint4*>(s_rows + (cti * H + w / CPR) * RS + (w % CPR) * 16) = pre[i];
  }
  const int r = tid;                       // this thread's row
  const int t_loc = (int)((unsigned)r / (unsigned)H);
  const int l_r = s_bl[2 * (t_loc < tpb ? t_loc : tpb - 1) + 1];
  __syncthreads();
  if (r < rows) {
    float v[HD];
#pragma unroll
    for (int c = 0; c < CPR; ++c) unpack8(*reinterpret_cast<const uint4*>(s_rows + r * RS + c * 16), v + c * 8);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < HD; ++j) ss += v[j] * v[j];
    const float rrms = rsqrtf(ss / (float)HD + eps);
    const float* sc = s_sc + (l_r < l_split ? 0 : HD);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {       // 4 scales per 16-byte LDS read
      const float4 w4 = *reinterpret_cast<const float4*>(sc + c * 4);
      // reference rounding points: (x*rrms).to(bf16) * scale(bf16) -> bf16   (layers.py:107-111); two values per
      // v_cvt_pk_bf16_f32, the halves widened back with one shift / one mask
      const unsigned a1 = pack_bf16x2(v[c * 4 + 0] * rrms, v[c * 4 + 1] * rrms);
      const unsigned b1 = pack_bf16x2(v[c * 4 + 2] * rrms, v[c * 4 + 3] * rrms);
      const unsigned a2 = pack_bf16x2(bf16_lo(a1) * w4.x, bf16_hi(a1) * w4.y);
      const unsigned b2 = pack_bf16x2(bf16_lo(b1) * w4.z, bf16_hi(b1) * w4.w);
      v[c * 4 + 0] = bf16_lo(a2); v[c * 4 + 1] = bf16_hi(a2);
      v[c * 4 + 2] = bf16_lo(b2); v[c * 4 + 3] = bf16_hi(b2);
    }
    const float* cr = s_cs + t_loc * HD;
    const float* sr = cr + HD / 2;
    float csv[HD / 2], snv[HD / 2];        // 4 angles per 16-byte LDS read (HD / 2 is a multiple of 4)
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      const float4 c4 = *reinterpret_cast<const float4*>(cr + c * 4);
      const float4 s4 = *reinterpret_cast<const float4*>(sr + c * 4);
      csv[c * 4 + 0] = c4.x; csv[c * 4 + 1] = c4.y; csv[c * 4 + 2] = c4.z; csv[c * 4 + 3] = c4.w;
      snv[c * 4 + 0] = s4.x; snv[c * 4 + 1] = s4.y; snv[c * 4 + 2] = s4.z; snv[c * 4 + 3] = s4.w;
    }
#pragma unroll
    for (int pj = 0; pj < HD / 2; ++pj) {
      const float cs = csv[pj], sn = snv[pj];
      if constexpr (MODE == 0) {           // pairs (2j, 2j+1)
        const float a = v[2 * pj], b2 = v[2 * pj + 1];
        v[2 * pj] = cs * a - sn * b2;
        v[2 * pj + 1] = sn * a + cs * b2;
      } else {                             // pairs (j, j + hd/2)
        const float a = v[pj], b2 = v[pj + HD / 2];
        v[pj] = a * cs - b2 * sn;
        v[pj + HD / 2] = b2 * cs + a * sn;
      }
    }
#pragma unroll
    for (int c = 0; c < CPR; ++c) *reinterpret_cast<uint4*>(s_rows + r * RS + c * 16) = pack8(v + c * 8);
  }
  __syncthreads();
  if (cvalid) {
    for (int w = cj; w < span_chunks; w += tpt)
      *reinterpret_cast<uint4*>(cspan + w * 8) =
          *reinterpret_cast<const uint4*>(s_rows + (ct * H + w / CPR) * RS + (w % CPR) * 16);
  }
}

extern "C" int osk_qknorm_rope_bf16(void* q, void* k, int64_t bs, int64_t rs, const void* qs0,
                                    const void* ks0, const void* qs1, const void* ks1, int l_split,
                                    const float* cos_t, const float* sin_t, int64_t csb, int B, int L,
                                    int H, int hd, int rope_mode, float eps, float q_mult, void* stream) {
  if ((!q && !k) || !qs0 || !ks0 || !qs1 || !ks1 || !cos_t || !sin_t) return OSK_EINVAL;
  if (B <= 0 || L <= 0 || H <= 0 || (bs & 7) || (rs & 7)) return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * L * H;
  {
    // row-per-thread kernel (coalesced spans through LDS)
    if (H >= 4 && H <= 256 && (hd == 64 || hd == 72) && (rope_mode == 0 || rope_mode == 1) && (int64_t)B * L < (1ll << 31) &&
        (((uintptr_t)cos_t | (uintptr_t)sin_t) & 15) == 0 && (csb & 3) == 0) {   // (16-byte cos / sin pieces; unaligned tables take the kernel below)
      // hd 128: the lane-group kernel below already uses every lane (16 x 16 B per row) and a whole row per thread would need 256 VGPRs
      // 128-row blocks (round 4: 21 KB of LDS -> 7 blocks per CU instead of 3 x 42 KB: the copy-in / compute / copy-out phases of
      // more, smaller blocks interleave better) unless the head count needs the 256-row block to hold a whole token
      const int NT_ = H <= 128 ? 128 : 256;
      const int tpb = NT_ / H;
      const int64_t ntok = (int64_t)B * L;
      dim3 grid((unsigned)((ntok + tpb - 1) / tpb), (q && k) ? 2 : 1), block(NT_);
#define LAUNCH_ROWS(HD, MODE)                                                                                   \
  {                                                                                                             \
    constexpr int RS_ = HD * 2 + ((HD % 32) == 0 ? 16 : 0);                                                     \
    const size_t smem = (size_t)NT_ * RS_ + (size_t)tpb * HD * 4 + 2 * HD * 4 + (size_t)tpb * 8;                \
    if (NT_ == 128)                                                                                             \
      hipLaunchKernelGGL((qknorm_rope_rows_kernel<HD, MODE, 128>), grid, block, smem, st, (unsigned short*)q,   \
                         (unsigned short*)k, bs, rs, (const unsigned short*)qs0, (const unsigned short*)ks0,    \
                         (const unsigned short*)qs1, (const unsigned short*)ks1, l_split, cos_t, sin_t, csb, B, \
                         L, H, eps, q_mult);                                                                    \
    else                                                                                                        \
    hipLaunchKernelGGL((qknorm_rope_rows_kernel<HD, MODE, 256>), grid, block, smem, st, (unsigned short*)q,     \
                       (unsigned short*)k, bs, rs, (const unsigned short*)qs0, (const unsigned short*)ks0,      \
                       (const unsigned short*)qs1, (const unsigned short*)ks1, l_split, cos_t, sin_t, csb, B,   \
                       L, H, eps, q_mult);                                                                      \
  }
      if (hd == 64) { if (rope_mode == 0) LAUNCH_ROWS(64, 0) else LAUNCH_ROWS(64, 1) }
      else { if (rope_mode == 0) LAUNCH_ROWS(72, 0) else LAUNCH_ROWS(72, 1) }
#undef LAUNCH_ROWS
      return (int)hipGetLastError();
    }
  }
#define LAUNCH(HD, CW, MODE)                                                                          \
  {                                                                                                   \
    constexpr int NCH = HD / CW;                                                                      \
    constexpr int LPR = NCH <= 8 ? 8 : (NCH <= 16 ? 16 : 32);                                         \
    constexpr int RPB = 256 / LPR;                                                                    \
    dim3 grid((unsigned)((total + RPB - 1) / RPB)), block(256);                                       \
    hipLaunchKernelGGL((qknorm_rope_kernel<HD, CW, MODE>), grid, block, 0, st, (unsigned short*)q,    \
                       (unsigned short*)k, bs, rs, (const unsigned short*)qs0,                        \
                       (const unsigned short*)ks0, (const unsigned short*)qs1,                        \
                       (const unsigned short*)ks1, l_split, cos_t, sin_t, csb, B, L, H, eps, q_mult); \
  }
  if (rope_mode == 0) {
    if (hd == 64) LAUNCH(64, 8, 0)
    else if (hd == 72) LAUNCH(72, 8, 0)
    else if (hd == 128) LAUNCH(128, 8, 0)
    else return OSK_EUNSUPPORTED;
  } else if (rope_mode == 1) {
    if (hd == 64) LAUNCH(64, 8, 1)
    else if (hd == 72) LAUNCH(72, 4, 1)
    else if (hd == 128) LAUNCH(128, 8, 1)
    else return OSK_EUNSUPPORTED;
  } else {
    return OSK_EINVAL;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

// =============================================================================================
// V [B, L, H, hd] (strided) -> VT [B, H, hd, Lp] with the per-16-key quad swap.  One block = 64 keys x
// one head; staged through LDS so both the read (hd contiguous) and the write (keys contiguous) are
// coalesced 16 B accesses.  Bytes: 2*hd read + 2*hd written per (key, head).
// =============================================================================================
template <int HD>
__global__ void __launch_bounds__(256) v_transpose_kernel(const unsigned short* __restrict__ v,
                                                          int64_t bs, int64_t rs,
                                                          unsigned short* __restrict__ vt, int L,
                                                          int Lp, int H) {
  __shared__ unsigned short tile[64][HD + 2];  // +2: odd dword stride -> conflict-free column reads
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
  // output: HD rows of 64 keys = 8 chunks of 8 keys, in the key order the attention kernel of this head_dim bakes into its P operand
  unsigned short* obase = vt + ((int64_t)(b * H + h) * HD) * Lp + key0;
  for (int i = threadIdx.x; i < HD * 8; i += 256) {
    const int d = i >> 3, pc = i & 7;  // pc: 8-position chunk within the 64-key row
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int key;
      if constexpr (HD == 72 || HD == 64) {
        // head_dim 72 and 64 (P.V on v_mfma_f32_16x16x32_bf16, attention_asm72.hip): chunk pc of the 64-key row = 32-key half pc / 4,
        // lane row r = pc % 4 of the MFMA operands, whose 8 keys are {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31}
        const int r = pc & 3;
        key = 32 * (pc >> 2) + ((r & 1) << 4) + ((r >> 1) << 2) + ((j >> 2) << 3) + (j & 3);
      } else {
        // head_dim 128 (P.V on 32x32x16): per 16 keys, positions 0-3 -> keys 0-3, 4-7 -> keys 8-11, 8-11 -> keys 4-7, 12-15 -> keys 12-15
        const int p = pc * 8 + j;                  // position within tile
        const int p16 = p & 15, grp = p & ~15;
        key = grp + ((p16 < 4 || p16 >= 12) ? p16 : (p16 < 8 ? p16 + 4 : p16 - 4));
      }
      e[j] = tile[key][d];
    }
    uint4 u;
    u.x = e[0] | ((unsigned)e[1] << 16);
    u.y = e[2] | ((unsigned)e[3] << 16);
    u.z = e[4] | ((unsigned)e[5] << 16);
    u.w = e[6] | ((unsigned)e[7] << 16);
    *reinterpret_cast<uint4*>(obase + (int64_t)d * Lp + pc * 8) = u;
  }
}

extern "C" int osk_v_transpose_bf16(const void* v, int64_t bs, int64_t rs, void* vt, int B, int L,
                                    int H, int hd, void* stream) {
  if (!v || !vt || B <= 0 || L <= 0 || H <= 0 || (bs & 7) || (rs & 7)) return OSK_EINVAL;
  const int Lp = (L + 63) / 64 * 64;
  dim3 grid(Lp / 64, H, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (hd == 64)
    hipLaunchKernelGGL(v_transpose_kernel<64>, grid, block, 0, st, (const unsigned short*)v, bs, rs, (unsigned short*)vt, L, Lp, H);
  else if (hd == 72)
    hipLaunchKernelGGL(v_transpose_kernel<72>, grid, block, 0, st, (const unsigned short*)v, bs, rs, (unsigned short*)vt, L, Lp, H);
  else if (hd == 128)
    hipLaunchKernelGGL(v_transpose_kernel<128>, grid, block, 0, st, (const unsigned short*)v, bs, rs, (unsigned short*)vt, L, Lp, H);
  else
    return OSK_EUNSUPPORTED;
  return (int)hipGetLastError();
}

// =============================================================================================
// Skinny GEMV over a task list.  One block (256 threads = 4 waves) per task of <= 64 weight rows;
// x (with optional SiLU) staged once per block in LDS as f32; each wave streams whole weight rows with
// 16 B loads.  Weight-bandwidth bound: bytes = 2*K per output row.
// =============================================================================================
template <int MB>
__global__ void __launch_bounds__(256) gemv_tasks_kernel(
    const float* __restrict__ x, int64_t xbs, int Bv, int K, const uint64_t* __restrict__ w_ptrs,
    const uint64_t* __restrict__ b_ptrs, const int* __restrict__ out_cols,
    const int* __restrict__ n_rows, float* __restrict__ out, int64_t obs, int act_in, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [MB][K]
  const int task = blockIdx.x;
  // 8 elements per thread and trip, loads first (clamped indices, unconditional LDS writes: a predicated write would pull its
  // load into the branch and make every element a dependent round trip)
  for (int i0 = threadIdx.x; i0 < MB * K; i0 += 256 * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = i0 + u * 256;
      i = i < MB * K ? i : MB * K - 1;
      const int b = i / K, kk = i - b * K;
      t[u] = x[(b < Bv ? b : Bv - 1) * xbs + kk];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = i0 + u * 256;
      i = i < MB * K ? i : MB * K - 1;
      const int b = i / K;
      float v = t[u];
      if (act_in == 1) v = silu(v);
      xs[i] = b < Bv ? v : 0.f;
    }
  }
  __syncthreads();
  const unsigned short* W = reinterpret_cast<const unsigned short*>(w_ptrs[task]);
  const unsigned short* bias = reinterpret_cast<const unsigned short*>(b_ptrs[task]);
  const int nr = n_rows[task];
  const int col0 = out_cols[task];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nchunk = K >> 3;
  // RU rows per wave at a time: their 16-byte weight loads of one k position are all issued before the first is used (round 5;
  // one row at a time made every row a dependent memory round trip: 0.9 TB/s over the 438 MB of adaLN weights of an XL step),
  // and the x pieces read from LDS serve all RU rows.  Per row the per-lane summation order is unchanged.
  constexpr int RU = MB == 8 ? 4 : 8;
  for (int r0 = wave * RU; r0 < nr; r0 += 4 * RU) {
    float acc[RU][MB];
    const unsigned short* wr[RU];
#pragma unroll
    for (int j = 0; j < RU; ++j) {
      const int rr = r0 + j < nr ? r0 + j : nr - 1;   // (clamped: a row past the task repeats its last row, result dropped)
      wr[j] = W + (int64_t)rr * K;
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[j][b] = 0.f;
    }
    for (int c = lane; c < nchunk; c += 64) {
      uint4 u[RU];
#pragma unroll
      for (int j = 0; j < RU; ++j) u[j] = *reinterpret_cast<const uint4*>(wr[j] + c * 8);
      float4 x0[MB], x1[MB];
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        x0[b] = *reinterpret_cast<const float4*>(&xs[b * K + c * 8]);
        x1[b] = *reinterpret_cast<const float4*>(&xs[b * K + c * 8 + 4]);
      }
#pragma unroll
      for (int j = 0; j < RU; ++j) {
        float w[8];
        unpack8(u[j], w);
#pragma unroll
        for (int b = 0; b < MB; ++b)
          acc[j][b] += w[0] * x0[b].x + w[1] * x0[b].y + w[2] * x0[b].z + w[3] * x0[b].w + w[4] * x1[b].x + w[5] * x1[b].y +
                       w[6] * x1[b].z + w[7] * x1[b].w;
      }
    }
#pragma unroll
    for (int j = 0; j < RU; ++j) {
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[j][b] = wave_sum(acc[j][b]);
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < RU; ++j) {
        const int r = r0 + j;
        if (r < nr) {
          const float bv = bias ? bf16_bits_to_f32(bias[r]) : 0.f;
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            if (b < Bv) {
              float* o = out + b * obs + col0 + r;
              const float val = acc[j][b] + bv;
              *o = accumulate ? (*o + val) : val;
            }
          }
        }
      }
    }
  }
}

extern "C" int osk_gemv_tasks_bf16(const float* x, int64_t xbs, int Bv, int K, const uint64_t* w_ptrs,
                                   const uint64_t* b_ptrs, const int32_t* out_cols,
                                   const int32_t* n_rows, int n_tasks, float* out, int64_t obs,
                                   int act_in, int accumulate, void* stream) {
  if (!x || !w_ptrs || !b_ptrs || !out_cols || !n_rows || !out) return OSK_EINVAL;
  if (Bv <= 0 || K <= 0 || (K & 7) || n_tasks <= 0) return OSK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(n_tasks), block(256);
  // The x rows of one launch live in LDS as f32 [MB][K] (<= 64 KiB): any batch is cut into slices of MB = 8 rows
  // (K <= 2048), 4 rows (K <= 4096) or 1 row (K <= 16384); the reference's Linear has no batch limit.
  const int mb = (size_t)8 * K * sizeof(float) <= 64 * 1024 && Bv > 4 ? 8
               : (size_t)4 * K * sizeof(float) <= 64 * 1024 ? 4
               : (size_t)K * sizeof(float) <= 64 * 1024 ? 1 : 0;
  if (mb == 0) return OSK_EUNSUPPORTED;
  const size_t sm = (size_t)mb * K * sizeof(float);
  for (int b0 = 0; b0 < Bv; b0 += mb) {
    const int nb = Bv - b0 < mb ? Bv - b0 : mb;
    const float* xb = x + (int64_t)b0 * xbs;
    float* ob = out + (int64_t)b0 * obs;
    if (mb == 8)
      hipLaunchKernelGGL(gemv_tasks_kernel<8>, grid, block, sm, st, xb, xbs, nb, K, w_ptrs, b_ptrs, out_cols, n_rows,
                         ob, obs, act_in, accumulate);
    else if (mb == 4)
      hipLaunchKernelGGL(gemv_tasks_kernel<4>, grid, block, sm, st, xb, xbs, nb, K, w_ptrs, b_ptrs, out_cols, n_rows,
                         ob, obs, act_in, accumulate);
    else
      hipLaunchKernelGGL(gemv_tasks_kernel<1>, grid, block, sm, st, xb, xbs, nb, K, w_ptrs, b_ptrs, out_cols, n_rows,
                         ob, obs, act_in, accumulate);
  }
  return (int)hipGetLastError();
}

// =============================================================================================
// timestep embedding and RoPE tables (tiny)
// =============================================================================================
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int B, int dim, float max_period,
                                          float time_factor, float* __restrict__ out) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float freq = expf(-logf(max_period) * (float)j / (float)half);
  const float a = (time_factor * t[b]) * freq;
  out[b * dim + j] = cosf(a);
  out[b * dim + half + j] = sinf(a);
  if ((dim & 1) && j == 0) out[b * dim + dim - 1] = 0.f;
}

extern "C" int osk_timestep_embedding(const float* t, int B, int dim, float max_period,
                                      float time_factor, float* out, void* stream) {
  if (!t || !out || B <= 0 || dim < 2) return OSK_EINVAL;
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     t, B, dim, max_period, time_factor, out);
  return (int)hipGetLastError();
}

struct AxesDesc {
  int n_axes;
  int dim[4];
  int start[4];  // pair-index start per axis
};

__global__ void rope_table_kernel(const float* __restrict__ ids, int64_t n_rows, AxesDesc ax, int half,
                                  double theta, int f32_angles, float* __restrict__ cos_out,
                                  float* __restrict__ sin_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * half) return;
  const int64_t row = i / half;
  const int j = (int)(i - row * half);
  int a = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t)
    if (t < ax.n_axes && j >= ax.start[t]) a = t;
  const int jj = j - ax.start[a];
  const float pos = ids[row * ax.n_axes + a];
  if (f32_angles) {
    // liger_rope, math.py:39-47: scale = arange(0,d,2,f32)/d; omega = 1/theta**scale; angle = pos*omega (f32)
    const float sc = (float)(2 * jj) / (float)ax.dim[a];
    const float omega = 1.0f / powf((float)theta, sc);
    const float ang = pos * omega;
    cos_out[i] = cosf(ang);
    sin_out[i] = sinf(ang);
  } else {
    // rope, math.py:50-57: everything in f64, result cast to f32
    const double sc = (double)(2 * jj) / (double)ax.dim[a];
    const double omega = 1.0 / pow(theta, sc);
    const double ang = (double)pos * omega;
    cos_out[i] = (float)cos(ang);
    sin_out[i] = (float)sin(ang);
  }
}

extern "C" int osk_rope_table(const float* ids, int64_t n_rows, int n_axes, const int32_t* axes_dim_host,
                              double theta, int f32_angles, float* cos_out, float* sin_out,
                              void* stream) {
  if (!ids || !axes_dim_host || !cos_out || !sin_out || n_rows <= 0 || n_axes <= 0 || n_axes > 4) return OSK_EINVAL;
  AxesDesc ax;
  ax.n_axes = n_axes;
  int half = 0;
  for (int a = 0; a < 4; ++a) {
    ax.dim[a] = a < n_axes ? axes_dim_host[a] : 2;
    ax.start[a] = half;
    if (a < n_axes) {
      if (axes_dim_host[a] <= 0 || (axes_dim_host[a] & 1)) return OSK_EINVAL;
      half += axes_dim_host[a] / 2;
    }
  }
  const int64_t n = n_rows * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, ids, n_rows, ax, half, theta, f32_angles, cos_out, sin_out);
  return (int)hipGetLastError();
}

// =============================================================================================
// CFG combine + Euler update.  Bytes: 3 pred reads + x read + x write = 10 B per element.
// =============================================================================================
__global__ void __launch_bounds__(256) cfg_euler_kernel(const unsigned short* __restrict__ pred, int64_t n,
                                                        const unsigned short* __restrict__ x,
                                                        unsigned short* __restrict__ xo, float g_txt,
                                                        float g_img, const float* __restrict__ g_img_vec,
                                                        float dt) {
  const int64_t nchunk = n >> 3;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * 256) {
    float pc[8], pu[8], p2[8], xv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(pred + c * 8), pc);
    unpack8(*reinterpret_cast<const uint4*>(pred + n + c * 8), pu);
    unpack8(*reinterpret_cast<const uint4*>(pred + 2 * n + c * 8), p2);
    unpack8(*reinterpret_cast<const uint4*>(x + c * 8), xv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gi = g_img_vec ? g_img_vec[c * 8 + j] : g_img;
      const float v = p2[j] + gi * (pu[j] - p2[j]) + g_txt * (pc[j] - pu[j]);
      o[j] = xv[j] + dt * v;
    }
    *reinterpret_cast<uint4*>(xo + c * 8) = pack8(o);
  }
}

extern "C" int osk_cfg_euler_bf16(const void* pred, int64_t n, const void* x, void* x_out, float g_txt,
                                  float g_img, const float* g_img_vec, float dt, void* stream) {
  if (!pred || !x || !x_out || n <= 0 || (n & 7)) return OSK_EINVAL;
  int64_t nb = (n / 8 + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)pred, n, (const unsigned short*)x, (unsigned short*)x_out,
                     g_txt, g_img, g_img_vec, dt);
  return (int)hipGetLastError();
}

// =============================================================================================
// Strided row copy (bf16): dst[j, b, l, 0:C] = src[j, b, l, 0:C] with independent chunk / batch / row strides on both sides; a
// source batch stride of 0 broadcasts one item over the batch.  The host glue of the denoise step without a torch kernel:
// sampler: latents -> the CFG triple's input; model: img / cond -> the K-padded img_in operand (sampling.py:196-201,
// model.py:170-176); sequence parallelism: "all heads of my tokens" [B, L/P, P x Dg] <-> "P head groups" [P, B, L/P, Dg] around
// the head all-to-all (the rearranges of distributed.py:473-495; chunk j = columns j Dg .. of the token-major side).
// Bytes: 4 C per row (read + write).  8-byte units: C and every stride are multiples of 4 elements.
// =============================================================================================
__global__ void __launch_bounds__(256) copy_rows_kernel(const unsigned short* __restrict__ src, int64_t scs, int64_t sbs, int64_t srs,
                                                        unsigned short* __restrict__ dst, int64_t dcs, int64_t dbs, int64_t drs,
                                                        int NC, int B, int L, int C4) {
  const int64_t per_b = (int64_t)L * C4, per_c = per_b * B, total = per_c * NC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i / per_c);
    const int64_t rj = i - j * per_c;
    const int b = (int)(rj / per_b);
    const int64_t r = rj - b * per_b;
    const int l = (int)(r / C4), c = (int)(r - (int64_t)l * C4) * 4;
    *reinterpret_cast<uint2*>(dst + j * dcs + b * dbs + l * drs + c) = *reinterpret_cast<const uint2*>(src + j * scs + b * sbs + l * srs + c);
  }
}

extern "C" int osk_copy_rows_bf16(const void* src, int64_t src_chunk_stride, int64_t src_batch_stride, int64_t src_row_stride, void* dst,
                                  int64_t dst_chunk_stride, int64_t dst_batch_stride, int64_t dst_row_stride, int n_chunks, int B, int L,
                                  int C, void* stream) {
  if (!src || !dst || n_chunks <= 0 || B <= 0 || L <= 0 || C <= 0) return OSK_EINVAL;
  if ((C & 3) || (src_chunk_stride & 3) || (src_batch_stride & 3) || (src_row_stride & 3) || (dst_chunk_stride & 3) ||
      (dst_batch_stride & 3) || (dst_row_stride & 3))
    return OSK_EINVAL;
  if (((uintptr_t)src & 7) || ((uintptr_t)dst & 7)) return OSK_EINVAL;
  const int64_t total = (int64_t)n_chunks * B * L * (C / 4);
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                     src_chunk_stride, src_batch_stride, src_row_stride, (unsigned short*)dst, dst_chunk_stride, dst_batch_stride,
                     dst_row_stride, n_chunks, B, L, C / 4);
  return (int)hipGetLastError();
}

extern "C" int osk_abi_version(void) { return OSK_ABI_VERSION; }
extern "C" const char* osk_arch(void) { return "gfx950"; }
