// CausalConv3d for gfx950 as an im2col-free implicit GEMM on MFMA bf16, NDHWC activations.
//
//   out[b, to, ho, wo, co] = bias[co] + res[...] +
//        sum_{dt,dh,dw,ci}  x[b, T(to,dt), H(ho,dh), W(wo,dw), ci] * w[co, (dt,dh,dw), ci]
//
// The reference pads first and convolves second (F.pad replicate (W 1,1 | H 1,1 | T 2,0) then an unpadded
// Conv3d, hunyuan_vae/unet_causal_3d_blocks.py:82-96) and, in the decoder, materialises a nearest-neighbour
// upsampled copy before the conv (:136-150).  Here neither copy exists: the replicate/causal padding is a CLAMP
// of the gathered coordinate and the upsample is a SHIFT of it (frame 0 is spatial-only: tu -> tu == 0 ? 0 :
// 1 + (tu-1)/2), both folded into the per-row source address of the A operand.  K runs tap-major / channel-minor
// (k = tap * Cin + ci), so with channels-last activations every 16-byte chunk of the A tile is 8 contiguous
// channels of ONE input voxel: coalesced 16 B global loads straight into LDS (global_load_lds_dwordx4), the same
// lane-linear, source-swizzled LDS image as the dense GEMM (gemm_bf16.hip).  The weight is pre-laid as
// [Cout][27 * Cin] (zero-padded to a multiple of 64) = the GEMM's W operand.
//
// Tile 128 (voxels) x 128 (Cout) x 64 (K), 4 waves (2 x 2) of 2 x 2 v_mfma_f32_32x32x16_bf16 tiles, operands
// swapped so an accumulator lane owns one output voxel and 4 consecutive channels: the epilogue (bias, residual
// add, bf16 pack) is lane-local and stores 8 B pieces of the NDHWC row.
//
// Roofline: MFMA bf16.  Algorithmic FLOPs = 2 * Cin * Cout * k^3 * B*To*Ho*Wo  (SURVEY.md §8(d)).
#include "conv_params.h"
#include "../../include/osk.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;
constexpr int SMEM_BYTES = 2 * 2 * TILE_BYTES;

using osk_conv::ConvParams;

OSK_DEV void glds16(const unsigned short* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

OSK_DEV int sel3(int a0, int a1, int a2, int d) { return d == 0 ? a0 : (d == 1 ? a1 : a2); }

// BIGC: Cin % 64 == 0 -> a K tile lies inside one tap (tap is wave-uniform, scalar decode)
template <bool BIGC>
__global__ void __launch_bounds__(256, 2) conv3d_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, nbm * nbn);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- staging rows of this lane: 4 row-blocks of 8 rows per wave per operand (as gemm_bf16.hip)
  const int srow8 = lane >> 3, spos = lane & 7;
  const unsigned short* gw[4];
  int lds_off[4], cch[4];
  int pT[4][3], pH[4][3], pW[4][3];  // per-axis voxel-index terms of the 3 taps (clamp + upsample folded in)
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = i * 4 + wave;
    const int r = rb * 8 + srow8;
    cch[i] = spos ^ ((r >> 1) & 7);  // source chunk that must land at LDS position spos
    lds_off[i] = rb * 1024;
    int n = n0 + r;
    n = n < p.Cout ? n : p.Cout - 1;
    gw[i] = p.w + (int64_t)n * p.wrs + cch[i] * 8;
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    const int wo = m % p.Wo;
    int q = m / p.Wo;
    const int ho = q % p.Ho;
    q /= p.Ho;
    const int to = q % p.To;
    const int b = q / p.To;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int dd = p.ks == 3 ? d : 0;
      int tu = to * p.st + dd - (p.ks - 1);   // causal: k-1 replicated frames in front, none behind
      tu = tu < 0 ? 0 : (tu > p.Tu - 1 ? p.Tu - 1 : tu);
      const int ts = p.up_t ? (tu == 0 ? 0 : 1 + ((tu - 1) >> 1)) : tu;
      int hu = ho * p.sh + dd - (p.ks >> 1);
      hu = hu < 0 ? 0 : (hu > p.Hu - 1 ? p.Hu - 1 : hu);
      const int hs = p.up_hw ? (hu >> 1) : hu;
      int wu = wo * p.sw + dd - (p.ks >> 1);
      wu = wu < 0 ? 0 : (wu > p.Wu - 1 ? p.Wu - 1 : wu);
      const int ws = p.up_hw ? (wu >> 1) : wu;
      pT[i][d] = (b * p.T + ts) * HW;
      pH[i][d] = hs * p.W;
      pW[i][d] = ws;
    }
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int cpt_mask = (1 << p.lg_cpt) - 1;
  const int sw = (l31 >> 1) & 7;
  const int a_row_off = (wm * 64 + l31) * 128;
  const int w_row_off = (wn * 64 + l31) * 128;

#define STAGE_ISSUE(BUFI, KT)                                                                     \
  {                                                                                               \
    unsigned char* ta_ = smem + (BUFI) * 2 * TILE_BYTES;                                          \
    unsigned char* tw_ = ta_ + TILE_BYTES;                                                        \
    int tap_u_ = 0, cc_u_ = 0, dt_u_ = 0, dh_u_ = 0, dw_u_ = 0;                                   \
    if constexpr (BIGC) {                                                                         \
      const int q_ = (KT) * 8;                                                                    \
      tap_u_ = q_ >> p.lg_cpt;                                                                    \
      cc_u_ = q_ & cpt_mask;                                                                      \
      if (p.ks == 3) {                                                                            \
        dt_u_ = tap_u_ / 9;                                                                       \
        const int r_ = tap_u_ - dt_u_ * 9;                                                        \
        dh_u_ = r_ / 3;                                                                           \
        dw_u_ = r_ - dh_u_ * 3;                                                                   \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      int dt_, dh_, dw_, cc_;                                                                     \
      if constexpr (BIGC) {                                                                       \
        dt_ = dt_u_; dh_ = dh_u_; dw_ = dw_u_; cc_ = cc_u_ + cch[i];                              \
      } else {                                                                                    \
        const int q_ = (KT) * 8 + cch[i];                                                         \
        int tap_ = q_ >> p.lg_cpt;                                                                \
        cc_ = q_ & cpt_mask;                                                                      \
        tap_ = tap_ < p.ntaps ? tap_ : 0; /* K padding: weights are zero there, any address */    \
        dt_ = 0; dh_ = 0; dw_ = 0;                                                                \
        if (p.ks == 3) {                                                                          \
          dt_ = tap_ / 9;                                                                         \
          const int r_ = tap_ - dt_ * 9;                                                          \
          dh_ = r_ / 3;                                                                           \
          dw_ = r_ - dh_ * 3;                                                                     \
        }                                                                                         \
      }                                                                                           \
      const int pos_ = sel3(pT[i][0], pT[i][1], pT[i][2], dt_) + sel3(pH[i][0], pH[i][1], pH[i][2], dh_) + \
                       sel3(pW[i][0], pW[i][1], pW[i][2], dw_);                                   \
      const unsigned short* ga_ = p.x + (((int64_t)pos_ << p.lg_cpt) + cc_) * 8;                  \
      glds16(ga_, ta_ + lds_off[i]);                                                              \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(gw[i] + (KT) * BK, tw_ + lds_off[i]);    \
  }

  STAGE_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < p.nk; ++kt) {
    const bool more = kt + 1 < p.nk;
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
#undef STAGE_ISSUE

  // ---- epilogue: lane owns voxel m, channels n = quad*8 + hi*4 + {0..3}
  const bool vec_ok = (p.Cout & 3) == 0;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 64 + tm * 32 + l31;
    if (m >= p.M) continue;
    const int64_t roff = (int64_t)m * p.Cout;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int n = n0 + wn * 64 + tn * 32 + qd * 8 + hi * 4;
        if (n >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][qd * 4 + j];
        if (vec_ok && n + 3 < p.Cout) {
          if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (p.res) {
            const uint2 rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
            v[0] += bf16_lo(rv.x); v[1] += bf16_hi(rv.x); v[2] += bf16_lo(rv.y); v[3] += bf16_hi(rv.y);
          }
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(p.out + roff + n) = o;
        } else {
          for (int j = 0; j < 4 && n + j < p.Cout; ++j) {
            float t = v[j] + (p.bias ? p.bias[n + j] : 0.f);
            if (p.res) t += bf16_bits_to_f32(p.res[roff + n + j]);
            p.out[roff + n + j] = f32_to_bf16_bits(t);
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Cout <= 4 (the decoder's conv_out: 128 -> 3 at full resolution, 2.0 of the VAE's 53 ms on the kernel above, whose 128-column
// MFMA tile spends 97 % of its work on zero columns).  The layer is 27 * Cin products per output channel: a reduction, not a
// GEMM.  A wave owns 64 consecutive voxels of one output row; lane = (16-voxel run lane / 16, 16-byte channel chunk lane % 16),
// so one load instruction fetches four whole 256-byte voxels (coalesced) and a lane meets the SAME 8 channels in every voxel
// it visits: its 4 weight dwords per (tap, output channel) come from LDS once per (dt, dh) and serve 16 voxels, and an input
// voxel is loaded once per (dt, dh) for the three dw taps that use it.  Products on
// v_dot2_f32_bf16 (packed bf16 pairs, f32 accumulate); the 16 chunk partials of a voxel are summed over the lane row at the end.
// Geometry: 3 x 3 x 3, stride 1, no upsample, Cin == 128, Wo % 64 == 0 (the row of a wave never wraps).
typedef __bf16 fo_bf16x2_t __attribute__((ext_vector_type(2)));

OSK_DEV float fo_row16_sum(float v) {
#define OSKF_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  OSKF_DPP_ADD(0xB1);
  OSKF_DPP_ADD(0x4E);
  OSKF_DPP_ADD(0x141);
  OSKF_DPP_ADD(0x140);
#undef OSKF_DPP_ADD
  return v;
}

template <int NC>
__global__ void __launch_bounds__(256, 2) conv_fewout_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) unsigned wl[27 * NC * 64];   // [tap][cout][64 dwords = 128 channels]
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * NC * 64; i += 256) {
    const int tap = i / (NC * 64), r = i - tap * NC * 64;
    const int c = r >> 6, d = r & 63;
    wl[i] = c < p.Cout ? *reinterpret_cast<const unsigned*>(p.w + (int64_t)c * p.wrs + tap * 128 + 2 * d) : 0u;
  }
  __syncthreads();
  const int lane = tid & 63, sub = lane >> 4, ch = lane & 15;
  // a block = the SAME 64-voxel column segment of 4 consecutive output rows, one per wave (round 6; before: 4 consecutive segments of
  // one row): the 4 waves read input rows ho - 1 .. ho + 4 -- 6 distinct rows for 12 row reads -- so half of the block's loads hit
  // the CU's L1 instead of going to L2 (the kernel was L2-bound: every input row is read by 3 output rows x 3 frames)
  const int segs_per_row = p.Wo >> 6, hblocks = (p.Ho + 3) >> 2;
  int q = blockIdx.x;
  const int cs = q % segs_per_row;
  q /= segs_per_row;
  const int ho = ((q % hblocks) << 2) + (tid >> 6);
  q /= hblocks;
  const int to = q % p.To, b = q / p.To;
  if (ho >= p.Ho || b >= p.B) return;
  const int w0 = (cs << 6) + 16 * sub;                         // this lane's 16 consecutive output voxels: w0 .. w0 + 15
  float acc[16][NC];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[i][c] = 0.f;
  for (int dt = 0; dt < 3; ++dt) {
    int ts = to + dt - 2;
    ts = ts < 0 ? 0 : ts;                                     // causal padding: the first frame repeats
    for (int dh = 0; dh < 3; ++dh) {
      int hs = ho + dh - 1;
      hs = hs < 0 ? 0 : (hs > p.H - 1 ? p.H - 1 : hs);
      const unsigned short* row = p.x + (((int64_t)b * p.T + ts) * p.H + hs) * p.W * 128 + ch * 8;
      uint4 wv[3][NC];                                        // the three dw taps of this (dt, dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int c = 0; c < NC; ++c) wv[dw][c] = *reinterpret_cast<const uint4*>(&wl[(((dt * 3 + dh) * 3 + dw) * NC + c) * 64 + ch * 4]);
      // input voxel w0 - 1 + jj of this row feeds outputs jj - dw (dw = 0, 1, 2): 18 loads serve 48 (output, tap) pairs -- the 27
      // taps of an output would otherwise re-read every input voxel through L2 (the first version of this kernel was L2-bound)
#pragma unroll
      for (int jj = 0; jj < 18; ++jj) {
        int ws = w0 + jj - 1;
        ws = ws < 0 ? 0 : (ws > p.W - 1 ? p.W - 1 : ws);
        const uint4 xv = *reinterpret_cast<const uint4*>(row + (int64_t)ws * 128);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int i = jj - dw;
          if (i < 0 || i > 15) continue;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            float a = acc[i][c];
            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fo_bf16x2_t, xv.x), __builtin_bit_cast(fo_bf16x2_t, wv[dw][c].x), a, false);
            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fo_bf16x2_t, xv.y), __builtin_bit_cast(fo_bf16x2_t, wv[dw][c].y), a, false);
            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fo_bf16x2_t, xv.z), __builtin_bit_cast(fo_bf16x2_t, wv[dw][c].z), a, false);
            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fo_bf16x2_t, xv.w), __builtin_bit_cast(fo_bf16x2_t, wv[dw][c].w), a, false);
            acc[i][c] = a;
          }
        }
      }
    }
  }
  const int64_t m0 = (((int64_t)b * p.To + to) * p.Ho + ho) * p.Wo + w0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float v = fo_row16_sum(acc[i][c]);
      if (ch == 0 && c < p.Cout) {
        const int64_t o = (m0 + i) * p.Cout + c;
        float r = v + (p.bias ? p.bias[c] : 0.f);
        if (p.res) r += bf16_bits_to_f32(p.res[o]);
        p.out[o] = f32_to_bf16_bits(r);
      }
    }
  }
}

bool fewout_supported(const ConvParams& p) {
  return p.ks == 3 && p.st == 1 && p.sh == 1 && p.sw == 1 && !p.up_t && !p.up_hw && p.Cin == 128 && p.Cout <= 4 && (p.Wo & 63) == 0 &&
         !p.gn_sums;
}

}  // namespace

static int conv_entry(const void* x, int B, int T, int H, int W, int Cin, const void* w, int64_t w_row_stride,
                      const float* bias, int Cout, int ksize, int stride_t, int stride_h, int stride_w, int up_t, int up_hw,
                      const void* res, void* out, int To, int Ho, int Wo, double* gn_sums, int gn_groups, void* stream,
                      const float* gn_in = nullptr) {
  if (!x || !w || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return OSK_EINVAL;
  if (ksize != 1 && ksize != 3) return OSK_EUNSUPPORTED;
  if (stride_t < 1 || stride_h < 1 || stride_w < 1 || stride_t > 2 || stride_h > 2 || stride_w > 2) return OSK_EINVAL;
  if ((Cin & 7) || (Cin & (Cin - 1))) return OSK_EUNSUPPORTED;  // Cin = 8 * 2^j (pad 3 -> 8 at the boundary)
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)out & 7) || ((uintptr_t)res & 7))
    return OSK_EINVAL;
  ConvParams p;
  p.x = (const unsigned short*)x; p.w = (const unsigned short*)w; p.bias = bias;
  p.res = (const unsigned short*)res; p.out = (unsigned short*)out;
  p.B = B; p.T = T; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.up_t = up_t ? 1 : 0; p.up_hw = up_hw ? 1 : 0;
  p.Tu = p.up_t ? 1 + 2 * (T - 1) : T;
  p.Hu = p.up_hw ? 2 * H : H;
  p.Wu = p.up_hw ? 2 * W : W;
  p.ks = ksize; p.st = stride_t; p.sh = stride_h; p.sw = stride_w;
  // output extent of "replicate-pad then unpadded conv" (unet_causal_3d_blocks.py:82-96)
  const int eTo = (p.Tu - 1) / stride_t + 1, eHo = (p.Hu - 1) / stride_h + 1, eWo = (p.Wu - 1) / stride_w + 1;
  if (To != eTo || Ho != eHo || Wo != eWo) return OSK_EINVAL;
  p.To = To; p.Ho = Ho; p.Wo = Wo;
  const int64_t M = (int64_t)B * To * Ho * Wo;
  if (M >= (int64_t)1 << 31 || (int64_t)B * T * H * W >= (int64_t)1 << 31) return OSK_EUNSUPPORTED;
  p.M = (int)M;
  int lg = 0;
  while ((8 << lg) < Cin) ++lg;
  p.lg_cpt = lg;
  p.ntaps = ksize * ksize * ksize;
  const int64_t K = (int64_t)p.ntaps * Cin;
  const int64_t Kp = (K + BK - 1) / BK * BK;
  if (w_row_stride < Kp || (w_row_stride & 7)) return OSK_EINVAL;  // weight rows zero-padded to a multiple of 64
  p.wrs = w_row_stride;
  p.nk = (int)(Kp / BK);
  hipStream_t s = (hipStream_t)stream;
  {
    // large-tile kernels with the hand-scheduled K loop (conv3d_256.hip) wherever the shape qualifies
    const int64_t x_bytes = (int64_t)B * T * H * W * Cin * 2;
    const bool big = osk_conv::conv256_supported(p, x_bytes, (int64_t)Cout * w_row_stride * 2);
    if (gn_sums) {   // fused statistics live in the large-tile kernels' epilogue only; nothing is launched otherwise
      p.gn_sums = gn_sums;
      p.gn_G = gn_groups;
      // (the statistics ride in the 16-byte-store path of the epilogue: the output must be 16-byte aligned)
      if (!big || !osk_conv::conv256_gn_supported(p) || ((uintptr_t)out & 15)) return OSK_EUNSUPPORTED;
    }
    if (gn_in) {     // input GroupNorm + SiLU folded into the halo refill: sliding-window kernels only; nothing is launched otherwise
      p.gn_in = gn_in;
      if (!big || !osk_conv::conv256_gn_in_supported(p)) return OSK_EUNSUPPORTED;
    }
    if (big) return osk_conv::launch_conv256(p, s);
  }
  if (fewout_supported(p)) {
    const dim3 g((unsigned)(B * To * ((Ho + 3) / 4) * (Wo >> 6))), blk(256);
    if (Cout == 1) hipLaunchKernelGGL((conv_fewout_kernel<1>), g, blk, 0, s, p);
    else if (Cout == 2) hipLaunchKernelGGL((conv_fewout_kernel<2>), g, blk, 0, s, p);
    else if (Cout == 3) hipLaunchKernelGGL((conv_fewout_kernel<3>), g, blk, 0, s, p);
    else hipLaunchKernelGGL((conv_fewout_kernel<4>), g, blk, 0, s, p);
    return (int)hipGetLastError();
  }
  const int nblk = ((p.M + BM - 1) / BM) * ((Cout + BN - 1) / BN);
  dim3 grid(nblk), block(256);
  if (Cin % 64 == 0) hipLaunchKernelGGL((conv3d_kernel<true>), grid, block, SMEM_BYTES, s, p);
  else hipLaunchKernelGGL((conv3d_kernel<false>), grid, block, SMEM_BYTES, s, p);
  return (int)hipGetLastError();
}

extern "C" int osk_causal_conv3d_ndhwc_bf16(const void* x, int B, int T, int H, int W, int Cin, const void* w,
                                            int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                            int stride_t, int stride_h, int stride_w, int up_t, int up_hw,
                                            const void* res, void* out, int To, int Ho, int Wo, void* stream) {
  return conv_entry(x, B, T, H, W, Cin, w, w_row_stride, bias, Cout, ksize, stride_t, stride_h, stride_w, up_t, up_hw, res, out,
                    To, Ho, Wo, nullptr, 0, stream);
}

extern "C" int osk_causal_conv3d_gn_ndhwc_bf16(const void* x, int B, int T, int H, int W, int Cin, const void* w,
                                               int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                               int stride_t, int stride_h, int stride_w, int up_t, int up_hw,
                                               const void* res, void* out, int To, int Ho, int Wo, double* gn_sums,
                                               int gn_groups, void* stream) {
  if (!gn_sums || gn_groups <= 0 || ((uintptr_t)gn_sums & 7)) return OSK_EINVAL;
  return conv_entry(x, B, T, H, W, Cin, w, w_row_stride, bias, Cout, ksize, stride_t, stride_h, stride_w, up_t, up_hw, res, out,
                    To, Ho, Wo, gn_sums, gn_groups, stream);
}

extern "C" int osk_causal_conv3d_gnin_ndhwc_bf16(const void* x, const float* gn_in_table, int B, int T, int H, int W, int Cin,
                                                 const void* w, int64_t w_row_stride, const float* bias, int Cout, int ksize,
                                                 int stride_t, int stride_h, int stride_w, const void* res, void* out, int To,
                                                 int Ho, int Wo, double* gn_sums, int gn_groups, void* stream) {
  if (!gn_in_table || ((uintptr_t)gn_in_table & 15)) return OSK_EINVAL;
  if (gn_sums && (gn_groups <= 0 || ((uintptr_t)gn_sums & 7))) return OSK_EINVAL;
  return conv_entry(x, B, T, H, W, Cin, w, w_row_stride, bias, Cout, ksize, stride_t, stride_h, stride_w, 0, 0, res, out, To, Ho, Wo,
                    gn_sums, gn_sums ? gn_groups : 0, stream, gn_in_table);
}
