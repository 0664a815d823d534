// Launch parameters shared by the CausalConv3d kernels (conv3d.hip, conv3d_256.hip).
#pragma once
#include "osk_common.h"

namespace osk_conv {

struct ConvParams {
  const unsigned short* x;
  const unsigned short* w;
  const float* bias;
  const unsigned short* res;
  unsigned short* out;
  int B, T, H, W;     // source (pre-upsample) dims
  int Tu, Hu, Wu;     // dims the conv sees (after the virtual nearest upsample)
  int To, Ho, Wo;
  int Cin, Cout;
  int ks, st, sh, sw, up_t, up_hw;
  int lg_cpt, ntaps, nk;
  int M;
  int64_t wrs;
  int brick = 0;      // 1 = a tile is a 16 x 16 spatial brick of one frame, tiles ordered frame-fastest (conv3d_256.hip)
  // fused GroupNorm statistics of the OUTPUT (conv3d_256.hip only): sums[b][g] += (sum y, sum y^2) over the bf16-rounded
  // outputs of group g (gn_G groups of Cout / gn_G adjacent channels); null = off
  double* gn_sums = nullptr;
  int gn_G = 0;
  // GroupNorm + SiLU of the INPUT folded into the halo refill (conv3d_256.hip, sliding-window kernels, plain geometry): the conv
  // sees bf16(silu(bf16(x a + d))); table f32 [B][Cin / 8][16] = 8 scales a then 8 shifts d per 8-channel chunk
  // (osk_groupnorm_table_f32); null = the input as stored
  const float* gn_in = nullptr;
};


// conv3d_256.hip: 256 voxels x {256,128} channels x 64 tile, 4 waves, all 27 taps in one hand-scheduled asm K loop
bool conv256_supported(const ConvParams& p, int64_t x_bytes, int64_t w_bytes);
bool conv256_gn_supported(const ConvParams& p);   // the fused-statistics epilogue takes this output geometry
bool conv256_gn_in_supported(const ConvParams& p);   // a kernel with the input GroupNorm + SiLU folded in takes this shape
int launch_conv256(const ConvParams& p, hipStream_t st);

}  // namespace osk_conv
