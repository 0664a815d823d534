// Launch parameters shared by the bf16 GEMM kernels (gemm_bf16.hip, gemm256.hip).
#pragma once
#include "osk_common.h"
#include <utility>

namespace osk_gemm {

struct GemmParams {
  const unsigned short* A;
  int64_t abs_, ars;
  int arpb;
  const unsigned short* W;
  int64_t wrs;
  const float* bias;
  void* C;
  int64_t cbs, crs;
  int crpb;
  const unsigned short* res;
  const float* gate;
  int64_t gbs;
  int M, N, K, gelu_from;
  int group;   // gemm256: row bands per tile group (L2 blocking of the tile order)
  // GEGLU (gemm256x.hip only): the N GEMM columns are N / 2 (value, gate) pairs in alternating 16-column blocks -- W / bias rows
  // packed by osk_geglu_pack_index -- and C has N / 2 columns: C[m, 16 j + i] = value * gelu_tanh(gate)   (gemm_epilogue16.h)
  int geglu = 0;
  const float* sa = nullptr;   // fp8 instantiation: per-row scales of A [M] and of W [N]
  const float* sw = nullptr;
  // ---- gemm256x.hip only (osk_gemm_group_bf16) ----
  // skipped column range: the logical GEMM columns n' in [0, N - skip_len) are the physical columns n = n' + (n' >= skip_from ?
  // skip_len : 0) of W, bias and C (a single-stream block's linear1 WITHOUT its V columns: [q | k | . | mlp]); skip_from % 256 == 0,
  // N is the PHYSICAL column count
  int skip_from = 0x7fffffff, skip_len = 0;
  // V^T task (vt != 0): the kernel's A operand is the WEIGHT (M = H * hd rows, one "batch"), its W operand the ACTIVATIONS -- column n
  // of the product = (batch n / wrpb, position n % wrpb) of the key axis, fed from activation row key = perm_vt(position) (the 64-key
  // order of osk_v_transpose_bf16 for this head dim: 1 = head_dim 64 / 72, 2 = head_dim 128), rows >= wvalid read clamped and stored
  // as zero; C[b * ccbs + m * crs + position]; rowbias[m] added in the epilogue.  Plain tasks: wrpb = INT_MAX (batch 0), wbs = 0.
  int vt = 0;
  int wrpb = 0x7fffffff, wvalid = 0x7fffffff;
  int64_t wbs = 0, ccbs = 0;
  const float* rowbias = nullptr;
};

// position within a 64-key group of V^T -> key within the group (the order the attention kernels' P operand holds its keys:
// elementwise.hip::v_transpose_kernel)
static __host__ __device__ __forceinline__ int vt_perm64(int pos, int kind) {
  const int g = pos & ~63, p = pos & 63;
  if (kind == 1) {   // head_dim 64 / 72: P.V on v_mfma_f32_16x16x32_bf16
    const int pc = p >> 3, j = p & 7, r = pc & 3;
    return g + 32 * (pc >> 2) + ((r & 1) << 4) + ((r >> 1) << 2) + ((j >> 2) << 3) + (j & 3);
  }
  const int p16 = p & 15, grp = p & ~15;   // head_dim 128: P.V on 32x32x16
  return g + grp + ((p16 < 4 || p16 >= 12) ? p16 : (p16 < 8 ? p16 + 4 : p16 - 4));
}

// large tiles need M >= 256, N >= 128, K % 64 == 0 and both operand tensors within the kernels' 32-bit per-lane offsets
bool gemm256_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems);
// gemm256p.hip: 256 x 128 tiles walked by one persistent 8-wave workgroup per CU (cross-tile prefetch, bias-initialised accumulators)
int launch_gemm256p(const GemmParams& p, int out_f32, hipStream_t st);
// gemm256x.hip: 256 x 256 tiles, 4 waves (one per SIMD, 128 x 128 wave tiles, 256 accumulator AGPRs) on v_mfma_f32_16x16x32_bf16
int launch_gemm256x(const GemmParams& p, int out_f32, hipStream_t st);
// the same kernel over TWO problems that share N, K, gelu_from, group (one tile list: the second problem fills the first one's last round)
int launch_gemm256x_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t st);
// one or two V^T problems (gemm_params.h: vt != 0) that share K as ONE tile list (gemm256x_vt_kernel)
int launch_gemm256x_vt(const GemmParams* ps, int n, hipStream_t st);
// gemm256.hip: the one-tile-per-workgroup frame on OCP e4m3 operands (A, W point at bytes; strides in elements = bytes)
bool gemm256_fp8_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems);
int launch_gemm256_fp8(const GemmParams& p, int bn, int out_f32, hipStream_t st);

}  // namespace osk_gemm
