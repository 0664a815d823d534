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
};

// large tiles need M >= 256, N >= 128, K % 64 == 0 and both operand tensors within the kernels' 32-bit per-lane offsets
bool gemm256_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems);
// gemm256p.hip: 256 x 128 tiles walked by one persistent 8-wave workgroup per CU (cross-tile prefetch, bias-initialised accumulators)
int launch_gemm256p(const GemmParams& p, int out_f32, hipStream_t st);
// gemm256x.hip: 256 x 256 tiles, 4 waves (one per SIMD, 128 x 128 wave tiles, 256 accumulator AGPRs) on v_mfma_f32_16x16x32_bf16
int launch_gemm256x(const GemmParams& p, int out_f32, hipStream_t st);
// the same kernel over TWO problems that share N, K, gelu_from, group (one tile list: the second problem fills the first one's last round)
int launch_gemm256x_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t st);
// gemm256.hip: the one-tile-per-workgroup frame on OCP e4m3 operands (A, W point at bytes; strides in elements = bytes)
bool gemm256_fp8_supported(const GemmParams& p, int64_t a_span_elems, int64_t w_span_elems);
int launch_gemm256_fp8(const GemmParams& p, int bn, int out_f32, hipStream_t st);

}  // namespace osk_gemm
