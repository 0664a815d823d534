// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the Open-Sora denoise path.
// gfx950 only: wave64, MFMA 32x32x16 bf16, 160 KiB LDS.  No portability layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define OSK_OK 0
#define OSK_EINVAL (-1)       // bad shape / alignment / unsupported configuration
#define OSK_EUNSUPPORTED (-2) // head_dim / dtype combination not compiled

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define OSK_DEV static __device__ __forceinline__

OSK_DEV float bf16_bits_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
OSK_DEV float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
OSK_DEV float bf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// round-to-nearest-even pack of two fp32 into one dword of two bf16 (lo = a): v_cvt_pk_bf16_f32 on gfx950
OSK_DEV unsigned pack_bf16x2(float a, float b) {
  f32x2_t v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
OSK_DEV unsigned short f32_to_bf16_bits(float a) {
  __bf16 r = (__bf16)a;
  return __builtin_bit_cast(unsigned short, r);
}

// 8 bf16 (one uint4) -> 8 fp32
OSK_DEV void unpack8(const uint4& u, float* f) {
  f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x);
  f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
  f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z);
  f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
OSK_DEV uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack_bf16x2(f[0], f[1]);
  u.y = pack_bf16x2(f[2], f[3]);
  u.z = pack_bf16x2(f[4], f[5]);
  u.w = pack_bf16x2(f[6], f[7]);
  return u;
}

OSK_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
OSK_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

OSK_DEV float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  == x * sigmoid(2 u) == x / (1 + 2^(-2 log2(e) u)):
  // two multiplies, one fma, v_exp_f32, one add, v_rcp_f32, one multiply (1-ulp transcendentals: the result is
  // rounded to bf16 by the caller).  x -> -inf: 2^(+inf) = inf, rcp = 0, result -0; x -> +inf: x.
  // (round 5: the exponent's constant factor folded into the cubic's coefficients -- one multiply less per element; the GELU class
  // costs the GEMM epilogue 11.5 k cycles per 256 x 256 tile, profiles/r05a_gemm_tile_timing_by_epilogue_class.jsonl)
  constexpr float K0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(0.044715f * K0, x * x, K0));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
OSK_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

// 4 x 4 transpose of (register, lane-of-a-quad): X[r] of lane j -> X[j] of lane r, within every quad of lanes; two DPP butterfly
// steps (lane bit 0 against register bit 0, then bit 1 against bit 1).  The 16 x 16 MFMA epilogues (gemm_epilogue16.h,
// conv3d_256.hip) use it to turn "a lane owns 16 bytes of ITS row in each of four column blocks" into "a quad owns 64 contiguous
// bytes of each of its four rows": stores that cover whole 128-byte row pieces.
OSK_DEV unsigned dpp_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }   // quad_perm [1,0,3,2]
OSK_DEV unsigned dpp_swap2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }   // quad_perm [2,3,0,1]
OSK_DEV void quad_transpose(unsigned& x0, unsigned& x1, unsigned& x2, unsigned& x3, bool odd, bool hi) {
  const unsigned s0 = dpp_swap1(x0), s1 = dpp_swap1(x1), s2 = dpp_swap1(x2), s3 = dpp_swap1(x3);
  const unsigned y0 = odd ? s1 : x0, y1 = odd ? x1 : s0, y2 = odd ? s3 : x2, y3 = odd ? x3 : s2;
  const unsigned t0 = dpp_swap2(y0), t1 = dpp_swap2(y1), t2 = dpp_swap2(y2), t3 = dpp_swap2(y3);
  x0 = hi ? t2 : y0;
  x1 = hi ? t3 : y1;
  x2 = hi ? y2 : t0;
  x3 = hi ? y3 : t1;
}

// Bijective XCD-aware remap of a 1-D block id (block b is observed on XCD b % 8): give every XCD a
// contiguous range of logical tiles so neighbouring tiles share that XCD's private 4 MiB L2.
OSK_DEV int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- per-DEVICE one-time launch state (host side).  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current
// device only, and the CU count differs between devices: a process that drives several GPUs, or two host threads making their
// first call at the same time, must not share one `static bool`.  One bit / one slot per device ordinal, atomics;
// setting the attribute twice is harmless, so a lost race costs one redundant call.
#define OSK_ENSURE_MAX_SMEM(KERNEL, BYTES)                                                                         \
  do {                                                                                                             \
    static std::atomic<uint64_t> osk_done_{0};                                                                     \
    int osk_dev_ = 0;                                                                                              \
    hipError_t osk_e_ = hipGetDevice(&osk_dev_);                                                                   \
    if (osk_e_ != hipSuccess) return (int)osk_e_;                                                                  \
    const uint64_t osk_bit_ = 1ull << (osk_dev_ & 63);                                                             \
    if (!(osk_done_.load(std::memory_order_acquire) & osk_bit_)) {                                                 \
      osk_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES)); \
      if (osk_e_ != hipSuccess) return (int)osk_e_;                                                                \
      osk_done_.fetch_or(osk_bit_, std::memory_order_release);                                                     \
    }                                                                                                              \
  } while (0)

// compute units of the CURRENT device (cached per ordinal; 256 if the query fails)
static inline int osk_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::atomic<int>& slot = cus[dev & 63];
  int n = slot.load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    slot.store(n, std::memory_order_relaxed);
  }
  return n;
}
