// Victim / aggressor micro-kernels for tools/interfere_probe.py: which hardware resource of a CU lets one workgroup's
// kernel disturb a co-resident workgroup of ANOTHER kernel (other stream / other process)?  Found in round 3: the batched
// adaLN GEMV returned a wrong batch-0 output now and then while a hand-scheduled LDS-DMA / MFMA kernel ran on the same GPU.
//
// Victims verify ONE resource each in a long loop and report mismatches; aggressors exercise ONE resource each.
//   build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/interfere_probe.hip -o tools/lib/libinterfere.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));

struct Report {           // device buffer: [0] = mismatch count, then up to 15 records of 4 words
  unsigned count;
  unsigned pad[3];
  unsigned rec[15][4];
};

__device__ void report(Report* r, unsigned a, unsigned b, unsigned c, unsigned d) {
  const unsigned i = atomicAdd(&r->count, 1u);
  if (i < 15) { r->rec[i][0] = a; r->rec[i][1] = b; r->rec[i][2] = c; r->rec[i][3] = d; }
}

// ---------------------------------------------------------------- victims
// LDS reads: NF floats of dynamic LDS written once, then read back `iters` times with 16-byte reads (the GEMV's x staging)
extern "C" __global__ void __launch_bounds__(256) victim_lds(Report* r, int nf, int iters) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int i = threadIdx.x; i < nf; i += 256) xs[i] = __uint_as_float(0x3F000000u + (unsigned)i * 7u + blockIdx.x);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int c = threadIdx.x; c * 4 < nf; c += 256) {
      const f4_t v = *reinterpret_cast<volatile f4_t*>(&xs[c * 4]);
      const unsigned e0 = 0x3F000000u + (unsigned)(c * 4) * 7u + blockIdx.x;
      if (__float_as_uint(v.x) != e0 || __float_as_uint(v.y) != e0 + 7u || __float_as_uint(v.z) != e0 + 14u || __float_as_uint(v.w) != e0 + 21u)
        report(r, 1u, (unsigned)(c * 4), __float_as_uint(v.x), e0);
    }
  }
}

// cross-lane reductions (ds_bpermute / DPP as hipcc emits them for __shfl_xor)
extern "C" __global__ void __launch_bounds__(256) victim_shfl(Report* r, int iters) {
  const int lane = threadIdx.x & 63;
  float base = (float)(lane + 1);
  for (int it = 0; it < iters; ++it) {
    float v = base + (float)(it & 7);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const float expect = 2080.0f + 64.0f * (float)(it & 7);
    if (v != expect) report(r, 2u, (unsigned)it, __float_as_uint(v), __float_as_uint(expect));
  }
}

// register residency: 24 VGPRs hold known values through a long dependent FMA chain (x = x * 1 + 0)
extern "C" __global__ void __launch_bounds__(256) victim_reg(Report* r, int iters) {
  float x[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) x[i] = (float)(threadIdx.x * 32 + i);
  float one = 1.0f, zero = 0.0f;
  asm volatile("" : "+v"(one), "+v"(zero));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) x[i] = __builtin_fmaf(x[i], one, zero);
  }
#pragma unroll
  for (int i = 0; i < 24; ++i)
    if (x[i] != (float)(threadIdx.x * 32 + i)) report(r, 3u, (unsigned)i, __float_as_uint(x[i]), threadIdx.x);
}

// global loads: a read-only buffer of known content, 16-byte loads (the GEMV's weight stream)
extern "C" __global__ void __launch_bounds__(256) victim_gld(Report* r, const unsigned* buf, int n16, int iters) {
  for (int it = 0; it < iters; ++it) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < n16; c += gridDim.x * 256) {
      const u4_t v = *reinterpret_cast<const volatile u4_t*>(buf + c * 4);
      const unsigned e0 = (unsigned)(c * 4) * 2654435761u;
      if (v.x != e0 || v.y != e0 + 2654435761u || v.z != e0 + 2u * 2654435761u || v.w != e0 + 3u * 2654435761u)
        report(r, 4u, (unsigned)c, v.x, e0);
    }
  }
}

// the GEMV's inner structure in one: x in LDS, weights from global, f32 FMA, wave reduction, lane 0 compares
extern "C" __global__ void __launch_bounds__(256) victim_gemvlike(Report* r, const unsigned* wbuf, int K, int iters) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [2][K]
  for (int i = threadIdx.x; i < 2 * K; i += 256) xs[i] = (float)((i * 37) & 15) * 0.0625f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ref0 = 0.f, ref1 = 0.f;
  for (int it = 0; it < iters; ++it) {
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane; c * 8 < K; c += 64) {
      const uint4 u = *reinterpret_cast<const uint4*>(wbuf + ((wave * 64 + (it & 15)) * (K / 2)) + c * 4);
      const float w[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u),
                          __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xFFFF0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xFFFF0000u)};
      const float4 x0 = *reinterpret_cast<const float4*>(&xs[c * 8]), x1 = *reinterpret_cast<const float4*>(&xs[c * 8 + 4]);
      const float4 y0 = *reinterpret_cast<const float4*>(&xs[K + c * 8]), y1 = *reinterpret_cast<const float4*>(&xs[K + c * 8 + 4]);
      a0 += w[0] * x0.x + w[1] * x0.y + w[2] * x0.z + w[3] * x0.w + w[4] * x1.x + w[5] * x1.y + w[6] * x1.z + w[7] * x1.w;
      a1 += w[0] * y0.x + w[1] * y0.y + w[2] * y0.z + w[3] * y0.w + w[4] * y1.x + w[5] * y1.y + w[6] * y1.z + w[7] * y1.w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
    if (it < 16) { if ((it & 15) == 0) { ref0 = a0; ref1 = a1; } }
    if ((it & 15) == 0 && it >= 16) {
      if (a0 != ref0) report(r, 5u, (unsigned)it, __float_as_uint(a0), __float_as_uint(ref0));
      if (a1 != ref1) report(r, 6u, (unsigned)it, __float_as_uint(a1), __float_as_uint(ref1));
    }
  }
}

// variants of victim_gemvlike that drop one ingredient each: mode bit 0 = no global loads (weights from the lane id),
// bit 1 = no cross-lane reduction (every lane checks its own partial sums), bit 2 = x of BOTH batches read with ds_read_b128
// from 16-byte aligned addresses (K multiple of 4 floats), bit 3 = batch 1 staged FIRST in LDS (low addresses), batch 0 behind
extern "C" __global__ void __launch_bounds__(256) victim_gemvvar(Report* r, const unsigned* wbuf, int K, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [2][K]
  for (int i = threadIdx.x; i < 2 * K; i += 256) xs[i] = (float)((i * 37) & 15) * 0.0625f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o0 = (mode & 8) ? K : 0, o1 = (mode & 8) ? 0 : K;
  float ref0 = 0.f, ref1 = 0.f;
  for (int it = 0; it < iters; ++it) {
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane; c * 8 < K; c += 64) {
      uint4 u;
      if (mode & 1) u = make_uint4(0x3F803F00u + lane, 0x3F003E80u + c, 0x3E803F80u, 0x3F803F80u);
      else u = *reinterpret_cast<const uint4*>(wbuf + ((wave * 64 + (it & 15)) * (K / 2)) + c * 4);
      const float w[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u),
                          __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xFFFF0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xFFFF0000u)};
      f4_t x0, x1, y0, y1;
      if (mode & 4) {
        x0 = *reinterpret_cast<volatile f4_t*>(&xs[o0 + c * 8]); x1 = *reinterpret_cast<volatile f4_t*>(&xs[o0 + c * 8 + 4]);
        y0 = *reinterpret_cast<volatile f4_t*>(&xs[o1 + c * 8]); y1 = *reinterpret_cast<volatile f4_t*>(&xs[o1 + c * 8 + 4]);
      } else {
        x0 = *reinterpret_cast<const f4_t*>(&xs[o0 + c * 8]); x1 = *reinterpret_cast<const f4_t*>(&xs[o0 + c * 8 + 4]);
        y0 = *reinterpret_cast<const f4_t*>(&xs[o1 + c * 8]); y1 = *reinterpret_cast<const f4_t*>(&xs[o1 + c * 8 + 4]);
      }
      a0 += w[0] * x0.x + w[1] * x0.y + w[2] * x0.z + w[3] * x0.w + w[4] * x1.x + w[5] * x1.y + w[6] * x1.z + w[7] * x1.w;
      a1 += w[0] * y0.x + w[1] * y0.y + w[2] * y0.z + w[3] * y0.w + w[4] * y1.x + w[5] * y1.y + w[6] * y1.z + w[7] * y1.w;
    }
    if (!(mode & 2)) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
    }
    if (it < 16) { if ((it & 15) == 0) { ref0 = a0; ref1 = a1; } }
    if ((it & 15) == 0 && it >= 16) {
      if (a0 != ref0) report(r, 0x50u + mode, (unsigned)it, __float_as_uint(a0), __float_as_uint(ref0));
      if (a1 != ref1) report(r, 0x60u + mode, (unsigned)it, __float_as_uint(a1), __float_as_uint(ref1));
    }
  }
}

// victim-side bisect: x from LDS (src 0) or from global memory (src 1); accumulation by scalar v_fma_f32 (op 0), by explicit
// v_pk_fma_f32 whose lo / hi halves are batch 0 / batch 1 (op 1).  Weights from the lane id, no reduction: a lane checks its own sums.
extern "C" __global__ void __launch_bounds__(256) victim_fmasrc(Report* r, const float* xg, int K, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [2][K]
  for (int i = threadIdx.x; i < 2 * K; i += 256) xs[i] = xg[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool from_global = mode & 1, pk = mode & 2;
  float ref0 = 0.f, ref1 = 0.f;
  for (int it = 0; it < iters; ++it) {
    float a0 = 0.f, a1 = 0.f;
    f2_t a01 = {0.f, 0.f};
    for (int c = lane; c * 4 < K; c += 64) {
      f4_t x, y;
      if (from_global) { x = *reinterpret_cast<const volatile f4_t*>(xg + c * 4); y = *reinterpret_cast<const volatile f4_t*>(xg + K + c * 4); }
      else { x = *reinterpret_cast<volatile f4_t*>(&xs[c * 4]); y = *reinterpret_cast<volatile f4_t*>(&xs[K + c * 4]); }
      const float w0 = 1.0f + lane * 0.015625f, w1 = 0.5f + c * 0.03125f;
      if (pk) {
        f2_t p0 = {x.x, y.x}, p1 = {x.y, y.y}, p2 = {x.z, y.z}, p3 = {x.w, y.w}, ww0 = {w0, w0}, ww1 = {w1, w1};
#ifndef PROBE_NOPK
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %3, %4, %0\n v_pk_fma_f32 %0, %5, %2, %0\n v_pk_fma_f32 %0, %6, %4, %0"
                     : "+v"(a01) : "v"(p0), "v"(ww0), "v"(p1), "v"(ww1), "v"(p2), "v"(p3));
#endif
      } else {
        asm volatile("v_fma_f32 %0, %2, %6, %0\n v_fma_f32 %0, %3, %7, %0\n v_fma_f32 %0, %4, %6, %0\n v_fma_f32 %0, %5, %7, %0\n"
                     "v_fma_f32 %1, %8, %6, %1\n v_fma_f32 %1, %9, %7, %1\n v_fma_f32 %1, %10, %6, %1\n v_fma_f32 %1, %11, %7, %1"
                     : "+v"(a0), "+v"(a1) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(w0), "v"(w1), "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w));
      }
    }
    if (pk) { a0 = a01.x; a1 = a01.y; }
    if (it == 0) { ref0 = a0; ref1 = a1; }
    else {
      if (a0 != ref0) report(r, 0x70u + mode, (unsigned)it, __float_as_uint(a0), __float_as_uint(ref0));
      if (a1 != ref1) report(r, 0x80u + mode, (unsigned)it, __float_as_uint(a1), __float_as_uint(ref1));
    }
  }
}

// ---------------------------------------------------------------- aggressors
// LDS-DMA only: every wave streams 1 KiB pieces of a global buffer into its workgroup's LDS at offset `base` + wave KiB
extern "C" __global__ void __launch_bounds__(512) aggr_dma(const unsigned* src, int iters, int lds_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned* g = src + (blockIdx.x * 512 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (it & 63) * 2048),
                                     (__attribute__((address_space(3))) void*)(smem + lds_off + wave * 1024), 16, 0, 0);
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 999) smem[0] = 1;
}

// MFMA only, accumulators pinned in AGPRs by the compiler's choice
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
extern "C" __global__ void __launch_bounds__(512) aggr_mfma(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * (i + 1)); }
  f32x16_t acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 123.456f) out[0] = s;
}

// LDS reads / writes only (ds_read_b128 / ds_write_b128 over `bytes` of dynamic LDS)
extern "C" __global__ void __launch_bounds__(512) aggr_lds(float* out, int bytes, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int n16 = bytes / 16;
  for (int it = 0; it < iters; ++it) {
    for (int c = threadIdx.x; c < n16; c += 512) {
      u4_t v = reinterpret_cast<volatile u4_t*>(smem)[c];
      acc.x += v.x; acc.y ^= v.y;
      u4_t w = {acc.x, (unsigned)it, (unsigned)c, acc.y};
      reinterpret_cast<volatile u4_t*>(smem)[(c + 257) % n16] = w;
    }
  }
  if (acc.x == 0x12345u) out[1] = 1.f;
}

// one instruction class per aggressor, 512 threads, `iters` repetitions of an 8-instruction group
#define AGGR_ASM(NAME, BODY)                                                                       \
  extern "C" __global__ void __launch_bounds__(512) NAME(float* out, int iters) {                  \
    unsigned a = threadIdx.x * 3u + 1u, b = threadIdx.x * 5u + 2u, c = a ^ b, d = a + b;           \
    for (int it = 0; it < iters; ++it) { BODY BODY BODY BODY BODY BODY BODY BODY }                 \
    if ((a ^ b ^ c ^ d) == 0x7654321u) out[2] = 1.f;                                               \
  }
AGGR_ASM(aggr_swap32, asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1" : "+v"(a), "+v"(b));)
AGGR_ASM(aggr_swap16, asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1\n s_nop 1" : "+v"(c), "+v"(d));)
AGGR_ASM(aggr_cvtpk, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b)); a += c;)
AGGR_ASM(aggr_accrw, asm volatile("v_accvgpr_write_b32 a0, %1\n s_nop 1\n v_accvgpr_read_b32 %0, a0" : "=v"(c) : "v"(a) : "a0"); a += c;)
AGGR_ASM(aggr_valu, asm volatile("v_add_u32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b)); a += c;)

// many live registers with lane-specific content, a long quiet spin, then a check of every register in every lane
extern "C" __global__ void __launch_bounds__(256) victim_reg64(Report* r, int iters) {
  unsigned x[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) { x[i] = (threadIdx.x << 8) + i; asm volatile("" : "+v"(x[i])); }
  unsigned spin = 0;
  for (int it = 0; it < iters; ++it) { asm volatile("v_add_u32 %0, %0, 1\n s_nop 3" : "+v"(spin)); }
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    asm volatile("" : "+v"(x[i]));
    if (x[i] != (threadIdx.x << 8) + i) report(r, 7u, (unsigned)i | (threadIdx.x << 8), x[i], spin);
  }
}

// packed f32 FMA chain with identical lo / hi inputs: the halves must stay identical
extern "C" __global__ void __launch_bounds__(256) victim_pkfma(Report* r, int iters) {
  f2_t acc = {0.f, 0.f};
  const float w0 = 1.0f + (threadIdx.x & 15) * 0.03125f, x0 = 0.5f + (threadIdx.x >> 4) * 0.0078125f;
  f2_t w = {w0, w0}, x = {x0, x0};
  asm volatile("" : "+v"(w), "+v"(x));
  for (int it = 0; it < iters; ++it) {
#ifndef PROBE_NOPK
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
#endif
    if ((it & 255) == 255) {
      if (acc.x != acc.y) report(r, 8u, (unsigned)it, __float_as_uint(acc.x), __float_as_uint(acc.y));
      acc.x = 0.f; acc.y = 0.f;
    }
  }
}

extern "C" int probe_launch(const char* name, void* stream, int grid, int smem, void* p0, void* p1, int i0, int i1, int i2) {
  hipStream_t st = (hipStream_t)stream;
  auto set = [&](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); };
#define IS(n) (__builtin_strcmp(name, n) == 0)
  if (IS("victim_lds")) { set((const void*)victim_lds); hipLaunchKernelGGL(victim_lds, dim3(grid), dim3(256), smem, st, (Report*)p0, i0, i1); }
  else if (IS("victim_shfl")) hipLaunchKernelGGL(victim_shfl, dim3(grid), dim3(256), 0, st, (Report*)p0, i0);
  else if (IS("victim_reg")) hipLaunchKernelGGL(victim_reg, dim3(grid), dim3(256), 0, st, (Report*)p0, i0);
  else if (IS("victim_gld")) hipLaunchKernelGGL(victim_gld, dim3(grid), dim3(256), 0, st, (Report*)p0, (const unsigned*)p1, i0, i1);
  else if (IS("victim_gemvlike")) { set((const void*)victim_gemvlike); hipLaunchKernelGGL(victim_gemvlike, dim3(grid), dim3(256), smem, st, (Report*)p0, (const unsigned*)p1, i0, i1); }
  else if (IS("victim_gemvvar")) { set((const void*)victim_gemvvar); hipLaunchKernelGGL(victim_gemvvar, dim3(grid), dim3(256), smem, st, (Report*)p0, (const unsigned*)p1, i0, i1, i2); }
  else if (IS("victim_fmasrc")) { set((const void*)victim_fmasrc); hipLaunchKernelGGL(victim_fmasrc, dim3(grid), dim3(256), smem, st, (Report*)p0, (const float*)p1, i0, i1, i2); }
  else if (IS("aggr_dma")) { set((const void*)aggr_dma); hipLaunchKernelGGL(aggr_dma, dim3(grid), dim3(512), smem, st, (const unsigned*)p1, i0, i1); }
  else if (IS("aggr_mfma")) hipLaunchKernelGGL(aggr_mfma, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("aggr_lds")) { set((const void*)aggr_lds); hipLaunchKernelGGL(aggr_lds, dim3(grid), dim3(512), smem, st, (float*)p0, i0, i1); }
  else if (IS("aggr_swap32")) hipLaunchKernelGGL(aggr_swap32, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("aggr_swap16")) hipLaunchKernelGGL(aggr_swap16, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("aggr_cvtpk")) hipLaunchKernelGGL(aggr_cvtpk, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("aggr_accrw")) hipLaunchKernelGGL(aggr_accrw, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("aggr_valu")) hipLaunchKernelGGL(aggr_valu, dim3(grid), dim3(512), 0, st, (float*)p0, i0);
  else if (IS("victim_reg64")) hipLaunchKernelGGL(victim_reg64, dim3(grid), dim3(256), 0, st, (Report*)p0, i0);
  else if (IS("victim_pkfma")) hipLaunchKernelGGL(victim_pkfma, dim3(grid), dim3(256), 0, st, (Report*)p0, i0);
  else return -1;
  return (int)hipGetLastError();
}
