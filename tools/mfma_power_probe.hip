// Sustained MFMA throughput under the board's power cap, by instruction shape and data (gfx950):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe tools/mfma_power_probe.hip && /tmp/mfma_probe
// Every SIMD runs WAVES waves of a register-only loop of independent MFMAs (no LDS, no memory) for ~1.5 s per
// configuration: what is measured is the matrix pipe's sustained rate at the clock the power management settles on.
// Operand data: "rand" = pseudo-random bf16 in [-1, 1) per lane and register, "zero" = 0 (a DVFS probe, never a result).
// Question behind it: do 32x32x16 and 16x16x32 bf16 MFMAs cost the same energy per flop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bf16x8 rnd8(unsigned seed, int zero) {
  bf16x8 v;
  for (int j = 0; j < 8; ++j) {
    unsigned h = hash(seed * 8 + j);
    float f = ((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f);
    v[j] = (__bf16)(zero ? 0.f : f);
  }
  return v;
}

// 4 independent 32x32 accumulator tiles (64 registers), NA x NB operand fragments
template <int ITER>
__global__ void __launch_bounds__(256) k32(float* out, int zero, int reps) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a0 = rnd8(t * 4 + 0, zero), a1 = rnd8(t * 4 + 1, zero), b0 = rnd8(t * 4 + 2, zero), b1 = rnd8(t * 4 + 3, zero);
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = c1[r] = c2[r] = c3[r] = 0.f; }
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[t] = s;
}

// 16 independent 16x16 accumulator tiles (64 registers), 4 x 4 operand fragments
template <int ITER>
__global__ void __launch_bounds__(256) k16(float* out, int zero, int reps) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(t * 8 + i, zero); b[i] = rnd8(t * 8 + 4 + i, zero); }
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) c[x * 4 + y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[x], b[y], c[x * 4 + y], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == 12345.678f) out[t] = s;
}

template <class F>
double run(F launch, double flops_per_launch, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  // ~1.5 s of back-to-back launches so the power management reaches its steady state; time the last 2/3
  int n = 0;
  float ms = 0.f;
  CK(hipEventRecord(e0));
  for (n = 0; n < 10; ++n) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  int total = (int)(1500.0f / (ms / 10)) + 1;
  for (int i = 0; i < total / 3; ++i) launch();
  CK(hipEventRecord(e0));
  int m = total * 2 / 3 + 1;
  for (int i = 0; i < m; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  double tf = flops_per_launch * m / (ms * 1e-3) / 1e12;
  printf("%-44s %8.1f TFLOP/s sustained  (%.3f ms per launch)\n", name, tf, ms / m);
  return tf;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, ncu);
  float* out; CK(hipMalloc(&out, (size_t)ncu * 8 * 256 * sizeof(float)));
  const int reps = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int grid = ncu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD and block
    for (int zero = 0; zero <= 1; ++zero) {
      char nm[128];
      const double f32 = (double)grid * 4 * reps * 8 * 4 * (2.0 * 32 * 32 * 16);
      snprintf(nm, sizeof nm, "32x32x16 bf16, %d wave(s)/SIMD, %s data", waves_per_simd, zero ? "zero" : "rand");
      run([&] { hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(256), 0, 0, out, zero, reps); }, f32, nm);
      const double f16 = (double)grid * 4 * reps * 2 * 16 * (2.0 * 16 * 16 * 32);
      snprintf(nm, sizeof nm, "16x16x32 bf16, %d wave(s)/SIMD, %s data", waves_per_simd, zero ? "zero" : "rand");
      run([&] { hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, out, zero, reps); }, f16, nm);
    }
  }
  return 0;
}
