// Standalone hardware-semantics probes for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probes tools/probes.hip && /tmp/probes > gpurun_out/probes.txt
// 1. v_mfma_f32_32x32x16_bf16 / 16x16x32 operand + accumulator lane maps (hypothesis test vs CPU matmul)
// 2. global_load_lds_dwordx4 destination rule   3. ds_read_b64_tr_b16 gather rule   4. v_permlane32_swap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_mfma32(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x, hi = l >> 5, r31 = l & 31;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r31 * 16 + hi * 8 + j]; b[j] = (__bf16)B[(hi * 8 + j) * 32 + r31]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * hi; D[row * 32 + r31] = c[r]; }
}
__global__ void k_mfma16(const float* A /*16x32*/, const float* B /*32x16*/, float* D /*16x16*/) {
  const int l = threadIdx.x, g = l >> 4, r15 = l & 15;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r15 * 32 + g * 8 + j]; b[j] = (__bf16)B[(g * 8 + j) * 16 + r15]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + r15] = c[r];
}
__global__ void k_glds(const unsigned* g /*256 dwords*/, unsigned* out /*512 dwords*/) {
  __shared__ __attribute__((aligned(16))) unsigned lds[512];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // lane l sources global chunk (63 - l) (reversed) so the landing slot of each lane is visible
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (63 - l) * 4),
                                   (__attribute__((address_space(3))) void*)(lds + 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}
__global__ void k_trread(unsigned short* out /*64*4*/, int stride_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * stride_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)t[j];
}
__global__ void k_permlane(unsigned* out /*128*/) {
  const unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d MHz lds/blk %zu\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.sharedMemPerBlock);
  srand(1);
  {  // ---- MFMA 32x32x16
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (auto& x : A) x = (float)(rand() % 7 - 3);
    for (auto& x : B) x = (float)(rand() % 7 - 3);
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("MFMA 32x32x16 bf16 hypothesis A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], D row=(r&3)+8*(r>>2)+4*(l>>5) col=l&31: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  }
  {  // ---- MFMA 16x16x32
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (auto& x : A) x = (float)(rand() % 7 - 3);
    for (auto& x : B) x = (float)(rand() % 7 - 3);
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) R[i * 16 + n] += A[i * 32 + k] * B[k * 16 + n];
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("MFMA 16x16x32 bf16 hypothesis A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15], D row=4*(l>>4)+r col=l&15: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  }
  {  // ---- global_load_lds dwordx4
    std::vector<unsigned> g(256), o(512);
    for (int i = 0; i < 256; ++i) g[i] = i;  // chunk c holds dwords 4c..4c+3
    unsigned *dg, *dout; CK(hipMalloc(&dg, 1024)); CK(hipMalloc(&dout, 2048));
    CK(hipMemcpy(dg, g.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 0, 0, dg, dout); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64 && ok; ++l) for (int j = 0; j < 4; ++j) if (o[64 + l * 4 + j] != (unsigned)((63 - l) * 4 + j)) ok = 0;
    for (int i = 0; i < 64; ++i) if (o[i] != 0xdeadbeefu) ok = 0;
    for (int i = 320; i < 512; ++i) if (o[i] != 0xdeadbeefu) ok = 0;
    printf("global_load_lds_dwordx4 hypothesis lds[base + lane*16B] <- lane's own 16 B: %s\n", ok ? "PASS" : "FAIL");
    if (!ok) { printf("  lds dump (dword idx: value):"); for (int i = 0; i < 512; ++i) if (o[i] != 0xdeadbeefu) printf(" %d:%u", i, o[i]); printf("\n"); }
  }
  for (int stride : {4, 16}) {  // ---- ds_read_b64_tr_b16
    std::vector<unsigned short> o(256);
    unsigned short* dout; CK(hipMalloc(&dout, 512));
    hipLaunchKernelGGL(k_trread, dim3(1), dim3(64), 0, 0, dout, stride); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16 with lane address = lds + lane*%d elements (lds[i] = i); lane: 4 results\n", stride);
    for (int l = 0; l < 64; ++l) { printf("  L%02d: %4u %4u %4u %4u%s", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l & 3) == 3 ? "\n" : " |"); }
  }
  {  // ---- permlane32_swap
    std::vector<unsigned> o(128);
    unsigned* dout; CK(hipMalloc(&dout, 512));
    hipLaunchKernelGGL(k_permlane, dim3(1), dim3(64), 0, 0, dout); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("permlane32_swap(a=1000+l, b=2000+l): r0[0]=%u r0[31]=%u r0[32]=%u r0[63]=%u | r1[0]=%u r1[31]=%u r1[32]=%u r1[63]=%u\n",
           o[0], o[31], o[32], o[63], o[64], o[95], o[96], o[127]);
  }
  return 0;
}
