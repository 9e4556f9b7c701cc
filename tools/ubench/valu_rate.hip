// valu_rate.hip -- what does each VALU instruction of the attention / GEMM inner loops cost, alone and beside a stream of bf16
// MFMAs (v_mfma_f32_32x32x16_bf16: 8 passes)?  Round 4 design input for attention_x3 (VALU-issue bound).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
// Output: cycles (s_memtime ticks scaled by the bare-MFMA stream = 32 cycles... printed raw) per instruction, 1 and 2 waves / SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define OPS(X) \
  X(0, "v_fma_f32 %0, %0, %1, %0") \
  X(1, "v_dot2c_f32_bf16 %0, %1, %1") \
  X(2, "v_cvt_pk_bf16_f32 %0, %0, %1") \
  X(3, "v_sub_f32 %0, %0, %1") \
  X(4, "v_and_b32 %0, 0xffff0000, %1") \
  X(5, "v_exp_f32 %0, %0") \
  X(6, "v_max3_f32 %0, %0, %1, %1") \
  X(7, "v_lshlrev_b32 %0, 16, %1") \
  X(8, "v_mul_f32 %0, %0, %1") \
  X(9, "v_pk_mul_f32 %0, %0, %1")

template <int KIND>
__device__ __forceinline__ void op(float& a, float b) {
#define X(k, s) if (KIND == k) asm volatile(s : "+v"(a) : "v"(b));
  OPS(X)
#undef X
}
__device__ __forceinline__ void op9(double& a, double b) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }

// FILL VALU instructions per MFMA (MFMA = 0: no MFMA at all, FILL instructions per "slot")
template <int KIND, int FILL, int MFMA>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
  f32x16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float f[8];
  double g[8];
  for (int q = 0; q < 8; ++q) { f[q] = (float)threadIdx.x * 0.001f + q; g[q] = q; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MFMA) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < FILL; ++v) {
        if (KIND == 9) op9(g[v & 7], g[(v + 1) & 7]);
        else op<KIND>(f[v & 7], f[(v + 1) & 7]);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 8; ++q) s += f[q] + (float)g[q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* d, long long* dc) {
  const int iters = 2000;
  auto go = [&](auto kern, int threads) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, iters);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    return (double)c / (iters * 4.0);
  };
  const double alone1 = go(k<KIND, 8, 0>, 256) / 8, alone2 = go(k<KIND, 8, 0>, 512) / 8;
  const double m4_1 = go(k<KIND, 4, 1>, 256), m8_1 = go(k<KIND, 8, 1>, 256), m12_1 = go(k<KIND, 12, 1>, 256);
  const double m4_2 = go(k<KIND, 4, 1>, 512), m8_2 = go(k<KIND, 8, 1>, 512), m12_2 = go(k<KIND, 12, 1>, 512);
  printf("%-34s alone: %.2f / %.2f ticks per instr (1 / 2 waves per SIMD) | per MFMA slot with +4/+8/+12 of them: 1 wave %.1f %.1f %.1f, 2 waves %.1f %.1f %.1f\n",
         name, alone1, alone2, m4_1, m8_1, m12_1, m4_2, m8_2, m12_2);
}

int main() {
  float* d; long long* dc;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&dc, 64);
  {
    hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(256), 0, 0, d, dc, 2000); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(256), 0, 0, d, dc, 2000);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("bare bf16 MFMA stream, 1 wave per SIMD: %.2f ticks per MFMA (= 32 shader cycles: 1 tick = %.2f cycles)\n", c / 8000.0, 32.0 / (c / 8000.0));
    hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(512), 0, 0, d, dc, 2000); hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("bare bf16 MFMA stream, 2 waves per SIMD: %.2f ticks per MFMA and wave\n", c / 8000.0);
  }
#define X(kk, s) run<kk>(s, d, dc);
  OPS(X)
#undef X
  return 0;
}
