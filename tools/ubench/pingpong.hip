// pingpong.hip -- round 4 design input for attention_x3: two waves per SIMD, each alternating a chain of 12 DEPENDENT bf16 MFMAs
// (the six-term products of one 32 x 32 x 32 block into one accumulator) with a block of VALU work (the softmax / split mix of one
// attention tile: simple ops, v_exp_f32, v_dot2c_f32_bf16).  How long does a "tile" (2 MFMA chains + 2 VALU blocks) take per SIMD
//   (a) free running, both waves started together                      (what two co-resident workgroups do),
//   (b) with the two waves of a SIMD held in ANTI-PHASE by a workgroup barrier per half tile: one wave's barrier sits after its MFMA
//       chain, the other's after its VALU block (512-thread workgroup, waves w and w + 4 share a SIMD),
//   (c) one wave per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pingpong.hip -o tools/ubench/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NS, int NE, int ND>
__device__ __forceinline__ void valu_block(float (&f)[8]) {
#pragma unroll
  for (int v = 0; v < NS; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"(f[(v + 3) & 7]));
#pragma unroll
  for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(f[v & 7]));
#pragma unroll
  for (int v = 0; v < ND; ++v) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(f[v & 7]) : "v"(f[(v + 3) & 7]));
}
__device__ __forceinline__ void mfma_chain(f32x16& c, bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int i = 0; i < 12; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// MODE 0: free running; MODE 1: anti-phase by barriers (needs 512 threads)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int tiles) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
  f32x16 s, t;
  for (int r = 0; r < 16; ++r) { s[r] = 0.f; t[r] = 0.f; }
  float f[8];
  for (int q = 0; q < 8; ++q) f[q] = (float)threadIdx.x * 0.001f + q;
  const int team = threadIdx.x >> 8;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int it = 0; it < tiles; ++it) {
      mfma_chain(s, a, b);
      valu_block<60, 17, 32>(f);          // softmax + P split
      mfma_chain(t, a, b);
      valu_block<30, 0, 16>(f);           // staging split, fold
    }
  } else if (team == 0) {
    for (int it = 0; it < tiles; ++it) {
      mfma_chain(s, a, b);
      __builtin_amdgcn_s_barrier();
      valu_block<60, 17, 32>(f);
      __builtin_amdgcn_s_barrier();
      mfma_chain(t, a, b);
      __builtin_amdgcn_s_barrier();
      valu_block<30, 0, 16>(f);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    valu_block<30, 0, 16>(f);
    for (int it = 0; it < tiles; ++it) {
      __builtin_amdgcn_s_barrier();
      mfma_chain(s, a, b);
      __builtin_amdgcn_s_barrier();
      valu_block<60, 17, 32>(f);
      __builtin_amdgcn_s_barrier();
      mfma_chain(t, a, b);
      __builtin_amdgcn_s_barrier();
      valu_block<30, 0, 16>(f);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += s[r] + t[r];
  for (int q = 0; q < 8; ++q) acc += f[q];
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));     // the slowest wave of the workgroup
}

int main() {
  float* d; long long* dc;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&dc, 64);
  const int tiles = 500;
  auto go = [&](auto kern, int threads, const char* what, int waves) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, tiles); hipDeviceSynchronize();
    hipMemset(dc, 0, 8);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, tiles);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.0f cycles per tile and wave = %6.0f per tile on the SIMD (24 MFMAs = 768 cycles of matrix pipe per tile)\n", what, (double)c / tiles, (double)c / tiles / waves);
  };
  go(k<0>, 256, "one wave per SIMD", 1);
  go(k<0>, 512, "two waves per SIMD, free running", 2);
  go(k<1>, 512, "two waves per SIMD, anti-phase by a barrier per half tile", 2);
  return 0;
}
