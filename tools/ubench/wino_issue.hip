// wino_issue.hip -- round 5 design input for the U-reuse Winograd kernels: what does VALU work cost beside a stream of 16-cycle
// fp16 MFMAs (v_mfma_f32_16x16x32_f16), with one and with two waves per SIMD, when the VALU instructions sit BETWEEN the wave's own
// MFMAs (fine interleave: FILL instructions after every MFMA) -- and what a B operand read from LDS per MFMA adds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wino_issue.hip -o tools/ubench/wino_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_fma_f32; 1: v_pk_fma_f32; 2: the transform's mix per 6 instructions (2 pk_fma, 1 cvt_pk_f16, 2 dot2c_f16, 1 cvt_pk_f16)
template <int KIND>
__device__ __forceinline__ void fill(int v, float* f, f32x2* g) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"(f[(v + 1) & 7]));
  if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(g[v & 7]) : "v"(g[(v + 1) & 7]));
  if (KIND == 2) {
    const int r = v % 6;
    if (r < 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(g[v & 7]) : "v"(g[(v + 1) & 7]));
    else if (r == 2 || r == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f[v & 7]) : "v"(f[(v + 1) & 7]), "v"(f[(v + 2) & 7]));
    else asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(f[v & 7]) : "v"(f[(v + 1) & 7]));
  }
}

// NM MFMAs per iteration on independent accumulators; FILL VALU after each; LDSR: one ds_read_b128 pair (two planes) per MFMA triple
template <int KIND, int FILL, int MFMA, int LDSR>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[24 * 512 * 2];
  for (int i = threadIdx.x; i < 24 * 512 * 2; i += blockDim.x) lds[i] = (_Float16)(float)(i & 15);
  f16x8 a, b, b2;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)((threadIdx.x + j) & 7); b[j] = (_Float16)(float)(j + 1); b2[j] = b[j]; }
  f32x4 c[12];
  for (int q = 0; q < 12; ++q) c[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float f[8];
  f32x2 g[8];
  for (int q = 0; q < 8; ++q) { f[q] = (float)threadIdx.x * 0.001f + q; g[q] = (f32x2){(float)q, 1.f}; }
  const _Float16* rd = lds + (threadIdx.x & 63) * 8;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      if (LDSR && (q % 3) == 0) {
        b = *reinterpret_cast<const f16x8*>(rd + ((q / 3 + it) % 24) * 512);
        b2 = *reinterpret_cast<const f16x8*>(rd + 24 * 512 + ((q / 3 + it) % 24) * 512);
      }
      if (MFMA) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, (q % 3) == 1 ? b2 : b, c[q], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < FILL; ++v) fill<KIND>(v + q * FILL, f, g);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int q = 0; q < 12; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  for (int q = 0; q < 8; ++q) s += f[q] + g[q][0] + g[q][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <typename K>
double go(K kern, int threads, float* d, long long* dc) {
  const int iters = 1000;
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, dc, iters);
  long long c[16]; hipMemcpy(c, dc, 128, hipMemcpyDeviceToHost);
  long long lo = c[0], hi = c[1];
  for (int w = 0; w < threads / 64; ++w) { lo = c[2 * w] < lo ? c[2 * w] : lo; hi = c[2 * w + 1] > hi ? c[2 * w + 1] : hi; }
  return (double)(hi - lo) / (iters * 12.0);   // the workgroup's span: every wave's MFMAs done
}

template <int KIND, int LDSR>
void row(const char* name, float* d, long long* dc) {
  printf("%-40s", name);
  printf(" 1 wave/SIMD:");
  printf(" %5.1f", go(k<KIND, 0, 1, LDSR>, 256, d, dc)); printf(" %5.1f", go(k<KIND, 2, 1, LDSR>, 256, d, dc));
  printf(" %5.1f", go(k<KIND, 4, 1, LDSR>, 256, d, dc)); printf(" %5.1f", go(k<KIND, 5, 1, LDSR>, 256, d, dc));
  printf(" %5.1f", go(k<KIND, 6, 1, LDSR>, 256, d, dc)); printf(" %5.1f", go(k<KIND, 8, 1, LDSR>, 256, d, dc));
  printf(" | 2 waves/SIMD (span per wave-slot):");
  printf(" %5.1f", go(k<KIND, 0, 1, LDSR>, 512, d, dc)); printf(" %5.1f", go(k<KIND, 2, 1, LDSR>, 512, d, dc));
  printf(" %5.1f", go(k<KIND, 4, 1, LDSR>, 512, d, dc)); printf(" %5.1f", go(k<KIND, 5, 1, LDSR>, 512, d, dc));
  printf(" %5.1f", go(k<KIND, 6, 1, LDSR>, 512, d, dc)); printf(" %5.1f", go(k<KIND, 8, 1, LDSR>, 512, d, dc));
  printf(" | no MFMA, 6 per slot: %5.1f / %5.1f\n", go(k<KIND, 6, 0, 0>, 256, d, dc), go(k<KIND, 6, 0, 0>, 512, d, dc));
}

int main() {
  float* d; long long* dc;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&dc, 256);
  printf("ticks (s_memtime) per MFMA slot with +0/+2/+4/+5/+6/+8 VALU instructions after every v_mfma_f32_16x16x32_f16\n");
  row<0, 0>("v_fma_f32", d, dc);
  row<1, 0>("v_pk_fma_f32", d, dc);
  row<2, 0>("transform mix (2 pk_fma,cvt,2 dot2c,cvt)", d, dc);
  row<2, 1>("transform mix + B operands from LDS", d, dc);
  return 0;
}
