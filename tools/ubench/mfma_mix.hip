// Microbenchmark 2: NW waves per SIMD, each alternating a block of MFMAs with a block of VALU work (the shape of a
// flash-attention tile: QK^T MFMAs -> softmax VALU -> PV MFMAs).  Compares v_mfma_f32_32x32x2_f32 (64 cycles) with
// v_mfma_f32_16x16x4_f32 (32 cycles) at equal matrix work.  Build: hipcc --offload-arch=gfx950 -O3 mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE, int NV>      // SHAPE 32: 16 x 32x32x2 per block; 16: 32 x 16x16x4 per block; NV = VALU (pk_fma) per block
__global__ void mix(unsigned long long* out, int iters, float seed) {
  const int tid = threadIdx.x;
  f32x16 a32[2];
  f32x4 a16[8];
  f32x2 v[8];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) a32[i][r] = seed * r;
  for (int i = 0; i < 8; ++i) { a16[i] = (f32x4){seed, seed, seed, seed}; v[i] = (f32x2){seed * i, seed}; }
  const float x = seed * tid, y = seed + tid;
  const f32x2 c = {seed, 0.5f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (SHAPE == 32) {
#pragma unroll
      for (int k = 0; k < 16; ++k) a32[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a32[k & 1], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) a16[k & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[k & 7], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_elementwise_fma(v[k & 7], c, c);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a16[i][0] + v[i][0] + v[i][1];
  s += a32[0][0] + a32[1][5];
  if (s == 12345.678f) out[1000] = 1;
  if ((tid & 63) == 0 && blockIdx.x == 0) out[tid >> 6] = t1 - t0;
}

template <int SHAPE, int NV>
void run(int waves_per_simd) {
  unsigned long long* d;
  (void)hipMalloc(&d, 2048 * 8);
  (void)hipMemset(d, 0, 2048 * 8);
  const int threads = 256 * waves_per_simd, IT = 500;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((mix<SHAPE, NV>), dim3(256), dim3(threads), 0, 0, d, IT, 1e-9f);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[16];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  const double per_it = (double)mx / IT;                 // cycles for one block of every wave on the SIMD
  const double mfma = 1024.0 * waves_per_simd;           // matrix-pipe cycles of that work
  printf("MFMA %-8s  %2d VALU/block  %d waves/SIMD:  %7.0f cycles per round  -> pipe busy %.2f\n", SHAPE == 32 ? "32x32x2" : "16x16x4", NV,
         waves_per_simd, per_it, mfma / per_it);
  (void)hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<32, 0>(w); run<16, 0>(w);
    run<32, 30>(w); run<16, 30>(w);
    run<32, 60>(w); run<16, 60>(w);
  }
  return 0;
}
