// Microbenchmark: how do VALU / LDS instructions co-issue with a saturated fp32 MFMA stream on one gfx950 SIMD?
// (design input for the Winograd kernels; results quoted in DESIGN.md).  Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MF(i) "v_mfma_f32_16x16x4_f32 %[a" #i "], %[x], %[y], %[a" #i "]\n"
#define VI(i) "v_pk_fma_f32 %[v" #i "], %[v" #i "], %[c], %[c]\n"          /* 8 independent chains */
#define VD    "v_pk_fma_f32 %[v0], %[v0], %[c], %[c]\n"                      /* one dependent chain */
#define LD(i) "ds_read_b128 %[l" #i "], %[addr]\n"

// SAME-WAVE patterns: 8 MFMAs (independent accumulators) with K VALU after each
template <int K, bool DEP>
__device__ __forceinline__ void body_same(f32x4 (&a)[8], f32x2 (&v)[8], float x, float y, f32x2 c) {
#define OPS [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), \
            [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7])
#define INS [x] "v"(x), [y] "v"(y), [c] "v"(c)
  if constexpr (K == 0) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) : OPS : INS);
  else if constexpr (DEP) {
    if constexpr (K == 1) asm volatile(MF(0) VD MF(1) VD MF(2) VD MF(3) VD MF(4) VD MF(5) VD MF(6) VD MF(7) VD : OPS : INS);
    if constexpr (K == 2) asm volatile(MF(0) VD VD MF(1) VD VD MF(2) VD VD MF(3) VD VD MF(4) VD VD MF(5) VD VD MF(6) VD VD MF(7) VD VD : OPS : INS);
    if constexpr (K == 4) asm volatile(MF(0) VD VD VD VD MF(1) VD VD VD VD MF(2) VD VD VD VD MF(3) VD VD VD VD MF(4) VD VD VD VD MF(5) VD VD VD VD MF(6) VD VD VD VD MF(7) VD VD VD VD : OPS : INS);
  } else {
    if constexpr (K == 1) asm volatile(MF(0) VI(0) MF(1) VI(1) MF(2) VI(2) MF(3) VI(3) MF(4) VI(4) MF(5) VI(5) MF(6) VI(6) MF(7) VI(7) : OPS : INS);
    if constexpr (K == 2) asm volatile(MF(0) VI(0) VI(1) MF(1) VI(2) VI(3) MF(2) VI(4) VI(5) MF(3) VI(6) VI(7) MF(4) VI(0) VI(1) MF(5) VI(2) VI(3) MF(6) VI(4) VI(5) MF(7) VI(6) VI(7) : OPS : INS);
    if constexpr (K == 4) asm volatile(MF(0) VI(0) VI(1) VI(2) VI(3) MF(1) VI(4) VI(5) VI(6) VI(7) MF(2) VI(0) VI(1) VI(2) VI(3) MF(3) VI(4) VI(5) VI(6) VI(7) MF(4) VI(0) VI(1) VI(2) VI(3) MF(5) VI(4) VI(5) VI(6) VI(7) MF(6) VI(0) VI(1) VI(2) VI(3) MF(7) VI(4) VI(5) VI(6) VI(7) : OPS : INS);
    if constexpr (K == 6) asm volatile(MF(0) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) MF(1) VI(6) VI(7) VI(0) VI(1) VI(2) VI(3) MF(2) VI(4) VI(5) VI(6) VI(7) VI(0) VI(1) MF(3) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(4) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) MF(5) VI(6) VI(7) VI(0) VI(1) VI(2) VI(3) MF(6) VI(4) VI(5) VI(6) VI(7) VI(0) VI(1) MF(7) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) : OPS : INS);
    if constexpr (K == 8) asm volatile(MF(0) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(1) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(2) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(3) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(4) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(5) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(6) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) MF(7) VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) : OPS : INS);
  }
#undef OPS
#undef INS
}

// MODE 0: every wave runs the same-wave pattern (K VALU per MFMA).  waves_per_simd = blockDim/256.
// MODE 1: waves 0-3 run MFMA only, waves 4-7 run VALU only (independent or dependent), both for a fixed iteration count.
// MODE 2: as MODE 1 but the second group runs ds_read_b128 streams.
template <int MODE, int K, bool DEP>
__global__ void bench(unsigned long long* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = seed * i;
  __syncthreads();
  f32x4 a[8];
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) { a[i] = (f32x4){seed, seed, seed, seed}; v[i] = (f32x2){seed * i, seed}; }
  const float x = seed * tid, y = seed + tid;
  const f32x2 c = {seed, 0.5f};
  unsigned long long t0 = 0, t1 = 0;
  if (MODE == 0 || wave < 4) {
    __syncthreads();
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 0) body_same<K, DEP>(a, v, x, y, c);
      else body_same<0, false>(a, v, x, y, c);
    }
    t1 = __builtin_readcyclecounter();
  } else {
    __syncthreads();
    t0 = __builtin_readcyclecounter();
    if constexpr (MODE == 1) {
      for (int it = 0; it < iters; ++it) {
        if constexpr (DEP) asm volatile(VD VD VD VD VD VD VD VD : [v0] "+v"(v[0]) : [c] "v"(c));
        else asm volatile(VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7)
                          : [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7])
                          : [c] "v"(c));
      }
    } else {
      const unsigned addr = (tid & 63) * 16;
      f32x4 l[4];
      for (int it = 0; it < iters; ++it) {
        asm volatile(LD(0) LD(1) LD(2) LD(3) "s_waitcnt lgkmcnt(0)\n"
                     : [l0] "=&v"(l[0]), [l1] "=&v"(l[1]), [l2] "=&v"(l[2]), [l3] "=&v"(l[3]) : [addr] "v"(addr) : "memory");
        v[0] += (f32x2){l[0][0] + l[1][1], l[2][2] + l[3][3]};
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][3] + v[i][0] + v[i][1];
  if (s == 12345.678f) out[1000] = 1;                 // keep everything live
  if ((tid & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int MODE, int K, bool DEP>
void run(const char* what, int threads, int iters, int per_iter_mfma, int per_iter_other) {
  unsigned long long* d;
  (void)hipMalloc(&d, 2048 * 8);
  (void)hipMemset(d, 0, 2048 * 8);
  hipLaunchKernelGGL((bench<MODE, K, DEP>), dim3(256), dim3(threads), 0, 0, d, iters, 1e-9f);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((bench<MODE, K, DEP>), dim3(256), dim3(threads), 0, 0, d, iters, 1e-9f);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s", what);
  const int nw = threads / 64;
  for (int w = 0; w < nw; w += 4) {
    const bool mf = (MODE == 0) || w < 4;
    const double cyc = (double)h[w] / iters;
    if (mf) printf("  wave%d: %7.1f cyc/iter = %5.1f per MFMA", w, cyc, cyc / per_iter_mfma);
    else printf("  wave%d: %7.1f cyc/iter = %5.1f per instr", w, cyc, cyc / per_iter_other);
  }
  printf("\n");
  (void)hipFree(d);
}

int main() {
  const int IT = 2000;
  run<0, 0, false>("1 wave/SIMD, MFMA only", 256, IT, 8, 0);
  run<0, 1, false>("1 wave/SIMD, MFMA + 1 independent pk_fma each", 256, IT, 8, 0);
  run<0, 2, false>("1 wave/SIMD, MFMA + 2 independent pk_fma each", 256, IT, 8, 0);
  run<0, 4, false>("1 wave/SIMD, MFMA + 4 independent pk_fma each", 256, IT, 8, 0);
  run<0, 6, false>("1 wave/SIMD, MFMA + 6 independent pk_fma each", 256, IT, 8, 0);
  run<0, 8, false>("1 wave/SIMD, MFMA + 8 independent pk_fma each", 256, IT, 8, 0);
  run<0, 1, true>("1 wave/SIMD, MFMA + 1 dependent pk_fma each", 256, IT, 8, 0);
  run<0, 2, true>("1 wave/SIMD, MFMA + 2 dependent pk_fma each", 256, IT, 8, 0);
  run<0, 4, true>("1 wave/SIMD, MFMA + 4 dependent pk_fma each", 256, IT, 8, 0);
  run<0, 0, false>("2 waves/SIMD, both MFMA only", 512, IT, 8, 0);
  run<0, 2, false>("2 waves/SIMD, both MFMA + 2 independent pk_fma each", 512, IT, 8, 0);
  run<0, 4, false>("2 waves/SIMD, both MFMA + 4 independent pk_fma each", 512, IT, 8, 0);
  run<1, 0, false>("2 waves/SIMD: MFMA-only wave | independent pk_fma wave", 512, IT, 8, 8);
  run<1, 0, true>("2 waves/SIMD: MFMA-only wave | dependent pk_fma wave", 512, IT, 8, 8);
  run<2, 0, false>("2 waves/SIMD: MFMA-only wave | ds_read_b128 x4 + wait wave", 512, IT, 8, 4);
  run<1, 0, false>("VALU reference: (MFMA waves present) see above", 512, 1, 8, 8);
  return 0;
}
