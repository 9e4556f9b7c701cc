// mfma_round.hip -- how do the fp32 MFMA instructions round?  (design input for the accumulation order of the GEMM /
// attention kernels: tests/util.py:assert_fp64_anchored measures the library against a float64 evaluation.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_round.hip -o tools/ubench/mfma_round && tools/ubench/mfma_round
// One wave computes D = A(32xK) * B(Kx32) four ways: v_mfma_f32_32x32x2_f32 chain, v_mfma_f32_16x16x4_f32 chain (on the
// top-left 16x16 block), a scalar fmaf chain in k order and the same chain split over 4 independent accumulators.
// The host compares each with a double-precision product: rms error, max error and MEAN SIGNED error (a non-zero mean
// = biased rounding, i.e. truncation instead of round-to-nearest).  Plus two deterministic probes of the accumulate step.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma32(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, i = l & 31, hi = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + hi], B[(k + hi) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = acc[r];
}
__global__ void k_mfma16(const float* A, const float* B, float* D, int K) {     // D[16][16] = A[0:16] * B[:, 0:16]
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc = {0};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + g], B[(k + g) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 32 + i] = acc[r];
}
__global__ void k_fma(const float* A, const float* B, float* D, float* D4, int K) {
  const int l = threadIdx.x;
  for (int e = l; e < 1024; e += 64) {
    const int i = e >> 5, j = e & 31;
    float s = 0.f, p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) { s = fmaf(A[i * K + k], B[k * 32 + j], s); p[k & 3] = fmaf(A[i * K + k], B[k * 32 + j], p[k & 3]); }
    D[e] = s;
    D4[e] = (p[0] + p[1]) + (p[2] + p[3]);
  }
}
// probes: out[0]: acc = 1, one product 0.75 ulp(1)  -> RNE gives 1 + 2^-23, truncation gives 1
//         out[1]: acc = 1, two products of 0.3 ulp in ONE instruction -> 1 + 2^-23 only if they are summed before the rounding
//         out[2]: acc = 1, 64 MFMAs each adding 0.75 ulp -> RNE: 1 + 64 ulp
__global__ void k_probe(float* out) {
  const int l = threadIdx.x, hi = l >> 5;
  const float ulp = 1.1920929e-07f;
  f32x16 one; for (int r = 0; r < 16; ++r) one[r] = 1.f;
  f32x16 a = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? 0.75f * ulp : 0.f, 1.f, one, 0, 0, 0);
  f32x16 b = __builtin_amdgcn_mfma_f32_32x32x2f32(0.3f * ulp, 1.f, one, 0, 0, 0);
  f32x16 c = one;
  for (int t = 0; t < 64; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? 0.75f * ulp : 0.f, 1.f, c, 0, 0, 0);
  if (l == 0) { out[0] = a[0]; out[1] = b[0]; out[2] = c[0]; }
}

int main() {
  for (int K : {32, 128, 256, 1024}) {
    std::vector<float> A(32 * K), B(K * 32);
    srand(7 + K);
    for (auto& v : A) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4 * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dD, 0, 4 * 1024 * 4);
    hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dD + 1024, K);
    hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, dA, dB, dD + 2048, dD + 3072, K);
    std::vector<float> D(4096);
    hipMemcpy(D.data(), dD, 4096 * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"mfma 32x32x2", "mfma 16x16x4", "fmaf chain", "fmaf 4 chains"};
    for (int v = 0; v < 4; ++v) {
      double se = 0, me = 0, mx = 0; int n = 0;
      for (int i = 0; i < (v == 1 ? 16 : 32); ++i)
        for (int j = 0; j < (v == 1 ? 16 : 32); ++j) {
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 32 + j];
          const double e = (double)D[v * 1024 + i * 32 + j] - ref;
          se += e * e; me += e; mx = fmax(mx, fabs(e)); ++n;
        }
      printf("K=%4d %-14s rms %.3e  max %.3e  mean signed %+.3e\n", K, names[v], sqrt(se / n), mx, me / n);
    }
    hipFree(dA); hipFree(dB); hipFree(dD);
  }
  float* dp; hipMalloc(&dp, 16);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dp);
  float p[3]; hipMemcpy(p, dp, 12, hipMemcpyDeviceToHost);
  printf("probe: 1 + 0.75ulp -> 1 + %.2f ulp;  1 + (0.3 + 0.3) ulp in one MFMA -> 1 + %.2f ulp;  64 x (+0.75 ulp) -> 1 + %.2f ulp\n",
         (p[0] - 1.f) / 1.1920929e-07f, (p[1] - 1.f) / 1.1920929e-07f, (p[2] - 1.f) / 1.1920929e-07f);
  return 0;
}
