// gemm_x3_bench.cpp -- times imx::launch_gemm_x3 (and the fp32-MFMA gemm_ws it replaces) on the GNN's three products at 64 pairs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -x hip tools/ubench/gemm_x3_bench.cpp image-matching_amd/csrc/gemm_x3.hip \
//         -o tools/ubench/gemm_x3_bench      (-DIMX_SPLIT_DOT2=0 for the shift/subtract split: A/B of csrc/split3.h)
#include "../../image-matching_amd/csrc/imx_kernels.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace imx;
namespace imx { thread_local const char* last_form = nullptr; }   // defined by imx_api.cpp in the library
static uint16_t bf16_rne(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf16_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }
int main() {
  const int M = 131072;
  struct Shape { const char* name; int K0, K1, N; bool res, relu; } shapes[] = {
      {"qkv  K=128 N=384", 128, 0, 384, false, false}, {"mlp1 K=256 N=256", 128, 128, 256, false, true}, {"mlp2 K=256 N=128", 256, 0, 128, true, false}};
  for (auto& sh : shapes) {
    const int K = sh.K0 + sh.K1, N = sh.N, nst = K / 16;
    std::vector<float> a0((size_t)M * (sh.K1 ? sh.K0 : K)), a1((size_t)M * (sh.K1 ? sh.K1 : 1)), w((size_t)K * N), b(N);
    srand(5);
    for (auto& v : a0) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : a1) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : b) v = rand() / (float)RAND_MAX;
    std::vector<uint16_t> pl((size_t)3 * N * K);
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) {
        const float x = w[(size_t)k * N + n];
        uint16_t t[3];
        t[0] = bf16_rne(x); const float r1 = x - bf16_f(t[0]); t[1] = bf16_rne(r1); t[2] = bf16_rne(r1 - bf16_f(t[1]));
        const int nb = n >> 5, st = k >> 4, lane = (n & 31) + 32 * ((k >> 3) & 1), j = k & 7;
        for (int q = 0; q < 3; ++q) pl[((((size_t)nb * nst + st) * 3 + q) * 64 + lane) * 8 + j] = t[q];
      }
    float *da0, *da1, *dw, *db, *dout, *dres; void* dwx;
    hipMalloc(&da0, a0.size() * 4); hipMalloc(&da1, a1.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&db, N * 4);
    hipMalloc(&dout, (size_t)M * N * 4); hipMalloc(&dres, (size_t)M * N * 4); hipMalloc(&dwx, pl.size() * 2);
    hipMemcpy(da0, a0.data(), a0.size() * 4, hipMemcpyHostToDevice); hipMemcpy(da1, a1.data(), a1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwx, pl.data(), pl.size() * 2, hipMemcpyHostToDevice); hipMemset(dres, 0, (size_t)M * N * 4);
    GemmArgs g{da0, sh.K1 ? sh.K0 : K, sh.K0, sh.K1 ? da1 : nullptr, sh.K1, sh.K1, dw, db, sh.res ? dres : nullptr, N, dout, N, M, N, N, sh.relu ? 1 : 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int form = 0; form < 1; ++form) {          // (the fp32-MFMA gemm_ws this was compared with was removed in round 3)
      auto run = [&]() { return launch_gemm_x3(g, dwx, 0); };
      for (int i = 0; i < 10; ++i) run();
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) run();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<float> out((size_t)64 * N);
      hipMemcpy(out.data(), dout + (size_t)1000 * N, out.size() * 4, hipMemcpyDeviceToHost);
      double se = 0, mx = 0;
      for (int r = 0; r < 64; ++r)
        for (int n = 0; n < N; ++n) {
          double ref = b[n];
          for (int k = 0; k < K; ++k) ref += (double)(k < sh.K0 ? a0[(size_t)(1000 + r) * (sh.K1 ? sh.K0 : K) + k] : a1[(size_t)(1000 + r) * sh.K1 + k - sh.K0]) * w[(size_t)k * N + n];
          if (sh.relu && ref < 0) ref = 0;
          const double e = out[(size_t)r * N + n] - ref; se += e * e; if (fabs(e) > mx) mx = fabs(e);
        }
      const double us = ms * 1000 / 20, gf = 2.0 * M * K * N * 1e-9;
      printf("%-18s %-8s %7.1f us  %6.1f TFLOP/s fp32-equivalent  %5.2f TB/s (A once + out)   rms err %.2e max %.2e\n", sh.name, form ? "gemm_ws" : "gemm_x3",
             us, gf / us * 1e-3 * 1e3, ((double)M * K * 4 + (double)M * N * 4 * (sh.res ? 2 : 1)) / us * 1e-6, sqrt(se / (64.0 * N)), mx);
    }
    hipFree(da0); hipFree(da1); hipFree(dw); hipFree(db); hipFree(dout); hipFree(dres); hipFree(dwx);
  }
  return 0;
}
