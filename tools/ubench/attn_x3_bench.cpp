// attn_x3_bench.cpp -- times imx::launch_attention_x3 against the fp32-MFMA launch_attention on C3's attention shape
// (64 pairs x 2 sides x 1024 keypoints, d = 128, 4 heads) and compares both with a float64 evaluation of a few query rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -x hip tools/ubench/attn_x3_bench.cpp image-matching_amd/csrc/attention.hip \
//         image-matching_amd/csrc/attention_x3.hip -o tools/ubench/attn_x3_bench
#include "../../image-matching_amd/csrc/imx_kernels.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace imx;
namespace imx { thread_local const char* last_form = nullptr; }   // defined by imx_api.cpp in the library
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, N = 1024, d = 128, heads = 4, hd = d / heads, ld = 3 * d;
  const size_t rows = (size_t)2 * B * N;
  std::vector<float> qkv(rows * ld);
  srand(9);
  for (auto& v : qkv) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *dq, *dout;
  hipMalloc(&dq, qkv.size() * 4); hipMalloc(&dout, rows * d * 4);
  hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice);
  AttnArgs a{dq, dout, B, N, N, d, heads, nullptr, nullptr, N - 5, N - 37, 1};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; ++form) {
    a.mfma_f32 = form;          // the form switch is a field of the arguments since round 3 (handle options)
    a.latency_forms = (!form && getenv("DEPHASE")) ? atoi(getenv("DEPHASE")) : 0;      // experiment hook (IMX_ATTN_DEPHASE_EXP builds)
    auto run = [&]() { return form ? launch_attention(a, 0) : launch_attention_x3(a, 0); };
    for (int i = 0; i < 20; ++i) run();      // clocks settle
    hipEventRecord(e0, 0);
    for (int i = 0; i < 40; ++i) run();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // float64 reference for pair 1, side 0 (cross: keys from side 1), head 2, queries 100..107
    const int b = B > 1 ? 1 : 0, head = 2;
    std::vector<float> out((size_t)8 * d);
    hipMemcpy(out.data(), dout + ((size_t)b * N + 100) * d, out.size() * 4, hipMemcpyDeviceToHost);
    double se = 0, mx = 0;
    for (int qi = 0; qi < 8; ++qi) {
      const float* q = &qkv[((size_t)b * N + 100 + qi) * ld + head * hd];
      std::vector<double> s(N - 37); double m = -1e300;
      for (int k = 0; k < N - 37; ++k) {
        const float* kk = &qkv[((size_t)B * N + (size_t)b * N + k) * ld + d + head * hd];
        double acc = 0; for (int t = 0; t < hd; ++t) acc += (double)q[t] * kk[t];
        s[k] = acc / sqrt((double)hd); if (s[k] > m) m = s[k];
      }
      double l = 0; std::vector<double> o(hd, 0.0);
      for (int k = 0; k < N - 37; ++k) {
        const double pr = exp(s[k] - m); l += pr;
        const float* vv = &qkv[((size_t)B * N + (size_t)b * N + k) * ld + 2 * d + head * hd];
        for (int t = 0; t < hd; ++t) o[t] += pr * vv[t];
      }
      for (int t = 0; t < hd; ++t) { const double e = out[(size_t)qi * d + head * hd + t] - o[t] / l; se += e * e; if (fabs(e) > mx) mx = fabs(e); }
    }
    if (!form && getenv("ATTN_DUMP")) {          // raw output of the first rows, to diff two builds of the kernel
      std::vector<float> full((size_t)4096 * d);
      hipMemcpy(full.data(), dout, full.size() * 4, hipMemcpyDeviceToHost);
      FILE* f = fopen(getenv("ATTN_DUMP"), "wb"); fwrite(full.data(), 4, full.size(), f); fclose(f);
    }
    const double us = ms * 1000 / 40, fl = 4.0 * 2 * B * heads * (double)N * N * hd;
    printf("%-22s %8.1f us   %6.1f TFLOP/s fp32-equivalent   rms err vs float64 %.2e  max %.2e\n", form ? "attention (fp32 MFMA)" : "attention_x3", us,
           fl / us * 1e-6, sqrt(se / (8.0 * hd)), mx);
  }
  return 0;
}
