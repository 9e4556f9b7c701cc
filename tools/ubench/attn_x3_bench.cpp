// attn_x3_bench.cpp -- times imx::launch_attention_x3 (bf16 x 3 planes, six term products; and fp16 x 2 planes, three term products)
// against the fp32-MFMA launch_attention on C3's attention shape (64 pairs x 2 sides x 1024 keypoints, d = 128, 4 heads; `D=256`
// in the environment: C5's head dim 64) and compares all three with a float64 evaluation of a few query rows.  `MAG=x` multiplies
// the inputs (fp16 range handling), `PEAK=x` the queries only (peaked softmax rows).
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
  const int B = argc > 1 ? atoi(argv[1]) : 64, N = 1024, d = getenv("D") ? atoi(getenv("D")) : 128, heads = 4, hd = d / heads, ld = 3 * d;
  const float mag = getenv("MAG") ? (float)atof(getenv("MAG")) : 1.f, peak = getenv("PEAK") ? (float)atof(getenv("PEAK")) : 1.f;
  const size_t rows = (size_t)2 * B * N;
  std::vector<float> qkv(rows * ld);
  srand(9);
  for (size_t i = 0; i < qkv.size(); ++i) qkv[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f * mag * ((int)(i % ld) < d ? peak : 1.f);
  if (getenv("GARBAGE"))                // rows past the valid counts (N - 5 / N - 37) hold huge finite values: no form may look at them
    for (int side = 0; side < 2; ++side)
      for (int b = 0; b < B; ++b)
        for (int r = side ? N - 37 : N - 5; r < N; ++r)
          for (int c = 0; c < ld; ++c) qkv[(((size_t)side * B + b) * N + r) * ld + c] = (c & 1) ? 3.0e30f : -1.0e30f;
  float *dq, *dout;
  hipMalloc(&dq, qkv.size() * 4); hipMalloc(&dout, rows * d * 4);
  hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice);
  AttnArgs a{dq, dout, B, N, N, d, heads, nullptr, nullptr, N - 5, N - 37, 1};
  a.qblocks = getenv("QB") ? atoi(getenv("QB")) : 1;          // two 32-query blocks per wave (attention_h2q2_kernel, round 6)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned* amax; hipMalloc(&amax, (size_t)2 * B * 16);
  {                                   // the maxima of the valid rows (three words), timed alone
    hipMemset(amax, 0, (size_t)2 * B * 16);
    for (int i = 0; i < 5; ++i) launch_qkv_amax(a, amax, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch_qkv_amax(a, amax, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned hm[3]; hipMemcpy(hm, amax + 4 * (B > 1 ? 1 : 0), 12, hipMemcpyDeviceToHost);
    float fm[3]; memcpy(fm, hm, 12);
    printf("qkv_amax               %8.1f us   max |q| %.4g  |k| %.4g  |v| %.4g\n", ms * 1000 / 20, fm[0], fm[1], fm[2]);
  }
  for (int form = 0; form < 3; ++form) {
    a.mfma_f32 = form == 1;     // the form switch is a field of the arguments since round 3 (handle options)
    a.amax = form == 2 ? amax : nullptr;
    a.latency_forms = 0;
    auto run = [&]() { return form == 1 ? launch_attention(a, 0) : launch_attention_x3(a, 0); };
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
    if (form != 1 && getenv("ATTN_DUMP") && (form == 2) == (getenv("ATTN_DUMP_H2") != nullptr)) {          // raw output of the first rows, to diff two builds of the kernel
      std::vector<float> full((size_t)4096 * d);
      hipMemcpy(full.data(), dout, full.size() * 4, hipMemcpyDeviceToHost);
      FILE* f = fopen(getenv("ATTN_DUMP"), "wb"); fwrite(full.data(), 4, full.size(), f); fclose(f);
    }
    const double us = ms * 1000 / 40, fl = 4.0 * 2 * B * heads * (double)N * N * hd;
    printf("%-22s %8.1f us   %6.1f TFLOP/s fp32-equivalent   rms err vs float64 %.2e  max %.2e\n", form == 1 ? "attention (fp32 MFMA)" : form == 2 ? "attention_h2 (f16x2)" : "attention_x3 (bf16x3)", us,
           fl / us * 1e-6, sqrt(se / (8.0 * hd)), mx);
  }
  return 0;
}
