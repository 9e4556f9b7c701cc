// gnn_tail_bench.cpp -- times imx::launch_gnn_tail_x3 (the fused GNN layer tail) against the three gemm_x3 launches it replaces, at the
// C3 step's row count (64 pairs: 131072 rows, d = 128), and checks both against a float64 evaluation of a few rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -x hip tools/ubench/gnn_tail_bench.cpp \
//         image-matching_amd/csrc/gnn_tail_x3.hip image-matching_amd/csrc/gnn_tail_h2.hip image-matching_amd/csrc/gemm_x3.hip -o tools/ubench/gnn_tail_bench
#include "../../image-matching_amd/csrc/imx_kernels.h"
#include "../../image-matching_amd/csrc/gnn_tail_pack.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace imx;
#ifdef GT_TRACE
namespace imx { void gnn_tail_trace_dump(); }
#endif
namespace imx { thread_local const char* last_form = nullptr; }
static std::vector<uint16_t> x3_planes(const std::vector<float>& w, int K, int N) {      // gemm_x3's B-fragment order (imx_api.cpp: split_bf16x3)
  const int nst = K / 16;
  std::vector<uint16_t> pl((size_t)3 * N * K);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      uint16_t t[3];
      gt_split(w[(size_t)k * N + n], t);
      const int nb = n >> 5, st = k >> 4, lane = (n & 31) + 32 * ((k >> 3) & 1), j = k & 7;
      for (int q = 0; q < 3; ++q) pl[((((size_t)nb * nst + st) * 3 + q) * 64 + lane) * 8 + j] = t[q];
    }
  return pl;
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 131072, d = 128, n3 = argc > 2 ? atoi(argv[2]) : 384;
  std::vector<float> x((size_t)M * d), att((size_t)M * d), w1((size_t)2 * d * 2 * d), w2((size_t)2 * d * d), w3((size_t)d * n3), b1(2 * d), b2(d), b3(n3);
  srand(11);
  auto rnd = [](float s) { return (rand() / (float)RAND_MAX - 0.5f) * s; };
  for (auto& v : x) v = rnd(2.f);
  for (auto& v : att) v = rnd(2.f);
  for (auto& v : w1) v = rnd(0.25f);
  for (auto& v : w2) v = rnd(0.25f);
  for (auto& v : w3) v = rnd(0.3f);
  for (auto& v : b1) v = rnd(1.f);
  for (auto& v : b2) v = rnd(1.f);
  for (auto& v : b3) v = rnd(1.f);
  const std::vector<uint16_t> stream = gnn_tail_pack(w1.data(), 2 * d, w2.data(), d, w3.data(), n3, d, n3);
  const auto p1 = x3_planes(w1, 2 * d, 2 * d), p2 = x3_planes(w2, 2 * d, d), p3 = x3_planes(w3, d, n3);
  float *dx, *dx0, *datt, *dhid, *dout, *db1, *db2, *db3, *dw1, *dw2, *dw3; void *dstream, *dp1, *dp2, *dp3;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dx0, x.size() * 4); hipMalloc(&datt, x.size() * 4); hipMalloc(&dhid, (size_t)M * 2 * d * 4); hipMalloc(&dout, (size_t)M * n3 * 4);
  hipMalloc(&db1, 2 * d * 4); hipMalloc(&db2, d * 4); hipMalloc(&db3, n3 * 4); hipMalloc(&dstream, stream.size() * 2);
  hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4); hipMalloc(&dw3, w3.size() * 4);
  hipMalloc(&dp1, p1.size() * 2); hipMalloc(&dp2, p2.size() * 2); hipMalloc(&dp3, p3.size() * 2);
  hipMemcpy(dx0, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(datt, att.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db1, b1.data(), 2 * d * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), d * 4, hipMemcpyHostToDevice); hipMemcpy(db3, b3.data(), n3 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dstream, stream.data(), stream.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw3, w3.data(), w3.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dp2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dp3, p3.data(), p3.size() * 2, hipMemcpyHostToDevice);
  // float64 reference of rows r0 .. r0 + 63 (two waves' worth, second half of a workgroup) and the last 32 rows
  const int nref = 96;
  std::vector<int> rows;
  for (int i = 0; i < 64; ++i) rows.push_back(1000 + 160 + i);
  for (int i = 0; i < 32; ++i) rows.push_back(M - 32 + i);
  std::vector<double> rx((size_t)nref * d), rout((size_t)nref * n3);
  for (int ri = 0; ri < nref; ++ri) {
    const int r = rows[ri];
    std::vector<double> hid(2 * d);
    for (int n = 0; n < 2 * d; ++n) {
      double a = b1[n];
      for (int k = 0; k < d; ++k) a += (double)x[(size_t)r * d + k] * w1[(size_t)k * 2 * d + n];
      for (int k = 0; k < d; ++k) a += (double)att[(size_t)r * d + k] * w1[(size_t)(d + k) * 2 * d + n];
      hid[n] = a > 0 ? a : 0;
    }
    for (int n = 0; n < d; ++n) {
      double a = b2[n] + x[(size_t)r * d + n];
      for (int k = 0; k < 2 * d; ++k) a += hid[k] * w2[(size_t)k * d + n];
      rx[(size_t)ri * d + n] = a;
    }
    for (int n = 0; n < n3; ++n) {
      double a = b3[n];
      for (int k = 0; k < d; ++k) a += rx[(size_t)ri * d + k] * w3[(size_t)k * n3 + n];
      rout[(size_t)ri * n3 + n] = a;
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned* damax = nullptr;
  const int PB = 64, PN = M / (2 * PB);                     // the amax form: M rows as 64 pairs x 2 sides (C3: 1024 rows each), 5 / 37 invalid rows
  hipMalloc(&damax, (size_t)M * 16 + 2 * PB * 16);          // (sized for the per-wave store experiment too)
  // form 3: gnn_tail_h2 (three fp16 plane products): its stream, constants and the (side, pair) maxima of x and of "v" (here: of att itself)
  GnnTailH2Consts hc{};
  const std::vector<uint16_t> stream_h2 = gnn_tail_pack_h2(w1.data(), 2 * d, w2.data(), d, w3.data(), n3, d, n3, &hc);
  void* dstream_h2; unsigned *dax_in, *dav, *dax_out;
  hipMalloc(&dstream_h2, stream_h2.size() * 2); hipMemcpy(dstream_h2, stream_h2.data(), stream_h2.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&dax_in, 2 * PB * 4); hipMalloc(&dav, 2 * PB * 16); hipMalloc(&dax_out, 2 * PB * 4);
  {
    std::vector<unsigned> ax(2 * PB), av(2 * PB * 4, 0);
    for (int sp = 0; sp < 2 * PB; ++sp) {
      float mx = 0, ma = 0;
      for (size_t i = (size_t)sp * PN * d; i < (size_t)(sp + 1) * PN * d; ++i) { mx = fmaxf(mx, fabsf(x[i])); ma = fmaxf(ma, fabsf(att[i])); }
      memcpy(&ax[sp], &mx, 4); memcpy(&av[sp * 4 + 2], &ma, 4);
    }
    hipMemcpy(dax_in, ax.data(), ax.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dav, av.data(), av.size() * 4, hipMemcpyHostToDevice);
  }
  float bm1 = 0, bm2 = 0;
  for (float v : b1) bm1 = fmaxf(bm1, fabsf(v));
  for (float v : b2) bm2 = fmaxf(bm2, fabsf(v));
  for (int form = 0; form < 4; ++form) {
    GnnTailArgs t{dx, datt, dstream, db1, db2, db3, dout, M, d, n3};
    if (form == 3) {
      if (M % (2 * PB * 32)) break;
      hipMemset(damax, 0, 2 * PB * 16); hipMemset(dax_out, 0, 2 * PB * 4);
      t.amax = n3 == 384 ? damax : nullptr; t.B = PB; t.N0p = PN; t.N1p = PN; t.N0 = PN; t.N1 = PN;
      t.stream_h2 = dstream_h2; t.w1_inv = hc.w1_inv; t.w2_inv = hc.w2_inv; t.w3_inv = hc.w3_inv; t.l1_1 = hc.l1_1; t.l1_2 = hc.l1_2; t.bmax_1 = bm1; t.bmax_2 = bm2;
      t.amax_x_in = dax_in; t.amax_v = dav; t.amax_x_out = dax_out; t.cross = 0;
    }
    if (form == 2) {
      if (n3 != 384 || M % (2 * PB * 32)) continue;
      hipMemset(damax, 0, 2 * PB * 16);
      t.amax = damax; t.B = PB; t.N0p = PN; t.N1p = PN; t.N0 = PN - 5; t.N1 = PN - 37;
    }
    GemmArgs g1{dx, d, d, datt, d, d, dw1, db1, nullptr, 0, dhid, 2 * d, M, 2 * d, 2 * d, 1};
    GemmArgs g2{dhid, 2 * d, 2 * d, nullptr, 0, 0, dw2, db2, dx, d, dx, d, M, d, d, 0};
    GemmArgs g3{dx, d, d, nullptr, 0, 0, dw3, db3, nullptr, 0, dout, n3, M, n3, n3, 0};
    auto run = [&]() {
      if (form == 3) return launch_gnn_tail_h2(t, 0);
      if (form != 1) return launch_gnn_tail_x3(t, 0);
      launch_gemm_x3(g1, dp1, 0); launch_gemm_x3(g2, dp2, 0); return launch_gemm_x3(g3, dp3, 0);
    };
    hipMemcpy(dx, dx0, x.size() * 4, hipMemcpyDeviceToDevice);
    hipError_t err = run();
    hipDeviceSynchronize();
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
    std::vector<float> gx((size_t)M * d), go((size_t)M * n3);
    hipMemcpy(gx.data(), dx, gx.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(go.data(), dout, go.size() * 4, hipMemcpyDeviceToHost);
    double sx = 0, mx = 0, so = 0, mo = 0;
    for (int ri = 0; ri < nref; ++ri) {
      for (int n = 0; n < d; ++n) { const double e = gx[(size_t)rows[ri] * d + n] - rx[(size_t)ri * d + n]; sx += e * e; if (fabs(e) > mx) mx = fabs(e); }
      for (int n = 0; n < n3; ++n) { const double e = go[(size_t)rows[ri] * n3 + n] - rout[(size_t)ri * n3 + n]; so += e * e; if (fabs(e) > mo) mo = fabs(e); }
    }
    std::vector<unsigned> am((size_t)2 * PB * 4), amx((size_t)2 * PB);
    if (form >= 2) hipMemcpy(am.data(), damax, am.size() * 4, hipMemcpyDeviceToHost);      // of the checked run (x is updated in place: later runs see other values)
    if (form == 3) hipMemcpy(amx.data(), dax_out, amx.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; ++i) { run(); }
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) run();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef GT_TRACE
    if (form == 0) gnn_tail_trace_dump();
#endif
    if (form == 3) {             // the maxima of x' and of q | k | v against the host's (every row valid here)
      int bad = 0;
      for (int sp = 0; sp < 2 * PB; ++sp) {
        float ref = 0;
        for (size_t i = (size_t)sp * PN * d; i < (size_t)(sp + 1) * PN * d; ++i) ref = fmaxf(ref, fabsf(gx[i]));
        float got; memcpy(&got, &amx[sp], 4);
        if (got != ref) { if (bad < 3) printf("  amax_x_out[%d] = %g, host %g\n", sp, got, ref); ++bad; }
        for (int q = 0; q < 3 && n3 == 384; ++q) {
          float r2 = 0;
          for (int r = 0; r < PN; ++r)
            for (int c = 0; c < d; ++c) r2 = fmaxf(r2, fabsf(go[((size_t)sp * PN + r) * n3 + q * d + c]));
          memcpy(&got, &am[(size_t)sp * 4 + q], 4);
          if (got != r2) { if (bad < 3) printf("  amax[%d][%d] = %g, host %g\n", sp, q, got, r2); ++bad; }
        }
      }
      printf("  h2 epilogues: %d maxima differ from the host's\n", bad);
    }
    if (form == 2) {             // the maxima against the host's, over the valid rows of this run's own output
      int bad = 0;
      for (int sp = 0; sp < 2 * PB; ++sp)
        for (int q = 0; q < 3; ++q) {
          float ref = 0;
          const int n = sp < PB ? PN - 5 : PN - 37;
          for (int r = 0; r < n; ++r)
            for (int c = 0; c < d; ++c) ref = fmaxf(ref, fabsf(go[((size_t)sp * PN + r) * n3 + q * d + c]));
          float got; memcpy(&got, &am[(size_t)sp * 4 + q], 4);
          if (got != ref) { if (bad < 4) printf("  amax[%d][%d] = %g, host %g\n", sp, q, got, ref); ++bad; }
        }
      printf("  amax epilogue: %d of %d maxima differ from the host's\n", bad, 2 * PB * 3);
    }
    printf("%-34s %7.1f us per layer tail   x' rms err vs float64 %.2e max %.2e | out rms %.2e max %.2e\n", form == 1 ? "3 x gemm_x3 (mlp1, mlp2, next)" : form == 2 ? "gnn_tail_x3 + q|k|v maxima" : form == 3 ? "gnn_tail_h2 (fp16 planes)" : "gnn_tail_x3 (one launch)",
           ms * 1000 / 20, sqrt(sx / (nref * d)), mx, sqrt(so / (nref * (double)n3)), mo);
  }
  return 0;
}
