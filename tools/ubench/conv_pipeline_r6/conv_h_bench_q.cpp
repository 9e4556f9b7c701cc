// conv_h_bench.cpp -- conv3x3_wino24h (Winograd F(2x4,3x3) with both transformed operands as two fp16 planes, three plane products on
// v_mfma_f32_16x16x32_f16) against conv3x3_wino24 (the same on the fp32 MFMA) on SuperPoint's layer shapes at C3's batch: times both,
// compares them with each other and with a float64 direct convolution of sample pixels, checks the per-image output maxima.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -x hip tools/ubench/conv_h_bench.cpp \
//         image-matching_amd/csrc/conv3x3_wino24.hip image-matching_amd/csrc/conv3x3_wino24h.hip image-matching_amd/csrc/conv1ab_wino24.hip \
//         image-matching_amd/csrc/conv1ab_wino24h.hip -o tools/ubench/conv_h_bench
//   usage: conv_h_bench [B H W Cin Cout pool blocked]     (default: every layer of the stack, B = 128)
#include "../../image-matching_amd/csrc/imx_kernels.h"
#include "../../image-matching_amd/csrc/wino24_pack.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace imx;
namespace imx { thread_local const char* last_form = nullptr; }
#ifdef H_TRACE
namespace imx { void conv_h_trace_read(long long* out); }
#endif
#ifdef P_TRACE
namespace imx { void conv_p_trace_read(long long* out); }
#endif

static int run(int B, int H, int W, int Cin, int Cout, int pool, int blocked, float mag) {
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  std::vector<float> x((size_t)B * H * W * Cin), w((size_t)9 * Cin * Cout), bias(Cout);
  srand(5);
  auto rnd = []() { return rand() / (float)RAND_MAX; };
  for (auto& v : x) { const float r = rnd(); v = r < 0.4f ? 0.f : (r - 0.4f) * mag; }        // post-ReLU-like: 40 % zeros
  for (auto& v : w) v = (rnd() - 0.5f) * 0.12f;
  for (auto& v : bias) v = (rnd() - 0.5f) * 0.2f;
  std::vector<float> xin = x;
  if (blocked) {           // (B, Cin/8, H, W, 8)
    for (int b = 0; b < B; ++b)
      for (int p = 0; p < H * W; ++p)
        for (int c = 0; c < Cin; ++c) xin[(((size_t)b * (Cin / 8) + c / 8) * H * W + p) * 8 + c % 8] = x[((size_t)b * H * W + p) * Cin + c];
  }
  std::vector<unsigned> amax(B);
  for (int b = 0; b < B; ++b) {
    float m = 0;
    for (size_t i = 0; i < (size_t)H * W * Cin; ++i) m = fmaxf(m, fabsf(x[(size_t)b * H * W * Cin + i]));
    memcpy(&amax[b], &m, 4);
  }
  const std::vector<float> u32 = wino24_transform(w, Cin, Cout);
  float su_inv = 0;
  const std::vector<uint16_t> uh = wino24h_pack(w, Cin, Cout, &su_inv);
  float *dx, *dw32, *db, *dout32, *douth, *doutu; void* duh; unsigned *damax, *damax_out;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dw32, u32.size() * 4); hipMalloc(&db, Cout * 4); hipMalloc(&duh, uh.size() * 2);
  hipMalloc(&dout32, (size_t)B * Ho * Wo * Cout * 4); hipMalloc(&douth, (size_t)B * Ho * Wo * Cout * 4); hipMalloc(&doutu, (size_t)B * Ho * Wo * Cout * 4); hipMalloc(&damax, 256 * 4); hipMalloc(&damax_out, 256 * 4);
  hipMemcpy(dx, xin.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw32, u32.data(), u32.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), Cout * 4, hipMemcpyHostToDevice); hipMemcpy(duh, uh.data(), uh.size() * 2, hipMemcpyHostToDevice);
  hipMemset(damax, 0, 256 * 4); hipMemcpy(damax, amax.data(), (B < 256 ? B : 256) * 4, hipMemcpyHostToDevice); hipMemset(damax_out, 0, 256 * 4);
  hipMemset(dout32, 0xff, (size_t)B * Ho * Wo * Cout * 4); hipMemset(douth, 0xff, (size_t)B * Ho * Wo * Cout * 4); hipMemset(doutu, 0xff, (size_t)B * Ho * Wo * Cout * 4);
  ConvArgs a{};
  a.in = dx; a.wu24 = dw32; a.bias = db; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = 1; a.pool = pool; a.in_blocked = blocked;
  a.wuh = duh; a.u_scale_inv = su_inv; a.amax_in = damax; a.amax_out = getenv("NOAMAX") ? nullptr : damax_out;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[3] = {0, 0, 0};
  std::vector<unsigned> amu(B);
  for (int form = 0; form < 3; ++form) {
    a.out = form == 2 ? doutu : form ? douth : dout32;
    if (form == 2) { std::vector<unsigned> am0(256); hipMemcpy(am0.data(), damax_out, 1024, hipMemcpyDeviceToHost); amu.assign(am0.begin(), am0.begin() + (B < 256 ? B : 256)); hipMemset(damax_out, 0, 1024); }
    auto go = [&]() { return form == 2 ? launch_conv3x3_wino24p(a, 0) : form ? launch_conv3x3_wino24h(a, 0) : launch_conv3x3_wino24(a, 0); };
    hipError_t err = go();
    hipDeviceSynchronize();
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed (form %d): %s\n", form, hipGetErrorString(err)); return 1; }
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[form], e0, e1);
    ms[form] /= 10;
  }
#ifdef P_TRACE
  {
    long long tr[16 * 16];
    conv_p_trace_read(tr);
    double s[16] = {0}; int n = 0;
    for (int g = 0; g < 16; ++g) { if (!tr[g * 16 + 2]) continue; ++n; for (int i = 0; i < 16; ++i) s[i] += (double)tr[g * 16 + i]; }
    const double items = (double)(((W + 15) / 16) * ((H + 7) / 8) * B / 2) * (Cout / 64) / 256.0, chunks = items * (Cin / 32);
    if (getenv("TRACE_WAVES"))
      for (int g = 0; g < 8; ++g)
{
        printf("    wave %d: slot A: rows 2-3 + raw store %.0f | MFMAs rows 0-1 %.0f | barrier %.0f || slot B: loads %.0f | rows 0-1 %.0f | MFMAs rows 2-3 %.0f | barrier %.0f | to slot A %.0f\n", g, tr[g * 16] / chunks,
               tr[g * 16 + 2] / chunks, tr[g * 16 + 1] / chunks, tr[g * 16 + 8] / chunks, tr[g * 16 + 9] / chunks, tr[g * 16 + 3] / chunks, tr[g * 16 + 4] / chunks, tr[g * 16 + 5] / chunks);
        printf("            per item: exchange %.0f | output transform %.0f | stores %.0f | maxima %.0f | to the next chunk step %.0f\n", tr[g * 16 + 6] / items, tr[g * 16 + 7] / items, tr[g * 16 + 13] / items, tr[g * 16 + 14] / items, tr[g * 16 + 5] / items);
      }
    printf("  P trace (wave 0 of %d workgroups; cycles per chunk of a tile PAIR): slot A T %.0f M %.0f barrier %.0f | slot B loads %.0f T %.0f M %.0f barrier %.0f | per item: exchange %.0f out %.0f stores %.0f maxima %.0f | total per chunk %.0f\n",
           n, s[0] / n / chunks, s[2] / n / chunks, s[1] / n / chunks, s[8] / n / chunks, s[9] / n / chunks, s[3] / n / chunks, s[4] / n / chunks, s[6] / n / items, s[7] / n / items, s[13] / n / items, s[14] / n / items,
           (s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7] + s[13] + s[14] + s[8] + s[9] + s[10] + s[11] + s[12]) / n / chunks);
  }
#endif
#ifdef H_TRACE
  {
    long long tr[16 * 8];
    conv_h_trace_read(tr);
    double s[8] = {0}; int n = 0;
    for (int g = 0; g < 16; ++g) { if (!tr[g * 8 + 2]) continue; ++n; for (int i = 0; i < 8; ++i) s[i] += (double)tr[g * 8 + i]; }
    const double items = (double)((W + 15) / 16) * ((H + 7) / 8) * B * (Cout / 64) / 512.0, chunks = items * (Cin / 32);
    printf("  trace (wave 0 of %d workgroups; cycles per chunk): transform %.0f | barrier %.0f | MFMA phase %.0f | raw store + loads %.0f | barrier %.0f | epilogue per item %.0f | total per chunk %.0f\n",
           n, s[0] / n / chunks, s[1] / n / chunks, s[2] / n / chunks, s[3] / n / chunks, s[4] / n / chunks, s[5] / n / items,
           (s[0] + s[1] + s[2] + s[3] + s[4] + s[5]) / n / chunks);
  }
#endif
  std::vector<float> o32((size_t)B * Ho * Wo * Cout), oh(o32.size());
  hipMemcpy(o32.data(), dout32, o32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), douth, oh.size() * 4, hipMemcpyDeviceToHost);
  double dmax = 0, omax = 0; size_t nan = 0;
  for (size_t i = 0; i < oh.size(); ++i) {
    if (!(oh[i] == oh[i])) { ++nan; continue; }
    dmax = fmax(dmax, fabs((double)oh[i] - o32[i])); omax = fmax(omax, fabs((double)o32[i]));
  }
  size_t udiff = 0;
  {
    std::vector<float> ou(oh.size());
    hipMemcpy(ou.data(), doutu, ou.size() * 4, hipMemcpyDeviceToHost);
    size_t byc[4] = {0, 0, 0, 0}, bypar[2] = {0, 0}; int shown = 0;
    for (size_t i = 0; i < oh.size(); ++i)
      if (memcmp(&ou[i], &oh[i], 4) != 0) {
        ++udiff;
        const int c = (int)(i % Cout), x = (int)((i / Cout) % Wo), y = (int)((i / Cout / Wo) % Ho), b = (int)(i / Cout / Wo / Ho);
        const int tx = (pool ? 2 * x : x) / 16, ty = (pool ? 2 * y : y) / 8, t = (b * ((H + 7) / 8) + ty) * ((W + 15) / 16) + tx;
        ++byc[(c % 64) / 16]; ++bypar[t & 1];
        if (getenv("SHOWDIFF") && shown++ < 12) printf("    diff b %d y %d x %d c %d (tile %d): pair %.6g  h %.6g\n", b, y, x, c, t, ou[i], oh[i]);
      }
    if (udiff && getenv("SHOWDIFF")) {
      size_t nz = 0, nan = 0, eq_nz = 0; double mx = 0;
      for (size_t i = 0; i < ou.size(); ++i) { if (ou[i] != ou[i]) ++nan; else if (ou[i] != 0) { ++nz; mx = fmax(mx, fabs(ou[i])); if (ou[i] == oh[i]) ++eq_nz; } }
      printf("    pair output: %zu nonzero (%zu equal to h), %zu NaN, max %.4g of %zu\n", nz, eq_nz, nan, mx, ou.size());
    }
    if (udiff && getenv("SHOWDIFF")) printf("    by channel block: %zu %zu %zu %zu; by tile parity: %zu %zu\n", byc[0], byc[1], byc[2], byc[3], bypar[0], bypar[1]);
    std::vector<unsigned> amn(256);
    hipMemcpy(amn.data(), damax_out, 1024, hipMemcpyDeviceToHost);
    for (int b = 0; b < B && b < 256; ++b) udiff += amn[b] != amu[b];
    hipMemcpy(damax_out, amu.data(), (B < 256 ? B : 256) * 4, hipMemcpyHostToDevice);     // (the check below reads the h form's maxima)
  }
  // float64 direct reference on sample output pixels (all channels)
  double e32 = 0, eh = 0; int ns = 0;
  for (int sidx = 0; sidx < 48; ++sidx) {
    const int b = (sidx * 37) % B, oy = (sidx * 53 + (sidx & 1 ? Ho - 1 : 0)) % Ho, ox = (sidx * 29 + (sidx & 2 ? Wo - 1 : 0)) % Wo;
    for (int co = 0; co < Cout; ++co) {
      double best = -1e300;
      for (int py = 0; py < (pool ? 2 : 1); ++py)
        for (int px = 0; px < (pool ? 2 : 1); ++px) {
          const int y = pool ? 2 * oy + py : oy, xx = pool ? 2 * ox + px : ox;
          double acc = bias[co];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = y + ky - 1, ix = xx + kx - 1;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              const float* xp = &x[(((size_t)b * H + iy) * W + ix) * Cin];
              const float* wp = &w[((size_t)(ky * 3 + kx) * Cin) * Cout + co];
              for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * wp[(size_t)ci * Cout];
            }
          best = fmax(best, acc);
        }
      best = fmax(best, 0.0);
      const size_t oi = (((size_t)b * Ho + oy) * Wo + ox) * Cout + co;
      e32 = fmax(e32, fabs(o32[oi] - best)); eh = fmax(eh, fabs(oh[oi] - best)); ++ns;
    }
  }
  std::vector<unsigned> am(B);
  hipMemcpy(am.data(), damax_out, B * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    float m = 0;
    for (size_t i = 0; i < (size_t)Ho * Wo * Cout; ++i) m = fmaxf(m, oh[(size_t)b * Ho * Wo * Cout + i]);
    float got; memcpy(&got, &am[b], 4);
    if (!(got >= m) || got > 1.5f * m + 1e-6f) { if (bad < 3) printf("  amax_out[%d] = %g, host max %g\n", b, got, m); ++bad; }
  }
  const double macs = (double)B * H * W * 9.0 * Cin * Cout;
  printf("%4dx%-4d %3d->%-3d pool %d %s | fp32 wino %7.1f us  f16x2 wino %7.1f us (x%.2f, %5.1f TFLOP/s direct-equivalent)  pair %7.1f us (x%.2f vs h; %zu words differ) | max |h - f32| %.2e of %.2e%s | vs float64 (%d samples): f32 %.2e  f16x2 %.2e | amax_out bad %d\n",
         H, W, Cin, Cout, pool, blocked ? "blocked" : "nhwc   ", ms[0] * 1e3, ms[1] * 1e3, ms[0] / ms[1], 2 * macs / (ms[1] * 1e-3) / 1e12, ms[2] * 1e3, ms[1] / ms[2], udiff, dmax, omax,
         nan ? " NaN!" : "", ns, e32, eh, bad);
  hipFree(dx); hipFree(dw32); hipFree(db); hipFree(duh); hipFree(dout32); hipFree(douth); hipFree(doutu); hipFree(damax); hipFree(damax_out);
  return 0;
}

// the fused first layer: conv1ab_wino24h against conv1ab_wino24 (outputs against each other and their per-image maxima)
static int run_first(int B, int H, int W, float mag) {
  const int Ho = H / 2, Wo = W / 2, C = 64;
  std::vector<float> im((size_t)B * H * W), w1(9 * 64), b1(64), w((size_t)9 * C * C), bias(C);
  srand(7);
  auto rnd = []() { return rand() / (float)RAND_MAX; };
  for (auto& v : im) v = rnd() * mag;
  for (auto& v : w1) v = (rnd() - 0.5f) * 0.8f;
  for (auto& v : b1) v = (rnd() - 0.5f) * 0.3f;
  for (auto& v : w) v = (rnd() - 0.5f) * 0.12f;
  for (auto& v : bias) v = (rnd() - 0.5f) * 0.2f;
  float l1 = 0, bmax = 0;
  for (int c = 0; c < 64; ++c) { float t = 0; for (int k = 0; k < 9; ++k) t += fabsf(w1[k * 64 + c]); l1 = fmaxf(l1, t); bmax = fmaxf(bmax, fabsf(b1[c])); }
  const std::vector<float> u32 = wino24_transform(w, C, C);
  float su_inv = 0;
  const std::vector<uint16_t> uh = wino24h_pack(w, C, C, &su_inv);
  float *dim_, *dw1, *db1, *dw32, *db, *dout32, *douth; void* duh; unsigned* damax_out;
  hipMalloc(&dim_, im.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&db1, 256); hipMalloc(&dw32, u32.size() * 4); hipMalloc(&db, 256); hipMalloc(&duh, uh.size() * 2);
  hipMalloc(&dout32, (size_t)B * Ho * Wo * C * 4); hipMalloc(&douth, (size_t)B * Ho * Wo * C * 4); hipMalloc(&damax_out, 1024);
  hipMemcpy(dim_, im.data(), im.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db1, b1.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dw32, u32.data(), u32.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), 256, hipMemcpyHostToDevice); hipMemcpy(duh, uh.data(), uh.size() * 2, hipMemcpyHostToDevice);
  ConvArgs a{};
  a.in = dim_; a.in2 = dim_; a.split = B; a.wu24 = dw32; a.bias = db; a.w1 = dw1; a.b1 = db1; a.B = B; a.H = H; a.W = W; a.Cin = 64; a.Cout = 64;
  a.relu = 1; a.pool = 1; a.first = 1; a.out_blocked = 1; a.wuh = duh; a.u_scale_inv = su_inv; a.c1a_l1 = l1; a.c1a_bmax = bmax; a.amax_out = damax_out;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[3] = {0, 0, 0};
  std::vector<unsigned> am[3];
  float* doutp; hipMalloc(&doutp, (size_t)B * Ho * Wo * C * 4); hipMemset(doutp, 0xff, (size_t)B * Ho * Wo * C * 4);
  for (int form = 0; form < 3; ++form) {
    a.out = form == 2 ? doutp : form ? douth : dout32;
    hipMemset(damax_out, 0, 1024);
    auto go = [&]() { return form == 2 ? launch_conv1ab_wino24p(a, 0) : form ? launch_conv1ab_wino24h(a, 0) : launch_conv1ab_wino24(a, 0); };
    hipError_t err = go();
    hipDeviceSynchronize();
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed (first, form %d): %s\n", form, hipGetErrorString(err)); return 1; }
    am[form].resize(256); hipMemcpy(am[form].data(), damax_out, 1024, hipMemcpyDeviceToHost);
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 5;      // (ITERS=60: sustained clocks, as inside the step)
    for (int i = 0; i < 2; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[form], e0, e1);
    ms[form] /= iters;
  }
  std::vector<float> o32((size_t)B * Ho * Wo * C), oh(o32.size());
  hipMemcpy(o32.data(), dout32, o32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), douth, oh.size() * 4, hipMemcpyDeviceToHost);
  double dmax = 0, omax = 0; size_t nan = 0;
  for (size_t i = 0; i < oh.size(); ++i) {
    if (!(oh[i] == oh[i])) { ++nan; continue; }
    dmax = fmax(dmax, fabs((double)oh[i] - o32[i])); omax = fmax(omax, fabs((double)o32[i]));
  }
  int bad = 0;
  for (int b = 0; b < B && b < 256; ++b) if (am[0][b] != am[1][b]) { float x, y; memcpy(&x, &am[0][b], 4); memcpy(&y, &am[1][b], 4); if (fabsf(x - y) > 1e-4f * fabsf(x)) ++bad; }
  size_t pdiff = 0;
  {
    std::vector<float> op(oh.size());
    hipMemcpy(op.data(), doutp, op.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < oh.size(); ++i) pdiff += memcmp(&op[i], &oh[i], 4) != 0;
    for (int b = 0; b < B && b < 256; ++b) pdiff += am[2][b] != am[1][b];
  }
  printf("first layer %dx%d B=%d | pair %8.1f us (x%.2f vs h; %zu words differ) | fp32 wino %8.1f us  f16x2 wino %8.1f us (x%.2f) | max |h - f32| %.2e of %.2e%s | per-image maxima differing by > 1e-4: %d\n",
         H, W, B, ms[2] * 1e3, ms[1] / ms[2], pdiff, ms[0] * 1e3, ms[1] * 1e3, ms[0] / ms[1], dmax, omax, nan ? " NaN!" : "", bad);
  return 0;
}

int main(int argc, char** argv) {
  const float mag = getenv("MAG") ? (float)atof(getenv("MAG")) : 3.f;
  if (argc > 4 && !strcmp(argv[1], "first")) return run_first(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), getenv("MAG") ? mag : 1.f);
  if (argc > 5) return run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 0, argc > 7 ? atoi(argv[7]) : 0, mag);
  const int B = argc > 1 ? atoi(argv[1]) : 128;
  run(B, 240, 320, 64, 64, 0, 1, mag);      // conv2a
  run(B, 240, 320, 64, 64, 1, 1, mag);      // conv2b + pool
  run(B, 120, 160, 64, 128, 0, 1, mag);     // conv3a
  run(B, 120, 160, 128, 128, 1, 1, mag);    // conv3b + pool
  run(B, 60, 80, 128, 128, 0, 1, mag);      // conv4a / conv4b
  run(B, 60, 80, 128, 512, 0, 1, mag);      // convPa | convDa
  run_first(B, 480, 640, 1.f);
  run_first(3, 123, 165, 1.f);              // ragged
  run(3, 37, 53, 64, 64, 1, 0, mag);        // ragged, NHWC, odd pooled size
  run(2, 49, 48, 128, 64, 0, 0, mag);
  return 0;
}
