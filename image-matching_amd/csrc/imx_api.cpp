// imx_api.cpp — C ABI of libimx.so (include/imx.h): handle, weight loading with BatchNorm
// folding and layout transforms, workspace, and the launch sequences of the SuperPoint /
// SuperGlue / Matching forwards.  All arithmetic of the path runs in the kernels declared in
// imx_kernels.h; this file only moves weights once and enqueues kernels on the caller's stream.
#include "../../include/imx.h"
#include "imx_kernels.h"
#include "gnn_tail_pack.h"
#include "wino24_pack.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <new>
#include <atomic>
#include <string>
#include <vector>

using namespace imx;

namespace imx {
thread_local const char* last_form = nullptr;
}

namespace {

constexpr double BN_EPS = 1e-5;   // nn.BatchNorm default (unet_parts.py:16; superglue_test.py:58)
constexpr int HEADS = 4;          // AttentionalPropagation(feature_dim, 4), superglue_test.py:126

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
struct ConvW {
  float* w = nullptr;    // direct form  [9][Cin][Cout] (conv3x3.hip: IMX_CONV=direct, and the fallback for shapes wino24 rejects)
  float* wu24 = nullptr; // Winograd F(2x4,3x3) form (conv1ab_wino24.hip / conv3x3_wino24.hip layout)
  float u_spread = 1.f;  // wino24h_pack: max |U| / median over output channels of their max |U| (the fp16-plane guard)
  void* wuh = nullptr;   // the same as two fp16 planes scaled by a power of two (conv3x3_wino24h.hip; wino24_pack.h: wino24h_pack), cin % 64 == 0 only
  float su_inv = 0.f;    // 1 / that power of two
  float* b = nullptr;
  int cin = 0, cout = 0;
};
struct GemmW {
  float* w = nullptr;
  float* wx3 = nullptr;   // the same weights as three bf16 terms per value in MFMA fragment order (split_bf16x3; gemm_x3.hip)
  float* wf = nullptr;    // the same fp32 weights in the B-fragment order of gnn_small.hip (fragment_order; K % 16 == 0 only)
  float* wh2 = nullptr;   // the same weights as two fp16 planes of w s in MFMA fragment order (split_f16x2; gemm_h2.hip), 1 / s, and how far
  float wh2_inv = 0.f;    // the largest |w| sits above the median over output columns of their largest |w| (the guard of the fp16 form:
  float wh2_spread = 0.f; // beyond kLinearSpreadMax the typical column loses bits of its low plane and the chain stays on gemm_x3)
  float* b = nullptr;
  int K = 0, N = 0, Npad = 0;
};
struct GnnLayer {
  GemmW qkv, merge, mlp1, mlp2;
  float* tail_stream = nullptr;    // gnn_tail_pack() of (mlp.0', mlp.3, the NEXT layer's q|k|v or final_proj): gnn_tail_x3.hip, d = 128 only
  void* tail_stream_h2 = nullptr;  // gnn_tail_pack_h2() of the same three matrices (two fp16 planes): gnn_tail_h2.hip
  GnnTailH2Consts h2c{};           // reciprocal weight scales and column L1 norms
  float bmax_1 = 0.f, bmax_2 = 0.f;   // largest |bias| of mlp.0' and mlp.3
  float qkv_spread = 1.f;          // over q | k | v: (largest column L2 norm) / (median column L2 norm) of that projection -- how far its strongest output
                                   // channel sits above the typical one.  The two-plane attention scales a whole (side, pair) by its ACTUAL maximum: beyond
                                   // kAttnSpreadMax the typical channel would lose its low plane and the layer's attention runs on three bf16 planes
};
struct Tap {
  const void* p;
  std::vector<int64_t> shape;      // what imx_debug_fetch returns (NHWC for activations)
  bool blocked = false;            // the device buffer is channel-blocked (B, C/8, H, W, 8): un-blocked on the host at fetch time
};
struct TimedEvent {
  std::string name;
  const char* form;      // what the launcher reported through imx::last_form (static strings), or nullptr
  hipEvent_t e0, e1;
};
struct TimingRow {
  std::string name, form;
  int64_t launches;
  double ms;
};

std::string g_create_error;

}  // namespace

struct imx_handle_s {
  int device = 0;
  imx_config_t cfg{};
  std::string err;
  std::map<std::string, std::vector<int64_t>> expected[2];
  std::map<std::string, HostTensor> raw[2];
  bool finalized[2] = {false, false};
  std::vector<void*> weight_allocs[2];   // per net: freed when that net's weights are finalized again
  int upload_net = 0;                    // net whose finalize is running (upload() files allocations under it)
  // SuperPoint
  float *w1 = nullptr, *b1 = nullptr;
  // the dense NMS map behind the "nms" tap when the detector ran the candidate-bit form: computed on demand by imx_debug_fetch
  struct { const float* smap = nullptr; float* nms = nullptr; void* scratch = nullptr; int B = 0, H = 0, W = 0, radius = 0; bool pending = false; } nms_lazy;
  float c1a_l1 = 0.f, c1a_bmax = 0.f, c1a_spread = 1.f;   // (c1a_spread: max over median of conv1a's output-channel maxima, the fp16-plane guard)
  ConvW conv[8];   // conv1b, 2a, 2b, 3a, 3b, 4a, 4b, heads (convPa | convDa)
  GemmW pb, db;
  // SuperGlue
  float *kenc0_w = nullptr, *kenc0_b = nullptr;
  int kenc_c1 = 0;
  std::vector<GemmW> kenc;
  std::vector<GnnLayer> layers;
  GemmW final_proj;
  float bin_score = 1.f;
  // workspace
  std::map<std::string, DevBuf> bufs;
  // state of the last detect
  int det_B = 0, det_H = 0, det_W = 0, det_Hc = 0, det_Wc = 0, det_Ksel = 0;
  // kernel-form options (imx_set_option; the environment only seeds them at imx_create)
  Options opt;
  std::string opt_text;      // backing store of imx_get_option's return value
  // debug / timing
  bool debug = false, timing = false;
  std::map<std::string, Tap> taps;
  std::vector<TimedEvent> events;
  std::vector<TimingRow> report;
};

namespace {

int fail(imx_handle_t h, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return -1;
}

#define HIP_OK(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

float* ws(imx_handle_t h, const std::string& name, size_t bytes) {
  DevBuf& b = h->bufs[name];
  if (b.bytes >= bytes && b.p) return static_cast<float*>(b.p);
  if (b.p) {
    (void)hipDeviceSynchronize();
    (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    b.p = nullptr;
    h->err = "hipMalloc failed for workspace '" + name + "' (" + std::to_string(want) + " bytes)";
    return nullptr;
  }
  b.bytes = want;
  return static_cast<float*>(b.p);
}
#define WS(var, type, name, bytes)                         \
  type* var = reinterpret_cast<type*>(ws(h, name, bytes)); \
  if (!var) return -1;

template <class F>
int run(imx_handle_t h, const char* name, hipStream_t s, F&& f) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->timing) {
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(h, "hipEventCreate failed");
    (void)hipEventRecord(e0, s);
  }
  last_form = nullptr;
  hipError_t e = f();
  if (h->timing) {
    (void)hipEventRecord(e1, s);
    h->events.push_back({name, last_form, e0, e1});
  }
  if (e != hipSuccess) return fail(h, "kernel '%s' launch failed: %s", name, hipGetErrorString(e));
  return 0;
}
#define RUN(name, expr)                                              \
  do {                                                               \
    if (run(h, name, s, [&]() -> hipError_t { return (expr); })) return -1; \
  } while (0)

void tap(imx_handle_t h, const char* name, const void* p, std::vector<int64_t> shape, bool blocked = false) {
  h->taps[name] = Tap{p, std::move(shape), blocked};
}

// ----------------------------------------------------------------------------- expected keys
void add_conv_keys(std::map<std::string, std::vector<int64_t>>& m, const std::string& conv, int cout, int cin, int k,
                   const std::string& bn) {
  if (k > 0) m[conv + ".weight"] = {cout, cin, k, k}; else m[conv + ".weight"] = {cout, cin, 1};
  m[conv + ".bias"] = {cout};
  if (!bn.empty())
    for (const char* leaf : {"weight", "bias", "running_mean", "running_var"}) m[bn + "." + leaf] = {cout};
}

void build_expected(imx_handle_t h) {
  const imx_config_t& c = h->cfg;
  auto& sp = h->expected[IMX_NET_SUPERPOINT];
  const int c1 = 64, c2 = 64, c3 = 128, c4 = 128, c5 = 256, d = c.descriptor_dim;
  if (c.sp_variant == IMX_SP_VARIANT_BN) {
    const char* blocks[4] = {"inc.conv.conv", "down1.mpconv.1.conv", "down2.mpconv.1.conv", "down3.mpconv.1.conv"};
    const int cin[4] = {1, c1, c2, c3}, cout[4] = {c1, c2, c3, c4};
    for (int i = 0; i < 4; ++i) {
      std::string p = blocks[i];
      add_conv_keys(sp, p + ".0", cout[i], cin[i], 3, p + ".1");
      add_conv_keys(sp, p + ".3", cout[i], cout[i], 3, p + ".4");
    }
    add_conv_keys(sp, "convPa", c5, c4, 3, "bnPa");
    add_conv_keys(sp, "convPb", 65, c5, 1, "bnPb");
    add_conv_keys(sp, "convDa", c5, c4, 3, "bnDa");
    add_conv_keys(sp, "convDb", d, c5, 1, "bnDb");
  } else {
    const char* names[8] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"};
    const int cin[8] = {1, c1, c1, c2, c2, c3, c3, c4}, cout[8] = {c1, c1, c2, c2, c3, c3, c4, c4};
    for (int i = 0; i < 8; ++i) add_conv_keys(sp, names[i], cout[i], cin[i], 3, "");
    add_conv_keys(sp, "convPa", c5, c4, 3, "");
    add_conv_keys(sp, "convPb", 65, c5, 1, "");
    add_conv_keys(sp, "convDa", c5, c4, 3, "");
    add_conv_keys(sp, "convDb", d, c5, 1, "");
  }
  auto& sg = h->expected[IMX_NET_SUPERGLUE];
  sg["bin_score"] = {};
  std::vector<int> ch = {3};
  for (int i = 0; i < c.kenc_n; ++i) ch.push_back(c.kenc_channels[i]);
  ch.push_back(d);
  for (size_t i = 1; i < ch.size(); ++i) {
    int j = 3 * (int)(i - 1);
    std::string p = "kenc.encoder." + std::to_string(j);
    add_conv_keys(sg, p, ch[i], ch[i - 1], 0, i + 1 < ch.size() ? "kenc.encoder." + std::to_string(j + 1) : "");
  }
  for (int l = 0; l < c.num_gnn_layers; ++l) {
    std::string p = "gnn.layers." + std::to_string(l);
    add_conv_keys(sg, p + ".attn.merge", d, d, 0, "");
    for (int k = 0; k < 3; ++k) add_conv_keys(sg, p + ".attn.proj." + std::to_string(k), d, d, 0, "");
    add_conv_keys(sg, p + ".mlp.0", 2 * d, 2 * d, 0, p + ".mlp.1");
    add_conv_keys(sg, p + ".mlp.3", d, 2 * d, 0, "");
  }
  add_conv_keys(sg, "final_proj", d, d, 0, "");
}

// ----------------------------------------------------------------------------- weight folding
bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

float* upload(imx_handle_t h, const std::vector<float>& v) {
  void* p = nullptr;
  if (hipMalloc(&p, v.size() * sizeof(float) + 16) != hipSuccess) return nullptr;
  if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p);
    return nullptr;
  }
  h->weight_allocs[h->upload_net].push_back(p);
  return static_cast<float*>(p);
}

void* upload_u16(imx_handle_t h, const std::vector<uint16_t>& v) {
  void* p = nullptr;
  if (hipMalloc(&p, v.size() * 2 + 16) != hipSuccess) return nullptr;
  if (hipMemcpy(p, v.data(), v.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p);
    return nullptr;
  }
  h->weight_allocs[h->upload_net].push_back(p);
  return p;
}

// per-output-channel scale s and shift t such that BN(conv + b) = conv*s + t
void fold_bn(const std::map<std::string, HostTensor>& raw, const std::string& conv, const std::string& bn, int cout,
             std::vector<double>& s, std::vector<double>& t) {
  const std::vector<float>& b = raw.at(conv + ".bias").data;
  s.assign(cout, 1.0);
  t.resize(cout);
  if (bn.empty()) {
    for (int c = 0; c < cout; ++c) t[c] = b[c];
    return;
  }
  const auto& g = raw.at(bn + ".weight").data;
  const auto& be = raw.at(bn + ".bias").data;
  const auto& mu = raw.at(bn + ".running_mean").data;
  const auto& var = raw.at(bn + ".running_var").data;
  for (int c = 0; c < cout; ++c) {
    s[c] = (double)g[c] / std::sqrt((double)var[c] + BN_EPS);
    t[c] = ((double)b[c] - (double)mu[c]) * s[c] + (double)be[c];
  }
}

// torch Conv2d (Cout,Cin,3,3) -> [tap = ky*3+kx][ci][co_off + co] of a [9][Cin][cout_total] buffer
void put_conv3(const std::map<std::string, HostTensor>& raw, const std::string& conv, const std::string& bn, int cin,
               int cout, int cout_total, int co_off, std::vector<float>& w, std::vector<float>& bias) {
  std::vector<double> s, t;
  fold_bn(raw, conv, bn, cout, s, t);
  const std::vector<float>& src = raw.at(conv + ".weight").data;
  for (int co = 0; co < cout; ++co) {
    for (int ci = 0; ci < cin; ++ci)
      for (int tp = 0; tp < 9; ++tp)
        w[((size_t)tp * cin + ci) * cout_total + co_off + co] = (float)((double)src[((size_t)co * cin + ci) * 9 + tp] * s[co]);
    bias[co_off + co] = (float)t[co];
  }
}

// bf16 helpers of the host-side splits (round to nearest even, bit patterns)
uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
// U = G2 g G4^T as in wino24_transform, every value split exactly into three bf16 terms (x = h + m + l) and laid out as

int make_conv(imx_handle_t h, ConvW& out, const std::map<std::string, HostTensor>& raw, const std::string& conv,
              const std::string& bn, int cin, int cout) {
  std::vector<float> w((size_t)9 * cin * cout), b(cout);
  put_conv3(raw, conv, bn, cin, cout, cout, 0, w, b);
  out.w = upload(h, w);
  out.wu24 = upload(h, wino24_transform(w, cin, cout));
  if (cin % 64 == 0 && cout % 64 == 0) out.wuh = upload_u16(h, wino24h_pack(w, cin, cout, &out.su_inv, &out.u_spread));
  out.b = upload(h, b);
  out.cin = cin;
  out.cout = cout;
  return (out.w && out.wu24 && out.b) ? 0 : fail(h, "weight upload failed (%s)", conv.c_str());
}

// linear weight (N,K[,1[,1]]) -> W[k_perm(k)][n_perm(n)] padded to Npad columns, folded BN
void build_gemm_host(const std::map<std::string, HostTensor>& raw, const std::string& conv, const std::string& bn, int K,
                     int N, const std::vector<int>* kperm, std::vector<float>& w, std::vector<float>& b, int& Npad) {
  Npad = ((N + 63) / 64) * 64;
  std::vector<double> s, t;
  fold_bn(raw, conv, bn, N, s, t);
  const std::vector<float>& src = raw.at(conv + ".weight").data;
  w.assign((size_t)K * Npad, 0.f);
  b.assign(Npad, 0.f);
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K; ++k) {
      const int kk = kperm ? (*kperm)[k] : k;   // kk = position of reference input channel k in OUR layout
      w[(size_t)kk * Npad + n] = (float)((double)src[(size_t)n * K + k] * s[n]);
    }
    b[n] = (float)t[n];
  }
}

// x = h + m + l with three bf16 terms (round-to-nearest-even residuals; exact for every finite fp32 whose last term does not
// underflow): W[k][n] -> planes [3][Npad][K], k contiguous, packed two bf16 per float slot (bit patterns only)
std::vector<float> split_bf16x3(const std::vector<float>& w, int K, int Npad) {
  // fragment order of v_mfma_f32_32x32x16_bf16's B operand: [column block of 32][16-k step][plane][lane = (col & 31) + 32 kb][8]
  // holds W[16 st + 8 kb + j][32 nb + (col & 31)], j = 0..7
  const int nst = K / 16;
  std::vector<uint16_t> pl((size_t)3 * Npad * K);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < Npad; ++n) {
      const float x = w[(size_t)k * Npad + n];
      uint16_t t[3];
      t[0] = bf16_rne(x);
      const float r1 = x - bf16_to_f32(t[0]);
      t[1] = bf16_rne(r1);
      t[2] = bf16_rne(r1 - bf16_to_f32(t[1]));
      const int nb = n >> 5, st = k >> 4, lane = (n & 31) + 32 * ((k >> 3) & 1), j = k & 7;
      for (int q = 0; q < 3; ++q) pl[((((size_t)nb * nst + st) * 3 + q) * 64 + lane) * 8 + j] = t[q];
    }
  std::vector<float> out((pl.size() + 1) / 2);
  memcpy(out.data(), pl.data(), pl.size() * sizeof(uint16_t));
  return out;
}

// W [K][Npad] -> [K/16][4 (kq)][Npad][4 (j)], element (t, kq, col, j) = W[16 t + 4 kq + j][col]: the four B values a lane of
// v_mfma_f32_16x16x4_f32 feeds to the four MFMAs of a 16-k group are one 16-byte load, sixteen lanes 256 contiguous bytes
std::vector<float> fragment_order(const std::vector<float>& w, int K, int Npad) {
  std::vector<float> f((size_t)K * Npad);
  for (int t = 0; t < K / 16; ++t)
    for (int kq = 0; kq < 4; ++kq)
      for (int col = 0; col < Npad; ++col)
        for (int j = 0; j < 4; ++j) f[(((size_t)t * 4 + kq) * Npad + col) * 4 + j] = w[(size_t)(16 * t + 4 * kq + j) * Npad + col];
  return f;
}

// w s = h + m with two fp16 planes, s = the power of two that brings max |w| to [2^13, 2^14) (round-to-nearest-even, subnormals kept:
// exact to 2^-38 of max |w|), in the fragment order of v_mfma_f32_32x32x16_f16's B operand: [column block of 32][16-k step][plane]
// [lane = (col & 31) + 32 kb][8] holds W[16 st + 8 kb + j][32 nb + (col & 31)].  Returns 1 / s and the spread (GemmW::wh2_spread).
std::vector<float> split_f16x2(const std::vector<float>& w, int K, int N, int Npad, float* s_inv, float* spread) {
  const int nst = K / 16;
  double mx = 0.0;
  std::vector<double> colmax(N > 0 ? N : 1, 0.0);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      const double a = std::fabs((double)w[(size_t)k * Npad + n]);
      colmax[n] = std::max(colmax[n], a);
      mx = std::max(mx, a);
    }
  std::nth_element(colmax.begin(), colmax.begin() + colmax.size() / 2, colmax.end());
  const double med = colmax[colmax.size() / 2];
  *spread = (float)std::min(1e30, med > 0 ? mx / med : (mx > 0 ? 1e30 : 1.0));
  int e = 0;
  if (mx > 0) std::frexp(mx, &e);
  const double sc = std::ldexp(1.0, 14 - e);
  *s_inv = (float)(1.0 / sc);
  std::vector<uint16_t> pl((size_t)2 * Npad * K);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < Npad; ++n) {
      const float x = (float)((double)w[(size_t)k * Npad + n] * sc);
      const uint16_t hh = gt_f16_rne(x), mm = gt_f16_rne(x - gt_f16_f(hh));
      const int nb = n >> 5, st = k >> 4, lane = (n & 31) + 32 * ((k >> 3) & 1), j = k & 7;
      pl[((((size_t)nb * nst + st) * 2 + 0) * 64 + lane) * 8 + j] = hh;
      pl[((((size_t)nb * nst + st) * 2 + 1) * 64 + lane) * 8 + j] = mm;
    }
  std::vector<float> out((pl.size() + 1) / 2);
  memcpy(out.data(), pl.data(), pl.size() * sizeof(uint16_t));
  return out;
}

int upload_gemm(imx_handle_t h, GemmW& out, const std::vector<float>& w, const std::vector<float>& b, int K, int N, int Npad,
                const char* what) {
  out.w = upload(h, w);
  out.wf = (K % 16 == 0 && K <= 512 && N == Npad) ? upload(h, fragment_order(w, K, Npad)) : nullptr;
  out.wx3 = upload(h, split_bf16x3(w, K, Npad));
  if (K % 32 == 0 && Npad % 64 == 0) out.wh2 = upload(h, split_f16x2(w, K, N, Npad, &out.wh2_inv, &out.wh2_spread));
  out.b = upload(h, b);
  out.K = K;
  out.N = N;
  out.Npad = Npad;
  return (out.w && out.wx3 && out.b) ? 0 : fail(h, "weight upload failed (%s)", what);
}

int make_gemm(imx_handle_t h, GemmW& out, const std::map<std::string, HostTensor>& raw, const std::string& conv,
              const std::string& bn, int K, int N, const std::vector<int>* kperm = nullptr) {
  std::vector<float> w, b;
  int Npad = 0;
  build_gemm_host(raw, conv, bn, K, N, kperm, w, b, Npad);
  return upload_gemm(h, out, w, b, K, N, Npad, conv.c_str());
}

int finalize_superpoint(imx_handle_t h) {
  const auto& raw = h->raw[IMX_NET_SUPERPOINT];
  const bool bn = h->cfg.sp_variant == IMX_SP_VARIANT_BN;
  const int d = h->cfg.descriptor_dim;
  std::string ck[8], bk[8];
  if (bn) {
    const char* blocks[4] = {"inc.conv.conv", "down1.mpconv.1.conv", "down2.mpconv.1.conv", "down3.mpconv.1.conv"};
    for (int i = 0; i < 4; ++i) {
      ck[2 * i] = std::string(blocks[i]) + ".0"; bk[2 * i] = std::string(blocks[i]) + ".1";
      ck[2 * i + 1] = std::string(blocks[i]) + ".3"; bk[2 * i + 1] = std::string(blocks[i]) + ".4";
    }
  } else {
    const char* names[8] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"};
    for (int i = 0; i < 8; ++i) ck[i] = names[i];
  }
  // conv1a: (64,1,3,3) -> [9][64]
  {
    std::vector<float> w(9 * 64), b(64);
    put_conv3(raw, ck[0], bk[0], 1, 64, 64, 0, w, b);
    h->w1 = upload(h, w);
    h->b1 = upload(h, b);
    if (!h->w1 || !h->b1) return fail(h, "weight upload failed (conv1a)");
    // |conv1a output| <= max|image patch| * c1a_l1 + c1a_bmax (conv1ab_wino24h.hip's per-tile scale)
    h->c1a_l1 = 0.f; h->c1a_bmax = 0.f;
    for (int c = 0; c < 64; ++c) {
      float l1 = 0.f;
      for (int t = 0; t < 9; ++t) l1 += std::fabs(w[t * 64 + c]);
      h->c1a_l1 = std::max(h->c1a_l1, l1);
      h->c1a_bmax = std::max(h->c1a_bmax, std::fabs(b[c]));
    }
    h->c1a_l1 = std::max(h->c1a_l1, 1e-30f);
    // conv1a's weights meet no fp16 scale (fp32 pipe), but an output channel far above the others is what conv1b's input transform then
    // sees -- one channel owning the tile's power of two, the typical channel 2^k below it.  wino24h_pack's statistic on the plain
    // weights (the rescale conv1a x 2^k / conv1b's column x 2^-k showed in no transformed-weight table)
    std::vector<double> comax(64, 0.0);
    double wmax = 0.0;
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < 64; ++c) {
        comax[c] = std::max(comax[c], (double)std::fabs(w[t * 64 + c]));
        wmax = std::max(wmax, comax[c]);
      }
    std::nth_element(comax.begin(), comax.begin() + 32, comax.end());
    h->c1a_spread = comax[32] > 0 ? (float)std::min(3.0e38, wmax / comax[32]) : (wmax > 0 ? 3.0e38f : 1.f);
  }
  const int cin[8] = {1, 64, 64, 64, 64, 128, 128, 128}, cout[8] = {64, 64, 64, 64, 128, 128, 128, 128};
  for (int i = 1; i < 8; ++i)
    if (make_conv(h, h->conv[i - 1], raw, ck[i], bk[i], cin[i], cout[i])) return -1;
  // heads: convPa | convDa merged into one 128 -> 512 convolution
  {
    std::vector<float> w((size_t)9 * 128 * 512), b(512);
    put_conv3(raw, "convPa", bn ? "bnPa" : "", 128, 256, 512, 0, w, b);
    put_conv3(raw, "convDa", bn ? "bnDa" : "", 128, 256, 512, 256, w, b);
    h->conv[7].w = upload(h, w);
    h->conv[7].wu24 = upload(h, wino24_transform(w, 128, 512));
    h->conv[7].wuh = upload_u16(h, wino24h_pack(w, 128, 512, &h->conv[7].su_inv, &h->conv[7].u_spread));
    h->conv[7].b = upload(h, b);
    h->conv[7].cin = 128;
    h->conv[7].cout = 512;
    if (!h->conv[7].w || !h->conv[7].wu24 || !h->conv[7].b) return fail(h, "weight upload failed (heads)");
  }
  if (make_gemm(h, h->pb, raw, "convPb", bn ? "bnPb" : "", 256, 65)) return -1;
  if (make_gemm(h, h->db, raw, "convDb", bn ? "bnDb" : "", 256, d)) return -1;
  return 0;
}

int finalize_superglue(imx_handle_t h) {
  const auto& raw = h->raw[IMX_NET_SUPERGLUE];
  const imx_config_t& c = h->cfg;
  const int d = c.descriptor_dim;
  if (d % HEADS != 0 || (d / HEADS != 16 && d / HEADS != 32 && d / HEADS != 64))
    return fail(h, "SuperGlue needs descriptor_dim/4 in {16,32,64} (got descriptor_dim=%d)", d);
  h->bin_score = raw.at("bin_score").data[0];
  std::vector<int> ch = {3};
  for (int i = 0; i < c.kenc_n; ++i) ch.push_back(c.kenc_channels[i]);
  ch.push_back(d);
  for (size_t i = 1; i < ch.size(); ++i)
    if (ch[i] % 32) return fail(h, "keypoint_encoder widths must be multiples of 32 (got %d)", ch[i]);
  // layer 0: (C1,3,1) + BN -> w[3][C1]
  {
    const int C1 = ch[1];
    std::vector<double> s, t;
    fold_bn(raw, "kenc.encoder.0", ch.size() > 2 ? "kenc.encoder.1" : "", C1, s, t);
    const auto& src = raw.at("kenc.encoder.0.weight").data;
    std::vector<float> w(3 * C1), b(C1);
    for (int n = 0; n < C1; ++n) {
      for (int k = 0; k < 3; ++k) w[k * C1 + n] = (float)((double)src[n * 3 + k] * s[n]);
      b[n] = (float)t[n];
    }
    h->kenc0_w = upload(h, w);
    h->kenc0_b = upload(h, b);
    h->kenc_c1 = C1;
    if (!h->kenc0_w || !h->kenc0_b) return fail(h, "weight upload failed (kenc0)");
  }
  h->kenc.clear();
  for (size_t i = 2; i < ch.size(); ++i) {
    int j = 3 * (int)(i - 1);
    GemmW g;
    if (make_gemm(h, g, raw, "kenc.encoder." + std::to_string(j),
                  i + 1 < ch.size() ? "kenc.encoder." + std::to_string(j + 1) : "", ch[i - 1], ch[i]))
      return -1;
    h->kenc.push_back(g);
  }
  // channel permutation: reference channel c = dim*HEADS + head (view(b, dim, heads, n),
  // superglue_test.py:104)  ->  ours = head*HD + dim
  const int HD = d / HEADS;
  std::vector<int> perm(d);
  for (int cidx = 0; cidx < d; ++cidx) perm[cidx] = (cidx % HEADS) * HD + cidx / HEADS;
  h->layers.clear();
  std::vector<std::vector<float>> host_qkv, host_w1, host_w2;      // host copies for the fused layer tail's weight stream (d = 128)
  std::vector<int> host_ld1, host_ld2;
  for (int l = 0; l < c.num_gnn_layers; ++l) {
    std::string p = "gnn.layers." + std::to_string(l);
    GnnLayer L;
    // fused q|k|v projection: W[k][which*d + perm(n)]
    {
      const int N = 3 * d;
      std::vector<float> w((size_t)d * N, 0.f), b(N, 0.f);
      for (int which = 0; which < 3; ++which) {
        const auto& src = raw.at(p + ".attn.proj." + std::to_string(which) + ".weight").data;
        const auto& bs = raw.at(p + ".attn.proj." + std::to_string(which) + ".bias").data;
        for (int n = 0; n < d; ++n) {
          for (int k = 0; k < d; ++k) w[(size_t)k * N + which * d + perm[n]] = src[(size_t)n * d + k];
          b[which * d + perm[n]] = bs[n];
        }
      }
      host_qkv.push_back(w);
      L.qkv_spread = 1.f;
      for (int which = 0; which < 3; ++which) {
        std::vector<double> l2(d, 0.0);
        for (int n = 0; n < d; ++n) {
          double sq = 0.0;
          for (int k = 0; k < d; ++k) sq += (double)w[(size_t)k * N + which * d + n] * w[(size_t)k * N + which * d + n];
          l2[n] = std::sqrt(sq);
        }
        const double mx = *std::max_element(l2.begin(), l2.end());
        std::nth_element(l2.begin(), l2.begin() + d / 2, l2.end());
        L.qkv_spread = std::max(L.qkv_spread, (float)std::min(1e30, l2[d / 2] > 0 ? mx / l2[d / 2] : (mx > 0 ? 1e30 : 1.0)));
      }
      L.qkv.w = upload(h, w);
      L.qkv.wx3 = upload(h, split_bf16x3(w, d, N));
      if (d % 32 == 0 && N % 64 == 0) L.qkv.wh2 = upload(h, split_f16x2(w, d, N, N, &L.qkv.wh2_inv, &L.qkv.wh2_spread));
      L.qkv.wf = (d % 16 == 0) ? upload(h, fragment_order(w, d, N)) : nullptr;
      L.qkv.b = upload(h, b);
      L.qkv.K = d;
      L.qkv.N = N;
      L.qkv.Npad = N;
      if (!L.qkv.w || !L.qkv.wx3 || !L.qkv.b) return fail(h, "weight upload failed (%s qkv)", p.c_str());
    }
    // attn.merge is linear and feeds only mlp.0's second input half (superglue_test.py:107,119):
    //   mlp.0([x ; Wm a + bm]) = W1x x + (W1m Wm) a + (W1m bm + b1)
    // so the merge GEMM is folded into the (BN-folded) mlp.0 weights once, in double, and the kernel
    // chain per layer is qkv -> attention -> mlp1([x | a]) -> mlp2 (+residual).
    {
      std::vector<float> wm, bm, w1, b1;
      int npm = 0, np1 = 0;
      build_gemm_host(raw, p + ".attn.merge", "", d, d, &perm, wm, bm, npm);      // wm[a_ch (ours)][c], bm[c]
      build_gemm_host(raw, p + ".mlp.0", p + ".mlp.1", 2 * d, 2 * d, nullptr, w1, b1, np1);
      std::vector<float> wf((size_t)2 * d * np1, 0.f), bf(b1);
      for (int k = 0; k < d; ++k)
        for (int n = 0; n < 2 * d; ++n) wf[(size_t)k * np1 + n] = w1[(size_t)k * np1 + n];
      for (int n = 0; n < 2 * d; ++n) {
        double accb = b1[n];
        for (int cch = 0; cch < d; ++cch) accb += (double)bm[cch] * (double)w1[(size_t)(d + cch) * np1 + n];
        bf[n] = (float)accb;
      }
      for (int kk = 0; kk < d; ++kk)
        for (int n = 0; n < 2 * d; ++n) {
          double acc = 0.0;
          for (int cch = 0; cch < d; ++cch) acc += (double)wm[(size_t)kk * npm + cch] * (double)w1[(size_t)(d + cch) * np1 + n];
          wf[(size_t)(d + kk) * np1 + n] = (float)acc;
        }
      if (upload_gemm(h, L.mlp1, wf, bf, 2 * d, 2 * d, np1, (p + ".mlp.0 (+merge)").c_str())) return -1;
      for (int n = 0; n < 2 * d; ++n) L.bmax_1 = std::max(L.bmax_1, std::fabs(bf[n]));
      host_w1.push_back(wf);
      host_ld1.push_back(np1);
    }
    {
      std::vector<float> w2, b2;
      int np2 = 0;
      build_gemm_host(raw, p + ".mlp.3", "", 2 * d, d, nullptr, w2, b2, np2);
      if (upload_gemm(h, L.mlp2, w2, b2, 2 * d, d, np2, (p + ".mlp.3").c_str())) return -1;
      for (int n = 0; n < d; ++n) L.bmax_2 = std::max(L.bmax_2, std::fabs(b2[n]));
      host_w2.push_back(w2);
      host_ld2.push_back(np2);
    }
    h->layers.push_back(L);
  }
  std::vector<float> wfin, bfin;
  int npf = 0;
  build_gemm_host(raw, "final_proj", "", d, d, nullptr, wfin, bfin, npf);
  if (upload_gemm(h, h->final_proj, wfin, bfin, d, d, npf, "final_proj")) return -1;
  // the fused layer tail of the throughput path (gnn_tail_x3.hip): one weight stream per layer, in consumption order
  if (d == 128) {
    for (int l = 0; l < c.num_gnn_layers; ++l) {
      const bool last = l + 1 == c.num_gnn_layers;
      const std::vector<uint16_t> st = gnn_tail_pack(host_w1[l].data(), host_ld1[l], host_w2[l].data(), host_ld2[l],
                                                     last ? wfin.data() : host_qkv[l + 1].data(), last ? npf : 3 * d, d, last ? d : 3 * d);
      std::vector<float> bits((st.size() + 1) / 2);
      memcpy(bits.data(), st.data(), st.size() * sizeof(uint16_t));
      h->layers[l].tail_stream = upload(h, bits);
      if (!h->layers[l].tail_stream) return fail(h, "weight upload failed (layer %d tail stream)", l);
      h->layers[l].tail_stream_h2 = upload_u16(h, gnn_tail_pack_h2(host_w1[l].data(), host_ld1[l], host_w2[l].data(), host_ld2[l],
                                                                   last ? wfin.data() : host_qkv[l + 1].data(), last ? npf : 3 * d, d, last ? d : 3 * d,
                                                                   &h->layers[l].h2c));
      if (!h->layers[l].tail_stream_h2) return fail(h, "weight upload failed (layer %d fp16 tail stream)", l);
    }
  }
  return 0;
}

// the guards of the fp16-plane forms: see wino24_pack.h (spread) and gnn_tail_pack.h (loose_h / loose_x)
constexpr float kConvSpreadMax = 16384.f, kTailLooseMax = 65536.f;
// (a q / k / v channel 2^12 above the median, times an activation crest factor of 2^4, puts the typical value 2^16 below the maximum)
constexpr float kAttnSpreadMax = 4096.f;

hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
int pad32(int n) { return ((n + 31) / 32) * 32; }

// `am` (optional; the q|k|v projection): where the maxima of the output's thirds over the valid rows go if the throughput form can
// write them from its epilogue; am->done says whether it did (else the caller runs launch_qkv_amax)
// What a linear layer of the GNN needs beside its operands: the (side, pair) row structure and, optionally, the q|k|v maxima its
// epilogue should write (amax: gemm_x3 / gemm_h2), the scale sources of gemm_h2's A operand (sa0 / sa1) and the word its epilogue
// should leave for the next kernel (amax_row).  done / h2: what the launch actually did.
struct GemmAmax {
  unsigned* amax; const int* n0; const int* n1; int B, N0p, N1p, N0, N1; bool done;
  const unsigned* sa0 = nullptr; int sa0_stride = 1, sa0_off = 0;
  const unsigned* sa1 = nullptr; int sa1_stride = 1, sa1_off = 0, sa1_cross = 0;
  unsigned* amax_row = nullptr; int amax_row_stride = 1, amax_row_off = 0;
  bool want_h2 = false, h2 = false;
};
constexpr float kLinearSpreadMax = 4096.f;       // the guard of gemm_h2's and gnn_tail_h2's weight planes (one power of two per matrix: beyond 2^12 the typical
                                                 // column's weights keep fewer than the scheme's 22 bits -- tests/test_gpu_heavy.py: a q / k channel at 2^14 used 0.8 of the tolerance)
bool gemm_h2_weights_ok(const GemmW& W) { return W.wh2 && W.wh2_inv > 0.f && W.wh2_spread <= kLinearSpreadMax && W.K % 32 == 0 && W.Npad % 64 == 0; }
int gemm(imx_handle_t h, hipStream_t s, const char* name, const GemmW& W, const float* a0, int lda0, int K0, const float* a1,
         int lda1, int K1, const float* res, int ldr, float* out, int ldo, int M, bool relu, GemmAmax* am = nullptr) {
  if (K0 + K1 != W.K) return fail(h, "internal: gemm '%s' K mismatch (%d+%d vs %d)", name, K0, K1, W.K);
  GemmArgs g{a0, lda0, K0, a1, lda1, K1, W.w, W.b, res, ldr, out, ldo, M, W.N, W.Npad, relu ? 1 : 0};
  if (am) { am->done = false; am->h2 = false; }
  // Four forms, each with its reason:
  //   gemm_small  M <= 4096 rows (one or two pairs): the latency form ("latency_forms": auto / off / on);
  //   gemm_h2     the GNN's plain linear layers in the throughput path: three fp16 plane products, operands scaled by their actual
  //               (side, pair) maxima ("linear" = auto / f16x2; the caller decides it for the whole chain: want_h2);
  //   gemm_x3     the throughput form of everything else: fp32 products as six bf16 term products on the bf16 matrix pipe;
  //   gemm_tiled  fp32 MFMA: the "mfma" = "f32" A/B reference of the parity tests and the fallback for shapes gemm_x3 rejects.
  // Measured per layer inside the C3 step (64 pairs, gemm_x3 vs the fp32-MFMA forms of round 2): mlp.0 1.85 vs 2.57 ms, mlp.3
  // 1.11 vs 1.45, convPb 0.23 vs 0.51, convDb 0.25 vs 0.35, q|k|v 2.06 vs 2.08.
  const Options& o = h->opt;
  const bool small = gemm_small_supported(g) && (o.latency_forms >= 0 ? o.latency_forms != 0 : M <= 4096);
  const bool x3 = !small && !o.mfma_f32 && W.wx3 && gemm_x3_supported(g);
  if (am && am->want_h2) {
    GemmArgs gh = g;
    gh.amax = am->amax; gh.an0 = am->n0; gh.an1 = am->n1; gh.aB = am->B; gh.aN0p = am->N0p; gh.aN1p = am->N1p; gh.aN0 = am->N0; gh.aN1 = am->N1;
    gh.sa0 = am->sa0; gh.sa0_stride = am->sa0_stride; gh.sa0_off = am->sa0_off;
    gh.sa1 = am->sa1; gh.sa1_stride = am->sa1_stride; gh.sa1_off = am->sa1_off; gh.sa1_cross = am->sa1_cross;
    gh.amax_row = am->amax_row; gh.amax_row_stride = am->amax_row_stride; gh.amax_row_off = am->amax_row_off;
    gh.w_inv = W.wh2_inv;
    if (small || o.mfma_f32 || !gemm_h2_weights_ok(W) || !gemm_h2_supported(gh))
      return fail(h, "internal: gemm '%s' was planned on fp16 planes but cannot run there", name);
    am->done = am->amax != nullptr;
    am->h2 = true;
    if (h->debug) {        // developer instrumentation: chunk stamps of the first 64 workgroups (all zeros unless gemm_h2.hip was built with -DGH2_TRACE); the LAST launch's stay
      WS(trc, unsigned long long, (std::string("sg.gh2_trace_") + name).c_str(), (size_t)64 * 128 * sizeof(unsigned long long));
      HIP_OK(h, hipMemsetAsync(trc, 0, (size_t)64 * 128 * sizeof(unsigned long long), s));
      gh.trace = trc;
      tap(h, (std::string("gh2_trace_") + name).c_str(), trc, {64, 256});
    }
    RUN(name, launch_gemm_h2(gh, W.wh2, s));
    return 0;
  }
  if (am && am->amax && x3) {
    GemmArgs ga = g;
    ga.amax = am->amax; ga.an0 = am->n0; ga.an1 = am->n1; ga.aB = am->B; ga.aN0p = am->N0p; ga.aN1p = am->N1p; ga.aN0 = am->N0; ga.aN1 = am->N1;
    if (gemm_x3_amax_supported(ga)) { g = ga; am->done = true; }
  }
  RUN(name, small ? launch_gemm_small(g, s) : x3 ? launch_gemm_x3(g, W.wx3, s) : launch_gemm(g, s));
  return 0;
}

// ----------------------------------------------------------------------------- SuperPoint
int sp_detect(imx_handle_t h, const float* img0, const float* img1, int split, int B, int H, int W, int32_t* counts_out,
              hipStream_t s, bool dense_only = false, int32_t* counts_lo = nullptr, int32_t* counts_hi = nullptr) {
  if (!h->finalized[IMX_NET_SUPERPOINT]) return fail(h, "SuperPoint weights not finalized");
  if (B <= 0 || H < 8 || W < 8) return fail(h, "bad image batch shape B=%d H=%d W=%d", B, H, W);
  const imx_config_t& c = h->cfg;
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, Hc = H4 / 2, Wc = W4 / 2, H8 = Hc * 8, W8 = Wc * 8;
  const int d = c.descriptor_dim;
  const size_t f = sizeof(float);
  WS(a1, float, "sp.a1", (size_t)B * H2 * W2 * 64 * f);
  // (a2a / a3a / a4a may be written tile-swizzled -- whole 8 x 16-pixel tiles, conv3x3_wino24p.hip: sized for the padded tile grid)
  auto padded = [](int hh, int ww) { return (size_t)((hh + 7) / 8 * 8) * ((ww + 15) / 16 * 16); };
  WS(a2a, float, "sp.a2a", (size_t)B * padded(H2, W2) * 64 * f);
  WS(a2, float, "sp.a2", (size_t)B * H4 * W4 * 64 * f);
  WS(a3a, float, "sp.a3a", (size_t)B * padded(H4, W4) * 128 * f);
  WS(a3, float, "sp.a3", (size_t)B * Hc * Wc * 128 * f);
  WS(a4a, float, "sp.a4a", (size_t)B * padded(Hc, Wc) * 128 * f);
  WS(x4, float, "sp.x4", (size_t)B * Hc * Wc * 128 * f);
  WS(hd, float, "sp.heads", (size_t)B * Hc * Wc * 512 * f);
  WS(semi, float, "sp.semi", (size_t)B * Hc * Wc * 65 * f);
  WS(dense, float, "sp.dense", (size_t)B * Hc * Wc * d * f);
  WS(smap, float, "sp.score_map", (size_t)B * H8 * W8 * f);
  WS(nms, float, "sp.nms", (size_t)B * H8 * W8 * f);

  // Activations between the Winograd layers are channel-blocked (B, C/8, H, W, 8) -- dense patch loads for the next layer
  // (conv3x3_wino24.hip); the last 3x3 layer writes NHWC rows for the 1x1-conv GEMMs.  IMX_CONV=direct (or a layer the Winograd
  // kernels reject) keeps NHWC everywhere: the direct kernel reads nothing else.
  bool blocked = !h->opt.conv_direct, h_shapes = true;
  {
    const int hs[7] = {H2, H2, H4, H4, Hc, Hc, Hc}, ws_[7] = {W2, W2, W4, W4, Wc, Wc, Wc};
    for (int i = 1; i < 8 && blocked; ++i) {
      ConvArgs t{};
      t.H = hs[i - 1]; t.W = ws_[i - 1]; t.Cin = h->conv[i].cin; t.Cout = h->conv[i].cout; t.wu24 = h->conv[i].wu24;
      blocked = conv3x3_wino24_supported(t);
      // the fp16-plane kernel's OWN predicate, in the same pre-pass (the chain used to be enabled on the fp32 kernel's
      // predicate and a layer the fp16 kernel then rejected was a hard error; now the whole chain falls back to the fp32 kernels)
      t.wuh = h->conv[i].wuh; t.u_scale_inv = h->conv[i].su_inv; t.amax_in = reinterpret_cast<const unsigned*>(h);      // (any non-null pointer: shape check only)
      h_shapes = h_shapes && blocked && conv3x3_wino24h_supported(t);
    }
  }
  // "conv" = "wino": the layers after the first run their Winograd products on the fp16 matrix pipe (conv3x3_wino24h.hip).  Each needs an
  // upper bound of |input| per image: one zeroed word per (layer, image), written by the PRODUCING layer's epilogue through atomicMax --
  // the fused first layer starts the chain, so without it (conv = direct, shapes the Winograd kernels reject) the fp32 forms run
  unsigned* amax = nullptr;
  bool all_h = true;
  // ... and weights whose transformed values fit ONE power-of-two scale per layer: a layer whose typical output channel sits more than
  // 2^14 below the largest |U| (a checkpoint with one runaway channel) would push that channel's weights under 2^-3 x 2^13 / 2^14 and
  // lose the low plane -- the whole chain then runs the fp32-MFMA Winograd kernels ("conv" reports it through imx_timing_form)
  for (int i = 1; i < 8; ++i) all_h = all_h && h->conv[i].wuh != nullptr && h->conv[i].u_spread <= kConvSpreadMax;
  all_h = all_h && h->conv[0].wuh != nullptr && h->conv[0].u_spread <= kConvSpreadMax && h->c1a_spread <= kConvSpreadMax;   // (conv1b; conv1a's plain output channels)
  // The maxima live in 256 slots per layer (the kernels' LDS tables): a batch of more than 256 images runs its 3x3 layers in slices of
  // 256 images, each with its own tables, so that an image's scales -- and with them its low-order bits -- never depend on which other
  // images share the call (tests/test_gpu_superpoint.py: image b of a 260-image batch equals the same image alone).
  constexpr int kSlice = 256;
  const int nslice = (B + kSlice - 1) / kSlice;
  if (blocked && all_h && h_shapes && h->opt.conv_f16 && !h->opt.mfma_f32) {
    WS(am, unsigned, "sp.amax", (size_t)nslice * 8 * 256 * 4);          // one word per (slice, layer, image slot b % 256)
    HIP_OK(h, hipMemsetAsync(am, 0, (size_t)nslice * 8 * 256 * 4, s));
    amax = am;
  }
  int layer = 0;
  // conv2a -> conv2b, conv3a -> conv3b, conv4a -> conv4b: the tensor between them is tile-swizzled (ConvArgs::out_blocked / in_blocked
  // = 2) when BOTH layers run the pair kernel on this slice -- its stores are then 1 KB contiguous per instruction
  // (conv3x3_wino24p.hip).  "conv_swizzle" = "off": never (the A/B switch of tools and tests; the results are bit-identical either way).
  const bool swz_on = h->opt.conv_swizzle != 0;
  auto swz_pair = [&](int li_prod, int hh, int ww, int nb) -> bool {
    if (!amax || !swz_on || h->opt.conv_f16 != 1 || (li_prod != 1 && li_prod != 3 && li_prod != 5)) return false;
    for (int k = 0; k < 2; ++k) {
      const ConvW& cw = h->conv[li_prod + k];
      ConvArgs t{};
      t.B = nb; t.H = hh; t.W = ww; t.Cin = cw.cin; t.Cout = cw.cout; t.pool = (k == 1 && li_prod != 5) ? 1 : 0; t.relu = 1;
      t.wu24 = cw.wu24; t.wuh = cw.wuh; t.u_scale_inv = cw.su_inv; t.amax_in = amax;
      if (!conv3x3_wino24p_preferred(t)) return false;
    }
    return true;
  };
  auto conv = [&](const char* name, const ConvW& w, const float* in, float* out, int hh, int ww, bool pool, bool first, bool last = false) -> int {
    const int li = layer++;
    for (int sl = 0; sl < (amax ? nslice : 1); ++sl) {
    ConvArgs a{};
    const int b0 = amax ? sl * kSlice : 0, nb = amax ? std::min(kSlice, B - b0) : B;
    const bool in_z = !first && swz_pair(li - 1, hh, ww, nb), out_z = !first && !pool && !last && swz_pair(li, hh, ww, nb);
    const size_t ztile = (size_t)((hh + 7) / 8) * ((ww + 15) / 16) * 2048;       // floats per image and 16-channel block of a swizzled tensor
    // (a slice's images are packed with their layout's own size; the SLICES of a tensor that may be swizzled are spaced by the padded
    // size whatever each slice's layout is -- a short last slice can fall back to the blocked layout, and its region must not start
    // inside the region of the slice before it)
    const bool in_cand = li == 2 || li == 4 || li == 6, out_cand = li == 1 || li == 3 || li == 5;
    const size_t in_img = in_cand ? ztile * (w.cin / 16) : (size_t)hh * ww * (first ? 1 : w.cin);
    const size_t out_img = out_cand ? ztile * (w.cout / 16) : (size_t)(pool ? hh / 2 : hh) * (pool ? ww / 2 : ww) * w.cout;
    if (amax) {
      unsigned* am = amax + (size_t)sl * 8 * 256;
      a.amax_out = last ? nullptr : am + (size_t)li * 256;
      a.amax_in = li > 0 ? am + (size_t)(li - 1) * 256 : nullptr;
      a.wuh = w.wuh; a.u_scale_inv = w.su_inv;
      a.c1a_l1 = h->c1a_l1; a.c1a_bmax = h->c1a_bmax;
    }
    a.in_blocked = in_z ? 2 : (blocked && !first) ? 1 : 0;
    a.out_blocked = out_z ? 2 : (blocked && !last) ? 1 : 0;
    if (first) {                       // images below `split` come from img0, the others from img1: the slice's view of that
      if (b0 >= split) { a.in = img1 + (size_t)(b0 - split) * in_img; a.in2 = nullptr; a.split = nb; }
      else { a.in = in + (size_t)b0 * in_img; a.in2 = img1; a.split = split - b0; }
    } else {
      a.in = in + (size_t)b0 * in_img; a.in2 = nullptr; a.split = 0;
    }
    a.w = w.w; a.wu24 = w.wu24; a.bias = w.b; a.w1 = h->w1; a.b1 = h->b1; a.out = out + (size_t)b0 * out_img;
    a.B = nb; a.H = hh; a.W = ww; a.Cin = w.cin; a.Cout = w.cout; a.relu = 1; a.pool = pool ? 1 : 0; a.first = first ? 1 : 0;
    const bool fused1 = a.first && a.pool && !h->opt.conv_direct;
    const bool wino = !a.first && !h->opt.conv_direct && conv3x3_wino24_supported(a);
    const bool winoh = wino && amax && conv3x3_wino24h_supported(a);
    if (amax && !fused1 && !winoh) return fail(h, "%s: the fp16-plane Winograd chain needs every layer to take part (internal)", name);
    const bool fused1h = fused1 && amax && conv1ab_wino24h_supported(a);
    // the fp16-plane layers: tile pairs with the positions split over two waves where that fills the chip (conv3x3_wino24p.hip: a U
    // fragment serves two tiles), one tile per workgroup otherwise ("conv" = "wino_h": always) -- the same arithmetic, bit for bit
    const bool winop = winoh && h->opt.conv_f16 == 1 && conv3x3_wino24p_preferred(a);
    const bool fused1p = fused1h && h->opt.conv_f16 == 1 && conv1ab_wino24p_preferred(a);
    if ((in_z || out_z) && !winop) return fail(h, "%s: a tile-swizzled tensor needs the pair kernel on both sides (internal)", name);
    RUN(name, fused1p ? launch_conv1ab_wino24p(a, s) : fused1h ? launch_conv1ab_wino24h(a, s) : fused1 ? launch_conv1ab_wino24(a, s) : winop ? launch_conv3x3_wino24p(a, s) : winoh ? launch_conv3x3_wino24h(a, s) :
              wino ? launch_conv3x3_wino24(a, s) : launch_conv3x3(a, s));
    }
    return 0;
  };
  if (conv("conv1ab_pool", h->conv[0], img0, a1, H, W, true, true)) return -1;
  if (conv("conv2a", h->conv[1], a1, a2a, H2, W2, false, false)) return -1;
  if (conv("conv2b_pool", h->conv[2], a2a, a2, H2, W2, true, false)) return -1;
  if (conv("conv3a", h->conv[3], a2, a3a, H4, W4, false, false)) return -1;
  if (conv("conv3b_pool", h->conv[4], a3a, a3, H4, W4, true, false)) return -1;
  if (conv("conv4a", h->conv[5], a3, a4a, Hc, Wc, false, false)) return -1;
  if (conv("conv4b", h->conv[6], a4a, x4, Hc, Wc, false, false)) return -1;
  if (conv("convPaDa", h->conv[7], x4, hd, Hc, Wc, false, false, true)) return -1;
  const int rows = B * Hc * Wc;
  if (gemm(h, s, "convPb", h->pb, hd, 512, 256, nullptr, 0, 0, nullptr, 0, semi, 65, rows, false)) return -1;
  if (gemm(h, s, "convDb", h->db, hd + 256, 512, 256, nullptr, 0, 0, nullptr, 0, dense, d, rows, false)) return -1;
  h->det_Hc = Hc; h->det_Wc = Wc;
  if (dense_only) { h->det_B = 0; h->nms_lazy.pending = false; return 0; }     // network only (imx_superpoint_dense)
  RUN("softmax_shuffle", launch_softmax_shuffle(semi, 65, smap, B, Hc, Wc, s));
  WS(nms_scratch, unsigned, "sp.nms_scratch", nms_scratch_bytes(B, H8, W8, c.nms_radius));
  // Round 6: NMS + threshold + remove_borders leave the detector as candidate BIT rows (20 words per 640-pixel row) and the keypoint kernels
  // count / scatter from those -- the dense where(max_mask, scores, 0) map (a 157-MB write and two reads at C3) is only materialised for
  // the debug tap, for "keypoints" = dense, and where the bit form does not apply (radius > 4 or 0, a negative threshold: there the
  // suppressed zeros pass `> threshold`)
  const bool kp_bits = nms_candidate_bits_supported(c.nms_radius, c.keypoint_threshold) && h->opt.keypoints != 0;
  if (kp_bits) RUN("nms", launch_nms_candidate_bits(smap, B, H8, W8, c.nms_radius, c.keypoint_threshold, c.remove_borders, s, nms_scratch));
  else RUN("nms", launch_nms(smap, nms, B, H8, W8, c.nms_radius, s, nms_scratch));

  const int Ksel = c.max_keypoints >= 0 ? (c.max_keypoints > 0 ? c.max_keypoints : 1) : H8 * W8;
  KeypointArgs k{};
  k.nms = kp_bits ? nullptr : nms; k.cand_bits = kp_bits ? nms_scratch : nullptr; k.scores = smap; k.B = B; k.H = H8; k.W = W8; k.threshold = c.keypoint_threshold; k.border = c.remove_borders;
  k.max_keypoints = c.max_keypoints; k.Ksel = Ksel;
  WS(row_count, int, "kp.row_count", (size_t)B * H8 * 4);
  WS(row_off, int, "kp.row_off", (size_t)B * H8 * 4);
  WS(cand_count, int, "kp.cand_count", (size_t)B * 4);
  WS(cand_idx, int, "kp.cand_idx", (size_t)B * H8 * W8 * 4);
  WS(cand_score, float, "kp.cand_score", (size_t)B * H8 * W8 * 4);
  WS(sel_count, int, "kp.sel_count", (size_t)B * 4);
  WS(sel_idx, int, "kp.sel_idx", (size_t)B * Ksel * 4);
  WS(sel_score, float, "kp.sel_score", (size_t)B * Ksel * 4);
  k.row_count = row_count; k.row_off = row_off; k.cand_count = cand_count; k.cand_idx = cand_idx; k.cand_score = cand_score;
  k.sel_count = sel_count; k.sel_idx = sel_idx; k.sel_score = sel_score;
  k.counts_out[0] = counts_out; k.counts_out[1] = counts_lo; k.counts_out[2] = counts_hi; k.counts_split = split;   // written by kp_topk itself
  if (c.max_keypoints > 16384) {      // beyond the LDS sort of kp_topk: its sort slots live in HBM (slow, but nothing is refused)
    size_t P = 1;
    while (P < (size_t)c.max_keypoints) P <<= 1;
    WS(slots, unsigned long long, "kp.sort_slots", (size_t)B * P * 8);
    k.sort_scratch = slots;
  }
  RUN("keypoints", launch_keypoints(k, s));

  h->det_B = B; h->det_H = H; h->det_W = W; h->det_Hc = Hc; h->det_Wc = Wc; h->det_Ksel = Ksel;
  tap(h, "a1", a1, {B, H2, W2, 64}, blocked);
  tap(h, "a2", a2, {B, H4, W4, 64}, blocked);
  tap(h, "a3", a3, {B, Hc, Wc, 128}, blocked);
  tap(h, "x4", x4, {B, Hc, Wc, 128}, blocked);
  tap(h, "semi", semi, {B, Hc, Wc, 65});
  tap(h, "desc_raw", dense, {B, Hc, Wc, d});
  tap(h, "score_map", smap, {B, H8, W8});
  // the "nms" tap: the dense map is materialised only if somebody fetches it (after the keypoints: the pass overwrites the bit rows)
  h->nms_lazy.smap = smap; h->nms_lazy.nms = nms; h->nms_lazy.scratch = nms_scratch; h->nms_lazy.B = B; h->nms_lazy.H = H8; h->nms_lazy.W = W8;
  h->nms_lazy.radius = c.nms_radius; h->nms_lazy.pending = kp_bits;
  tap(h, "nms", nms, {B, H8, W8});
  return 0;
}

int sp_describe(imx_handle_t h, int b0, int B, int Kcap, float* kpts, float* scores, float* desc, hipStream_t s,
                int split = -1, float* kpts2 = nullptr, float* scores2 = nullptr, float* desc2 = nullptr) {
  if (h->det_B <= 0 || b0 + B > h->det_B) return fail(h, "describe: no matching imx_superpoint_detect (B=%d, detect B=%d)", B, h->det_B);
  DescribeArgs a{};
  a.dense = static_cast<const float*>(h->bufs["sp.dense"].p);
  a.ld = h->cfg.descriptor_dim; a.d = h->cfg.descriptor_dim; a.Hc = h->det_Hc; a.Wc = h->det_Wc; a.W8 = h->det_Wc * 8;
  a.sel_count = static_cast<const int*>(h->bufs["kp.sel_count"].p);
  a.sel_idx = static_cast<const int*>(h->bufs["kp.sel_idx"].p);
  a.sel_score = static_cast<const float*>(h->bufs["kp.sel_score"].p);
  a.Ksel = h->det_Ksel; a.b0 = b0; a.B = B; a.Kcap = Kcap; a.kpts = kpts; a.scores = scores; a.desc = desc;
  a.split = split >= 0 ? split : B; a.kpts2 = kpts2; a.scores2 = scores2; a.desc2 = desc2;
  a.align_corners = h->cfg.align_corners; a.dense_eps = h->cfg.sp_variant == IMX_SP_VARIANT_OFFICIAL ? 1 : 0;
  RUN("describe", launch_describe(a, s));
  return 0;
}

// ----------------------------------------------------------------------------- SuperGlue
struct SgSide {
  const float* kpts; const float* scores; const float* desc;
  int64_t sb, sc, sn;
  const int32_t* n; int N, H, W;
};

int sg_forward(imx_handle_t h, int B, const SgSide sd[2], int64_t* m0, int64_t* m1, float* ms0, float* ms1, hipStream_t s) {
  if (!h->finalized[IMX_NET_SUPERGLUE]) return fail(h, "SuperGlue weights not finalized");
  const imx_config_t& c = h->cfg;
  const int d = c.descriptor_dim;
  const int N0 = sd[0].N, N1 = sd[1].N;
  if (B <= 0 || N0 < 0 || N1 < 0) return fail(h, "bad SuperGlue shapes B=%d N0=%d N1=%d", B, N0, N1);
  const size_t f = sizeof(float);
  if (N0 == 0 || N1 == 0) {   // superglue_test.py:235-242 (dtype handling is the Python side's)
    if (N0) { HIP_OK(h, hipMemsetAsync(m0, 0xFF, (size_t)B * N0 * 8, s)); HIP_OK(h, hipMemsetAsync(ms0, 0, (size_t)B * N0 * f, s)); }
    if (N1) { HIP_OK(h, hipMemsetAsync(m1, 0xFF, (size_t)B * N1 * 8, s)); HIP_OK(h, hipMemsetAsync(ms1, 0, (size_t)B * N1 * f, s)); }
    return 0;
  }
  const int N0p = pad32(N0), N1p = pad32(N1);
  const int R = B * (N0p + N1p);
  const size_t off1 = (size_t)B * N0p;   // first row of side 1
  int maxw = d;
  for (int i = 0; i < c.kenc_n; ++i) maxw = std::max(maxw, c.kenc_channels[i]);
  WS(x, float, "sg.x", (size_t)R * d * f);
  WS(ta, float, "sg.ta", (size_t)R * maxw * f);
  WS(tb, float, "sg.tb", (size_t)R * maxw * f);
  WS(qkv, float, "sg.qkv", (size_t)R * 3 * d * f);
  WS(att, float, "sg.att", (size_t)R * d * f);
  WS(hid, float, "sg.hid", (size_t)R * 2 * d * f);
  WS(mdesc, float, "sg.mdesc", (size_t)R * d * f);
  WS(S, float, "sg.S", (size_t)B * N0p * N1p * f);
  WS(uv, float, "sg.uv", ((size_t)B * (N0p + 1) + (size_t)B * (N1p + 1)) * f);     // u then v, contiguous: ONE memset zeroes both
  float* u = uv;
  float* v = uv + (size_t)B * (N0p + 1);
  WS(max0, float, "sg.max0", (size_t)B * N0p * f);
  WS(max1, float, "sg.max1", (size_t)B * N1p * f);
  WS(idx0, int, "sg.idx0", (size_t)B * N0p * 4);
  WS(idx1, int, "sg.idx1", (size_t)B * N1p * 4);

  // descriptors -> rows; first keypoint-encoder layer (superglue_test.py:245-250): both sides, one launch
  {
    SgPrologueArgs pa{};
    pa.d = d;
    for (int sidx = 0; sidx < 2; ++sidx) {
      const SgSide& q = sd[sidx];
      const int Np = sidx ? N1p : N0p;
      pa.desc[sidx] = q.desc; pa.sb[sidx] = (long)q.sb; pa.sc[sidx] = (long)q.sc; pa.sn[sidx] = (long)q.sn;
      pa.xrow[sidx] = x + (sidx ? off1 : 0) * d;
      Kenc0Args& k = pa.k[sidx];
      k.kpts = q.kpts; k.scores = q.scores; k.B = B; k.N = q.N; k.Np = Np;
      k.cx = (float)q.W / 2.0f; k.cy = (float)q.H / 2.0f;
      k.scaling = (float)std::max(q.W, q.H) * 0.7f;
      k.w = h->kenc0_w; k.bias = h->kenc0_b; k.C1 = h->kenc_c1;
      k.out = ta + (sidx ? off1 : 0) * h->kenc_c1;
    }
    RUN("sg_prologue", launch_sg_prologue(pa, s));
  }
  {
    float* cur = ta;
    float* nxt = tb;
    int curw = h->kenc_c1;
    for (size_t i = 0; i < h->kenc.size(); ++i) {
      const GemmW& g = h->kenc[i];
      const bool last = i + 1 == h->kenc.size();
      if (last) {
        if (gemm(h, s, "kenc", g, cur, curw, g.K, nullptr, 0, 0, x, d, x, d, R, false)) return -1;   // desc + kenc(...)
      } else {
        if (gemm(h, s, "kenc", g, cur, curw, g.K, nullptr, 0, 0, nullptr, 0, nxt, g.N, R, true)) return -1;
        std::swap(cur, nxt);
        curw = g.N;
      }
    }
  }
  if (h->debug) {
    WS(tk, float, "tap.kenc", (size_t)R * d * f);
    HIP_OK(h, hipMemcpyAsync(tk, x, (size_t)R * d * f, hipMemcpyDeviceToDevice, s));
    tap(h, "kenc", tk, {R, d});
  }
  // attentional GNN (superglue_test.py:122-138)
  // Latency form (one or two pairs; "latency_forms"): the three products after the attention -- mlp.0', mlp.3 + residual and the
  // NEXT layer's q|k|v (final_proj after the last layer) -- are ONE launch per layer (gnn_small.hip; same arithmetic, bit for bit,
  // as the three gemm_small launches it replaces).
  const bool small_form = h->opt.latency_forms >= 0 ? h->opt.latency_forms != 0 : R <= 4096;
  bool have_next = false, have_mdesc = false, have_amax = false;
  // "attention" = f16x2: the two-plane fp16 form of the throughput attention scales q, k, v by powers of two taken from their maxima
  // over the valid rows of every (side, pair) -- [2 B][4] words per layer, zeroed once per forward; written by the fused layer tail that produces the layer's
  // q|k|v (gnn_tail_x3's epilogue), else by qkv_amax (layer 0, whose q|k|v is a plain GEMM; the unfused A/B forms)
  // (the same buffer carries, behind the q / k / v tables, one word per (layer, side, pair) for max |x|: gnn_tail_h2.hip's bounds)
  unsigned* amax = nullptr;
  unsigned* amax_x = nullptr;
  if (h->opt.attention != 0 && !h->opt.mfma_f32) {
    const size_t nl = h->layers.size(), words = nl * 2 * B * 4 + (nl + 1) * 2 * B;
    WS(am, unsigned, "sg.amax", words * 4);
    HIP_OK(h, hipMemsetAsync(am, 0, words * 4, s));
    amax = am;
    amax_x = am + nl * 2 * B * 4;
  }
  long x_max_layer = -1;       // amax_x + 2 B x_max_layer holds max |x| of the CURRENT x (-1: not computed)
  // ("qkv_amax" = "kernel": the maxima of a projected q|k|v by the separate pass even where the projection's epilogue can write them --
  // the A/B switch of tests/test_gpu_superglue.py; the two must agree bit for bit)
  const bool amax_by_kernel = h->opt.qkv_amax != 0;
  // "linear" = auto / f16x2 (round 6): the plain linear layers of the GNN -- every layer's q|k|v, mlp.0' and mlp.3 where the tail is
  // not fused (descriptor_dim 256: C5), layer 0's q|k|v and final_proj otherwise -- as three fp16 plane products (gemm_h2.hip), each
  // operand scaled by its ACTUAL (side, pair) maximum: max |x| from rows_amax (layer 0) or the producing mlp.3's epilogue, max |v|
  // (which bounds the attention output) from the q|k|v epilogue, max |hidden| from mlp.0's.  Decided for the whole chain: the
  // two-plane attention's tables exist and its kernel runs, the throughput forms apply, every matrix passes the spread guard.
  bool lin_h2 = amax && !small_form && h->opt.linear != 0 && N0p % 128 == 0 && N1p % 128 == 0 && R == B * (N0p + N1p) && gemm_h2_weights_ok(h->final_proj);
  for (const GnnLayer& L : h->layers) lin_h2 = lin_h2 && gemm_h2_weights_ok(L.qkv) && gemm_h2_weights_ok(L.mlp1) && gemm_h2_weights_ok(L.mlp2);
  {
    AttnArgs a{};
    a.B = B; a.N0p = N0p; a.N1p = N1p; a.d = d; a.heads = HEADS; a.mfma_f32 = h->opt.mfma_f32; a.latency_forms = h->opt.latency_forms;
    lin_h2 = lin_h2 && attention_takes_x3(a);
  }
  const size_t nl_ = h->layers.size();
  auto x_max_now = [&](size_t l) -> int {      // max |x| of the rows about to be projected, unless the kernel that produced x left it
    if (x_max_layer != (long)l) RUN("rows_amax", launch_rows_amax_any(x, d, B, N0p, N1p, sd[0].n, sd[1].n, N0, N1, amax_x + (size_t)2 * B * l, s));
    x_max_layer = (long)l;
    return 0;
  };
  for (size_t l = 0; l < h->layers.size(); ++l) {
    const GnnLayer& L = h->layers[l];
    if (!have_next) {
      // (the maxima of this q|k|v, if the two-plane attention will want them, out of the projection's epilogue where it can)
      GemmAmax gam{amax && !amax_by_kernel ? amax + 8 * B * l : nullptr, sd[0].n, sd[1].n, B, N0p, N1p, N0, N1, false};
      if (lin_h2) {
        if (x_max_now(l)) return -1;
        gam.want_h2 = true; gam.sa0 = amax_x + (size_t)2 * B * l;
      }
      if (gemm(h, s, "qkv_proj", L.qkv, x, d, d, nullptr, 0, 0, nullptr, 0, qkv, 3 * d, R, false, &gam)) return -1;
      have_amax = gam.done;
    }
    have_next = false;
    AttnArgs a{};
    a.qkv = qkv; a.out = att; a.B = B; a.N0p = N0p; a.N1p = N1p; a.d = d; a.heads = HEADS;
    a.n0 = sd[0].n; a.n1 = sd[1].n; a.N0 = N0; a.N1 = N1; a.cross = c.gnn_layer_is_cross[l];
    a.mfma_f32 = h->opt.mfma_f32; a.latency_forms = h->opt.latency_forms; a.qblocks = h->opt.attention_qblocks;
    const bool f16x2 = amax && attention_takes_x3(a) && (L.qkv_spread <= kAttnSpreadMax || h->opt.attention == 1);   // ("attention" = f16x2 forces it: the guard's A/B)
    if (f16x2) {
      if (!have_amax) RUN("qkv_amax", launch_qkv_amax(a, amax + 8 * B * l, s));
      a.amax = amax + 8 * B * l;
    }
    have_amax = false;
    RUN("attention", launch_attention(a, s));
    const bool last = l + 1 == h->layers.size();
    const GemmW& nx = last ? h->final_proj : h->layers[l + 1].qkv;
    GnnSmallArgs ga{x, att, L.mlp1.wf, L.mlp1.b, L.mlp2.wf, L.mlp2.b, nx.wf, nx.b, last ? mdesc : qkv, R, d, nx.N};
    // Throughput form: the same three products in one launch on the bf16 pipe (gnn_tail_x3.hip): "gnn_tail" = auto takes it whenever the
    // latency forms do not apply (M > 4096 rows; measured against three gemm_x3 launches: 40 vs 50 us at 8224 rows, 65 vs 86 at 32768,
    // 256 vs 300 at 131072) -- so results do not depend on the batch size under "latency_forms" = off.
    GnnTailArgs ta{x, att, L.tail_stream, L.mlp1.b, L.mlp2.b, nx.b, last ? mdesc : qkv, R, d, nx.N};
    if (f16x2 && !last) {              // the next layer's attention takes the same form (same shapes): its maxima come out of this tail
      ta.amax = amax + 8 * B * (l + 1);
      ta.n0 = sd[0].n; ta.n1 = sd[1].n; ta.B = B; ta.N0p = N0p; ta.N1p = N1p; ta.N0 = N0; ta.N1 = N1;
    }
    const bool tail_ok = !small_form && !h->opt.mfma_f32 && L.tail_stream && nx.Npad == nx.N && gnn_tail_x3_supported(ta);
    const bool tail = tail_ok && h->opt.gnn_tail != 0;
    // "gnn_tail" = auto / fused: the same launch as three fp16 plane products (gnn_tail_h2.hip) where the two-plane attention runs (its
    // v maxima bound att) -- the maxima of x come from the previous layer's tail, for layer 0 from rows_amax
    bool tail_h2 = false;
    // ("auto": only where the bounds that scale the operands are tight enough for both fp16 planes -- L.h2c.loose_*, computed from the
    // weights at imx_finalize_weights; "fused" forces the fp16 form, "bf16x3" the other)
    const bool h2_safe = L.h2c.loose_h <= kTailLooseMax && L.h2c.loose_x <= kTailLooseMax && L.h2c.w_spread <= kLinearSpreadMax;
    if (tail && f16x2 && h->opt.gnn_tail != 2 && L.tail_stream_h2 && (h2_safe || h->opt.gnn_tail == 1)) {
      ta.stream_h2 = L.tail_stream_h2;
      ta.w1_inv = L.h2c.w1_inv; ta.w2_inv = L.h2c.w2_inv; ta.w3_inv = L.h2c.w3_inv;
      ta.l1_1 = L.h2c.l1_1; ta.l1_2 = L.h2c.l1_2; ta.bmax_1 = L.bmax_1; ta.bmax_2 = L.bmax_2;
      ta.amax_x_in = amax_x + (size_t)2 * B * l; ta.amax_v = amax + 8 * B * l; ta.amax_x_out = last ? nullptr : amax_x + (size_t)2 * B * (l + 1);
      ta.cross = c.gnn_layer_is_cross[l];
      ta.n0 = sd[0].n; ta.n1 = sd[1].n; ta.B = B; ta.N0p = N0p; ta.N1p = N1p; ta.N0 = N0; ta.N1 = N1;
      tail_h2 = gnn_tail_h2_supported(ta);
      if (tail_h2 && x_max_now(l)) return -1;
    }
    if (small_form && h->opt.latency_forms != 2 && L.mlp1.Npad == 2 * d && L.mlp2.Npad == d && nx.Npad == nx.N && gnn_layer_small_supported(ga)) {
      RUN("gnn_layer", launch_gnn_layer_small(ga, s));
      have_next = !last;
      have_mdesc = last;
      x_max_layer = -1;
    } else if (tail) {
      RUN("gnn_tail", tail_h2 ? launch_gnn_tail_h2(ta, s) : launch_gnn_tail_x3(ta, s));
      have_next = !last;
      have_mdesc = last;
      have_amax = ta.amax != nullptr;
      x_max_layer = tail_h2 && !last ? (long)l + 1 : -1;     // (the tail's epilogue leaves max |x'| for the next layer)
    } else {
      GemmAmax g1{nullptr, sd[0].n, sd[1].n, B, N0p, N1p, N0, N1, false}, g2 = g1;
      if (lin_h2) {                      // (x's maximum: this layer's q|k|v projection had it; v's: the attention's table)
        if (x_max_now(l)) return -1;
        g1.want_h2 = true; g1.sa0 = amax_x + (size_t)2 * B * l;
        g1.sa1 = amax + 8 * B * l; g1.sa1_stride = 4; g1.sa1_off = 2; g1.sa1_cross = c.gnn_layer_is_cross[l] ? 1 : 0;
        g1.amax_row = amax + 8 * B * l; g1.amax_row_stride = 4; g1.amax_row_off = 3;               // max |hidden|: the table's fourth word
        g2.want_h2 = true; g2.sa0 = amax + 8 * B * l; g2.sa0_stride = 4; g2.sa0_off = 3;
        g2.amax_row = amax_x + (size_t)2 * B * (l + 1);                                              // max |x'|: the next projection's scale
      }
      if (gemm(h, s, "gnn_mlp1", L.mlp1, x, d, d, att, d, d, nullptr, 0, hid, 2 * d, R, true, &g1)) return -1;   // merge folded in
      if (gemm(h, s, "gnn_mlp2", L.mlp2, hid, 2 * d, 2 * d, nullptr, 0, 0, x, d, x, d, R, false, &g2)) return -1;
      x_max_layer = lin_h2 ? (long)l + 1 : -1;
    }
    if (h->debug) {
      std::string nm = "gnn" + std::to_string(l);
      WS(tg, float, "tap." + nm, (size_t)R * d * f);
      HIP_OK(h, hipMemcpyAsync(tg, x, (size_t)R * d * f, hipMemcpyDeviceToDevice, s));
      tap(h, nm.c_str(), tg, {R, d});
    }
  }
  if (!have_mdesc) {
    GemmAmax gf{nullptr, sd[0].n, sd[1].n, B, N0p, N1p, N0, N1, false};
    if (lin_h2) {
      if (x_max_now(nl_)) return -1;
      gf.want_h2 = true; gf.sa0 = amax_x + (size_t)2 * B * nl_;
    }
    if (gemm(h, s, "final_proj", h->final_proj, x, d, d, nullptr, 0, 0, nullptr, 0, mdesc, d, R, false, &gf)) return -1;
  }
  ScoreArgs sc{mdesc, mdesc + off1 * d, S, B, N0p, N1p, d, (float)(1.0 / std::sqrt((double)d))};
  RUN("score_gemm", launch_score_gemm(sc, s));
  float* part = nullptr;
  if (const int Rs = sinkhorn_slab_rows(N1p)) {
    WS(pt, float, "sg.part", (size_t)B * (N0p / Rs + 1) * (N1p + 1) * 2 * f);
    part = pt;
  }
  SinkhornArgs sk{S, u, v, B, N0p, N1p, sd[0].n, sd[1].n, N0, N1, h->bin_score, c.sinkhorn_iterations, part, h->opt.sinkhorn_group, h->opt.sinkhorn_prefetch};
  if (part && h->opt.sinkhorn_merge > 0) {       // "sinkhorn_merge" = fused: the slab kernel merges its own column partials (auto = kernel: measured, sg_misc.hip)
    WS(mc, unsigned, "sg.sk_merge_cnt", ((size_t)B + 1) * sizeof(unsigned));
    sk.merge_cnt = mc;
    tap(h, "sk_merge_cnt", mc, {(int64_t)B + 1});          // (word [B] != 0: a merging workgroup gave up waiting -- never seen; the tests read it)
  }
  if (h->debug && part) {        // developer instrumentation: the slab kernel's workgroup lives (all zeros unless sg_misc.hip was built with -DSK_TRACE)
    const int Rs = sinkhorn_slab_rows(N1p), ng = N0p / Rs + 1;
    WS(trc, unsigned long long, "sg.sk_trace", (size_t)B * ng * 8 * sizeof(unsigned long long));
    HIP_OK(h, hipMemsetAsync(trc, 0, (size_t)B * ng * 8 * sizeof(unsigned long long), s));
    sk.trace = trc;
    tap(h, "sk_trace", trc, {(int64_t)B * ng, 16});
  }
  RUN("sinkhorn", launch_sinkhorn(sk, s));
  MatchArgs ma{S, u, v, B, N0p, N1p, sd[0].n, sd[1].n, N0, N1, h->bin_score, c.match_threshold,
               max0, idx0, max1, idx1, m0, m1, ms0, ms1};
  RUN("matches", launch_matches(ma, s));
  tap(h, "x", x, {R, d});
  tap(h, "qkv", qkv, {R, 3 * d});                                           // the LAST layer's q|k|v ...
  if (amax) tap(h, "amax", amax + 8 * (size_t)B * (h->layers.size() - 1), {2 * B, 4});      // ... and its maxima (bit patterns; fetched as floats)
  tap(h, "mdesc", mdesc, {R, d});
  tap(h, "scores_in", S, {B, N0p, N1p});
  tap(h, "u", u, {B, N0p + 1});
  tap(h, "v", v, {B, N1p + 1});
  tap(h, "max0", max0, {B, N0p});
  tap(h, "max1", max1, {B, N1p});
  return 0;
}

// "mfma" = x3 | f32, "latency_forms" = auto | off | on | unfused, "conv" = wino | wino_h | wino32 | direct, "gnn_tail" = auto | fused | bf16x3 | unfused, "attention" = auto | f16x2 | bf16x3,
// "linear" = auto | f16x2 | bf16x3;
// the A/B switches "conv_swizzle" = on | off, "qkv_amax" = epilogue | kernel, "sinkhorn_group" = auto | 1 | 2 | 4, "sinkhorn_prefetch" = auto | off | on.  Returns 0, or -1 for an unknown key / value.
int apply_option(imx_handle_t h, const std::string& key, const std::string& v) {
  Options& o = h->opt;
  if (key == "mfma") {
    if (v == "x3") o.mfma_f32 = 0; else if (v == "f32") o.mfma_f32 = 1; else return -1;
  } else if (key == "latency_forms") {
    if (v == "auto") o.latency_forms = -1; else if (v == "off" || v == "0") o.latency_forms = 0; else if (v == "on" || v == "1") o.latency_forms = 1; else if (v == "unfused") o.latency_forms = 2; else return -1;
  } else if (key == "gnn_tail") {
    if (v == "auto") o.gnn_tail = -1; else if (v == "unfused" || v == "0") o.gnn_tail = 0; else if (v == "fused" || v == "1") o.gnn_tail = 1;
    else if (v == "bf16x3") o.gnn_tail = 2; else return -1;
  } else if (key == "attention") {
    if (v == "auto") o.attention = -1; else if (v == "bf16x3" || v == "x3" || v == "0") o.attention = 0; else if (v == "f16x2" || v == "1") o.attention = 1; else return -1;
  } else if (key == "conv") {
    if (v == "wino") { o.conv_direct = 0; o.conv_f16 = 1; }
    else if (v == "wino_h") { o.conv_direct = 0; o.conv_f16 = 2; }
    else if (v == "wino32") { o.conv_direct = 0; o.conv_f16 = 0; }
    else if (v == "wx3") {             // round 3's bf16-plane experiment, deleted in round 4: its nearest living form
      fprintf(stderr, "imx: conv = wx3 was removed (round 4); using wino32\n");
      o.conv_direct = 0; o.conv_f16 = 0;
    }
    else if (v == "direct") o.conv_direct = 1;
    else return -1;
  } else if (key == "attention_qblocks") {
    if (v == "auto") o.attention_qblocks = -1; else if (v == "1") o.attention_qblocks = 1; else if (v == "2") o.attention_qblocks = 2; else return -1;
  } else if (key == "linear") {
    if (v == "auto") o.linear = -1; else if (v == "f16x2" || v == "1") o.linear = 1; else if (v == "bf16x3" || v == "x3" || v == "0") o.linear = 0; else return -1;
  } else if (key == "conv_swizzle") {
    if (v == "on" || v == "1") o.conv_swizzle = 1; else if (v == "off" || v == "0") o.conv_swizzle = 0; else return -1;
  } else if (key == "qkv_amax") {
    if (v == "epilogue") o.qkv_amax = 0; else if (v == "kernel") o.qkv_amax = 1; else return -1;
  } else if (key == "sinkhorn_group") {
    if (v == "auto" || v == "0") o.sinkhorn_group = 0; else if (v == "1" || v == "2" || v == "4") o.sinkhorn_group = v[0] - '0'; else return -1;
  } else if (key == "sinkhorn_merge") {
    if (v == "auto") o.sinkhorn_merge = -1; else if (v == "kernel") o.sinkhorn_merge = 0; else if (v == "fused") o.sinkhorn_merge = 1; else return -1;
  } else if (key == "keypoints") {
    if (v == "auto") o.keypoints = -1; else if (v == "dense") o.keypoints = 0; else if (v == "bits") o.keypoints = 1; else return -1;
  } else if (key == "sinkhorn_prefetch") {
    if (v == "auto") o.sinkhorn_prefetch = -1; else if (v == "off" || v == "0") o.sinkhorn_prefetch = 0; else if (v == "on" || v == "1") o.sinkhorn_prefetch = 1; else return -1;
  } else {
    return -1;
  }
  return 0;
}

// Nothing may throw across the C ABI (include/imx.h): every entry point runs inside this guard.  The handle owns
// std::string / std::map / std::vector state, so std::bad_alloc (or any other exception) is possible in principle;
// it becomes an error code + imx_last_error text.  Recording the text must not throw either.
int fail_nothrow(imx_handle_t h, const char* where, const char* what) noexcept {
  try {
    return fail(h, "%s: exception: %s", where, what);
  } catch (...) {
    return -1;
  }
}
template <class F>
int guarded(imx_handle_t h, const char* where, F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail_nothrow(h, where, "out of host memory (std::bad_alloc)");
  } catch (const std::exception& e) {
    return fail_nothrow(h, where, e.what());
  } catch (...) {
    return fail_nothrow(h, where, "unknown C++ exception");
  }
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

#ifndef IMX_BUILD_ID
#define IMX_BUILD_ID "dev"
#endif
const char* imx_version(void) { return "imx 0.4 gfx950 hip-7.2 fp32 build " IMX_BUILD_ID; }

int imx_create(int device_id, const imx_config_t* cfg, imx_handle_t* out) {
  return guarded(nullptr, "imx_create", [&]() -> int {
    if (!cfg || !out) return fail(nullptr, "imx_create: null argument");
    if (cfg->num_gnn_layers < 0 || cfg->num_gnn_layers > IMX_MAX_GNN_LAYERS) return fail(nullptr, "imx_create: bad num_gnn_layers %d", cfg->num_gnn_layers);
    if (cfg->kenc_n < 1 || cfg->kenc_n > IMX_MAX_KENC) return fail(nullptr, "imx_create: bad keypoint_encoder length %d", cfg->kenc_n);
    // SuperPoint alone takes any descriptor_dim that keeps rows float4-aligned; SuperGlue's own limits (4 heads of 16/32/64
    // dims) are checked when ITS weights are finalized
    if (cfg->descriptor_dim <= 0 || cfg->descriptor_dim % 4 || cfg->descriptor_dim > 512) return fail(nullptr, "imx_create: descriptor_dim must be a multiple of 4 in [4,512] (got %d)", cfg->descriptor_dim);
    if (cfg->nms_radius < 0) return fail(nullptr, "imx_create: nms_radius must be >= 0 (got %d)", cfg->nms_radius);
    // the top-k stage rounds max_keypoints up to a power of two in 32-bit arithmetic (sp_tail.hip: launch_keypoints)
    if (cfg->max_keypoints > (1 << 30)) return fail(nullptr, "imx_create: max_keypoints must be <= 2^30 (got %d); use -1 for 'all'", cfg->max_keypoints);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "imx_create: no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, "imx_create: device %d out of range (%d devices)", device_id, ndev);
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, "imx_create: hipSetDevice(%d) failed", device_id);
    std::unique_ptr<imx_handle_s> h(new imx_handle_s());      // released only once nothing below can throw
    h->device = device_id;
    h->cfg = *cfg;
    // the environment seeds the options once, here; afterwards only imx_set_option changes them
    for (const char* key : {"mfma", "latency_forms", "conv", "gnn_tail", "attention", "attention_qblocks", "linear"}) {
      std::string env = std::string("IMX_") + key;
      for (char& ch : env) ch = (char)toupper((unsigned char)ch);
      if (const char* e = getenv(env.c_str()))
        if (apply_option(h.get(), key, e)) return fail(nullptr, "imx_create: bad value '%s' in the environment variable %s", e, env.c_str());
    }
    // switches of earlier rounds that no longer exist: say so once instead of silently measuring the default path
    static std::atomic<bool> warned{false};
    for (const char* old : {"IMX_GEMM", "IMX_GEMM_SMALL", "IMX_ATTN", "IMX_ATTN_SPLIT", "IMX_NMS", "IMX_CONV_BLOCKED", "IMX_WINO_EXP", "IMX_WINO_WGS",
                            "IMX_X3_WGS", "IMX_SINKHORN_WAVES", "IMX_CONV_SWZ", "IMX_QKV_AMAX", "IMX_SINKHORN_GROUP", "IMX_SINKHORN_PREFETCH", "IMX_SINKHORN_PREFETCH_NOW"})
      if (getenv(old) && !warned.exchange(true))
        fprintf(stderr, "libimx: the environment variable %s (and the other per-kernel switches of rounds 1-5) was removed; it is ignored. "
                        "Kernel forms are handle options now: imx_set_option(h, \"mfma\" | \"latency_forms\" | \"conv\", ...), seeded from IMX_MFMA / "
                        "IMX_LATENCY_FORMS / IMX_CONV at imx_create (an unknown VALUE of those three makes imx_create fail).\n", old);
    build_expected(h.get());
    *out = h.release();
    return 0;
  });
}

int imx_destroy(imx_handle_t h) {
  return guarded(h, "imx_destroy", [&]() -> int {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto& va : h->weight_allocs) for (void* p : va) (void)hipFree(p);
    for (auto& kv : h->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& e : h->events) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    delete h;
    return 0;
  });
}

const char* imx_last_error(imx_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }   // c_str() does not throw

int imx_load_weight(imx_handle_t h, int net, const char* name, const float* host, int ndim, const int64_t* shape) {
  return guarded(h, "imx_load_weight", [&]() -> int {
    if (!h) return -1;
    if (net != IMX_NET_SUPERPOINT && net != IMX_NET_SUPERGLUE) return fail(h, "imx_load_weight: bad net id %d", net);
    if (!name || !host) return fail(h, "imx_load_weight: null argument");
    std::string key = name;
    if (ends_with(key, "num_batches_tracked")) return 0;
    auto it = h->expected[net].find(key);
    if (it == h->expected[net].end()) return fail(h, "imx_load_weight: unexpected key '%s' for %s", name, net ? "SuperGlue" : "SuperPoint");
    int64_t want = 1, got = 1;
    for (int64_t v : it->second) want *= v;
    for (int i = 0; i < ndim; ++i) got *= shape[i];
    bool same = (size_t)ndim == it->second.size();
    for (int i = 0; same && i < ndim; ++i) same = shape[i] == it->second[i];
    if (!same && !(want == got && want == 1)) {
      std::string ws_, gs_;
      for (int64_t v : it->second) ws_ += std::to_string(v) + ",";
      for (int i = 0; i < ndim; ++i) gs_ += std::to_string(shape[i]) + ",";
      return fail(h, "imx_load_weight: size mismatch for %s: expected (%s) got (%s)", name, ws_.c_str(), gs_.c_str());
    }
    HostTensor t;
    t.data.assign(host, host + got);
    t.shape.assign(shape, shape + ndim);
    h->raw[net][key] = std::move(t);
    h->finalized[net] = false;
    return 0;
  });
}

int imx_finalize_weights(imx_handle_t h, int net) {
  return guarded(h, "imx_finalize_weights", [&]() -> int {
    if (!h) return -1;
    if (net != IMX_NET_SUPERPOINT && net != IMX_NET_SUPERGLUE) return fail(h, "imx_finalize_weights: bad net id %d", net);
    HIP_OK(h, hipSetDevice(h->device));
    for (auto& kv : h->expected[net])
      if (!h->raw[net].count(kv.first)) return fail(h, "Missing key in state_dict: \"%s\"", kv.first.c_str());
    // a second load_state_dict (checkpoint sweeps): the previous uploads of THIS net are released first
    if (!h->weight_allocs[net].empty()) {
      HIP_OK(h, hipDeviceSynchronize());
      for (void* p : h->weight_allocs[net]) (void)hipFree(p);
      h->weight_allocs[net].clear();
    }
    h->finalized[net] = false;
    h->upload_net = net;
    int rc = net == IMX_NET_SUPERPOINT ? finalize_superpoint(h) : finalize_superglue(h);
    if (rc) return rc;
    h->finalized[net] = true;
    return 0;
  });
}

int imx_superpoint_detect(imx_handle_t h, const float* img_dev, int B, int H, int W, int32_t* counts_dev, void* stream) {
  return guarded(h, "imx_superpoint_detect", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    return sp_detect(h, img_dev, nullptr, B, B, H, W, counts_dev, as_stream(stream));
  });
}

int imx_superpoint_dense(imx_handle_t h, const float* img_dev, int B, int H, int W, float* semi_dev, float* desc_dev, void* stream) {
  return guarded(h, "imx_superpoint_dense", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    if (sp_detect(h, img_dev, nullptr, B, B, H, W, nullptr, s, true)) return -1;
    RUN("dense_export", launch_dense_export(static_cast<const float*>(h->bufs["sp.semi"].p), 65,
                                            static_cast<const float*>(h->bufs["sp.dense"].p), h->cfg.descriptor_dim, semi_dev,
                                            desc_dev, B, h->det_Hc, h->det_Wc, h->cfg.sp_variant == IMX_SP_VARIANT_OFFICIAL, s));
    return 0;
  });
}

int imx_superpoint_describe(imx_handle_t h, int B, int Kcap, float* kpts_dev, float* scores_dev, float* desc_dev, void* stream) {
  return guarded(h, "imx_superpoint_describe", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    if (Kcap <= 0) return 0;
    return sp_describe(h, 0, B, Kcap, kpts_dev, scores_dev, desc_dev, as_stream(stream));
  });
}

int imx_superglue_forward(imx_handle_t h, int B, const float* kpts0_dev, const float* scores0_dev, const float* desc0_dev,
                          int64_t desc0_stride_b, int64_t desc0_stride_c, int64_t desc0_stride_n, const int32_t* n0_dev,
                          int N0, int H0, int W0, const float* kpts1_dev, const float* scores1_dev, const float* desc1_dev,
                          int64_t desc1_stride_b, int64_t desc1_stride_c, int64_t desc1_stride_n, const int32_t* n1_dev,
                          int N1, int H1, int W1, int64_t* matches0_dev, int64_t* matches1_dev, float* mscores0_dev,
                          float* mscores1_dev, void* stream) {
  return guarded(h, "imx_superglue_forward", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    SgSide sd[2] = {{kpts0_dev, scores0_dev, desc0_dev, desc0_stride_b, desc0_stride_c, desc0_stride_n, n0_dev, N0, H0, W0},
                    {kpts1_dev, scores1_dev, desc1_dev, desc1_stride_b, desc1_stride_c, desc1_stride_n, n1_dev, N1, H1, W1}};
    return sg_forward(h, B, sd, matches0_dev, matches1_dev, mscores0_dev, mscores1_dev, as_stream(stream));
  });
}

int imx_match_pairs(imx_handle_t h, const float* img0_dev, const float* img1_dev, int B, int H, int W, float* kpts0_dev,
                    float* kpts1_dev, float* scores0_dev, float* scores1_dev, int32_t* counts0_dev, int32_t* counts1_dev,
                    float* desc0_dev, float* desc1_dev, int64_t* matches0_dev, int64_t* matches1_dev, float* mscores0_dev,
                    float* mscores1_dev, void* stream) {
  return guarded(h, "imx_match_pairs", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    const int K = h->cfg.max_keypoints, d = h->cfg.descriptor_dim;
    if (K <= 0) return fail(h, "imx_match_pairs needs max_keypoints > 0 (fixed-size outputs); got %d", K);
    WS(counts, int32_t, "mp.counts", (size_t)2 * B * 4);
    if (sp_detect(h, img0_dev, img1_dev, B, 2 * B, H, W, counts, s, false, counts0_dev, counts1_dev)) return -1;
    if (!desc0_dev) { WS(t0, float, "mp.desc0", (size_t)B * K * d * 4); desc0_dev = t0; }
    if (!desc1_dev) { WS(t1, float, "mp.desc1", (size_t)B * K * d * 4); desc1_dev = t1; }
    if (sp_describe(h, 0, 2 * B, K, kpts0_dev, scores0_dev, desc0_dev, s, B, kpts1_dev, scores1_dev, desc1_dev)) return -1;   // both sides, one launch
    const int H8 = h->det_Hc * 8, W8 = h->det_Wc * 8;
    (void)H8; (void)W8;
    SgSide sd[2] = {{kpts0_dev, scores0_dev, desc0_dev, (int64_t)K * d, 1, d, counts, K, H, W},
                    {kpts1_dev, scores1_dev, desc1_dev, (int64_t)K * d, 1, d, counts + B, K, H, W}};
    return sg_forward(h, B, sd, matches0_dev, matches1_dev, mscores0_dev, mscores1_dev, s);
  });
}

int imx_pack_records(imx_handle_t h, const int32_t* pair_ids_dev, int B, int K, const float* kpts0_dev, const float* kpts1_dev,
                     const int32_t* counts0_dev, const int32_t* counts1_dev, const int64_t* matches0_dev, const int64_t* matches1_dev,
                     const float* mscores0_dev, const float* mscores1_dev, int32_t* rec_dev, int rows, void* stream) {
  return guarded(h, "imx_pack_records", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    if (B < 0 || K <= 0 || rows < B) return fail(h, "imx_pack_records: bad shape B=%d K=%d rows=%d", B, K, rows);
    if (B > 0 && (!pair_ids_dev || !kpts0_dev || !kpts1_dev || !counts0_dev || !counts1_dev || !matches0_dev || !matches1_dev ||
                  !mscores0_dev || !mscores1_dev)) return fail(h, "imx_pack_records: null argument");
    if (!rec_dev) return fail(h, "imx_pack_records: null record buffer");
    PackArgs a{pair_ids_dev, kpts0_dev, kpts1_dev, counts0_dev, counts1_dev, reinterpret_cast<const long long*>(matches0_dev),
               reinterpret_cast<const long long*>(matches1_dev), mscores0_dev, mscores1_dev, rec_dev, B, K, rows};
    RUN("pack_records", launch_pack_records(a, s));
    return 0;
  });
}

// RCCL entry points, resolved at the first call from the RCCL already in the process (a torch process has its own copy loaded;
// a C host links one), else from librccl.so: libimx.so itself carries no link-time dependency on a particular RCCL build.
extern "C++" {
namespace {
struct Rccl {
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const Rccl& rccl() {
  static Rccl r = [] {
    Rccl t;
    void* lib = nullptr;
    auto sym = [&](const char* name) -> void* {
      if (void* p = dlsym(RTLD_DEFAULT, name)) return p;
      if (!lib) {
        // the host names the RCCL build that created its communicators with IMX_RCCL_LIBRARY (a path) when that library is not
        // visible to dlsym(RTLD_DEFAULT) -- e.g. loaded RTLD_LOCAL under another soname; otherwise the usual names are tried
        if (const char* path = getenv("IMX_RCCL_LIBRARY")) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        for (const char* so : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
          if (!lib && (lib = dlopen(so, RTLD_NOW | RTLD_GLOBAL))) break;
      }
      return lib ? dlsym(lib, name) : nullptr;
    };
    t.CommCount = reinterpret_cast<decltype(t.CommCount)>(sym("ncclCommCount"));
    t.CommUserRank = reinterpret_cast<decltype(t.CommUserRank)>(sym("ncclCommUserRank"));
    t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(sym("ncclGroupStart"));
    t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(sym("ncclGroupEnd"));
    t.Send = reinterpret_cast<decltype(t.Send)>(sym("ncclSend"));
    t.Recv = reinterpret_cast<decltype(t.Recv)>(sym("ncclRecv"));
    t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(sym("ncclGetErrorString"));
    t.ok = t.CommCount && t.CommUserRank && t.GroupStart && t.GroupEnd && t.Send && t.Recv;
    return t;
  }();
  return r;
}
}  // namespace
}  // extern "C++"

int imx_gather_records(imx_handle_t h, const int32_t* rec_dev, int rows, int width, int32_t* out_dev, int dst, void* nccl_comm, void* stream) {
  return guarded(h, "imx_gather_records", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    if (!nccl_comm || !rec_dev || rows < 0 || width <= 0) return fail(h, "imx_gather_records: bad arguments");
    const Rccl& n = rccl();
    if (!n.ok) return fail(h, "imx_gather_records: no RCCL in this process and librccl.so could not be loaded");
    int world = 0, rank = 0, rc = 0;
    auto chk = [&](int e, const char* what) { if (e && !rc) { rc = e; fail(h, "imx_gather_records: %s failed: %s", what, n.GetErrorString ? n.GetErrorString(e) : "RCCL error"); } };
    chk(n.CommCount(nccl_comm, &world), "ncclCommCount");
    chk(n.CommUserRank(nccl_comm, &rank), "ncclCommUserRank");
    if (rc) return -1;
    if (dst < 0 || dst >= world) return fail(h, "imx_gather_records: destination rank %d outside the communicator (%d ranks)", dst, world);
    if (rank == dst && !out_dev) return fail(h, "imx_gather_records: the destination rank needs an output buffer");
    const size_t count = (size_t)rows * width;
    constexpr int kInt32 = 2;                       // ncclInt32
    hipStream_t s = as_stream(stream);
    // one group: every rank sends its rows to `dst`; `dst` posts one receive per rank into its slot (rank order = gather order)
    chk(n.GroupStart(), "ncclGroupStart");
    if (count) {
      chk(n.Send(rec_dev, count, kInt32, dst, nccl_comm, s), "ncclSend");
      if (rank == dst)
        for (int r = 0; r < world; ++r) chk(n.Recv(out_dev + (size_t)r * count, count, kInt32, r, nccl_comm, s), "ncclRecv");
    }
    chk(n.GroupEnd(), "ncclGroupEnd");
    return rc ? -1 : 0;
  });
}

int imx_op_nms(imx_handle_t h, const float* scores_dev, float* out_dev, int B, int H, int W, int radius, void* stream) {
  return guarded(h, "imx_op_nms", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    if (radius < 0) return fail(h, "imx_op_nms: radius must be >= 0 (got %d)", radius);
    WS(nms_scratch, unsigned, "op.nms_scratch", nms_scratch_bytes(B, H, W, radius));
    RUN("nms", launch_nms(scores_dev, out_dev, B, H, W, radius, s, nms_scratch));
    return 0;
  });
}

int imx_estimate_affine_partial(imx_handle_t h, const float* kpts0_dev, const float* kpts1_dev, const int64_t* matches0_dev,
                                const int32_t* counts0_dev, int B, int K, float ransac_threshold, int hypotheses, uint32_t seed,
                                float* M_dev, uint8_t* inlier_dev, int32_t* n_inliers_dev, void* stream) {
  return guarded(h, "imx_estimate_affine_partial", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    if (B <= 0 || K <= 0) return fail(h, "imx_estimate_affine_partial: bad shape B=%d K=%d", B, K);
    float* scratch = nullptr;
    if (K > 8192) {      // the compacted coordinates no longer fit LDS: they live in HBM (slower; nothing is refused)
      WS(rs, float, "ransac.scratch", (size_t)B * 4 * K * sizeof(float));
      scratch = rs;
    }
    RansacArgs a{kpts0_dev, kpts1_dev, reinterpret_cast<const long long*>(matches0_dev), counts0_dev, B, K, ransac_threshold,
                 hypotheses, seed, M_dev, inlier_dev, n_inliers_dev, scratch};
    RUN("ransac", launch_ransac(a, s));
    return 0;
  });
}

int imx_knn_ratio_match(imx_handle_t h, int B, const float* desc0_dev, int64_t s0b, int64_t s0c, int64_t s0n, const int32_t* n0_dev,
                        int N0, const float* desc1_dev, int64_t s1b, int64_t s1c, int64_t s1n, const int32_t* n1_dev, int N1,
                        float ratio, int64_t* matches_dev, float* dist1_dev, float* dist2_dev, void* stream) {
  return guarded(h, "imx_knn_ratio_match", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    hipStream_t s = as_stream(stream);
    const int d = h->cfg.descriptor_dim;
    if (B <= 0 || N0 <= 0 || N1 < 0) return fail(h, "imx_knn_ratio_match: bad shape B=%d N0=%d N1=%d", B, N0, N1);
    const int N0p = pad32(N0), N1p = pad32(std::max(N1, 1));
    const size_t f = sizeof(float);
    WS(r0, float, "knn.rows0", (size_t)B * N0p * d * f);
    WS(r1, float, "knn.rows1", (size_t)B * N1p * d * f);
    WS(nr0, float, "knn.norm0", (size_t)B * N0p * f);
    WS(nr1, float, "knn.norm1", (size_t)B * N1p * f);
    WS(dots, float, "knn.dots", (size_t)B * N0p * N1p * f);
    RUN("gather_desc", launch_gather_desc(desc0_dev, s0b, s0c, s0n, B, N0, N0p, d, r0, s));
    RUN("gather_desc", launch_gather_desc(desc1_dev, s1b, s1c, s1n, B, N1, N1p, d, r1, s));
    RUN("rownorm2", launch_rownorm2(r0, d, (long)B * N0p, nr0, s));
    RUN("rownorm2", launch_rownorm2(r1, d, (long)B * N1p, nr1, s));
    ScoreArgs sc{r0, r1, dots, B, N0p, N1p, d, 1.0f};
    RUN("knn_dots", launch_score_gemm(sc, s));
    KnnArgs k{dots, nr0, nr1, B, N0, N1, N0p, N1p, n0_dev, n1_dev, ratio, reinterpret_cast<long long*>(matches_dev), dist1_dev, dist2_dev};
    RUN("knn2", launch_knn2(k, s));
    return 0;
  });
}

int imx_ingest_resize_u8(imx_handle_t h, const uint8_t* src_dev, int B, int Hs, int Ws, int64_t src_stride_b, float* dst_dev, int H,
                         int W, void* stream) {
  return guarded(h, "imx_ingest_resize_u8", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    if (B <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return fail(h, "imx_ingest_resize_u8: bad shape %dx%dx%d -> %dx%d", B, Hs, Ws, H, W);
    hipStream_t s = as_stream(stream);
    RUN("ingest_resize", launch_resize_u8_unit(src_dev, (long)src_stride_b, B, Hs, Ws, dst_dev, H, W, s));
    return 0;
  });
}

int imx_warp_affine_u8(imx_handle_t h, const uint8_t* src_dev, int Hs, int Ws, const double* M_host, uint8_t* dst_dev, int H, int W,
                       void* stream) {
  return guarded(h, "imx_warp_affine_u8", [&]() -> int {
    if (!h) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    if (Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || !M_host) return fail(h, "imx_warp_affine_u8: bad arguments");
    // cv2.warpAffine without WARP_INVERSE_MAP inverts the 2x3 matrix in double precision (imgwarp.cpp invertAffineTransform path)
    const double* M = M_host;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    double inv[6];
    inv[0] = A11; inv[1] = M[1] * (-D); inv[3] = M[3] * (-D); inv[4] = A22;
    inv[2] = -inv[0] * M[2] - inv[1] * M[5];
    inv[5] = -inv[3] * M[2] - inv[4] * M[5];
    hipStream_t s = as_stream(stream);
    RUN("warp_affine", launch_warp_affine_u8(src_dev, Hs, Ws, dst_dev, H, W, inv, s));
    return 0;
  });
}

int imx_set_debug(imx_handle_t h, int enable) {
  return guarded(h, "imx_set_debug", [&]() -> int {
    if (!h) return -1;
    h->debug = enable != 0;
    return 0;
  });
}

int imx_debug_fetch(imx_handle_t h, const char* name, float* host_out, int64_t capacity, int64_t* shape_out, int* ndim_out) {
  return guarded(h, "imx_debug_fetch", [&]() -> int {
    if (!h) return -1;
    auto it = h->taps.find(name ? name : "");
    if (it == h->taps.end()) return fail(h, "imx_debug_fetch: no tap named '%s'", name ? name : "(null)");
    int64_t n = 1;
    for (int64_t v : it->second.shape) n *= v;
    if (ndim_out) *ndim_out = (int)it->second.shape.size();
    if (shape_out) for (size_t i = 0; i < it->second.shape.size() && i < 4; ++i) shape_out[i] = it->second.shape[i];
    if (!host_out) return 0;   // shape query
    if (capacity < n) return fail(h, "imx_debug_fetch: capacity %lld < %lld elements", (long long)capacity, (long long)n);
    HIP_OK(h, hipSetDevice(h->device));
    HIP_OK(h, hipDeviceSynchronize());
    if (h->nms_lazy.pending && name && std::string(name) == "nms") {
      auto& z = h->nms_lazy;
      HIP_OK(h, launch_nms(z.smap, z.nms, z.B, z.H, z.W, z.radius, nullptr, z.scratch));
      HIP_OK(h, hipDeviceSynchronize());
      z.pending = false;
    }
    if (!it->second.blocked) {
      HIP_OK(h, hipMemcpy(host_out, it->second.p, (size_t)n * 4, hipMemcpyDeviceToHost));
      return 0;
    }
    // channel-blocked activation (B, C/8, H, W, 8) -> the NHWC tensor the tap promises
    std::vector<float> raw((size_t)n);
    HIP_OK(h, hipMemcpy(raw.data(), it->second.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    const int64_t Bn = it->second.shape[0], Hn = it->second.shape[1], Wn = it->second.shape[2], Cn = it->second.shape[3];
    for (int64_t bb = 0; bb < Bn; ++bb)
      for (int64_t ck = 0; ck < Cn / 8; ++ck)
        for (int64_t yx = 0; yx < Hn * Wn; ++yx)
          for (int64_t c = 0; c < 8; ++c)
            host_out[(bb * Hn * Wn + yx) * Cn + ck * 8 + c] = raw[((bb * (Cn / 8) + ck) * Hn * Wn + yx) * 8 + c];
    return 0;
  });
}

int imx_set_timing(imx_handle_t h, int enable) {
  return guarded(h, "imx_set_timing", [&]() -> int {
    if (!h) return -1;
    h->timing = enable != 0;
    return 0;
  });
}

int imx_timing_reset(imx_handle_t h) {
  return guarded(h, "imx_timing_reset", [&]() -> int {
    if (!h) return -1;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto& e : h->events) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    h->events.clear();
    h->report.clear();
    return 0;
  });
}

int imx_timing_report(imx_handle_t h, int index, const char** name_out, int64_t* launches_out, double* total_ms_out) {
  return guarded(h, "imx_timing_report", [&]() -> int {
    if (!h) return -1;
    if (index < 0) {   // (re)build the report; returns the number of rows
      HIP_OK(h, hipSetDevice(h->device));
      HIP_OK(h, hipDeviceSynchronize());
      std::map<std::string, TimingRow> agg;      // keyed by name + form: a name whose launches took different forms gets one row per form
      std::vector<std::string> order;
      for (auto& e : h->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
        const std::string form = e.form ? e.form : "", key = e.name + "|" + form;
        auto it = agg.find(key);
        if (it == agg.end()) { agg[key] = TimingRow{e.name, form, 1, ms}; order.push_back(key); }
        else { it->second.launches++; it->second.ms += ms; }
      }
      h->report.clear();
      for (auto& n : order) h->report.push_back(agg[n]);
      return (int)h->report.size();
    }
    if ((size_t)index >= h->report.size()) return fail(h, "imx_timing_report: index %d out of range", index);
    if (name_out) *name_out = h->report[index].name.c_str();
    if (launches_out) *launches_out = h->report[index].launches;
    if (total_ms_out) *total_ms_out = h->report[index].ms;
    return 0;
  });
}

const char* imx_timing_form(imx_handle_t h, int index) {
  if (!h || index < 0 || (size_t)index >= h->report.size()) return "";
  return h->report[index].form.c_str();
}

int imx_set_option(imx_handle_t h, const char* key, const char* value) {
  return guarded(h, "imx_set_option", [&]() -> int {
    if (!h) return -1;
    if (!key || !value) return fail(h, "imx_set_option: null argument");
    if (apply_option(h, key, value)) return fail(h, "imx_set_option: unknown option or value '%s' = '%s' (mfma = x3|f32, latency_forms = auto|off|on|unfused, conv = wino|wino_h|wino32|direct, gnn_tail = auto|fused|bf16x3|unfused, attention = auto|f16x2|bf16x3, linear = auto|f16x2|bf16x3, attention_qblocks = auto|1|2, conv_swizzle = on|off, qkv_amax = epilogue|kernel, sinkhorn_group = auto|1|2|4, sinkhorn_prefetch = auto|off|on, sinkhorn_merge = auto|kernel|fused, keypoints = auto|dense|bits)", key, value);
    return 0;
  });
}

const char* imx_get_option(imx_handle_t h, const char* key) {
  if (!h || !key) return "";
  try {
    const std::string k = key;
    const Options& o = h->opt;
    if (k == "mfma") h->opt_text = o.mfma_f32 ? "f32" : "x3";
    else if (k == "latency_forms") h->opt_text = o.latency_forms < 0 ? "auto" : o.latency_forms == 2 ? "unfused" : o.latency_forms ? "on" : "off";
    else if (k == "conv") h->opt_text = o.conv_direct ? "direct" : o.conv_f16 == 2 ? "wino_h" : o.conv_f16 ? "wino" : "wino32";
    else if (k == "gnn_tail") h->opt_text = o.gnn_tail < 0 ? "auto" : o.gnn_tail == 2 ? "bf16x3" : o.gnn_tail ? "fused" : "unfused";
    else if (k == "attention") h->opt_text = o.attention < 0 ? "auto" : o.attention ? "f16x2" : "bf16x3";
    else if (k == "attention_qblocks") h->opt_text = o.attention_qblocks < 0 ? std::string("auto") : std::to_string(o.attention_qblocks);
    else if (k == "linear") h->opt_text = o.linear < 0 ? "auto" : o.linear ? "f16x2" : "bf16x3";
    else if (k == "conv_swizzle") h->opt_text = o.conv_swizzle ? "on" : "off";
    else if (k == "qkv_amax") h->opt_text = o.qkv_amax ? "kernel" : "epilogue";
    else if (k == "sinkhorn_group") h->opt_text = o.sinkhorn_group ? std::to_string(o.sinkhorn_group) : std::string("auto");
    else if (k == "sinkhorn_prefetch") h->opt_text = o.sinkhorn_prefetch < 0 ? "auto" : o.sinkhorn_prefetch ? "on" : "off";
    else if (k == "sinkhorn_merge") h->opt_text = o.sinkhorn_merge < 0 ? "auto" : o.sinkhorn_merge ? "fused" : "kernel";
    else if (k == "keypoints") h->opt_text = o.keypoints < 0 ? "auto" : o.keypoints ? "bits" : "dense";
    else if (k == "arith_guard") {      // read-only: what the weights-derived guards decided (after imx_finalize_weights)
      char buf[96];
      float sp = h->c1a_spread;
      for (int i = 0; i < 8; ++i) sp = std::max(sp, h->conv[i].u_spread);
      snprintf(buf, sizeof buf, "conv: max spread 2^%.1f -> %s; gnn_tail bf16x3 layers:", std::log2(std::max(sp, 1.f)), sp <= kConvSpreadMax ? "f16x2" : "f32");
      h->opt_text = buf;
      for (size_t l = 0; l < h->layers.size(); ++l)
        if (h->layers[l].tail_stream_h2 && !(h->layers[l].h2c.loose_h <= kTailLooseMax && h->layers[l].h2c.loose_x <= kTailLooseMax && h->layers[l].h2c.w_spread <= kLinearSpreadMax)) h->opt_text += " " + std::to_string(l);
      float lx = 0.f;
      for (const auto& L : h->layers) lx = std::max(lx, std::max(L.h2c.loose_h, L.h2c.loose_x));
      snprintf(buf, sizeof buf, " (largest bound looseness 2^%.1f)", std::log2(std::max(lx, 1.f)));
      h->opt_text += buf;
      h->opt_text += "; attention bf16x3 layers:";
      float qs = 1.f;
      for (size_t l = 0; l < h->layers.size(); ++l) {
        qs = std::max(qs, h->layers[l].qkv_spread);
        if (h->layers[l].qkv_spread > kAttnSpreadMax) h->opt_text += " " + std::to_string(l);
      }
      snprintf(buf, sizeof buf, " (largest q|k|v channel spread 2^%.1f)", std::log2(qs));
      h->opt_text += buf;
      float ws = 1.f;
      bool lin_ok = gemm_h2_weights_ok(h->final_proj);
      for (const auto& L : h->layers) {
        ws = std::max(ws, std::max(L.qkv.wh2_spread, std::max(L.mlp1.wh2_spread, L.mlp2.wh2_spread)));
        lin_ok = lin_ok && gemm_h2_weights_ok(L.qkv) && gemm_h2_weights_ok(L.mlp1) && gemm_h2_weights_ok(L.mlp2);
      }
      snprintf(buf, sizeof buf, "; linear: max spread 2^%.1f -> %s", std::log2(ws), lin_ok ? "f16x2" : "bf16x3");
      h->opt_text += buf;
    } else h->opt_text.clear();
    return h->opt_text.c_str();
  } catch (...) {
    return "";
  }
}

}  // extern "C"
