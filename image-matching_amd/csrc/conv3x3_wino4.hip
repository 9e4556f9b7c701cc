// conv3x3_wino4.hip — Winograd F(2x2,3x3) 3x3 convolution, software-pipelined variant:
// v_mfma_f32_32x32x2_f32, 256 accumulator registers per wave, one workgroup per CU.
//
// Same contract and arithmetic as conv3x3_wino.hip (replaces F.conv2d + folded BatchNorm + ReLU
// (+ MaxPool2d(2)) of superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123).
// Why a second variant: the 16x16x4 form needs 512 B of operands per 32-cycle MFMA and, with 128
// accumulators, two co-resident workgroups whose staging cannot hide behind their own matrix work
// (measured 50 % MFMA-pipe utilisation).  Here
//   * a wave owns a 32 wtile x 32 channel block for ALL 16 transform positions (16 x f32x16 = 256
//     VGPRs; one wave per SIMD), so a 64-cycle MFMA needs the same 512 B: half the operand rate;
//   * V and U are double buffered in LDS and every piece of the next chunk's staging (U -> LDS, input
//     transform raw -> V, raw patch store, global prefetch, and in FIRST mode the conv1a evaluation)
//     is issued in small slices BETWEEN the 64 MFMAs of the current chunk (sched_barrier pins the
//     interleave), so the matrix pipe keeps running while the wave stages;
//   * the 32x32 D layout still gives each lane one output channel with all 16 positions of its 16
//     wtiles: output transform, bias, ReLU and 2x2 max-pool stay in-lane.
// Workgroup = 256 threads (4 waves, 2 wtile halves x 2 channel halves) -> 4x16 wtiles (8x32 output
// pixels) x 64 output channels; K in chunks of 8 input channels; LDS 152 KB:
//   V[2][16 pos][8 ch][68]   A operands: lane (wtile l&31, k l>>5) reads 32 consecutive floats
//   U[2][16 pos][8 ch][64]   B operands, same shape (global layout identical: float4 copies)
//   raw[10][34][12]          input patch (pixel stride 12: conflict-free (4 wtiles x 8 ch) reads)
#include "imx_kernels.h"
#include <cstdio>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int TR = 4, TC = 16, OH = 2 * TR, OW = 2 * TC;   // wtiles / output pixels per workgroup
constexpr int RH = OH + 2, RW = OW + 2, RS = 12;           // raw patch 10 x 34, pixel stride 12 floats
constexpr int RAW = RH * RW * RS;                          // 4080
constexpr int CK = 8, NT = 64;
constexpr int VP = 68, VSZ = 16 * CK * VP;                 // 8704
constexpr int USZ = 16 * CK * NT;                          // 8192
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;              // FIRST: image patch 12 x 36
constexpr int NRAW4 = RH * RW * 2;                         // 680 float4 items of the raw patch
constexpr int PD = 4;                                      // operand prefetch distance (MFMAs)

template <bool POOL, bool RELU, bool FIRST, bool TRACE = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino4(ConvArgs p, int tiles_x, int tiles_y, unsigned* trace = nullptr) {
  unsigned long long tstart = 0, tloop = 0, tepi = 0;
  if constexpr (TRACE) tstart = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vb = smem;                    // [2][VSZ]
  float* Ub = Vb + 2 * VSZ;            // [2][USZ]
  float* raw = Ub + 2 * USZ;           // [RAW]
  float* img = raw + RAW;              // FIRST only
  float* w1s = img + IMG_H * IMG_W;
  float* b1s = w1s + 9 * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ih = wave & 1, jh = wave >> 1;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * OW, y0 = ty * OH;
  const int n0 = blockIdx.y * NT;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK;
  const float* ublk = p.wu4 + (size_t)blockIdx.y * nchunk * USZ;

  // ---- chunk-invariant staging maps
  // raw patch float4 items e = tid + 256*it (it < 3): pixel e>>1, channel half e&1
  const float* rsrc[3];
  int rdst[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int e = tid + it * 256;
    rsrc[it] = nullptr;
    rdst[it] = -1;
    if (!FIRST && e < NRAW4) {
      const int pix = e >> 1, half = e & 1, py = pix / RW, px = pix % RW;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) rsrc[it] = p.in + ((size_t)(b * H + gy) * W + gx) * Cin + 4 * half;
      rdst[it] = pix * RS + 4 * half;
    }
  }
  // transform items: channel tc, wtiles tw and tw + 32
  const int tc = tid & 7, tw = tid >> 3;
  int roff[2], voff[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int w = tw + 32 * it, wr = w >> 4, wc = w & 15;
    roff[it] = ((2 * wr) * RW + 2 * wc) * RS + tc;
    voff[it] = tc * VP + w;
  }
  // FIRST: conv1a items = halo pixels tid and tid + 256 (< 340)
  int c1pix[2], c1img[2];
  bool c1in[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pix = tid + 256 * it;
    const int py = pix / RW, px = pix % RW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    c1pix[it] = pix < RH * RW ? pix : -1;
    c1img[it] = py * IMG_W + px;
    c1in[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
  }

  float4 rr0 = make_float4(0.f, 0.f, 0.f, 0.f), rr1 = rr0, rr2 = rr0;     // raw patch prefetch registers
  float4 uu0, uu1, uu2, uu3, uu4, uu5, uu6, uu7;                           // U prefetch registers

#define W4_GRAW(reg_, it_, chunk_) \
  reg_ = rsrc[it_] ? *reinterpret_cast<const float4*>(rsrc[it_] + (chunk_) * CK) : make_float4(0.f, 0.f, 0.f, 0.f);
#define W4_SRAW(reg_, it_) \
  if (rdst[it_] >= 0) *reinterpret_cast<float4*>(raw + rdst[it_]) = reg_;
#define W4_GU(reg_, it_, chunk_) \
  reg_ = *reinterpret_cast<const float4*>(ublk + (size_t)(chunk_) * USZ + (tid + (it_) * 256) * 4);
#define W4_SU(reg_, it_, buf_) \
  *reinterpret_cast<float4*>(Ub + (buf_) * USZ + (tid + (it_) * 256) * 4) = reg_;

  // conv1a (1 -> 64, 3x3, folded BN, ReLU) for 8 channels of chunk `chunk_` at halo pixel item it_
  auto conv1a_item = [&](int it, int chunk) {
    if (c1pix[it] < 0) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (c1in[it]) {
      const int cb = chunk * CK;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = b1s[cb + j];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const float iv = img[c1img[it] + (tp / 3) * IMG_W + tp % 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(iv, w1s[tp * 64 + cb + j], v[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    float4* dst = reinterpret_cast<float4*>(raw + c1pix[it] * RS);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
  };
  // input transform of one (channel, wtile) item: raw -> V[buf]
  auto transform_item = [&](int it, int buf) {
    const float* rp = raw + roff[it];
    float d[4][4], tt[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) d[a][bb] = rp[(a * RW + bb) * RS];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      tt[0][bb] = d[0][bb] - d[2][bb];
      tt[1][bb] = d[1][bb] + d[2][bb];
      tt[2][bb] = d[2][bb] - d[1][bb];
      tt[3][bb] = d[1][bb] - d[3][bb];
    }
    float* vp = Vb + buf * VSZ + voff[it];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      vp[((xi * 4 + 0) * CK) * VP] = tt[xi][0] - tt[xi][2];
      vp[((xi * 4 + 1) * CK) * VP] = tt[xi][1] + tt[xi][2];
      vp[((xi * 4 + 2) * CK) * VP] = tt[xi][2] - tt[xi][1];
      vp[((xi * 4 + 3) * CK) * VP] = tt[xi][1] - tt[xi][3];
    }
  };

  // ================================================================ prologue
  if constexpr (FIRST) {
    const float* im = (b < p.split) ? p.in + (size_t)b * H * W : p.in2 + (size_t)(b - p.split) * H * W;
    for (int e = tid; e < IMG_H * IMG_W; e += 256) {
      const int py = e / IMG_W, px = e % IMG_W;
      const int gy = y0 + py - 2, gx = x0 + px - 2;
      img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    }
    for (int e = tid; e < 9 * 64; e += 256) w1s[e] = p.w1[e];
    if (tid < 64) b1s[tid] = p.b1[tid];
    __syncthreads();
    conv1a_item(0, 0);
    conv1a_item(1, 0);
  }
  // all prologue loads are issued back to back (one exposed memory latency instead of three)
  const int pc1 = nchunk > 1 ? 1 : 0, pc2 = nchunk > 2 ? 2 : nchunk - 1;
  float4 ra0 = rr0, ra1 = rr0, ra2 = rr0, rb0 = rr0, rb1 = rr0, rb2 = rr0;
  float4 ua0, ua1, ua2, ua3, ua4, ua5, ua6, ua7;
  if constexpr (!FIRST) {
    W4_GRAW(ra0, 0, 0) W4_GRAW(ra1, 1, 0) W4_GRAW(ra2, 2, 0)
    W4_GRAW(rb0, 0, pc1) W4_GRAW(rb1, 1, pc1) W4_GRAW(rb2, 2, pc1)
    W4_GRAW(rr0, 0, pc2) W4_GRAW(rr1, 1, pc2) W4_GRAW(rr2, 2, pc2)          // raw(2) stays in registers
  }
  W4_GU(ua0, 0, 0) W4_GU(ua1, 1, 0) W4_GU(ua2, 2, 0) W4_GU(ua3, 3, 0)
  W4_GU(ua4, 4, 0) W4_GU(ua5, 5, 0) W4_GU(ua6, 6, 0) W4_GU(ua7, 7, 0)
  W4_GU(uu0, 0, pc1) W4_GU(uu1, 1, pc1) W4_GU(uu2, 2, pc1) W4_GU(uu3, 3, pc1)      // U(1) stays in registers
  W4_GU(uu4, 4, pc1) W4_GU(uu5, 5, pc1) W4_GU(uu6, 6, pc1) W4_GU(uu7, 7, pc1)
  if constexpr (!FIRST) { W4_SRAW(ra0, 0) W4_SRAW(ra1, 1) W4_SRAW(ra2, 2) }
  W4_SU(ua0, 0, 0) W4_SU(ua1, 1, 0) W4_SU(ua2, 2, 0) W4_SU(ua3, 3, 0)
  W4_SU(ua4, 4, 0) W4_SU(ua5, 5, 0) W4_SU(ua6, 6, 0) W4_SU(ua7, 7, 0)
  __syncthreads();                       // raw(0) visible
  transform_item(0, 0);
  transform_item(1, 0);
  __syncthreads();                       // raw(0) consumed, V[0]/U[0] visible
  if constexpr (FIRST) {
    conv1a_item(0, pc1);
    conv1a_item(1, pc1);
  } else {
    W4_SRAW(rb0, 0) W4_SRAW(rb1, 1) W4_SRAW(rb2, 2)
  }
  __syncthreads();                       // raw(1) visible

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int aoff = (lane >> 5) * VP + ih * 32 + (lane & 31);
  const int boff = (lane >> 5) * NT + jh * 32 + (lane & 31);

  if constexpr (TRACE) tloop = __builtin_readcyclecounter();
  // ================================================================ main loop
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cb = ch & 1, nb = cb ^ 1;
    const int cu2 = ch + 2 < nchunk ? ch + 2 : nchunk - 1;   // chunk whose U is fetched this iteration
    const int cr3 = ch + 3 < nchunk ? ch + 3 : nchunk - 1;   // chunk whose raw patch is fetched (non-FIRST)
    const int cr2 = ch + 2 < nchunk ? ch + 2 : nchunk - 1;   // chunk whose conv1a patch is computed (FIRST)
    const float* va = Vb + cb * VSZ + aoff;
    const float* ub = Ub + cb * USZ + boff;
    float aop[2 * PD], bop[2 * PD];
    float d0[16], d1[16];                  // input-transform state of the two (channel, wtile) items
    float c0[8], c1[8], iv0[9], iv1[9];    // FIRST: conv1a state of the two halo-pixel items
    (void)c0; (void)c1; (void)iv0; (void)iv1;
#pragma unroll
    for (int m = 0; m < PD; ++m) {         // MFMA m: position m & 15, k-step m >> 4
      aop[m] = va[((m & 15) * CK + 2 * (m >> 4)) * VP];
      bop[m] = ub[((m & 15) * CK + 2 * (m >> 4)) * NT];
    }
    // ---- staging slices (each is issued right after one MFMA so the matrix pipe keeps draining)
#define W4_TR_READ(dd_, it_, a0_)                                                   \
  _Pragma("unroll") for (int a = (a0_); a < (a0_) + 2; ++a)                        \
    _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) dd_[a * 4 + bb] = raw[roff[it_] + (a * RW + bb) * RS];
#define W4_TR_ROWS(dd_)                                                             \
  _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) {                               \
    const float r0_ = dd_[0 + bb], r1_ = dd_[4 + bb], r2_ = dd_[8 + bb], r3_ = dd_[12 + bb]; \
    dd_[0 + bb] = r0_ - r2_; dd_[4 + bb] = r1_ + r2_; dd_[8 + bb] = r2_ - r1_; dd_[12 + bb] = r1_ - r3_; \
  }
#define W4_TR_COLS(dd_)                                                             \
  _Pragma("unroll") for (int xi = 0; xi < 4; ++xi) {                               \
    const float q0_ = dd_[xi * 4], q1_ = dd_[xi * 4 + 1], q2_ = dd_[xi * 4 + 2], q3_ = dd_[xi * 4 + 3]; \
    dd_[xi * 4] = q0_ - q2_; dd_[xi * 4 + 1] = q1_ + q2_; dd_[xi * 4 + 2] = q2_ - q1_; dd_[xi * 4 + 3] = q1_ - q3_; \
  }
#define W4_TR_WRITE(dd_, it_, xi_)                                                  \
  _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                 \
    Vb[nb * VSZ + voff[it_] + (((xi_) * 4 + nu) * CK) * VP] = dd_[(xi_) * 4 + nu];
#define W4_C1_BEGIN(cc_, iv_, it_)                                                  \
  if (c1pix[it_] >= 0) {                                                            \
    _Pragma("unroll") for (int tp = 0; tp < 9; ++tp) iv_[tp] = img[c1img[it_] + (tp / 3) * IMG_W + tp % 3]; \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) cc_[j] = b1s[cr2 * CK + j];      \
  }
#define W4_C1_TAP(cc_, iv_, it_, tp_)                                               \
  if (c1pix[it_] >= 0) {                                                            \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) cc_[j] = fmaf(iv_[tp_], w1s[(tp_) * 64 + cr2 * CK + j], cc_[j]); \
  }
#define W4_C1_END(cc_, it_)                                                         \
  if (c1pix[it_] >= 0) {                                                            \
    float4* dst_ = reinterpret_cast<float4*>(raw + c1pix[it_] * RS);                \
    const bool in_ = c1in[it_];                                                     \
    dst_[0] = in_ ? make_float4(fmaxf(cc_[0], 0.f), fmaxf(cc_[1], 0.f), fmaxf(cc_[2], 0.f), fmaxf(cc_[3], 0.f)) \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                \
    dst_[1] = in_ ? make_float4(fmaxf(cc_[4], 0.f), fmaxf(cc_[5], 0.f), fmaxf(cc_[6], 0.f), fmaxf(cc_[7], 0.f)) \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                \
  }
#pragma clang loop unroll(full)
    for (int m = 0; m < 64; ++m) {
      if (m + PD < 64) {
        const int m2 = m + PD;
        aop[m2 % (2 * PD)] = va[((m2 & 15) * CK + 2 * (m2 >> 4)) * VP];
        bop[m2 % (2 * PD)] = ub[((m2 & 15) * CK + 2 * (m2 >> 4)) * NT];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[m % (2 * PD)], bop[m % (2 * PD)], acc[m & 15], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      switch (m) {
        case 0: W4_SU(uu0, 0, nb) break;       // U(ch+1) registers -> LDS
        case 1: W4_SU(uu1, 1, nb) break;
        case 2: W4_SU(uu2, 2, nb) break;
        case 3: W4_SU(uu3, 3, nb) break;
        case 4: W4_SU(uu4, 4, nb) break;
        case 5: W4_SU(uu5, 5, nb) break;
        case 6: W4_SU(uu6, 6, nb) break;
        case 7: W4_SU(uu7, 7, nb) break;
        case 8: W4_TR_READ(d0, 0, 0) break;     // input transform raw(ch+1) -> V[nb], two items
        case 9: W4_TR_READ(d0, 0, 2) break;
        case 10: W4_TR_READ(d1, 1, 0) break;
        case 11: W4_TR_READ(d1, 1, 2) break;
        case 13: W4_TR_ROWS(d0) break;
        case 14: W4_TR_COLS(d0) break;
        case 15: W4_TR_WRITE(d0, 0, 0) break;
        case 16: W4_TR_WRITE(d0, 0, 1) break;
        case 17: W4_TR_WRITE(d0, 0, 2) break;
        case 18: W4_TR_WRITE(d0, 0, 3) break;
        case 19: W4_TR_ROWS(d1) break;
        case 20: W4_TR_COLS(d1) break;
        case 21: W4_TR_WRITE(d1, 1, 0) break;
        case 22: W4_TR_WRITE(d1, 1, 1) break;
        case 23: W4_TR_WRITE(d1, 1, 2) break;
        case 24: W4_TR_WRITE(d1, 1, 3) break;
        case 27: __syncthreads(); break;        // every thread is done reading raw(ch+1)
        case 28: if constexpr (FIRST) { W4_C1_BEGIN(c0, iv0, 0) } else { W4_SRAW(rr0, 0) } break;   // raw(ch+2) -> LDS
        case 29: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 0) } else { W4_SRAW(rr1, 1) } break;
        case 30: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 1) } else { W4_SRAW(rr2, 2) } break;
        case 31: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 2) } break;
        case 32: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 3) } W4_GU(uu0, 0, cu2) break;       // U(ch+2) -> registers
        case 33: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 4) } W4_GU(uu1, 1, cu2) break;
        case 34: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 5) } W4_GU(uu2, 2, cu2) break;
        case 35: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 6) } W4_GU(uu3, 3, cu2) break;
        case 36: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 7) } W4_GU(uu4, 4, cu2) break;
        case 37: if constexpr (FIRST) { W4_C1_TAP(c0, iv0, 0, 8) } W4_GU(uu5, 5, cu2) break;
        case 38: if constexpr (FIRST) { W4_C1_END(c0, 0) } W4_GU(uu6, 6, cu2) break;
        case 39: if constexpr (FIRST) { W4_C1_BEGIN(c1, iv1, 1) } W4_GU(uu7, 7, cu2) break;
        case 40: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 0) } else { W4_GRAW(rr0, 0, cr3) } break;   // raw(ch+3) -> registers
        case 41: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 1) } else { W4_GRAW(rr1, 1, cr3) } break;
        case 42: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 2) } else { W4_GRAW(rr2, 2, cr3) } break;
        case 43: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 3) } break;
        case 44: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 4) } break;
        case 45: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 5) } break;
        case 46: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 6) } break;
        case 47: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 7) } break;
        case 48: if constexpr (FIRST) { W4_C1_TAP(c1, iv1, 1, 8) } break;
        case 49: if constexpr (FIRST) { W4_C1_END(c1, 1) } break;
        default: break;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef W4_TR_READ
#undef W4_TR_ROWS
#undef W4_TR_COLS
#undef W4_TR_WRITE
#undef W4_C1_BEGIN
#undef W4_C1_TAP
#undef W4_C1_END
    __syncthreads();       // V[nb]/U[nb] and raw(ch+2) visible; everyone is done with V[cb]/U[cb]
  }
#undef W4_GRAW
#undef W4_SRAW
#undef W4_GU
#undef W4_SU

  if constexpr (TRACE) tepi = __builtin_readcyclecounter();
  // ================================================================ output transform + epilogue
  // acc[q][r]: wtile ih*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), channel n0 + jh*32 + (lane&31).
  // The transformed tile goes through LDS (free after the loop's last barrier) so that HBM sees whole
  // 256-byte channel rows written with float4 stores: per-lane dword stores at a pixel stride cost
  // ~250 cycles per wave-instruction here (measured: 19k-cycle epilogue for 64 of them).
  constexpr int OS = NT + 4;                      // staging row stride (floats), 16-B aligned rows
  float* Ot = smem;
  const int col = jh * 32 + (lane & 31);
  const float bs = p.bias[n0 + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int w = ih * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int wr = w >> 4, wc = w & 15;
    float t0[4], t1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t0[nu] = acc[0 * 4 + nu][r] + acc[1 * 4 + nu][r] + acc[2 * 4 + nu][r];
      t1[nu] = acc[1 * 4 + nu][r] - acc[2 * 4 + nu][r] - acc[3 * 4 + nu][r];
    }
    float y00 = t0[0] + t0[1] + t0[2] + bs, y01 = t0[1] - t0[2] - t0[3] + bs;
    float y10 = t1[0] + t1[1] + t1[2] + bs, y11 = t1[1] - t1[2] - t1[3] + bs;
    if (RELU) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
    if constexpr (POOL) {
      Ot[(wr * TC + wc) * OS + col] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
    } else {
      float* o = Ot + ((2 * wr) * OW + 2 * wc) * OS + col;
      o[0] = y00;
      o[OS] = y01;
      o[OW * OS] = y10;
      o[OW * OS + OS] = y11;
    }
  }
  __syncthreads();
  {
    constexpr int PH_ = POOL ? TR : OH, PW_ = POOL ? TC : OW;      // staged tile in pixels
    const int Hout = POOL ? (H >> 1) : H, Wout = POOL ? (W >> 1) : W;
    const int oy0 = POOL ? (y0 >> 1) : y0, ox0 = POOL ? (x0 >> 1) : x0;
#pragma unroll
    for (int it = 0; it < PH_ * PW_ * (NT / 4) / 256; ++it) {
      const int e = tid + it * 256;
      const int pix = e / (NT / 4), v4 = e % (NT / 4);
      const int py = pix / PW_, px = pix % PW_;
      const int oy = oy0 + py, ox = ox0 + px;
      if (oy < Hout && ox < Wout)
        *reinterpret_cast<float4*>(p.out + ((size_t)(b * Hout + oy) * Wout + ox) * Cout + n0 + 4 * v4) =
            *reinterpret_cast<const float4*>(Ot + pix * OS + 4 * v4);
    }
  }
  if constexpr (TRACE) {
    const unsigned long long tend = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
      trace[blockIdx.x * 4 + 0] = (unsigned)(tloop - tstart);
      trace[blockIdx.x * 4 + 1] = (unsigned)(tepi - tloop);
      trace[blockIdx.x * 4 + 2] = (unsigned)(tend - tepi);
    }
  }
}

template <bool POOL, bool RELU, bool FIRST>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  dim3 grid((unsigned)(tiles_x * tiles_y * a.B), (unsigned)(a.Cout / NT));
  const size_t lds = (size_t)(2 * VSZ + 2 * USZ + RAW + (FIRST ? IMG_H * IMG_W + 9 * 64 + 64 : 0)) * sizeof(float);
  auto k = conv3x3_wino4<POOL, RELU, FIRST>;
  static bool attr_set = false;
  if (!attr_set) {    // > 64 KB of dynamic LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (getenv("IMX_WINO_TRACE")) {
    static unsigned* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 4 * sizeof(unsigned));
    (void)hipMemsetAsync(dbuf, 0, 4096 * 4 * sizeof(unsigned), s);
    auto kt = conv3x3_wino4<POOL, RELU, FIRST, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kt, grid, dim3(256), lds, s, a, tiles_x, tiles_y, dbuf);
    (void)hipStreamSynchronize(s);
    static unsigned host[4096 * 4];
    (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
    const int n = grid.x < 4096 ? (int)grid.x : 4096;
    double sum[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) sum[j] += host[i * 4 + j];
    fprintf(stderr, "[wino4 trace] H=%d W=%d Cin=%d Cout=%d pool=%d first=%d grid=%u | prologue %.0f  loop %.0f (%.0f / chunk)  epilogue %.0f cycles\n",
            a.H, a.W, a.Cin, a.Cout, (int)POOL, (int)FIRST, grid.x, sum[0] / n, sum[1] / n, sum[1] / n / (a.Cin / CK), sum[2] / n);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y, (unsigned*)nullptr);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3_wino4(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % CK || a.Cout % NT || (a.first && a.Cin != 64) || !a.wu4) return hipErrorInvalidValue;
  if (a.first) return a.pool ? launch_t<true, true, true>(a, s) : launch_t<false, true, true>(a, s);
  if (a.pool) return a.relu ? launch_t<true, true, false>(a, s) : launch_t<true, false, false>(a, s);
  return a.relu ? launch_t<false, true, false>(a, s) : launch_t<false, false, false>(a, s);
}

}  // namespace imx
