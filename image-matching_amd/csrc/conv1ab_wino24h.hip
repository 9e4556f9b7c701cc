// conv1ab_wino24h.hip -- conv1ab_wino24.hip's fused first layer (conv1a + conv1b + folded BN + ReLU + MaxPool2d(2);
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-114) with conv1b's Winograd F(2x4, 3x3) products on the FP16 matrix
// pipe, as conv3x3_wino24h.hip does for the later layers (round 4): V = B2^T d B4 and U = G2 g G4^T as two fp16 planes each, three
// plane products per 32 input channels on v_mfma_f32_16x16x32_f16 (144 MFMAs of 16 cycles per tile and wave where the fp32 form
// spends 384 of 32).
//
// What changes against conv1ab_wino24.hip:
//   * a chunk is 32 channels, so V is 48 KB; the conv1a patch is therefore produced 32 CHANNELS AT A TIME (26 KB: 10x18 pixels x
//     (32 + 2) floats) -- half 0 before chunk 0's transform, half 1 behind chunk 0's MFMAs -- 76 KB per workgroup, two per CU;
//   * the scale of V: conv1a's outputs are bounded by max|image patch| x (largest column L1 norm of the folded conv1a weights) +
//     max|bias| (ConvArgs::c1a_l1, c1a_bmax, host); the TILE's power of two s_v brings 32 x that bound to 2^13 and rides in the
//     conv1a GEMM's B operand (the image values and the bias tap are multiplied by it), so the patch arrives in LDS scaled; it is
//     undone, with U's host-side scale, in the epilogue's bias fma;
//   * per tile: image patch + its maximum | barrier | conv1a half 0 | barrier | transform 0 | barrier | MFMAs 0, conv1a half 1 |
//     barrier | transform 1 | barrier | MFMAs 1, output transform, pool, stores -- five barriers where the fp32 form has ten.
// The transform / split / MFMA phases are conv3x3_wino24h.hip's; the conv1a GEMM, the epilogue and the tile walk conv1ab_wino24.hip's.
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per workgroup (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // conv1a patch (pad-1 halo)
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;  // image patch 12 x 20
constexpr int RSH = 34;                        // conv1a half patch: pixel stride (32 channels + 2: wtile columns 4 px apart land 8 banks apart)
constexpr int RAWSZ = 192 * RSH;               // 180 pixels + 12 pad (the conv1a GEMM's twelfth pixel block stores unmasked)
constexpr int NPOS = 24;
constexpr int VPLANE = NPOS * 4 * 16 * 8;      // halves per plane
constexpr int UPOS = 2 * 4 * 64 * 8;           // halves of U per (chunk, position): [plane][wave][lane][8]
constexpr int RING = 6;

template <bool V>
struct BoolC { static constexpr bool value = V; };

__device__ __forceinline__ void split_h2(f32x2 x, f16x2& h, f16x2& m) {      // conv3x3_wino24h.hip
  unsigned lo_u, hi_u;
  asm("s_mov_b32 %0, 0x0000bc00" : "=s"(lo_u));
  asm("s_mov_b32 %0, 0xbc000000" : "=s"(hi_u));
  const f16x2 lo = __builtin_bit_cast(f16x2, lo_u), hi = __builtin_bit_cast(f16x2, hi_u);
  h[0] = (_Float16)x[0]; h[1] = (_Float16)x[1];
  const float r0 = __builtin_amdgcn_fdot2(h, lo, x[0], false);
  const float r1 = __builtin_amdgcn_fdot2(h, hi, x[1], false);
  m[0] = (_Float16)r0; m[1] = (_Float16)r1;
}
// the power of two that brings 32 x `bound` (>= 20 max|d| >= |V|) to 2^13
__device__ __forceinline__ float v_scale_of_bound(float bound) {
  unsigned e = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (261u - e) << 23);
}

__global__ __launch_bounds__(256, 2) void conv1ab_wino24h(ConvArgs p, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_1h[];
  _Float16* Vp = reinterpret_cast<_Float16*>(smem_1h);                          // [2][VPLANE]
  float* raw = reinterpret_cast<float*>(smem_1h + 2 * VPLANE * 2);              // [192][RSH]: 32 channels of the conv1a patch
  float* img = raw + RAWSZ;                                                     // [12][20]
  float* wmax = img + IMG_H * IMG_W;                                            // [4]: the waves' maxima of |image patch|

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;
  const int H = p.H, W = p.W, Cout = p.Cout;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wuh, 0, 2 * NPOS * UPOS * 2, 0x00020000);
  const int uoff_lane = (cb * 64 + lane) * 16;

  // ---- per-lane constants of the conv1a GEMM (conv1ab_wino24.hip): A = weights, 12 registers, loaded once per workgroup
  const int n = lane & 15, kq = lane >> 4;
  const float c2 = kq == 0 ? 1.f : 0.f, a2 = kq == 1 ? 1.f : 0.f;      // third k-step: tap 8 | the bias "tap" (input 1) | zero padding
  float wa[4][3];
  int toff[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = 4 * ks + kq;
    toff[ks] = tap < 9 ? (tap / 3) * IMG_W + tap % 3 : 0;
#pragma unroll
    for (int cbk = 0; cbk < 4; ++cbk) {
      const float* src = tap < 9 ? p.w1 + tap * 64 + cbk * 16 + n : p.b1 + cbk * 16 + n;
      const float v = *(tap <= 9 ? src : p.b1);
      wa[cbk][ks] = tap <= 9 ? v : 0.f;
    }
  }
  // ---- input transform roles (conv3x3_wino24h.hip): lane = (channel pair tk, wtile tw); transformed row i = wave
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  const f32x2 sg2 = {sg, sg};
  const f32x2 m5 = {-5.f, -5.f};
  const float* rpa = raw + ((2 * twr + ra) * RW + 4 * twc) * RSH + 2 * tk;
  const float* rpb = raw + ((2 * twr + rb) * RW + 4 * twc) * RSH + 2 * tk;
  _Float16* vwr = Vp + (wave * 4 * 16 + tw) * 8 + 2 * tk;
  const _Float16* vrd = Vp + lane * 8;

  // ---- persistent over a contiguous range of tiles; the NEXT tile's image patch element is fetched a whole tile ahead
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, b = t_begin / (tiles_x * tiles_y);
  const int ipy = tid / IMG_W - 2, ipx = tid % IMG_W - 2;
  auto fetch_px = [&](int ftx, int fty, int fb, bool live) -> float {
    const int gy = fty * OH + ipy, gx = ftx * OW + ipx;
    const bool ok = live && tid < IMG_H * IMG_W && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* im = (fb < p.split) ? p.in + (size_t)fb * H * W : p.in2 + (size_t)(fb - p.split) * H * W;
    const float* src = ok ? im + (size_t)gy * W + gx : p.in;
    const float v = *src;
    return ok ? v : 0.f;
  };
  float pre = fetch_px(tx, ty, b, t_begin < t_end);
  int gpy[3], gpx[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pp = (3 * wave + j) * 16 + n;
    const int pc = pp < RH * RW ? pp : RH * RW - 1;
    gpy[j] = pc / RW;
    gpx[j] = pc % RW;
  }
  unsigned am_run = 0;           // ConvArgs::amax_out: flushed when the image changes (conv1ab_wino24.hip)
  int am_b = -1;
  auto am_flush = [&]() {
    unsigned mb = am_run;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if (lane == 0 && mb && am_b >= 0) atomicMax(p.amax_out + (am_b & 255), mb);
    am_run = 0;
  };
  const f32x4 bs4 = *reinterpret_cast<const f32x4*>(p.bias + cb * 16 + 4 * (lane >> 4));
  u32x4v ub[RING][2];
  auto u_load = [&](int slot, int chv, int pos) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane((chv * NPOS + pos) * (UPOS * 2));
    ub[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so, 0);
    ub[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so + 4 * 64 * 16, 0);
  };
#pragma unroll
  for (int g = 0; g < RING; ++g) u_load(g, 0, g);
  __builtin_amdgcn_s_waitcnt(0x0F70);

  f32x4 acc[NPOS];
  const f32x2 k8 = {8.f, 8.f};
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};
  float sv = 1.f;                // this tile's power of two

  // conv1a + folded BN + ReLU for 32 channels (16-channel blocks 2 half, 2 half + 1) of the 10x18 halo patch ON THE MATRIX CORES,
  // scaled by sv (image values and the bias tap carry it), into raw
  auto conv1a_half = [&](int half, int x0, int y0) __attribute__((always_inline)) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pp = (3 * wave + j) * 16 + n;
      const int py = gpy[j], px = gpx[j];
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      const float m = (pp < RH * RW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? sv : 0.f;   // branch-free: finite * 0; a pixel outside the image is relu(0)
      const float* ip = img + py * IMG_W + px;
      float bv[3];
      bv[0] = ip[toff[0]] * m;
      bv[1] = ip[toff[1]] * m;
      bv[2] = (ip[toff[2]] * c2 + a2) * m;
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        f32x4 d = zero4;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(half ? wa[2 + cl][ks] : wa[cl][ks], bv[ks], d, 0, 0, 0);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        d = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, d), (i32x4){0, 0, 0, 0}));      // ReLU on the bit pattern
        float* o = raw + pp * RSH + cl * 16 + 4 * kq;
        *reinterpret_cast<f32x2*>(o) = (f32x2){d[0], d[1]};
        *reinterpret_cast<f32x2*>(o + 2) = (f32x2){d[2], d[3]};
      }
    }
  };
  auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x2 o[6], T[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb)
        o[bb] = pk_fma(sg2, *reinterpret_cast<const f32x2*>(rpb + 8 * q + bb * RSH), *reinterpret_cast<const f32x2*>(rpa + 8 * q + bb * RSH));
      const W24Half hb = w24_batch_a(o, m5);
      w24_batch_b(o, hb, T);
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        f16x2 h, m;
        split_h2(T[jj], h, m);
        _Float16* d = vwr + ((jj * 4 * 4 + q) * 16) * 8;
        *reinterpret_cast<f16x2*>(d) = h;
        *reinterpret_cast<f16x2*>(d + VPLANE) = m;
      }
    }
  };
  auto mfma_phase = [&](auto firstc, int c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
#pragma unroll
    for (int pp = 0; pp < NPOS; pp += 2) {
      f16x8 bh[2], bm[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        bh[e] = *reinterpret_cast<const f16x8*>(vrd + (pp + e) * 512);
        bm[e] = *reinterpret_cast<const f16x8*>(vrd + VPLANE + (pp + e) * 512);
      }
      const f16x8 ah0 = __builtin_bit_cast(f16x8, ub[pp % RING][0]), am0 = __builtin_bit_cast(f16x8, ub[pp % RING][1]);
      const f16x8 ah1 = __builtin_bit_cast(f16x8, ub[(pp + 1) % RING][0]), am1 = __builtin_bit_cast(f16x8, ub[(pp + 1) % RING][1]);
      acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bm[0], FIRST ? zero4c : acc[pp], 0, 0, 0);
      acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bm[1], FIRST ? zero4c : acc[pp + 1], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am0, bh[0], acc[pp], 0, 0, 0);
      acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am1, bh[1], acc[pp + 1], 0, 0, 0);
      acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh[0], acc[pp], 0, 0, 0);
      acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh[1], acc[pp + 1], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int np = pp + e + RING;              // the other chunk follows (U is tile independent)
        if (np < NPOS) u_load((pp + e) % RING, c, np);
        else u_load((pp + e) % RING, c ^ 1, np - NPOS);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int t = t_begin; t < t_end; ++t) {
    const int x0 = tx * OW, y0 = ty * OH, bcur = b;
    if (tid < IMG_H * IMG_W) img[tid] = pre;
    {                                           // the patch's largest |value|: wave maxima -> LDS
      float mx = fabsf(pre);
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      if (lane == 0) wmax[wave] = mx;
    }
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++b; } }      // -> next tile
    pre = fetch_px(tx, ty, b, t + 1 < t_end);
    __syncthreads();             // image patch and maxima visible (and every wave is past the previous tile's conv1a half 1)
    {
      const float imax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      sv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v_scale_of_bound(fmaf(imax, p.c1a_l1, p.c1a_bmax)))));
    }
    conv1a_half(0, x0, y0);
    __syncthreads();             // channels 0..31 of the patch complete
    transform();
    __syncthreads();             // V complete; raw free
    mfma_phase(BoolC<true>{}, 0);
    conv1a_half(1, x0, y0);
    __syncthreads();             // channels 32..63 complete; V free
    transform();
    __syncthreads();             // V complete
    mfma_phase(BoolC<false>{}, 1);

    // ---- output transform, 2x2 max-pool (commutes with the positive scale), un-scale + bias, ReLU, stores (conv1ab_wino24.hip)
    {
      const float inv = p.u_scale_inv / sv;
      const f32x4 inv4 = {inv, inv, inv, inv};
      f32x4 y[2][4];
      w24_output_transform(acc, k8, y);
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int wr = (lane & 15) >> 2, wc = lane & 3;
      const int Ho = H >> 1, Wo = W >> 1;
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)bcur * Ho * Wo * Cout), 0, Ho * Wo * Cout * 4, 0x00020000);
      const bool outb = p.out_blocked != 0;
      const int opx = outb ? 32 : Cout * 4;
      const int choff = outb ? (cb * 2 + (lane >> 5)) * (Ho * Wo * 32) + ((lane >> 4) & 1) * 16 : (cb * 16 + 4 * (lane >> 4)) * 4;
      typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int oy = (y0 >> 1) + wr;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 mx4 = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]), __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1]));
        const f32x4 v = __builtin_elementwise_max(__builtin_elementwise_fma(mx4, inv4, bs4), zero4);
        const int ox = (x0 >> 1) + 2 * wc + hh;
        const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * opx + choff) : 0x7ffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)off, 0, 0);
        if (p.amax_out) {
          if (bcur != am_b) { am_flush(); am_b = bcur; }           // uniform
          am_run = max(am_run, __builtin_bit_cast(unsigned, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))));
        }
      }
    }
  }
  if (p.amax_out) am_flush();
}
}  // namespace

bool conv1ab_wino24h_supported(const ConvArgs& a) {
  return a.first && a.pool && a.Cin == 64 && a.Cout == 64 && a.wuh && a.u_scale_inv > 0.f && a.c1a_l1 > 0.f;
}

hipError_t launch_conv1ab_wino24h(const ConvArgs& a, hipStream_t s) {
  if (!conv1ab_wino24h_supported(a)) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH, ntiles = tiles_x * tiles_y * a.B;
  const size_t lds = (size_t)2 * VPLANE * 2 + (size_t)(RAWSZ + IMG_H * IMG_W + 4) * sizeof(float);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(conv1ab_wino24h), (int)lds, attr);
  const dim3 grid((unsigned)(ntiles < 2 * ncu ? ntiles : 2 * ncu));     // persistent: two workgroups per CU
  last_form = "conv1ab_wino24h:f16x2";
  hipLaunchKernelGGL(conv1ab_wino24h, grid, dim3(256), lds, s, a, tiles_x, tiles_y, ntiles);
  return hipGetLastError();
}

}  // namespace imx
