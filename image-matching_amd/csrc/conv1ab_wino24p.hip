// conv1ab_wino24p.hip -- conv1ab_wino24h.hip's fused first layer (conv1a + conv1b + folded BN + ReLU + MaxPool2d(2);
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-114) on PAIRS of tiles, conv1b's 24 Winograd positions split over two
// waves -- conv3x3_wino24p.hip's structure (round 5): a transformed-weight fragment U of conv1b meets the B operands of two tiles, so
// the layer pulls half the bytes through the L1 data path that the counters name as its busiest unit (profiles/r05_*_pmc_limiter.json:
// 0.55 of the launch's cycles for conv1ab_wino24h, 121 GB per 128-image launch).
//
//   workgroup = 8 waves = two 8x16-pixel tiles x 64 channels, one workgroup per CU, a contiguous range of tile pairs.
//   per pair: input transform of chunk 0, split by (8-channel sub-patch, tile) | barrier | 72 MFMAs per wave (transformed columns 3 ph .. 3 ph + 2 of
//   both tiles) with conv1a channels 32..63 in three pieces between them | barrier | the NEXT pair's image patches (2 x 12x20) and
//   their maxima to LDS, transform 1 | barrier | MFMAs 1 with the next pair's conv1a channels 0..31 between them (conv1a on the fp32
//   matrix cores, four waves per tile, conv1ab_wino24h's GEMM: weights = A, im2col = B, the tile's power of two s_v riding in B) |
//   barrier | output transform's row stage, exchange of its six results (write | barrier | read | barrier) |
//   conv1ab_wino24h's epilogue (output transform, 2x2 max-pool, un-scale + bias, ReLU, store) by wave (cb, ph) for tile ph.
// Every output sees conv1ab_wino24h's arithmetic in the same order: the two kernels agree bit for bit (tests/test_gpu_superpoint.py).
// LDS: V 96 KB + conv1a half patches 2 x 26 KB + image patches + maxima table = 151 KB.
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdint>
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
constexpr int OH = 8, OW = 16;                 // output pixels per tile (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // conv1a patch (pad-1 halo)
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;  // image patch 12 x 20
constexpr int IMG_N = IMG_H * IMG_W;           // 240
constexpr int RSH = 34;                        // conv1a half patch: pixel stride (32 channels + 2), conv1ab_wino24h.hip
constexpr int RAWSZ = 192 * RSH;               // floats per tile: 180 pixels + 12 pad
constexpr int NPOS = 24, NLP = 12, NG = 2;
constexpr int VPLANE = NPOS * 4 * 16 * 8;      // halves per plane of a tile
constexpr int VGRP = 2 * VPLANE;               // halves per tile
constexpr int UPOS = 2 * 4 * 64 * 8;           // halves of U per (chunk, position): [plane][channel block][lane][8]
constexpr int RING = 6;
constexpr int XCH = 6 * 64 * 16;               // bytes of one wave's exchange block: the row stage's six results for the partner's tile (conv3x3_wino24p.hip)
constexpr int AMAX_SLOTS = 256;

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
__device__ __forceinline__ int lane_now() {                                    // conv3x3_wino24p.hip
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// the power of two that brings 32 x `bound` (>= 20 max|d| >= |V|) to 2^13 (conv1ab_wino24h.hip)
__device__ __forceinline__ float v_scale_of_bound(float bound) {
  unsigned e = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (261u - e) << 23);
}

__global__ __launch_bounds__(512) void conv1ab_wino24p(ConvArgs p, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_1p[];
  float* raw = reinterpret_cast<float*>(smem_1p + NG * VGRP * 2);               // [NG][192][RSH]: 32 channels of each conv1a patch
  float* img = raw + NG * RAWSZ;                                                // [NG][12][20]
  float* wmax = img + NG * IMG_N;                                               // [8]: the waves' maxima of |image patch|
  unsigned* amax_tab = reinterpret_cast<unsigned*>(wmax + 8);                   // [AMAX_SLOTS]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_1p;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave_s & 3, ph = wave_s >> 2;         // matrix role: channel block, position half
  const int tq = wave_s & 3, tg = wave_s >> 2;         // transform role: 8-channel sub-patch, tile; conv1a / image / epilogue: tile tg (== ph), quarter tq
  const int H = p.H, W = p.W, Cout = p.Cout;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wuh, 0, 2 * NPOS * UPOS * 2, 0x00020000);
  const int uoff_lane = (cb * 64 + lane) * 16 + ph * NLP * (UPOS * 2);         // the wave's positions: transformed columns 3 ph .. 3 ph + 2 = positions 12 ph + lp (conv3x3_wino24p.hip)
  typedef const float __attribute__((address_space(4)))* cf32p;
  const cf32p bias_c = (cf32p)(uintptr_t)p.bias;

  // ---- per-lane constants of the conv1a GEMM (conv1ab_wino24h.hip): A = weights, 12 registers, loaded once per workgroup
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
  // ---- input transform: lane = (channel pair tk, wtile tw) of sub-patch tq of tile tg (conv3x3_wino24p.hip)
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const f32x2 m5 = {-5.f, -5.f}, one2 = {1.f, 1.f}, mone2 = {-1.f, -1.f};
  const float* rp = raw + tg * RAWSZ + ((2 * twr) * RW + 4 * twc) * RSH + 8 * tq + 2 * tk;      // + (row * RW + column) * RSH
  _Float16* const vwr = (_Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)(lds0 + (unsigned)(tg * (VGRP * 2) + tq * 256 + lane * 4)));
  const _Float16 *vrdK, *vrdS;
  {
    unsigned a = lds0 + (unsigned)(ph * (VGRP * 2) + ph * NLP * 1024 + lane * 16);
    unsigned b = lds0 + (unsigned)((1 - ph) * (VGRP * 2) + ph * NLP * 1024 + lane * 16);
    asm volatile("" : "+v"(a), "+v"(b));
    vrdK = (const _Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)a);
    vrdS = (const _Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)b);
  }

  // ---- persistent over a contiguous range of tile PAIRS; this wave's tile of pair q is 2 q + tg; the next pair's image patch element
  // is fetched a whole pair ahead (thread t8 < 240 of the tile's 256 threads)
  const int npairs = (ntiles + NG - 1) / NG;
  const int per = (npairs + (int)gridDim.x - 1) / (int)gridDim.x;
  const int q_begin = (int)blockIdx.x * per, q_end = q_begin + per < npairs ? q_begin + per : npairs;
  if (q_begin >= q_end) return;
  int tx, ty, b, tlive;
  {
    const int t = NG * q_begin + tg;
    tlive = t < ntiles;
    const int tt = tlive ? t : 0;
    tx = tt % tiles_x; ty = (tt / tiles_x) % tiles_y; b = tt / (tiles_x * tiles_y);
  }
  auto next_tile = [&]() __attribute__((always_inline)) {          // two tiles on
    tx += NG;
    while (tx >= tiles_x) { tx -= tiles_x; if (++ty == tiles_y) { ty = 0; ++b; } }
    tlive = b < p.B;
  };
  const int t8 = tid & 255;
  auto fetch_px = [&](bool live) __attribute__((always_inline)) -> float {
    // (the element's patch coordinates from the lane index, recomputed here: kept across the loop they are spilled, and a scratch
    // reload is waited for with vmcnt(0) -- behind the U refills the matrix phase has just requested)
    const int t8n = (wave_s & 3) * 64 + lane_now();
    const int ipy = t8n / IMG_W - 2, ipx = t8n % IMG_W - 2;
    const int gy = ty * OH + ipy, gx = tx * OW + ipx;
    const bool ok = live && tlive && t8 < IMG_N && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int fb = tlive ? b : 0;
    const float* im = (fb < p.split) ? p.in + (size_t)fb * H * W : p.in2 + (size_t)(fb - p.split) * H * W;
    const float* src = ok ? im + (size_t)gy * W + gx : p.in;
    const float v = *src;
    return ok ? v : 0.f;
  };
  float pre = fetch_px(true);
  int gpy[3], gpx[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pp = (3 * tq + j) * 16 + n;
    const int pc = pp < RH * RW ? pp : RH * RW - 1;
    gpy[j] = pc / RW;
    gpx[j] = pc % RW;
  }
  u32x4v ub[RING][2];
  auto u_load = [&](int slot, int chv, int lp) __attribute__((always_inline)) {
    const int pos = lp;
    const int so = __builtin_amdgcn_readfirstlane((chv * NPOS + pos) * (UPOS * 2));
    ub[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so, 0);
    ub[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so + 4 * 64 * 16, 0);
  };
  for (int i = tid; i < AMAX_SLOTS; i += 512) amax_tab[i] = 0;
#pragma unroll
  for (int g = 0; g < RING; ++g) u_load(g, 0, g);

  f32x4 accK[NLP], accS[NLP];
  const f32x2 k8 = {8.f, 8.f};
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};
  float sv = 1.f;                // this wave's tile's power of two
  float* const rawt = raw + tg * RAWSZ;
  const float* const imgt = img + tg * IMG_N;

  // conv1a + folded BN + ReLU for 32 channels (16-channel blocks 2 half, 2 half + 1) of this wave's share of its tile's 10x18 halo
  // patch ON THE MATRIX CORES, scaled by sv, into raw (conv1ab_wino24h.hip)
  auto conv1a_block = [&](int half, int j, int x0, int y0, float svb) __attribute__((always_inline)) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    {
      const int pp = (3 * tq + j) * 16 + n;
      const int py = gpy[j], px = gpx[j];
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      const float m = (pp < RH * RW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? svb : 0.f;
      const float* ip = imgt + py * IMG_W + px;
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
        float* o = rawt + pp * RSH + cl * 16 + 4 * kq;
        *reinterpret_cast<f32x2*>(o) = (f32x2){d[0], d[1]};
        *reinterpret_cast<f32x2*>(o + 2) = (f32x2){d[2], d[3]};
      }
    }
  };
  auto conv1a_half = [&](int half, int x0, int y0, float svb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) conv1a_block(half, j, x0, y0, svb);
  };
  auto v_store2 = [&](int pos, f16x2 h, f16x2 m) __attribute__((always_inline)) {
    *reinterpret_cast<f16x2*>(vwr + pos * 512) = h;
    *reinterpret_cast<f16x2*>(vwr + VPLANE + pos * 512) = m;
  };
  auto transform = [&]() __attribute__((always_inline)) {
    f32x2 r1[6], r2[6], rx[6];
    auto row_load = [&](f32x2 (&d)[6], int row) __attribute__((always_inline)) {
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) d[bb] = *reinterpret_cast<const f32x2*>(rp + (row * RW + bb) * RSH);
    };
    auto row_out = [&](int i, f32x2 sg2, const f32x2 (&bq)[6], const f32x2 (&aq)[6]) __attribute__((always_inline)) {
      f32x2 o[6], T[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, bq[bb], aq[bb]);
      const W24Half hb = w24_batch_a(o, m5);
      w24_batch_b(o, hb, T);
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        f16x2 h, m;
        split_h2(T[jj], h, m);
        v_store2(jj * 4 + i, h, m);
      }
    };
    row_load(r1, 1);
    row_load(r2, 2);
    row_load(rx, 0);
    row_out(1, one2, r2, r1);        // r1 + r2
    row_out(2, mone2, r1, r2);       // r2 - r1
    row_out(0, mone2, r2, rx);       // r0 - r2
    row_load(rx, 3);
    row_out(3, mone2, rx, r1);       // r1 - r3
  };
  // (each phase carries half a conv1a in three pieces, after positions 1, 5 and 9 -- chunk 0's this pair's channels 32..63, chunk
  // 1's the NEXT pair's channels 0..31 (`half` = 0, the next pair's coordinates and scale): their fp32 MFMAs take their operands from
  // LDS and registers, and fill matrix-pipe time in which the fp16 MFMAs wait for U on the L1 data path)
  auto mfma_phase = [&](auto firstc, int c, int x0, int y0, float svb) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    f16x8 bq[2][4];              // [buffer][K h, K m, S h, S m]
    auto b_load = [&](int buf, int lp) __attribute__((always_inline)) {
      const int po = lp * 512;
      bq[buf][0] = *reinterpret_cast<const f16x8*>(vrdK + po);
      bq[buf][1] = *reinterpret_cast<const f16x8*>(vrdK + VPLANE + po);
      bq[buf][2] = *reinterpret_cast<const f16x8*>(vrdS + po);
      bq[buf][3] = *reinterpret_cast<const f16x8*>(vrdS + VPLANE + po);
    };
    b_load(0, 0);
#pragma unroll
    for (int lp = 0; lp < NLP; ++lp) {
      const int buf = lp & 1;
      if (lp + 1 < NLP) b_load(buf ^ 1, lp + 1);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 bKh = bq[buf][0], bKm = bq[buf][1], bSh = bq[buf][2], bSm = bq[buf][3];
      const f16x8 ah = __builtin_bit_cast(f16x8, ub[lp % RING][0]), am = __builtin_bit_cast(f16x8, ub[lp % RING][1]);
      accK[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bKm, FIRST ? zero4c : accK[lp], 0, 0, 0);
      accS[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bSm, FIRST ? zero4c : accS[lp], 0, 0, 0);
      accK[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bKh, accK[lp], 0, 0, 0);
      accS[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bSh, accS[lp], 0, 0, 0);
      accK[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bKh, accK[lp], 0, 0, 0);
      accS[lp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bSh, accS[lp], 0, 0, 0);
      {
        const int np = lp + RING;                  // the other chunk follows (U is tile independent)
        if (np < NLP) u_load(lp % RING, c, np);
        else u_load(lp % RING, c ^ 1, np - NLP);
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((lp & 3) == 1) {                         // (5790 us against 5817 with the three pieces behind the phase)
        conv1a_block(FIRST ? 1 : 0, lp >> 2, x0, y0, svb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  unsigned amax_run = 0;
  const int Ho = H >> 1, Wo = W >> 1;
  const bool outb = p.out_blocked != 0;
  const int opx = outb ? 32 : Cout * 4;
  __syncthreads();               // the maxima table is zeroed
  // Software pipeline over pairs (round 5, second form): the image patch of pair q + 1 goes to LDS and its conv1a channels 0..31 are
  // computed INSIDE pair q's second matrix phase (raw is free there: transform 1 has consumed channels 32..63), so a pair costs six
  // barriers instead of nine and no conv1a phase of its own.  x0 / y0 / bcur / lcur / sv: the pair being multiplied; tx / ty / b /
  // tlive: the tile whose image element `pre` holds (one pair ahead of the image in LDS until it is stored).
  auto image_to_lds = [&]() __attribute__((always_inline)) {
    // (addresses from the lane index recomputed here, the wave maximum by DPP: a kept LDS address or a shuffle's bpermute address is
    // spilled across the loop, and its scratch reload is waited for with vmcnt(0) behind the U refills just requested)
    const int ln = lane_now(), t8n = (wave_s & 3) * 64 + ln;
    if (t8n < IMG_N) img[tg * IMG_N + t8n] = pre;
    unsigned mb = __builtin_bit_cast(unsigned, fabsf(pre));       // the patch's largest |value| (bit patterns of non-negative floats order as integers)
    mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x141, 0xf, 0xf, true));   // row_half_mirror
    mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x140, 0xf, 0xf, true));   // row_mirror
    const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 0), (unsigned)__builtin_amdgcn_readlane((int)mb, 16));
    const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 32), (unsigned)__builtin_amdgcn_readlane((int)mb, 48));
    if (ln == 0) wmax[wave_s] = __builtin_bit_cast(float, max(m01, m23));
  };
  auto scale_from_lds = [&]() __attribute__((always_inline)) -> float {
    const float* wm = wmax + 4 * tg;
    const float imax = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v_scale_of_bound(fmaf(imax, p.c1a_l1, p.c1a_bmax)))));
  };
  int x0 = tx * OW, y0 = ty * OH, bcur = tlive ? b : 0, lcur = tlive;
  image_to_lds();
  next_tile();
  pre = fetch_px(q_begin + 1 < q_end);
  __syncthreads();               // image patches and maxima visible
  sv = scale_from_lds();
  conv1a_half(0, x0, y0, sv);
  __syncthreads();               // channels 0..31 of both patches complete
  for (int q = q_begin; q < q_end; ++q) {
    transform();
    __syncthreads();             // V complete; raw free
    mfma_phase(BoolC<true>{}, 0, x0, y0, sv);  // (+ conv1a channels 32..63)
    __syncthreads();             // channels 32..63 complete; V free; the image patches have been read for the last time
    const int x0n = tx * OW, y0n = ty * OH, bn = tlive ? b : 0, ln = tlive;     // the pair after this one (dead past the range: zeros)
    image_to_lds();
    next_tile();
    pre = fetch_px(q + 2 < q_end);
    transform();
    __syncthreads();             // V complete; raw free; the next pair's image patches and maxima visible
    const float svn = scale_from_lds();
    mfma_phase(BoolC<false>{}, 1, x0n, y0n, svn);     // (+ the next pair's conv1a channels 0..31)
    __syncthreads();             // every wave is past its B-operand reads: the V region takes the exchange; raw holds the next pair's channels 0..31

    // ---- the accumulators of the partner's tile go to LDS, the partner's of THIS wave's tile come back (conv3x3_wino24p.hip)
    typedef __attribute__((address_space(3))) f32x4* lds4p;
    const int lq = lane_now();
    const unsigned xw = lds0 + (unsigned)(wave_s * XCH + lq * 16), xr = lds0 + (unsigned)((wave_s ^ 4) * XCH + lq * 16);
    f32x4 bs4;
    {
      const int bo = __builtin_amdgcn_readfirstlane(cb * 16);
      typedef const f32x4 __attribute__((address_space(4)))* cf4p;
      const cf4p b4 = (cf4p)(bias_c + bo);
      const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
      const int kq2 = lq >> 4;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) bs4[qq] = kq2 == 0 ? b0[qq] : kq2 == 1 ? b1[qq] : kq2 == 2 ? b2[qq] : b3[qq];
    }
    // the output transform's row stage on this wave's three columns of both tiles; the six results for the partner's tile through
    // LDS, the partner's for this wave's tile back; then the column stage (conv3x3_wino24p.hip)
    f32x4 sK0[3], sK1[3];
    {
      f32x4 sS0[3], sS1[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        w24_out_rows(accS[jj * 4 + 0], accS[jj * 4 + 1], accS[jj * 4 + 2], accS[jj * 4 + 3], sS0[jj], sS1[jj]);
        *(lds4p)(uintptr_t)(xw + (2 * jj) * 1024) = sS0[jj];
        *(lds4p)(uintptr_t)(xw + (2 * jj + 1) * 1024) = sS1[jj];
      }
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) w24_out_rows(accK[jj * 4 + 0], accK[jj * 4 + 1], accK[jj * 4 + 2], accK[jj * 4 + 3], sK0[jj], sK1[jj]);
    }
    __syncthreads();
    f32x4 gs0[3], gs1[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      gs0[jj] = *(lds4p)(uintptr_t)(xr + (2 * jj) * 1024);
      gs1[jj] = *(lds4p)(uintptr_t)(xr + (2 * jj + 1) * 1024);
    }
    __syncthreads();             // (the next pair's transform overwrites the region)

    // ---- column stage, 2x2 max-pool (commutes with the positive scale), un-scale + bias, ReLU, stores (conv1ab_wino24h.hip)
    f32x4 y[2][4];
    {
      // (the column stage is instantiated in both branches of the wave-uniform order test: assembling one s0 / s1 in the branches and
      // transforming after the join cost ~60 register copies per pair and wave -- 6 % of the kernel's VALU instructions, round 6)
      auto cols = [&](const f32x4 (&a0)[3], const f32x4 (&a1)[3], const f32x4 (&b0)[3], const f32x4 (&b1)[3]) __attribute__((always_inline)) {
        const f32x4 s0[6] = {a0[0], a0[1], a0[2], b0[0], b0[1], b0[2]}, s1[6] = {a1[0], a1[1], a1[2], b1[0], b1[1], b1[2]};
        w24_out_cols(s0, s1, k8, y);
      };
      if (ph == 0) { cols(sK0, sK1, gs0, gs1); asm volatile("" ::: "memory"); }
      else { cols(gs0, gs1, sK0, sK1); asm volatile("; order 1" ::: "memory"); }
    }
    {
      const float inv = p.u_scale_inv / sv;
      const f32x4 inv4 = {inv, inv, inv, inv};
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int wr = (lq & 15) >> 2, wc = lq & 3;
      const int bu = __builtin_amdgcn_readfirstlane(bcur), lv = __builtin_amdgcn_readfirstlane(lcur);
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)bu * Ho * Wo * Cout), 0, lv ? Ho * Wo * Cout * 4 : 0, 0x00020000);
      const int choff = outb ? (cb * 2 + (lq >> 5)) * (Ho * Wo * 32) + ((lq >> 4) & 1) * 16 : (cb * 16 + 4 * (lq >> 4)) * 4;
      typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int oy = (y0 >> 1) + wr;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 mx4 = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]), __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1]));
        const f32x4 v = __builtin_elementwise_max(__builtin_elementwise_fma(mx4, inv4, bs4), zero4);
        const int ox = (x0 >> 1) + 2 * wc + hh;
        const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * opx + choff) : 0x7ffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)off, 0, 0);
        amax_run = max(amax_run, __builtin_bit_cast(unsigned, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))));
      }
      if (p.amax_out) {          // wave maximum by DPP, into the workgroup's LDS table (conv3x3_wino24p.hip)
        unsigned mb = amax_run;
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xb1, 0xf, 0xf, true));
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4e, 0xf, 0xf, true));
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x141, 0xf, 0xf, true));
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x140, 0xf, 0xf, true));
        const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 0), (unsigned)__builtin_amdgcn_readlane((int)mb, 16));
        const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 32), (unsigned)__builtin_amdgcn_readlane((int)mb, 48));
        const unsigned mw = max(m01, m23);
        if (lv && mw && lq == 0) atomicMax(amax_tab + (bu & (AMAX_SLOTS - 1)), mw);
        amax_run = 0;
      }
    }
    x0 = x0n; y0 = y0n; bcur = bn; lcur = ln; sv = svn;
  }
  if (p.amax_out) {
    __syncthreads();
    for (int i = tid; i < AMAX_SLOTS; i += 512)
      if (amax_tab[i]) atomicMax(p.amax_out + i, amax_tab[i]);
  }
}
}  // namespace

bool conv1ab_wino24p_supported(const ConvArgs& a) { return conv1ab_wino24h_supported(a); }

// at least TWO tile pairs per CU (the software pipeline over pairs needs a second pair to fill; below that conv1ab_wino24h, with twice the
// workgroups, fills more of the chip)
bool conv1ab_wino24p_preferred(const ConvArgs& a) {
  if (!conv1ab_wino24p_supported(a)) return false;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    ncu = prop.multiProcessorCount;
  }
  const long long tiles = (long long)((a.W + OW - 1) / OW) * ((a.H + OH - 1) / OH) * a.B;
  return (tiles + NG - 1) / NG >= 2LL * ncu;
}

hipError_t launch_conv1ab_wino24p(const ConvArgs& a, hipStream_t s) {
  if (!conv1ab_wino24p_supported(a)) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH, ntiles = tiles_x * tiles_y * a.B;
  const int npairs = (ntiles + NG - 1) / NG;
  const size_t lds = (size_t)NG * VGRP * 2 + (size_t)(NG * RAWSZ + NG * IMG_N + 8) * sizeof(float) + AMAX_SLOTS * sizeof(unsigned);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(conv1ab_wino24p), (int)lds, attr);
  const dim3 grid((unsigned)(npairs < ncu ? npairs : ncu));     // persistent: one workgroup per CU
  last_form = "conv1ab_wino24p:f16x2";
  hipLaunchKernelGGL(conv1ab_wino24p, grid, dim3(512), lds, s, a, tiles_x, tiles_y, ntiles);
  return hipGetLastError();
}

}  // namespace imx
