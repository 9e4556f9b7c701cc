// conv3x3_wino24p.hip -- conv3x3_wino24h.hip's layer (Winograd F(2x4, 3x3), both transformed operands as two fp16 planes, three plane
// products on v_mfma_f32_16x16x32_f16; superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123) with every U fragment used
// for TWO tiles and the 24 positions SPLIT over two waves (round 5).
//
// Why.  The phase clocks of conv3x3_wino24h (tools/ubench/conv_h_bench.cpp -DH_TRACE) say its 72-MFMA phase takes
// ~4 k cycles per chunk against 1.2 k of MFMA issue, and 5.6 k with a ring of four positions instead of six: the phase is paced by
// the L2 latency of the wave's own U stream -- 24 positions x ~1 k cycles / ring depth -- and the ring cannot grow (96 accumulators +
// 48 ring registers + the patch in flight + the transform = 242 of the 256 registers two waves per SIMD leave each).  A first
// attempt, conv3x3_wino24u (one wave per SIMD with 512 registers, a U fragment against the B operands of two tiles: half the U
// bytes, twice the time per position), was correct and SLOWER (2.35 against 2.08 ms on conv2a): with one wave per SIMD every
// instruction -- scalar address arithmetic and waits included -- costs its 4-5 issue cycles, and the transform and the epilogue are
// 1.9 k and 2.3 k instructions per tile pair.  This kernel keeps two waves per SIMD AND the reuse:
//   * workgroup = 8 waves = one PAIR of 8x16-pixel tiles x 64 output channels, one workgroup per CU; wave (cb, ph) owns channel
//     block cb and the positions of transformed COLUMNS 3 ph .. 3 ph + 2 (twelve of the 24) of BOTH tiles: 96 accumulators as before;
//   * its U fragment of a position meets the B operands of both tiles: 24 U loads per chunk and wave for 72 MFMAs (48 before), and a
//     ring of six positions is now half of the wave's positions -- a refill has ~1.2 k cycles (six positions x six MFMAs x two
//     waves) to arrive;
//   * the input transform is split by (8-channel sub-patch, tile) over the eight waves: a wave reads each row of its sub-patch
//     ONCE and produces all four transformed rows (the row-per-wave split of conv3x3_wino24h reads every row twice);
//   * at the end of an item each wave runs the ROW stage of the output transform on its three columns of both tiles, the two waves
//     of a channel block exchange the six results of the tile they do not finish through LDS (6 KB each way, in the V region, which
//     is free then; the first form exchanged the twelve accumulators) and each runs the column stage and the rest of
//     conv3x3_wino24h's epilogue on ITS tile: every output sees the arithmetic of conv3x3_wino24h in the same order -- the two
//     kernels agree bit for bit (tests/test_gpu_superpoint.py), so every parity statement made for that kernel holds for this one;
//   * the per-image maxima and the bias come through the scalar cache (uniform addresses; as vector loads they were waited for with
//     vmcnt(0) behind the U refills and the patch loads just issued); the V stores are plain ds_write2st64_b32 (ds_write_addtid_b32
//     was tried: twice the rate, four instructions per store with its m0 write and wait state, no gain).
// U layout, scales (ConvArgs::amax_in / amax_out, u_scale_inv) and the blocked / NHWC activation layouts are conv3x3_wino24h's.
//
// LDS: V 2 tiles x 48 KB + raw patches 2 x 30 KB + 1 KB of maxima = 157 KB, one workgroup per CU.
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per tile (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // input patch (pad-1 halo)
constexpr int RSC = 10;                        // raw sub-patch: pixel stride (8 channels + 2), as in conv3x3_wino24.hip
constexpr int RAWC = 192 * RSC;                // 180 pixels + pad, floats per 8-channel sub-patch
constexpr int NSUB = 4;                        // 8-channel sub-patches per chunk
constexpr int NG = 2;                          // tiles per workgroup
constexpr int CKH = 32, NT = 64, NPOS = 24;
constexpr int NLP = 12;                        // positions per wave: transformed COLUMNS 3 ph .. 3 ph + 2, all four rows = positions 12 ph + lp, lp = jj*4 + i
constexpr int VPLANE = NPOS * 4 * 16 * 8;      // halves per plane of a tile (24576 bytes)
constexpr int VGRP = 2 * VPLANE;               // halves per tile
constexpr int UPOS = 2 * 4 * 64 * 8;           // halves of U per (item block, chunk, position): [plane][channel block][lane][8]
constexpr int AMAX_SLOTS = 256;                // image b -> slot b % 256 (conv3x3_wino24h.hip)
constexpr int RING = 6;                        // of the wave's twelve positions, in flight (NLP % RING == 0)
constexpr int XCH = 6 * 64 * 16;               // bytes of one wave's exchange block: the row stage's six results for the partner's tile
constexpr unsigned OOB = 0x7ffffff0u;          // byte offset beyond any image: buffer loads return 0
#ifdef P_TRACE
// phase clocks (tools/ubench/conv_h_bench.cpp, -DP_TRACE): cycles of wave 0 of every 16th workgroup in each phase of chunk_step
__device__ long long p_trace_buf[16 * 16];
#define P_STAMP(i_) { const long long t_ = __builtin_amdgcn_s_memtime(); tr[i_] += t_ - tlast; tlast = t_; }
#else
#define P_STAMP(i_)
#endif

struct Tile { int b, y0, x0, live; };
template <bool V>
struct BoolC { static constexpr bool value = V; };

// x = h + m in fp16, two values at a time (conv3x3_wino24h.hip).  (The residual and its conversion as one mixed-precision fma each --
// v_fma_mixlo_f16 / v_fma_mixhi_f16, three instructions instead of four, the same bits -- measured the same: 1917 vs 1916 us on conv2a.)
__device__ __forceinline__ void split_h2(f32x2 x, f16x2& h, f16x2& m) {
  unsigned lo_u, hi_u;
  asm("s_mov_b32 %0, 0x0000bc00" : "=s"(lo_u));
  asm("s_mov_b32 %0, 0xbc000000" : "=s"(hi_u));
  const f16x2 lo = __builtin_bit_cast(f16x2, lo_u), hi = __builtin_bit_cast(f16x2, hi_u);
  h[0] = (_Float16)x[0]; h[1] = (_Float16)x[1];
  const float r0 = __builtin_amdgcn_fdot2(h, lo, x[0], false);
  const float r1 = __builtin_amdgcn_fdot2(h, hi, x[1], false);
  m[0] = (_Float16)r0; m[1] = (_Float16)r1;
}

// the lane index, recomputed where it is called (a volatile asm is not hoisted out of the main loop: values derived from a kept
// lane index are spilled to scratch there, and a scratch reload is a vector-memory operation that waits for the loads in flight)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// s_v of an image: 32 x its largest |input| (>= the bound 20 max|d| of the transformed patch) goes to 2^13 (conv3x3_wino24h.hip),
// and its reciprocal (a power of two either way: equal to conv3x3_wino24h's division)
__device__ __forceinline__ float v_scale(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (262u - e) << 23);
}
__device__ __forceinline__ float v_scale_inv(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (e - 8u) << 23);
}

template <bool POOL, bool RELU, bool FASTW, bool INZ>
__global__ __launch_bounds__(512) void conv3x3_wino24p(ConvArgs p, int tiles_x, int tiles_y, int ntiles, int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
  float* raw = reinterpret_cast<float*>(smem_p + NG * VGRP * 2);               // [NG][NSUB][RAWC]   (V [NG][2][VPLANE] halves sits at 0)
  unsigned* amax_tab = reinterpret_cast<unsigned*>(raw + NG * NSUB * RAWC);    // [AMAX_SLOTS]: this workgroup's output maxima per image slot
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_p;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave_s & 3, ph = wave_s >> 2;         // matrix role: channel block, position half (transformed columns 3 ph .. 3 ph + 2)
  const int tq = wave_s & 3, tg = wave_s >> 2;         // transform / loader / epilogue role: 8-channel sub-patch, tile (tg == ph)
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CKH, ncob = Cout / NT;
  const int grid = (int)gridDim.x;
  // XCD-aware start index (conv3x3_wino24.hip)
  const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= nitems) return;
#ifdef P_PRIO
  if (ph == P_PRIO - 1) __builtin_amdgcn_s_setprio(2);      // experiment: the issue arbiter favours the older four waves
#endif
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wuh, 0, ncob * nchunk * NPOS * UPOS * 2, 0x00020000);
  const int uoff_lane = (cb * 64 + lane) * 16 + ph * NLP * (UPOS * 2);         // bytes: plane 0 of position 12 ph of a (block, chunk)
  const int img_bytes = H * W * Cin * 4;
  typedef const unsigned __attribute__((address_space(4)))* cu32p;
  typedef const float __attribute__((address_space(4)))* cf32p;
  const cu32p amax_c = (cu32p)(uintptr_t)p.amax_in;                            // written by the producing layer's launch: the scalar cache is clean at kernel start
  const cf32p bias_c = (cf32p)(uintptr_t)p.bias;

  // ---- input transform: lane = (channel pair tk, wtile tw) of sub-patch tq of tile tg; every transformed row i of B2^T
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const f32x2 m5 = {-5.f, -5.f}, one2 = {1.f, 1.f}, mone2 = {-1.f, -1.f};
  const float* rp = raw + (tg * NSUB + tq) * RAWC + ((2 * twr) * RW + 4 * twc) * RSC + 2 * tk;     // + (row * RW + column) * RSC

  // B-operand reads: lane = (wtile n = lane & 15, group kg = lane >> 4) -> 16 bytes at position * 1024 + lane * 16 of a plane.  One
  // opaque base per tile (the second tile's planes lie beyond the 64 KB an LDS offset field reaches from the first tile's base:
  // hipcc would materialise an address register per position), the wave's 12 ph folded in; K = the tile this wave finishes (tile
  // ph), S = the tile whose accumulators it sends to its partner
  const _Float16 *vrdK, *vrdS;
  {
    unsigned a = lds0 + (unsigned)(ph * (VGRP * 2) + ph * NLP * 1024 + lane * 16);
    unsigned b = lds0 + (unsigned)((1 - ph) * (VGRP * 2) + ph * NLP * 1024 + lane * 16);
    asm volatile("" : "+v"(a), "+v"(b));
    vrdK = (const _Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)a);
    vrdS = (const _Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)b);
  }

  // ---- loader: the 256 threads of tile tg -> two (pixel, channel half) float4 of each of its 10x18x8 sub-patches
  // (conv3x3_wino24.hip's table; the pixel coordinates are recomputed in loader_tile, once per item, not kept across the main loop)
  auto loader_slot = [&](int t, int k, int& py, int& px_, int& half) __attribute__((always_inline)) {
    const int e = (k == 1 && t + 256 < RH * RW * 2) ? t + 256 : t;
    const int px = p.in_blocked ? e >> 1 : e % (RH * RW);
    half = p.in_blocked ? e & 1 : e / (RH * RW);
    py = px / RW - 1;
    px_ = px % RW - 1;
    return px * RSC + half * 4;
  };
  // Tile-swizzled input (in_blocked == 2; round 5): the producer (this kernel, out_blocked == 2) wrote the tensor as
  // [image][tile row][tile column][16-channel block][r][x][channel quarter][wtile][4] -- its accumulators' own lane order, 1 KB per store
  // instruction -- so a tile's 32 channels of a chunk are 16 KB contiguous: thread t takes the float4 t, t + 256, t + 512, t + 768 of
  // them (pixel (2 wr + r, 4 wc + x): x = t >> 6, r = k & 1, block k >> 1) and up to two of the 52 x 8 float4 of the one-pixel halo
  // from the neighbouring tiles.
  constexpr bool inz = INZ;                  // (a template parameter: with both loaders in one kernel their state cost 16 spilled registers)
  auto halo_slot = [&](int t, int j, int& py, int& px_, int& piece) __attribute__((always_inline)) -> bool {
    const int hh = t + 256 * j, hp = hh >> 3;
    piece = hh & 7;
    py = hp < 18 ? 0 : hp < 36 ? RH - 1 : hp < 44 ? hp - 35 : hp - 43;
    px_ = hp < 18 ? hp : hp < 36 ? hp - 18 : hp < 44 ? 0 : RW - 1;
    return hh < 52 * 8;
  };
  int ldst[2], ldmain = 0;
  if constexpr (inz) {
    const int t = tid & 255, n = t & 15, kq = (t >> 4) & 3, x = t >> 6;
    ldmain = tg * NSUB * RAWC + (kq >> 1) * RAWC + ((1 + 2 * (n >> 2)) * RW + 1 + 4 * (n & 3) + x) * RSC + (kq & 1) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int py, px, piece;
      halo_slot(t, j, py, px, piece);
      ldst[j] = tg * NSUB * RAWC + ((piece >> 2) * 2 + ((piece & 3) >> 1)) * RAWC + (py * RW + px) * RSC + (piece & 1) * 4;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) { int a, b, c; ldst[k] = tg * NSUB * RAWC + loader_slot(tid & 255, k, a, b, c); }
  }
  // an item = (pair of consecutive tiles, 64-channel output block); a wave only ever needs ITS tile of the pair.  Items vb + k grid:
  // the host makes grid a multiple of ncob whenever a workgroup has more than one item, so the output block is fixed (vb % ncob) and
  // the tile advances by dt = NG grid / ncob per item -- stepped with carries (three divisions per item and wave, plus the loader's
  // three, were a tenth of the epilogue's instructions)
  const int cob = __builtin_amdgcn_readfirstlane(vb % ncob);       // (a division leaves its uniform result in a vector register)
  const int dt = NG * (grid / ncob), dtx = __builtin_amdgcn_readfirstlane(dt % tiles_x), dty = __builtin_amdgcn_readfirstlane((dt / tiles_x) % tiles_y),
            dtb = __builtin_amdgcn_readfirstlane(dt / (tiles_x * tiles_y));
  Tile cur, nxt;                 // cur: the item whose chunks are multiplied; nxt: the one after it = the loader's, once it has wrapped
  {
    const int t = NG * (vb / ncob) + tg;
    cur.live = t < ntiles;
    cur.x0 = __builtin_amdgcn_readfirstlane(t % tiles_x);        // (tile coordinates; pixels = * OW, * OH)
    cur.y0 = __builtin_amdgcn_readfirstlane((t / tiles_x) % tiles_y);
    cur.b = __builtin_amdgcn_readfirstlane(t / (tiles_x * tiles_y));
  }
  auto step_tile = [&](const Tile& t) __attribute__((always_inline)) -> Tile {
    Tile r;
    int x = t.x0 + dtx, y = t.y0 + dty, b = t.b + dtb;
    if (x >= tiles_x) { x -= tiles_x; ++y; }
    if (y >= tiles_y) { y -= tiles_y; ++b; }
    r.x0 = x; r.y0 = y; r.b = b;
    r.live = b < p.B;
    return r;
  };
  int item_c = vb, lchunk = 0;
  __amdgpu_buffer_rsrc_t lrs;
  unsigned goff[2], gmain = 0, mmask = 0xfu;                      // (swizzled input: the halo slots, the tile's own block, its rows' validity)
  float lsv;                                                      // s_v of the loader's tile
  const int nblk_in = Cin / 16;                                   // swizzled input: 16-channel blocks of 8 KB per tile
  const int zimg = tiles_x * tiles_y * nblk_in * 8192;            // bytes per image
  const bool inb = p.in_blocked != 0;
  const int pxb = inb ? 8 * 4 : Cin * 4;                          // bytes from one pixel to the next
  const int sub_step = inb ? H * W * 8 * 4 : 8 * 4;               // bytes from one 8-channel group to the next
  auto loader_tile = [&](const Tile& t) __attribute__((always_inline)) {
    const int t8 = (wave_s & 3) * 64 + lane_now();
    int lpy[2], lpx[2], lhalf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) loader_slot(t8, k, lpy[k], lpx[k], lhalf[k]);
    const bool lv = t.live != 0;
    const int b = __builtin_amdgcn_readfirstlane(lv ? t.b : 0);
    lsv = v_scale(amax_c[b & (AMAX_SLOTS - 1)]);
    if constexpr (inz) {
      lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * (zimg >> 2)), 0, lv ? zimg : 0, 0x00020000);
      const int n = t8 & 15, x = t8 >> 6;
      gmain = (unsigned)((t.y0 * tiles_x + t.x0) * nblk_in * 8192 + t8 * 16);
      const bool xin = t.x0 * OW + 4 * (n & 3) + x < W;
      const int gy0 = t.y0 * OH + 2 * (n >> 2);
      mmask = ((xin && gy0 < H) ? 5u : 0u) | ((xin && gy0 + 1 < H) ? 10u : 0u);        // bit k: slot k (row r = k & 1) lies inside the image
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int py, px, piece;
        const bool used = halo_slot(t8, j, py, px, piece);
        const int gy = t.y0 * OH - 1 + py, gx = t.x0 * OW - 1 + px;
        const bool in = (int)used & (int)lv & (int)((unsigned)gy < (unsigned)H) & (int)((unsigned)gx < (unsigned)W);
        goff[j] = in ? (unsigned)((((gy >> 3) * tiles_x + (gx >> 4)) * nblk_in + (piece >> 2)) * 8192 + ((gy & 1) * 4 + (gx & 3)) * 1024 + (piece & 3) * 256 +
                                  (((gy & 7) >> 1) * 4 + ((gx & 15) >> 2)) * 16)
                     : OOB;
      }
      return;
    }
    lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * H * W * Cin), 0, lv ? img_bytes : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int gy = t.y0 * OH + lpy[k], gx = t.x0 * OW + lpx[k];
      const bool in = (int)lv & (int)((unsigned)gy < (unsigned)H) & (int)((unsigned)gx < (unsigned)W);
      goff[k] = in ? (unsigned)((gy * W + gx) * pxb + lhalf[k] * 16) : OOB;
    }
  };
  f32x4 rr[NSUB][2];
  float rr_sv = 1.f;                                              // the scale that goes with the registers' chunk
  unsigned rr_mask = 0xfu;                                        // (swizzled input: the validity bits that go with it)
  auto issue_load = [&]() __attribute__((always_inline)) {
    if constexpr (inz) {                                          // rr[k][0]: the tile's own block, slot k; rr[j][1]: the halo slots
      const int so = __builtin_amdgcn_readfirstlane(lchunk * 2 * 8192);
#pragma unroll
      for (int k = 0; k < 4; ++k) rr[k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)gmain, so + k * 4096, 0));
#pragma unroll
      for (int j = 0; j < 2; ++j) rr[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[j], so, 0));
      rr_sv = lsv;
      rr_mask = mmask;
      return;
    }
    const int so = __builtin_amdgcn_readfirstlane(lchunk * NSUB * sub_step);
#pragma unroll
    for (int q = 0; q < NSUB; ++q) {
      rr[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[0], so + q * sub_step, 0));
      rr[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[1], so + q * sub_step, 0));
    }
    rr_sv = lsv;
  };
  // (called right before issue_load: the registers of the previous patch are dead by then)
  auto advance_loader = [&]() __attribute__((always_inline)) {
    if (__builtin_expect(++lchunk == nchunk, 0)) {   // the loader moves on to this workgroup's next item
      lchunk = 0;
      nxt = step_tile(cur);
      if (item_c + grid >= nitems) nxt.live = 0;
      loader_tile(nxt);
      asm volatile("" ::: "memory");
    }
  };
  auto store_raw = [&]() __attribute__((always_inline)) {
    const f32x4 s4 = {rr_sv, rr_sv, rr_sv, rr_sv};
    if constexpr (inz) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {                               // (pixels of the padded tile outside the image hold the producer's values: zero here)
        const float mk = (rr_mask >> k) & 1u ? 1.f : 0.f;
        const f32x4 v = rr[k][0] * s4 * (f32x4){mk, mk, mk, mk};
        float* d = raw + ldmain + (k & 1) * (RW * RSC) + (k >> 1) * (2 * RAWC);
        *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[1]};
        *reinterpret_cast<f32x2*>(d + 2) = (f32x2){v[2], v[3]};
      }
      const int t8 = (wave_s & 3) * 64 + lane_now();
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (t8 + 256 * j < 52 * 8) {
          const f32x4 v = rr[j][1] * s4;
          float* d = raw + ldst[j];
          *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[1]};
          *reinterpret_cast<f32x2*>(d + 2) = (f32x2){v[2], v[3]};
        }
      return;
    }
#pragma unroll
    for (int q = 0; q < NSUB; ++q)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 v = rr[q][k] * s4;
        float* d = raw + q * RAWC + ldst[k];
        *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[1]};
        *reinterpret_cast<f32x2*>(d + 2) = (f32x2){v[2], v[3]};
      }
  };

  // ---- U ring: slot lp % RING holds local position lp's two planes (eight halves each per lane)
  u32x4 ub[RING][2];
  auto u_load = [&](int slot, int cobv, int chv, int lp) __attribute__((always_inline)) {
    const int pos = lp;                                                            // (+ 12 ph: in uoff_lane)
    const int so = __builtin_amdgcn_readfirstlane(((cobv * nchunk + chv) * NPOS + pos) * (UPOS * 2));
    ub[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so, 0);
    ub[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so + 4 * 64 * 16, 0);
  };

  for (int i = tid; i < AMAX_SLOTS; i += 512) amax_tab[i] = 0;       // (visible after the fill's barrier)
  // ---- pipeline fill: chunk 0 of the first item into raw, the U ring of chunk 0
  loader_tile(cur);
  issue_load();
  store_raw();
#pragma unroll
  for (int g = 0; g < RING; ++g) u_load(g, cob, 0, g);
  __syncthreads();

  f32x4 accK[NLP], accS[NLP];    // this wave's positions of the tile it finishes / of its partner's tile; an item's first chunk
                                 // starts every accumulator from a literal-zero C operand
  const f32x2 k8 = {8.f, 8.f};
  // ---- store geometry (conv3x3_wino24.hip)
  const int Ho_k = POOL ? H >> 1 : H, Wo_k = POOL ? W >> 1 : W;
  const bool outb = p.out_blocked != 0;
  constexpr bool fastw = FASTW;    // whole tiles only: (W % OW) == 0 && (!out_blocked || (H % OH) == 0) -- the stores need no mask
  const int opx = outb ? 8 * 4 : Cout * 4;
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};

  // V stores: h and m of a channel pair, 4 bytes each, at (position * 4 + tq) * 256 + lane * 4 of their planes (hipcc pairs the two
  // planes' stores into one ds_write2st64_b32).  (ds_write_addtid_b32 -- no address register, twice the LDS rate -- was built first:
  // with its m0 setup and the wait state a scalar write of m0 needs before an add-TID instruction it is FOUR instructions per store,
  // and this phase is bound by instruction issue, not by the LDS.)
  _Float16* const vwr = (_Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)(lds0 + (unsigned)(tg * (VGRP * 2) + tq * 256 + lane * 4)));
  auto v_store2 = [&](int pos, f16x2 h, f16x2 m) __attribute__((always_inline)) {
    *reinterpret_cast<f16x2*>(vwr + pos * 512) = h;
    *reinterpret_cast<f16x2*>(vwr + VPLANE + pos * 512) = m;
  };
  // phase A: this wave's sub-patch (scaled) -> its four transformed rows in the V planes.  Rows of B2^T: r0 - r2, r1 + r2, r2 - r1,
  // r1 - r3 (as fma(+-1, b, a): conv3x3_wino24h.hip's instruction)
#ifdef P_TRACE
  long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  auto transform = [&]() __attribute__((always_inline)) {
    f32x2 r1[6], r2[6], rx[6];
    auto row_load = [&](f32x2 (&d)[6], int row) __attribute__((always_inline)) {
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) d[bb] = *reinterpret_cast<const f32x2*>(rp + (row * RW + bb) * RSC);
    };
    auto row_out = [&](int i, f32x2 sg2, const f32x2 (&b)[6], const f32x2 (&a)[6]) __attribute__((always_inline)) {
      f32x2 o[6], T[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, b[bb], a[bb]);
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
    P_STAMP(9)
    row_out(2, mone2, r1, r2);       // r2 - r1
    P_STAMP(10)
    row_out(0, mone2, r2, rx);       // r0 - r2
    P_STAMP(11)
    row_load(rx, 3);
    row_out(3, mone2, rx, r1);       // r1 - r3
    P_STAMP(12)
  };
  // phase B: 72 MFMAs; per position one U fragment (two planes) against the B operands of both tiles; the U slot is refilled in
  // place with the wave's position lp + RING (of this chunk, or of the next chunk / the next item's block)
  auto mfma_phase = [&](auto firstc, int c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    // (c through an opaque scalar: for an item's first chunk, c == 0, hipcc hoisted (cob nchunk + nch) out of the item loop as a VECTOR
    // value, spilled it, and reloaded it in the middle of this phase -- a scratch reload is a vector-memory operation that is waited
    // for with vmcnt(0), i.e. behind the six U refills just issued)
    int cs = c;
    asm volatile("" : "+s"(cs));
    const bool lastc = cs + 1 == nchunk;
    const int nch = lastc ? 0 : cs + 1;
    // the B operands of position lp + 1 are requested beneath the MFMAs of position lp (without this the phase waits for the LDS
    // once per position -- ~270 cycles per position and wave against 96 of MFMAs: the trace of the first build)
    f16x8 bq[2][4];              // [buffer][K h, K m, S h, S m]
    auto b_load = [&](int buf, int lp) __attribute__((always_inline)) {
      const int po = lp * 512;                               // halves (the wave's 12 ph sits in the bases)
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
        const int np = lp + RING;
        if (np < NLP) u_load(lp % RING, cob, c, np);
        else u_load(lp % RING, cob, nch, np - NLP);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The patches of chunk s + 1 are requested at the START of chunk step s (after the barrier that closes step s - 1, and after the
  // epilogue when that step ended an item): their registers are dead through the epilogue and the loader's item bookkeeping.
  auto chunk_step = [&](auto firstc, int c) __attribute__((always_inline)) {
    P_STAMP(5)                     // (epilogue and item bookkeeping)
    advance_loader();
    issue_load();
    P_STAMP(8)                     // item bookkeeping + patch loads issued
    transform();
    P_STAMP(0)
    __syncthreads();               // V complete; raw free
    P_STAMP(1)
    mfma_phase(firstc, c);
    P_STAMP(2)
    store_raw();                   // the next chunk's patch (requested at the start of this step)
    P_STAMP(3)
    __syncthreads();               // raw complete; V free
    P_STAMP(4)
  };

  unsigned amax_run = 0;           // this lane's largest stored value of the current tile (bit pattern; values >= 0 after ReLU, |.| otherwise)
#pragma unroll 1
  for (;;) {
    chunk_step(BoolC<true>{}, 0);
#pragma unroll 1
    for (int c = 1; c < nchunk; ++c) chunk_step(BoolC<false>{}, c);

    // ---- item done.  The accumulators of the partner's tile go to LDS (the V region: every wave is past the barrier that closed the
    // last matrix phase), the partner's accumulators of THIS wave's tile come back: all 24 positions of 16 channels x 16 wtiles
    typedef __attribute__((address_space(3))) f32x4* lds4p;
    const int lq = lane_now();
    // (requested here, ahead of the exchange's barriers: scalar-cache latency)
    // the item's bias (sixteen scalars of this wave's channel block; a lane keeps the four of its quarter) and un-scale factor
    f32x4 bs4;
    {
      const int bo = __builtin_amdgcn_readfirstlane(cob * NT + cb * 16);
      const int kq = lq >> 4;
      typedef const f32x4 __attribute__((address_space(4)))* cf4p;
      const cf4p b4 = (cf4p)(bias_c + bo);
      const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
#pragma unroll
      for (int q = 0; q < 4; ++q) bs4[q] = kq == 0 ? b0[q] : kq == 1 ? b1[q] : kq == 2 ? b2[q] : b3[q];
    }
    Tile tl;                       // (through readfirstlane: hipcc does not see that these are wave-uniform and wraps every store
    tl.b = __builtin_amdgcn_readfirstlane(cur.b);            // whose descriptor derives from them in a waterfall loop)
    tl.y0 = __builtin_amdgcn_readfirstlane(cur.y0) * OH;
    tl.x0 = __builtin_amdgcn_readfirstlane(cur.x0) * OW;
    tl.live = __builtin_amdgcn_readfirstlane(cur.live);
    const float inv = p.u_scale_inv * v_scale_inv(amax_c[tl.b & (AMAX_SLOTS - 1)]);


    // The output transform's ROW stage (s0[j] = m0j + m1j + m2j, s1[j] = m1j - m2j - m3j: down the four rows of a transformed column)
    // needs one column's four accumulators -- which one wave holds, for both tiles.  It runs here, before the exchange, on this
    // wave's three columns of both tiles; the six results for the PARTNER's tile go through LDS (half of the twelve accumulators that
    // went before round 5's second form), the partner's six for this wave's tile come back, and the column stage runs on all six
    // columns: the instructions of conv3x3_wino24h's output transform on the same values -- bit-identical.
    const unsigned xw = lds0 + (unsigned)(wave_s * XCH + lq * 16), xr = lds0 + (unsigned)((wave_s ^ 4) * XCH + lq * 16);
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
    __syncthreads();               // (the next transform overwrites the region)
    P_STAMP(6)

    // ---- column stage, un-scale + bias, ReLU, (2x2 max-pool), stores straight from registers (conv3x3_wino24h.hip); columns
    // 3 ph .. 3 ph + 2 are this wave's, the other three the partner's
    f32x4 y[2][4];
    {
      // (the column stage is instantiated in both branches of the wave-uniform order test: assembling one s0 / s1 in the branches and
      // transforming after the join cost ~60 register copies per pair and wave, round 6)
      auto cols = [&](const f32x4 (&a0)[3], const f32x4 (&a1)[3], const f32x4 (&b0)[3], const f32x4 (&b1)[3]) __attribute__((always_inline)) {
        const f32x4 s0[6] = {a0[0], a0[1], a0[2], b0[0], b0[1], b0[2]}, s1[6] = {a1[0], a1[1], a1[2], b1[0], b1[1], b1[2]};
        w24_out_cols(s0, s1, k8, y);
      };
      if (ph == 0) { cols(sK0, sK1, gs0, gs1); asm volatile("" ::: "memory"); }
      else { cols(gs0, gs1, sK0, sK1); asm volatile("; order 1" ::: "memory"); }
    }
    P_STAMP(7)
    {
      const int lwr = (lq & 15) >> 2, lwc = lq & 3;
      const int chl = outb ? (cb * 2 + (lq >> 5)) * (Ho_k * Wo_k * 8 * 4) + ((lq >> 4) & 1) * 16 : (cb * 16 + 4 * (lq >> 4)) * 4;
      const f32x4 inv4 = {inv, inv, inv, inv};
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int Ho = Ho_k, Wo = Wo_k;
      typedef unsigned su32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int ibase = __builtin_amdgcn_readfirstlane(((POOL ? tl.y0 >> 1 : tl.y0) * Wo + (POOL ? tl.x0 >> 1 : tl.x0)) * opx +
                                                       (outb ? cob * (NT / 8) * (Ho * Wo * 8 * 4) : cob * NT * 4));
      const int fbase = fastw ? ibase : 0;
      // a dead tile (the odd tile out at the end of the grid) stores through an empty descriptor
      // Tile-swizzled output (out_blocked == 2, layers without a pool whose consumer is this kernel too; round 5): the accumulators'
      // own lane order -- [image][tile][16-channel block][r][x][channel quarter = lane >> 4][wtile = lane & 15][4] -- so that a store
      // instruction's 64 lanes write 1 KB contiguous: 16 cycles of the CU's store unit instead of the 56 it takes for 64 separate
      // 16-byte pieces (tools/ubench/store_rate.hip; with a pixel-major layout the pieces are four pixels apart).  Whole padded tiles
      // are written: the consumer's loader zeroes what lies outside the image.
      const bool outz = !POOL && p.out_blocked == 2;
      const int zbase = outz ? __builtin_amdgcn_readfirstlane((((tl.y0 / OH) * tiles_x + tl.x0 / OW) * (Cout / 16) + cob * 4 + cb) * 8192) : 0;
      const size_t zimg_out = (size_t)tiles_x * tiles_y * (Cout / 16) * 2048;        // floats per image
      const __amdgpu_buffer_rsrc_t ors =
          outz ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)tl.b * zimg_out + (zbase >> 2)), 0, tl.live ? 8192 : 0, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)tl.b * Ho * Wo * Cout + (fbase >> 2)), 0, tl.live ? Ho * Wo * Cout * 4 - fbase : 0, 0x00020000);
      auto note = [&](const f32x4& v) __attribute__((always_inline)) {
        const float mm = RELU ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax_run = max(amax_run, __builtin_bit_cast(unsigned, mm));
      };
      if constexpr (POOL) {
        const int oy = (tl.y0 >> 1) + lwr;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4 mx4 = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]), __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1]));
          f32x4 v = __builtin_elementwise_fma(mx4, inv4, bs4);
          if (RELU) v = __builtin_elementwise_max(v, zero4);
          note(v);
          const int ox = (tl.x0 >> 1) + 2 * lwc + hh;
          const int so_ = (lwr * Wo + 2 * lwc + hh) * opx + chl;
          const unsigned off = fastw ? (unsigned)so_ : (oy < Ho && ox < Wo) ? (unsigned)(so_ + ibase) : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, (int)off, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            f32x4 v = __builtin_elementwise_fma(y[r][x], inv4, bs4);
            if (RELU) v = __builtin_elementwise_max(v, zero4);
            note(v);
            const int oy = tl.y0 + 2 * lwr + r, ox = tl.x0 + 4 * lwc + x;
            const int so_ = ((2 * lwr + r) * Wo + 4 * lwc + x) * opx + chl;
            const unsigned off = outz ? (unsigned)((r * 4 + x) * 1024 + lq * 16) : fastw ? (unsigned)so_ : (oy < Ho && ox < Wo) ? (unsigned)(so_ + ibase) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, (int)off, 0, 0);
          }
      }
      P_STAMP(13)
      // the image's output maximum for the NEXT layer's s_v: into this workgroup's LDS table, flushed once at the end of the kernel
      // (conv3x3_wino24h.hip).  Wave maximum by DPP (row butterflies, then the four rows' lane 0 through SGPRs): a shuffle
      // reduction keeps six bpermute addresses alive across the main loop, and an atomic from every lane is turned by hipcc's atomic
      // optimizer into a 64-iteration scan loop
      if (p.amax_out) {
        unsigned mb = amax_run;
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x141, 0xf, 0xf, true));   // row_half_mirror
        mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x140, 0xf, 0xf, true));   // row_mirror
        const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 0), (unsigned)__builtin_amdgcn_readlane((int)mb, 16));
        const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 32), (unsigned)__builtin_amdgcn_readlane((int)mb, 48));
        const unsigned mw = max(m01, m23);
        if (tl.live && mw && lq == 0) atomicMax(amax_tab + (tl.b & (AMAX_SLOTS - 1)), mw);
        amax_run = 0;
      }
      P_STAMP(14)
    }
    item_c += grid;
    if (item_c >= nitems) break;
    cur = nxt;
  }
  if (p.amax_out) {
    __syncthreads();
    for (int i = tid; i < AMAX_SLOTS; i += 512)
      if (amax_tab[i]) atomicMax(p.amax_out + i, amax_tab[i]);
  }
#ifdef P_TRACE
#ifdef P_TRACE_WAVES          // the eight waves of workgroup 0 instead of wave 0 of sixteen workgroups
  if (lane == 0 && blockIdx.x == 0)
    for (int i = 0; i < 16; ++i) p_trace_buf[wave_s * 16 + i] = tr[i];
#else
  if (tid == 0 && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 16)
    for (int i = 0; i < 16; ++i) p_trace_buf[(blockIdx.x >> 4) * 16 + i] = tr[i];
#endif
#endif
}

template <bool POOL, bool RELU, bool FASTW, bool INZ>
hipError_t launch_p2(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int nitems = ((ntiles + NG - 1) / NG) * (a.Cout / NT);
  const size_t lds = (size_t)NG * VGRP * 2 + (size_t)NG * NSUB * RAWC * sizeof(float) + AMAX_SLOTS * sizeof(unsigned);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  auto k = conv3x3_wino24p<POOL, RELU, FASTW, INZ>;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(k), (int)lds, attr);
  const int ncob = a.Cout / NT;
  const dim3 grid((unsigned)(nitems <= ncu ? nitems : ncu - ncu % ncob));     // persistent: one workgroup per CU; a multiple of ncob (item stepping)
  if (grid.x == 0) return hipErrorInvalidValue;
  last_form = "conv3x3_wino24p:f16x2";
  hipLaunchKernelGGL(k, grid, dim3(512), lds, s, a, tiles_x, tiles_y, ntiles, nitems);
  return hipGetLastError();
}
template <bool POOL, bool RELU, bool FASTW>
hipError_t launch_p(const ConvArgs& a, hipStream_t s) {
  return a.in_blocked == 2 ? launch_p2<POOL, RELU, FASTW, true>(a, s) : launch_p2<POOL, RELU, FASTW, false>(a, s);
}
}  // namespace

#ifdef P_TRACE
void conv_p_trace_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(p_trace_buf), sizeof(long long) * 16 * 16); }
#endif

// (the tile-swizzled tensors are addressed with 32-bit byte offsets of the PADDED tile grid, which must stay below the out-of-bounds
// sentinel too: an image just under the limit with H % 8 or W % 16 != 0 would wrap -- ADVICE r5)
bool conv3x3_wino24p_supported(const ConvArgs& a) {
  if (!conv3x3_wino24h_supported(a)) return false;
  const size_t padded = (size_t)((a.H + OH - 1) / OH * OH) * ((a.W + OW - 1) / OW * OW) * 4;
  return padded * a.Cin < (size_t)OOB && padded * a.Cout < (size_t)OOB;
}

// The pair form has one workgroup per CU: below one item per CU (one or two image pairs at the coarse layers) the tile-per-workgroup
// form, with twice the items and two workgroups per CU, fills more of the chip.
bool conv3x3_wino24p_preferred(const ConvArgs& a) {
  if (!conv3x3_wino24p_supported(a)) return false;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    ncu = prop.multiProcessorCount;
  }
  const long long tiles = (long long)((a.W + OW - 1) / OW) * ((a.H + OH - 1) / OH) * a.B;
  return ((tiles + NG - 1) / NG) * (a.Cout / NT) >= ncu;
}

hipError_t launch_conv3x3_wino24p(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wino24p_supported(a)) return hipErrorInvalidValue;
  const bool fastw = (a.W % OW) == 0 && (!a.out_blocked || (a.H % OH) == 0);
  if (fastw) {
    if (a.pool) return a.relu ? launch_p<true, true, true>(a, s) : launch_p<true, false, true>(a, s);
    return a.relu ? launch_p<false, true, true>(a, s) : launch_p<false, false, true>(a, s);
  }
  if (a.pool) return a.relu ? launch_p<true, true, false>(a, s) : launch_p<true, false, false>(a, s);
  return a.relu ? launch_p<false, true, false>(a, s) : launch_p<false, false, false>(a, s);
}

}  // namespace imx
