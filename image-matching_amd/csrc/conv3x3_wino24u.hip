// conv3x3_wino24u.hip -- conv3x3_wino24h.hip's layer (Winograd F(2x4, 3x3), both transformed operands as two fp16 planes, three plane
// products on v_mfma_f32_16x16x32_f16; superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123) with every U fragment used
// for TWO tiles (round 5).
//
// Why.  In conv3x3_wino24h a wave streams its own U fragments from L2 -- 48 KB per 32-channel chunk and wave for 72 MFMAs -- and
// holds a ring of a few positions of them: the phase clocks of that kernel (tools/ubench/conv_h_bench.cpp -DH_TRACE) show 14.4 k cycles
// per chunk on conv2a against 1.2 k of MFMA issue, 1.2 k of transform issue per wave, and wherever a vmcnt(0) or an LDS wait is removed
// the waiting moves to the next phase: the chunk is paced by the U stream (L2 latency against a ring that covers ~200 cycles, and two
// waves per SIMD whose packed-fp32 transform instructions cost 10 cycles each beside the other wave's MFMAs:
// tools/ubench/wino_issue.hip).  Here a workgroup owns a PAIR of 8x16-pixel tiles x 64 output channels and there is one workgroup
// (four waves, one per SIMD, up to 512 registers each) per CU:
//   * a wave's U fragment of a position meets the B operands of both tiles: half the U bytes per MFMA, and twelve MFMAs (192 cycles)
//     per position instead of six -- with a ring of eight positions the refill of a slot has ~1.5 k cycles to arrive;
//   * one wave per SIMD: the transform phase runs its packed instructions at their own rate (4.5 cycles) instead of beside another
//     wave's MFMAs; every LDS read is requested a step ahead of its use (the B operands of position p + 1 beneath the MFMAs of
//     position p, the patch rows of sub-patch q + 1 beneath the arithmetic of q), because no second wave hides it;
//   * the per-image maxima and the bias come through the scalar cache (uniform addresses): as vector loads they were waited for
//     with vmcnt(0), behind the U refills and the patch loads just issued.
// The arithmetic -- transforms, the split, the order of the three plane products and of the chunks, the epilogue -- is
// conv3x3_wino24h.hip's, instruction for instruction per output: the two kernels agree bit for bit (tests/test_gpu_superpoint.py).
// U layout, scales (ConvArgs::amax_in / amax_out, u_scale_inv) and the blocked / NHWC activation layouts are shared with it.
//
// LDS: V 2 tiles x 48 KB + raw patches 2 x 30 KB + 1 KB of maxima = 157 KB, one workgroup per CU.
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// (v_store below names m0 as clobbered: hipcc reserves m0 and warns about any mention of it; nothing else in this kernel uses it)
#pragma clang diagnostic ignored "-Winline-asm"

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
constexpr int VPLANE = NPOS * 4 * 16 * 8;      // halves per plane of a tile (24576 bytes)
constexpr int VGRP = 2 * VPLANE;               // halves per tile
constexpr int UPOS = 2 * 4 * 64 * 8;           // halves of U per (item block, chunk, position): [plane][wave][lane][8]
constexpr int AMAX_SLOTS = 256;                // image b -> slot b % 256 (conv3x3_wino24h.hip)
constexpr int RING = 6;                        // positions of U in flight (24 % RING == 0; with 8 the 256 AGPRs -- 192 accumulators + the ring -- leave hipcc no register to park a VGPR in, and it spills to scratch)
constexpr unsigned OOB = 0x7ffffff0u;          // byte offset beyond any image: buffer loads return 0
#ifndef U_EXP
#define U_EXP 0                                // timing experiments (tools/ubench/conv_h_bench.cpp): 1 no U refills, 2 no transform, 3 no patch loads / stores
#endif
#ifdef U_TRACE
// phase clocks (tools/ubench/conv_h_bench.cpp, -DU_TRACE): cycles of wave 0 of every 16th workgroup in each phase of chunk_step
__device__ long long u_trace_buf[16 * 8];
#define U_STAMP(i_) { const long long t_ = __builtin_amdgcn_s_memtime(); tr[i_] += t_ - tlast; tlast = t_; }
#else
#define U_STAMP(i_)
#endif

struct Tile { int b, y0, x0, live; };
struct Item { Tile t[NG]; int cob; };
template <bool V>
struct BoolC { static constexpr bool value = V; };

// x = h + m in fp16, two values at a time (conv3x3_wino24h.hip)
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
// lane index were spilled to scratch there)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// s_v of an image: 32 x its largest |input| (>= the bound 20 max|d| of the transformed patch) goes to 2^13 (conv3x3_wino24h.hip)
__device__ __forceinline__ float v_scale(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (262u - e) << 23);
}

// 1 / v_scale (a power of two either way)
__device__ __forceinline__ float v_scale_inv(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (e - 8u) << 23);
}

template <bool POOL, bool RELU>
__global__ __launch_bounds__(256, 1) void conv3x3_wino24u(ConvArgs p, int tiles_x, int tiles_y, int ntiles, int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];
  _Float16* Vp = reinterpret_cast<_Float16*>(smem_u);                          // [NG][2][VPLANE]
  float* raw = reinterpret_cast<float*>(smem_u + NG * VGRP * 2);               // [NG][NSUB][RAWC]
  unsigned* amax_tab = reinterpret_cast<unsigned*>(raw + NG * NSUB * RAWC);    // [AMAX_SLOTS]: this workgroup's output maxima per image slot

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int cb = wave_s;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CKH, ncob = Cout / NT;
  const int grid = (int)gridDim.x;
  // XCD-aware start index (conv3x3_wino24.hip)
  const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= nitems) return;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wuh, 0, ncob * nchunk * NPOS * UPOS * 2, 0x00020000);
  const int uoff_lane = (cb * 64 + lane) * 16;                                 // bytes inside a plane of a position
  const int img_bytes = H * W * Cin * 4;
  typedef const unsigned __attribute__((address_space(4)))* cu32p;
  typedef const float __attribute__((address_space(4)))* cf32p;
  const cu32p amax_c = (cu32p)(uintptr_t)p.amax_in;                            // written by the producing layer's launch: the scalar cache is clean at kernel start
  const cf32p bias_c = (cf32p)(uintptr_t)p.bias;

  // ---- input transform roles: lane = (channel pair tk, wtile tw); transformed row i = wave (rows of B2^T)
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  const f32x2 sg2 = {sg, sg};
  const f32x2 m5 = {-5.f, -5.f};
  const float* rpa = raw + ((2 * twr + ra) * RW + 4 * twc) * RSC + 2 * tk;
  const float* rpb = raw + ((2 * twr + rb) * RW + 4 * twc) * RSC + 2 * tk;
  // V stores: position p = j*4 + wave, group q, wtile tw, channels 2 tk, 2 tk + 1 of the group -> halves ((p*4 + q)*16 + tw)*8 + 2 tk
  // (= bytes (p*4 + q) * 256 + lane * 4: v_store below)
  // B-operand reads: lane = (wtile n = lane & 15, group kg = lane >> 4) -> 16 bytes at position p * 1024 + lane * 16
  // (one opaque base per tile: the second tile's planes lie beyond the 64 KB an LDS instruction's offset field reaches from the first
  // tile's base, and hipcc then materialises a separate address register for every position -- 32 VGPRs)
  const _Float16* vrdg[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_u + (unsigned)(g * (VGRP * 2) + lane * 16);
    asm volatile("" : "+v"(a));
    vrdg[g] = (const _Float16*)((__attribute__((address_space(3))) unsigned char*)(uintptr_t)a);
  }

  // ---- loader: thread -> two (pixel, channel half) float4 of every 10x18x8 sub-patch of either tile (conv3x3_wino24.hip's table)
  // (the pixel coordinates are recomputed in loader_item, once per item, from an opaque copy of tid: kept across the main loop
  // they were spilled, and a scratch reload is a vector-memory operation that waits for the patch loads in flight)
  auto loader_slot = [&](int t, int k, int& py, int& px_, int& half) __attribute__((always_inline)) {
    const int e = (k == 1 && t + 256 < RH * RW * 2) ? t + 256 : t;
    const int px = p.in_blocked ? e >> 1 : e % (RH * RW);
    half = p.in_blocked ? e & 1 : e / (RH * RW);
    py = px / RW - 1;
    px_ = px % RW - 1;
    return px * RSC + half * 4;
  };
  int ldst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { int a, b, c; ldst[k] = loader_slot(tid, k, a, b, c); }
  auto decode = [&](int it) __attribute__((always_inline)) -> Item {
    Item r;
    r.cob = it % ncob;
    const int pair = it / ncob;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int t = NG * pair + g;
      r.t[g].live = t < ntiles;
      const int tt = r.t[g].live ? t : 0;
      r.t[g].x0 = (tt % tiles_x) * OW;
      r.t[g].y0 = ((tt / tiles_x) % tiles_y) * OH;
      r.t[g].b = tt / (tiles_x * tiles_y);
    }
    return r;
  };
  // cur: the item whose chunks are multiplied; nxt_cob: the output block of the one after it (its U block is prefetched during cur's
  // last chunk); litem: the loader's item (the loader runs one chunk ahead: chunk s + 1 is requested at the start of step s and stored at its end)
  int item_c = vb;
  Item cur = decode(vb);
  int nxt_cob = vb + grid < nitems ? (vb + grid) % ncob : cur.cob;
  int litem = item_c, lchunk = 0;
  __amdgpu_buffer_rsrc_t lrs[NG];
  unsigned goff[NG][2];
  float lsv[NG];                                                  // s_v of the loader's tiles
  const bool inb = p.in_blocked != 0;
  const int pxb = inb ? 8 * 4 : Cin * 4;                          // bytes from one pixel to the next
  const int sub_step = inb ? H * W * 8 * 4 : 8 * 4;               // bytes from one 8-channel group to the next
  auto loader_item = [&](const Item& it, bool live) __attribute__((always_inline)) {
    const int tq = wave_s * 64 + lane_now();                      // = tid, recomputed here (not kept alive across the main loop)
    int lpy[2], lpx[2], lhalf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) loader_slot(tq, k, lpy[k], lpx[k], lhalf[k]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool lv = live && it.t[g].live;
      const int b = __builtin_amdgcn_readfirstlane(lv ? it.t[g].b : 0);
      lrs[g] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * H * W * Cin), 0, lv ? img_bytes : 0, 0x00020000);
      lsv[g] = v_scale(amax_c[b & (AMAX_SLOTS - 1)]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gy = it.t[g].y0 + lpy[k], gx = it.t[g].x0 + lpx[k];
        goff[g][k] = (lv && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * pxb + lhalf[k] * 16) : OOB;
      }
    }
  };
  f32x4 rr[NG][NSUB][2];
  float rr_sv[NG];                                                // the scales that go with the registers' chunk
  auto issue_load = [&]() __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(lchunk * NSUB * sub_step);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int q = 0; q < NSUB; ++q) {
        rr[g][q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs[g], (int)goff[g][0], so + q * sub_step, 0));
        rr[g][q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs[g], (int)goff[g][1], so + q * sub_step, 0));
      }
      rr_sv[g] = lsv[g];
    }
  };
  // (called right before issue_load: the registers of the previous patch are dead by then)
  auto advance_loader = [&]() __attribute__((always_inline)) {
    if (__builtin_expect(++lchunk == nchunk, 0)) {   // the loader moves on to this workgroup's next item
      lchunk = 0;
      litem += grid;
      const bool live = litem < nitems;
      const Item lit = decode(live ? litem : vb);
      loader_item(lit, live);
      asm volatile("" ::: "memory");
    }
  };
  auto store_raw = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 s4 = {rr_sv[g], rr_sv[g], rr_sv[g], rr_sv[g]};
#pragma unroll
      for (int q = 0; q < NSUB; ++q)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x4 v = rr[g][q][k] * s4;
          float* d = raw + (g * NSUB + q) * RAWC + ldst[k];
          *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[1]};
          *reinterpret_cast<f32x2*>(d + 2) = (f32x2){v[2], v[3]};
        }
    }
  };

  // ---- U ring: slot p % RING holds position p's two planes (eight halves each per lane)
  u32x4 ub[RING][2];
  auto u_load = [&](int slot, int cobv, int chv, int pos) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(((cobv * nchunk + chv) * NPOS + pos) * (UPOS * 2));
    ub[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so, 0);
    ub[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so + 4 * 64 * 16, 0);
  };

  for (int i = tid; i < AMAX_SLOTS; i += 256) amax_tab[i] = 0;       // (visible after the fill's barrier)
  // ---- pipeline fill: chunk 0 of the first item into raw, the U ring of chunk 0
  loader_item(cur, true);
  issue_load();
  store_raw();
#pragma unroll
  for (int g = 0; g < RING; ++g) u_load(g, cur.cob, 0, g);
  __syncthreads();

  f32x4 acc[NG][NPOS];      // an item's first chunk starts every accumulator from a literal-zero C operand
  const f32x2 k8 = {8.f, 8.f};
  // ---- store offsets (conv3x3_wino24.hip)
  const int Ho_k = POOL ? H >> 1 : H, Wo_k = POOL ? W >> 1 : W;
  const bool outb = p.out_blocked != 0;
  const bool fastw = (W % OW) == 0 && (!outb || (H % OH) == 0);
  const int opx = outb ? 8 * 4 : Cout * 4;
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};

  const unsigned v_m0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_u + (unsigned)wave_s * 1024u;
  auto v_store = [&](int g, int plane, int jj, int q, unsigned bits) __attribute__((always_inline)) {
    const unsigned a = v_m0 + (unsigned)(g * (VGRP * 2) + plane * (VPLANE * 2) + jj * 4096 + q * 256);     // (one s_add per store)
    asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0" : : "v"(bits), "s"(a) : "memory", "m0");
  };
  // phase A: raw (scaled) -> V planes of both tiles; the twelve row reads of sub-patch k + 1 are requested before the arithmetic of k
  auto transform = [&]() __attribute__((always_inline)) {
    f32x2 xa[2][6], xb[2][6];
    auto t_load = [&](int k, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) {
        xa[buf][bb] = *reinterpret_cast<const f32x2*>(rpa + k * RAWC + bb * RSC);     // k = g * NSUB + q
        xb[buf][bb] = *reinterpret_cast<const f32x2*>(rpb + k * RAWC + bb * RSC);
      }
    };
    t_load(0, 0);
#pragma unroll
    for (int k = 0; k < NG * NSUB; ++k) {
      const int g = k / NSUB, q = k % NSUB, buf = k & 1;
      if (k + 1 < NG * NSUB) t_load(k + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x2 o[6], T[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, xb[buf][bb], xa[buf][bb]);
      const W24Half hb = w24_batch_a(o, m5);
      w24_batch_b(o, hb, T);
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        f16x2 h, m;
        split_h2(T[jj], h, m);
        // position jj*4 + wave, group q: bytes (pos * 4 + q) * 256 + lane * 4 of the plane -- a wave's store is 256 contiguous bytes
        // in lane order, which is what ds_write_addtid_b32 writes (address = M0 + offset + 4 * lane, no address register: twice
        // ds_write_b32's rate)
        v_store(g, 0, jj, q, __builtin_bit_cast(unsigned, h));
        v_store(g, 1, jj, q, __builtin_bit_cast(unsigned, m));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // phase B: 144 MFMAs; per position one U fragment (two planes) against the B operands of both tiles, which were requested beneath
  // the previous position's MFMAs; the U slot is refilled in place with position p + RING (of this chunk, or of the next chunk /
  // the next item's block)
  auto mfma_phase = [&](auto firstc, int c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    const bool lastc = c + 1 == nchunk;
    const int ncb = lastc ? nxt_cob : cur.cob, nch = lastc ? 0 : c + 1;
    f16x8 bq[2][NG][2];          // [buffer][tile][plane]
    auto b_load = [&](int buf, int pos) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bq[buf][g][0] = *reinterpret_cast<const f16x8*>(vrdg[g] + pos * 512);
        bq[buf][g][1] = *reinterpret_cast<const f16x8*>(vrdg[g] + VPLANE + pos * 512);
      }
    };
    b_load(0, 0);
#pragma unroll
    for (int pp = 0; pp < NPOS; ++pp) {
      const int buf = pp & 1;
      if (pp + 1 < NPOS) b_load(buf ^ 1, pp + 1);
      __builtin_amdgcn_sched_barrier(0);           // (left alone the scheduler sinks these reads to one MFMA before their use)
      const f16x8 ah = __builtin_bit_cast(f16x8, ub[pp % RING][0]), am = __builtin_bit_cast(f16x8, ub[pp % RING][1]);
      acc[0][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[buf][0][1], FIRST ? zero4c : acc[0][pp], 0, 0, 0);
      acc[1][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[buf][1][1], FIRST ? zero4c : acc[1][pp], 0, 0, 0);
      acc[0][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bq[buf][0][0], acc[0][pp], 0, 0, 0);
      acc[1][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bq[buf][1][0], acc[1][pp], 0, 0, 0);
      acc[0][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[buf][0][0], acc[0][pp], 0, 0, 0);
      acc[1][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[buf][1][0], acc[1][pp], 0, 0, 0);
      if (U_EXP != 1) {
        const int np = pp + RING;
        if (np < NPOS) u_load(pp % RING, cur.cob, c, np);
        else u_load(pp % RING, ncb, nch, np - NPOS);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#ifdef U_TRACE
  long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  // The patches of chunk s + 1 are requested at the START of chunk step s (after the barrier that closes step s - 1, and after the
  // epilogue when that step ended an item): the 64 registers they land in are dead through the epilogue and the loader's item
  // bookkeeping, which is where hipcc spilled (scratch reloads are vector-memory operations: each one waited for the patch loads
  // just issued -- 22 k cycles per epilogue in the first build).
  auto chunk_step = [&](auto firstc, int c) __attribute__((always_inline)) {
    U_STAMP(5)                     // (epilogue and item bookkeeping)
    advance_loader();
    if (U_EXP != 3) issue_load();
    if (U_EXP != 2) transform();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the V stores are inline assembly, which hipcc's wait counting does not see
    U_STAMP(0)
    __syncthreads();               // V complete; raw free
    U_STAMP(1)
    mfma_phase(firstc, c);
    U_STAMP(2)
    if (U_EXP != 3) store_raw();   // the next chunk's patches (requested at the start of the previous step)
    U_STAMP(3)
    __syncthreads();               // raw complete; V free
    U_STAMP(4)
  };

  unsigned amax_run = 0;           // this lane's largest stored value of the current tile (bit pattern; values >= 0 after ReLU, |.| otherwise)
#pragma unroll 1
  for (;;) {
    // the item's bias (sixteen scalars of this wave's channel block; a lane keeps the four of its quarter) and un-scale factors
    f32x4 bs4;
    {
      const int bo = __builtin_amdgcn_readfirstlane(cur.cob * NT + cb * 16);
      const int kq = lane >> 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float b0 = bias_c[bo + q], b1 = bias_c[bo + 4 + q], b2 = bias_c[bo + 8 + q], b3 = bias_c[bo + 12 + q];
        bs4[q] = kq == 0 ? b0 : kq == 1 ? b1 : kq == 2 ? b2 : b3;
      }
    }
    float inv[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) inv[g] = p.u_scale_inv * v_scale_inv(amax_c[__builtin_amdgcn_readfirstlane(cur.t[g].b & (AMAX_SLOTS - 1))]);
    chunk_step(BoolC<true>{}, 0);
#pragma unroll 1
    for (int c = 1; c < nchunk; ++c) chunk_step(BoolC<false>{}, c);

    // ---- item done: per tile the output transform, un-scale + bias, ReLU, (2x2 max-pool), stores straight from registers
    // (conv3x3_wino24h.hip); the lane's store offsets are derived here, from an opaque copy of the lane index (see loader_item)
    const int lq = lane_now();
    const int lwr = (lq & 15) >> 2, lwc = lq & 3;
    const int chl = outb ? (cb * 2 + (lq >> 5)) * (Ho_k * Wo_k * 8 * 4) + ((lq >> 4) & 1) * 16 : (cb * 16 + 4 * (lq >> 4)) * 4;
    int soff[POOL ? 2 : 8];
#pragma unroll
    for (int e = 0; e < (POOL ? 2 : 8); ++e) {
      const int oy = POOL ? lwr : 2 * lwr + (e >> 2), ox = POOL ? 2 * lwc + e : 4 * lwc + (e & 3);
      soff[e] = (oy * Wo_k + ox) * opx + chl;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // (through readfirstlane: hipcc does not see that an item's fields are wave-uniform, and wraps every store whose descriptor
      // derives from them in a waterfall loop)
      Tile tl;
      tl.b = __builtin_amdgcn_readfirstlane(cur.t[g].b);
      tl.y0 = __builtin_amdgcn_readfirstlane(cur.t[g].y0);
      tl.x0 = __builtin_amdgcn_readfirstlane(cur.t[g].x0);
      tl.live = __builtin_amdgcn_readfirstlane(cur.t[g].live);
      const f32x4 inv4 = {inv[g], inv[g], inv[g], inv[g]};
      f32x4 y[2][4];
      w24_output_transform(acc[g], k8, y);
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int Ho = Ho_k, Wo = Wo_k;
      typedef unsigned su32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int ibase = __builtin_amdgcn_readfirstlane(((POOL ? tl.y0 >> 1 : tl.y0) * Wo + (POOL ? tl.x0 >> 1 : tl.x0)) * opx +
                                                       (outb ? cur.cob * (NT / 8) * (Ho * Wo * 8 * 4) : cur.cob * NT * 4));
      const int fbase = fastw ? ibase : 0;
      // a dead tile (the odd tile out at the end of the grid) stores through an empty descriptor
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)tl.b * Ho * Wo * Cout + (fbase >> 2)), 0,
                                                                           tl.live ? Ho * Wo * Cout * 4 - fbase : 0, 0x00020000);
      auto note = [&](const f32x4& v) __attribute__((always_inline)) {
        const float m = RELU ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax_run = max(amax_run, __builtin_bit_cast(unsigned, m));
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
          const unsigned off = fastw ? (unsigned)soff[hh] : (oy < Ho && ox < Wo) ? (unsigned)(soff[hh] + ibase) : OOB;
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
            const unsigned off = fastw ? (unsigned)soff[r * 4 + x] : (oy < Ho && ox < Wo) ? (unsigned)(soff[r * 4 + x] + ibase) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, (int)off, 0, 0);
          }
      }
      // the image's output maximum for the NEXT layer's s_v: into this workgroup's LDS table, flushed once at the end of the kernel
      // (conv3x3_wino24h.hip)
      if (p.amax_out) {
        // wave maximum by DPP (row butterflies, then the four rows' lane 0 through SGPRs): a shuffle reduction keeps six bpermute
        // addresses alive across the main loop (spills), and an atomic from every lane is turned by hipcc's atomic optimizer into
        // a 64-iteration scan loop
        {
          unsigned mb = amax_run;
          mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
          mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
          mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x141, 0xf, 0xf, true));   // row_half_mirror
          mb = max(mb, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0x140, 0xf, 0xf, true));   // row_mirror
          const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 0), (unsigned)__builtin_amdgcn_readlane((int)mb, 16));
          const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)mb, 32), (unsigned)__builtin_amdgcn_readlane((int)mb, 48));
          const unsigned mw = max(m01, m23);
          if (tl.live && mw && lq == 0) atomicMax(amax_tab + (tl.b & (AMAX_SLOTS - 1)), mw);
        }
        amax_run = 0;
      }
    }
    item_c += grid;
    if (item_c >= nitems) break;
    cur = decode(item_c);
    if (item_c + grid < nitems) nxt_cob = (item_c + grid) % ncob;
  }
  if (p.amax_out) {
    __syncthreads();
    for (int i = tid; i < AMAX_SLOTS; i += 256)
      if (amax_tab[i]) atomicMax(p.amax_out + i, amax_tab[i]);
  }
#ifdef U_TRACE
  if (tid == 0 && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 16)
    for (int i = 0; i < 8; ++i) u_trace_buf[(blockIdx.x >> 4) * 8 + i] = tr[i];
#endif
}

template <bool POOL, bool RELU>
hipError_t launch_u(const ConvArgs& a, hipStream_t s) {
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
  auto k = conv3x3_wino24u<POOL, RELU>;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(k), (int)lds, attr);
  const dim3 grid((unsigned)(nitems < ncu ? nitems : ncu));     // persistent: one workgroup per CU
  last_form = "conv3x3_wino24u:f16x2";
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y, ntiles, nitems);
  return hipGetLastError();
}
}  // namespace

#ifdef U_TRACE
void conv_u_trace_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(u_trace_buf), sizeof(long long) * 16 * 8); }
#endif

bool conv3x3_wino24u_supported(const ConvArgs& a) { return conv3x3_wino24h_supported(a); }

hipError_t launch_conv3x3_wino24u(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wino24u_supported(a)) return hipErrorInvalidValue;
  if (a.pool) return a.relu ? launch_u<true, true>(a, s) : launch_u<true, false>(a, s);
  return a.relu ? launch_u<false, true>(a, s) : launch_u<false, false>(a, s);
}

}  // namespace imx
