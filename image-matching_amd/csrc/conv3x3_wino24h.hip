// conv3x3_wino24h.hip -- conv3x3_wino24.hip's Winograd F(2x4, 3x3) layer (superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123)
// with its 24 per-position GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]  on the FP16 matrix pipe (round 4): both
// transformed operands are cut into TWO fp16 planes (x s = h + m, 22 bits; s a power of two) and three plane products are kept,
// (h,m) (m,h) (h,h) on v_mfma_f32_16x16x32_f16 -- 72 MFMAs of 16 cycles per 32 input channels where the fp32 form spends 192 of 32
// (16 / 3 of the rate), with the same operand bytes (two 2-byte planes = one fp32) through LDS and from L2.
// Accuracy (tools/wino_accuracy_emul.py, the SuperPoint stack against the reference goldens): x4 / semi / desc use 0.271 / 0.386 /
// 0.033 of the 1e-4 + 1e-4|ref| tolerance against 0.253 / 0.331 / 0.034 for the fp32 Winograd form (rms error vs float64 6.97e-6
// against 6.65e-6): V and U are rounded to 24 bits there, to 22 here, and the partial sums are rounded once per 32 channels instead
// of once per 4.
//
// Scales.  fp16 has five exponent bits: V = B2^T d B4 is bounded by 20 max|d| (row sums 2 x 10 of the transform matrices), so the
// raw patch is multiplied by s_v = the power of two that brings 32 x the IMAGE's largest |input| to 2^13 on its way into LDS --
// the maximum comes from the layer that produced the input (ConvArgs::amax_in, one word per image slot b % 256, written by that layer's
// epilogue through atomicMax: this kernel's, or conv1ab_wino24's); U is scaled on the host (its maximum to [2^13, 2^14)).  The
// product of the two powers of two is undone in the epilogue's bias add (one fma instead of an add).
//
// Structure: conv3x3_wino24.hip's items (8x16-pixel tile x 64 output channels, persistent, two workgroups per CU, four waves of 16
// channels x 16 wtiles x 24 positions), output transform, pooling and stores unchanged.  A chunk is 32 input channels:
//   phase A  the input transform of the chunk (four 8-channel sub-patches, lane = (channel pair, wtile), the four waves take the
//            four transformed rows -- the fp32 kernel's code), each transformed pair split (v_cvt_pk_f16_f32, two v_dot2c_f32_f16,
//            v_cvt_pk_f16_f32) and written as 4 bytes per plane: V[plane][position][8-channel group][wtile][8] -- a wave's
//            stores cover 256 contiguous bytes, a B-operand read (lane = (wtile, group): 16 bytes) 1 KB contiguous;
//   barrier
//   phase B  72 MFMAs (two positions interleaved); the A operands (U: [item block][chunk][position][plane][wave][lane][8]) come
//            from L2 through a ring of six positions refilled in place; then the NEXT chunk's raw patch goes from registers to
//            LDS (scaled) and the one after is requested;
//   barrier
// LDS: V 48 KB + raw 30 KB + 1 KB = 79 KB per workgroup, two per CU.
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per item (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // input patch (pad-1 halo)
constexpr int RSC = 10;                        // raw sub-patch: pixel stride (8 channels + 2), as in conv3x3_wino24.hip
constexpr int RAWC = 192 * RSC;                // 180 pixels + pad, floats per 8-channel sub-patch
constexpr int NSUB = 4;                        // 8-channel sub-patches per chunk
constexpr int CKH = 32, NT = 64, NPOS = 24;
constexpr int VPLANE = NPOS * 4 * 16 * 8;      // halves per plane (24576 bytes)
constexpr int UPOS = 2 * 4 * 64 * 8;           // halves of U per (item block, chunk, position): [plane][wave][lane][8]
constexpr int AMAX_SLOTS = 256;                // image b -> slot b % 256 (ConvArgs::amax_in / amax_out hold upper bounds, so sharing a slot is safe)
constexpr int RING = 6;                        // positions of U in flight
constexpr unsigned OOB = 0x7ffffff0u;          // byte offset beyond any image: buffer loads return 0
#ifndef H_EXP
#define H_EXP 0                                // timing experiments (tools/ubench/conv_h_bench.cpp): 1 no U refills, 2 no transform, 3 no patch loads / stores
#endif

struct Item { int b, y0, x0, cob; };
#ifdef H_TRACE
// phase clocks (tools/ubench/conv_h_bench.cpp, -DH_TRACE): cycles of wave 0 of every 32nd workgroup in each phase of chunk_step
__device__ long long h_trace_buf[16 * 8];
#define H_STAMP(i_) { const long long t_ = __builtin_amdgcn_s_memtime(); tr[i_] += t_ - tlast; tlast = t_; }
#else
#define H_STAMP(i_)
#endif
template <bool V>
struct BoolC { static constexpr bool value = V; };

// x = h + m in fp16, two values at a time (attention_x3.hip's FmtH2::split; constants through SGPRs: hipcc 7.2 folds a packed
// {-1, 0} into the inline constant -1.0)
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

// s_v of an image: 32 x its largest |input| (>= the bound 20 max|d| of the transformed patch) goes to 2^13
__device__ __forceinline__ float v_scale(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 60u ? 60u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (262u - e) << 23);
}

template <bool POOL, bool RELU>
__global__ __launch_bounds__(256, 2) void conv3x3_wino24h(ConvArgs p, int tiles_x, int tiles_y, int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  _Float16* Vp = reinterpret_cast<_Float16*>(smem_h);                          // [2][VPLANE]
  float* raw = reinterpret_cast<float*>(smem_h + 2 * VPLANE * 2);              // [NSUB][RAWC]
  unsigned* amax_tab = reinterpret_cast<unsigned*>(raw + NSUB * RAWC);         // [AMAX_SLOTS]: this workgroup's output maxima per image slot

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CKH, ncob = Cout / NT;
  const int grid = (int)gridDim.x;
  // XCD-aware start index (conv3x3_wino24.hip)
  const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= nitems) return;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wuh, 0, ncob * nchunk * NPOS * UPOS * 2, 0x00020000);
  const int uoff_lane = (cb * 64 + lane) * 16;                                 // bytes inside a plane of a position
  const int img_bytes = H * W * Cin * 4;

  // ---- input transform roles: lane = (channel pair tk, wtile tw); transformed row i = wave (rows of B2^T)
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  const f32x2 sg2 = {sg, sg};
  const f32x2 m5 = {-5.f, -5.f};
  const float* rpa = raw + ((2 * twr + ra) * RW + 4 * twc) * RSC + 2 * tk;
  const float* rpb = raw + ((2 * twr + rb) * RW + 4 * twc) * RSC + 2 * tk;
  // V stores: position p = j*4 + wave, group q, wtile tw, channels 2 tk, 2 tk + 1 of the group -> halves ((p*4 + q)*16 + tw)*8 + 2 tk
  _Float16* vwr = Vp + (wave * 4 * 16 + tw) * 8 + 2 * tk;
  // B-operand reads: lane = (wtile n = lane & 15, group kg = lane >> 4) -> 16 bytes at position p * 1024 + lane * 16
  const _Float16* vrd = Vp + lane * 8;

  // ---- loader: thread -> two (pixel, channel half) float4 of every 10x18x8 sub-patch (conv3x3_wino24.hip's table)
  int lpy[2], lpx[2], ldst[2], lhalf[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (k == 1 && tid + 256 < RH * RW * 2) ? tid + 256 : tid;
    const int px = p.in_blocked ? e >> 1 : e % (RH * RW), half = p.in_blocked ? e & 1 : e / (RH * RW);
    lpy[k] = px / RW - 1;
    lpx[k] = px % RW - 1;
    ldst[k] = px * RSC + half * 4;
    lhalf[k] = half;
  }
  auto decode = [&](int it) -> Item {
    Item r;
    r.cob = it % ncob;
    const int tile = it / ncob;
    r.x0 = (tile % tiles_x) * OW;
    r.y0 = ((tile / tiles_x) % tiles_y) * OH;
    r.b = tile / (tiles_x * tiles_y);
    return r;
  };
  // cur: the item whose chunks are multiplied; nxt: the one after it (its U block is prefetched during cur's last chunk); lit: the
  // loader's item (the loader runs two chunks ahead: with two chunks per item it is a whole item ahead of cur)
  int item_c = vb;
  Item cur = decode(vb), nxt = vb + grid < nitems ? decode(vb + grid) : cur, lit = cur;
  int litem = item_c, lchunk = 0;
  __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  unsigned goff[2];
  float lsv = 1.f;                                                // s_v of the loader's item
  const bool inb = p.in_blocked != 0;
  const int pxb = inb ? 8 * 4 : Cin * 4;                          // bytes from one pixel to the next
  const int sub_step = inb ? H * W * 8 * 4 : 8 * 4;               // bytes from one 8-channel group to the next
  auto loader_item = [&](const Item& it, bool live) {
    lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)(live ? it.b : 0) * H * W * Cin), 0, live ? img_bytes : 0, 0x00020000);
    lsv = v_scale(p.amax_in[(live ? it.b : 0) & (AMAX_SLOTS - 1)]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int gy = it.y0 + lpy[k], gx = it.x0 + lpx[k];
      goff[k] = (live && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * pxb + lhalf[k] * 16) : OOB;
    }
  };
  f32x4 rr[NSUB][2];
  float rr_sv = 1.f;                                              // the scale that goes with the registers' chunk
  auto issue_load = [&]() {
    const int so = __builtin_amdgcn_readfirstlane(lchunk * NSUB * sub_step);
#pragma unroll
    for (int q = 0; q < NSUB; ++q) {
      rr[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[0], so + q * sub_step, 0));
      rr[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[1], so + q * sub_step, 0));
    }
    rr_sv = lsv;
  };
  auto advance_loader = [&]() {
    if (__builtin_expect(++lchunk == nchunk, 0)) {   // the loader moves on to this workgroup's next item
      lchunk = 0;
      litem += grid;
      const bool live = litem < nitems;
      if (live) lit = decode(litem);
      loader_item(lit, live);
      asm volatile("" ::: "memory");
    }
  };
  auto store_raw = [&]() {
    const f32x4 s4 = {rr_sv, rr_sv, rr_sv, rr_sv};
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

  // ---- U ring: slot p % RING holds position p's two planes (eight halves each per lane)
  u32x4 ub[RING][2];
  auto u_load = [&](int slot, int cobv, int chv, int pos) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(((cobv * nchunk + chv) * NPOS + pos) * (UPOS * 2));
    ub[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so, 0);
    ub[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(ur, uoff_lane, so + 4 * 64 * 16, 0);
  };

  for (int i = tid; i < AMAX_SLOTS; i += 256) amax_tab[i] = 0;       // (visible after the fill's barrier)
  // ---- pipeline fill: chunk 0 of the first item into raw, chunk 1 in flight, the U ring of chunk 0
  loader_item(cur, true);
  issue_load(); advance_loader();
  store_raw();
  issue_load(); advance_loader();
#pragma unroll
  for (int g = 0; g < RING; ++g) u_load(g, cur.cob, 0, g);
  __syncthreads();

  f32x4 acc[NPOS];          // an item's first chunk starts every accumulator from a literal-zero C operand
  const f32x2 k8 = {8.f, 8.f};
  // ---- store offsets (conv3x3_wino24.hip)
  const int Ho_k = POOL ? H >> 1 : H, Wo_k = POOL ? W >> 1 : W;
  const bool outb = p.out_blocked != 0;
  const bool fastw = (W % OW) == 0 && (!outb || (H % OH) == 0);
  const int lwr = (lane & 15) >> 2, lwc = lane & 3;
  const int opx = outb ? 8 * 4 : Cout * 4;
  const int chl = outb ? (cb * 2 + (lane >> 5)) * (Ho_k * Wo_k * 8 * 4) + ((lane >> 4) & 1) * 16 : (cb * 16 + 4 * (lane >> 4)) * 4;
  int soff[POOL ? 2 : 8];
#pragma unroll
  for (int e = 0; e < (POOL ? 2 : 8); ++e) {
    const int oy = POOL ? lwr : 2 * lwr + (e >> 2), ox = POOL ? 2 * lwc + e : 4 * lwc + (e & 3);
    soff[e] = (oy * Wo_k + ox) * opx + chl;
  }
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};

  // phase A: raw (scaled) -> V planes
  auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NSUB; ++q) {
      f32x2 o[6], T[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb)
        o[bb] = pk_fma(sg2, *reinterpret_cast<const f32x2*>(rpb + q * RAWC + bb * RSC), *reinterpret_cast<const f32x2*>(rpa + q * RAWC + bb * RSC));
      const W24Half hb = w24_batch_a(o, m5);
      w24_batch_b(o, hb, T);
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        f16x2 h, m;
        split_h2(T[jj], h, m);
        _Float16* d = vwr + ((jj * 4 * 4 + q) * 16) * 8;          // position jj*4 + wave (the wave part sits in vwr), group q
        *reinterpret_cast<f16x2*>(d) = h;
        *reinterpret_cast<f16x2*>(d + VPLANE) = m;
      }
    }
  };
  // phase B: 72 MFMAs, two positions interleaved; U slots refilled in place with position p + RING (of this chunk, or of the next
  // chunk / the next item's block)
  auto mfma_phase = [&](auto firstc, int c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;
    const bool lastc = c + 1 == nchunk;
    const int ncb = lastc ? nxt.cob : cur.cob, nch = lastc ? 0 : c + 1;
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
        const int np = pp + e + RING;
        if (H_EXP == 1) continue;
        if (np < NPOS) u_load((pp + e) % RING, cur.cob, c, np);
        else u_load((pp + e) % RING, ncb, nch, np - NPOS);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#ifdef H_TRACE
  long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  auto chunk_step = [&](auto firstc, int c) __attribute__((always_inline)) {
    H_STAMP(5)                     // (epilogue and item bookkeeping)
    if (H_EXP != 2) transform();
    H_STAMP(0)
    __syncthreads();               // V complete; raw free
    H_STAMP(1)
    mfma_phase(firstc, c);
    H_STAMP(2)
    if (H_EXP != 3) {
      store_raw();                 // the next chunk's patch (requested a chunk ago)
      issue_load();
    }
    advance_loader();
    H_STAMP(3)
    __syncthreads();               // raw complete; V free
    H_STAMP(4)
  };

  unsigned amax_run = 0;           // this lane's largest stored value of the current item (bit pattern; values >= 0 after ReLU, |.| otherwise)
#pragma unroll 1
  for (;;) {
    // the item's bias and un-scale factor are requested HERE: loaded in the epilogue they cost it an L2 round trip per item
    const f32x4 bs4 = *reinterpret_cast<const f32x4*>(p.bias + cur.cob * NT + cb * 16 + 4 * (lane >> 4));
    const float inv = p.u_scale_inv / v_scale(p.amax_in[cur.b & (AMAX_SLOTS - 1)]);
    chunk_step(BoolC<true>{}, 0);
#pragma unroll 1
    for (int c = 1; c < nchunk; ++c) chunk_step(BoolC<false>{}, c);

    // ---- item done: output transform, un-scale + bias, ReLU, (2x2 max-pool), stores straight from registers (conv3x3_wino24.hip)
    {
      const f32x4 inv4 = {inv, inv, inv, inv};
      f32x4 y[2][4];
      w24_output_transform(acc, k8, y);
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int Ho = Ho_k, Wo = Wo_k;
      typedef unsigned su32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int ibase = __builtin_amdgcn_readfirstlane(((POOL ? cur.y0 >> 1 : cur.y0) * Wo + (POOL ? cur.x0 >> 1 : cur.x0)) * opx +
                                                       (outb ? cur.cob * (NT / 8) * (Ho * Wo * 8 * 4) : cur.cob * NT * 4));
      const int fbase = fastw ? ibase : 0;
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)cur.b * Ho * Wo * Cout + (fbase >> 2)), 0,
                                                                           Ho * Wo * Cout * 4 - fbase, 0x00020000);
      auto note = [&](const f32x4& v) __attribute__((always_inline)) {
        const float m = RELU ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax_run = max(amax_run, __builtin_bit_cast(unsigned, m));
      };
      if constexpr (POOL) {
        const int oy = (cur.y0 >> 1) + lwr;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4 mx4 = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]), __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1]));
          f32x4 v = __builtin_elementwise_fma(mx4, inv4, bs4);
          if (RELU) v = __builtin_elementwise_max(v, zero4);
          note(v);
          if (fastw) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, soff[hh], 0, 0);
          } else {
            const int ox = (cur.x0 >> 1) + 2 * lwc + hh;
            const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)(soff[hh] + ibase) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, (int)off, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            f32x4 v = __builtin_elementwise_fma(y[r][x], inv4, bs4);
            if (RELU) v = __builtin_elementwise_max(v, zero4);
            note(v);
            if (fastw) {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, soff[r * 4 + x], 0, 0);
            } else {
              const int oy = cur.y0 + 2 * lwr + r, ox = cur.x0 + 4 * lwc + x;
              const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)(soff[r * 4 + x] + ibase) : OOB;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), ors, (int)off, 0, 0);
            }
          }
      }
      // the image's output maximum for the NEXT layer's s_v (an upper bound is what is needed: pixels of a partial tile past the
      // image edge are included): into this workgroup's LDS table (one ds_max per wave and item -- a global atomic here sits in
      // the wave's VMEM queue and is waited for at the next barrier: +18 % on conv2a, 4 x at C5 where 16 images share the words);
      // the table goes out once, at the end of the kernel
      if (p.amax_out) {
        unsigned mb = amax_run;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
        if (lane == 0 && mb) atomicMax(amax_tab + (cur.b & (AMAX_SLOTS - 1)), mb);
        amax_run = 0;
      }
    }
    item_c += grid;
    if (item_c >= nitems) break;
    cur = nxt;
    if (item_c + grid < nitems) nxt = decode(item_c + grid);
  }
  if (p.amax_out) {
    __syncthreads();
    for (int i = tid; i < AMAX_SLOTS; i += 256)
      if (amax_tab[i]) atomicMax(p.amax_out + i, amax_tab[i]);
  }
#ifdef H_TRACE
  if (tid == 0 && (blockIdx.x & 31) == 0 && (blockIdx.x >> 5) < 16)
    for (int i = 0; i < 8; ++i) h_trace_buf[(blockIdx.x >> 5) * 8 + i] = tr[i];
#endif
}

template <bool POOL, bool RELU>
hipError_t launch_h(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  const int nitems = tiles_x * tiles_y * a.B * (a.Cout / NT);
  const size_t lds = (size_t)2 * VPLANE * 2 + (size_t)NSUB * RAWC * sizeof(float) + AMAX_SLOTS * sizeof(unsigned);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  auto k = conv3x3_wino24h<POOL, RELU>;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(k), (int)lds, attr);
  const dim3 grid((unsigned)(nitems < 2 * ncu ? nitems : 2 * ncu));     // persistent: two workgroups per CU
  last_form = "conv3x3_wino24h:f16x2";
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y, nitems);
  return hipGetLastError();
}
}  // namespace

#ifdef H_TRACE
void conv_h_trace_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(h_trace_buf), sizeof(long long) * 16 * 8); }
#endif

bool conv3x3_wino24h_supported(const ConvArgs& a) {
  if (a.first || a.Cin % 64 || a.Cout % NT || !a.wuh || !a.amax_in || !(a.u_scale_inv > 0.f)) return false;
  const size_t Ho = a.pool ? a.H / 2 : a.H, Wo = a.pool ? a.W / 2 : a.W;
  return (size_t)a.H * a.W * a.Cin * 4 < (size_t)OOB && Ho * Wo * (size_t)a.Cout * 4 < (size_t)OOB;
}

hipError_t launch_conv3x3_wino24h(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wino24h_supported(a)) return hipErrorInvalidValue;
  if (a.pool) return a.relu ? launch_h<true, true>(a, s) : launch_h<true, false>(a, s);
  return a.relu ? launch_h<false, true>(a, s) : launch_h<false, false>(a, s);
}

}  // namespace imx
