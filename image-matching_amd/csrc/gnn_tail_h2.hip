// gnn_tail_h2.hip -- gnn_tail_x3.hip's fused GNN layer tail (superglue_test.py:110-119,134-137 and the next layer's projections :99-100,
// or final_proj :256) with every fp32 product as THREE fp16 plane products of two-plane operands (round 4; the format of attention_x3.hip's
// FmtH2 and of conv3x3_wino24h.hip) instead of six bf16 plane products of three-plane operands: 864 instead of 1728 MFMAs per wave and 32
// rows, 16-KB instead of 24-KB weight images.  Structure, data flow and barriers are gnn_tail_x3.hip's (every product transposed, the
// hidden activations and x' never leave the wave's registers, weights by LDS-DMA in consumption order: gnn_tail_pack.h).
//
// Scales.  fp16 has five exponent bits, so every operand is multiplied by a power of two.  The weights' are fixed on the host
// (gnn_tail_pack_h2: each matrix's largest |value| to [2^13, 2^14)).  The activations' come from BOUNDS that are uniform over a wave (its
// 32 rows belong to one (side, pair)):
//     |[x | att]| <= bound_in = max(amax_x, amax_v)      amax_x: the (side, pair)'s largest |x| over its valid rows, written by the kernel that
//                                                        produced x (this kernel's epilogue for the previous layer; rows_amax (gemm_h2.hip) for layer 0);
//                                                        amax_v: the key side's largest |v| (att is a convex combination of v rows)
//     |hidden|    <= bound_h  = bound_in L1(W1') + max|b1|           (largest column L1 norm of the folded mlp.0)
//     |x'|        <= bound_x  = amax_x + bound_h L1(W2) + max|b2|
// A bound that is loose by 2^k costs nothing while the typical value stays above 2^-3 after scaling (its low plane is then a normal
// fp16 number): these are loose by ~2^6 and ~2^11 of 2^17.  Rows past the valid count are not covered by the maxima: their inputs are
// zeroed before the split (they are never read by the attention's softmax or by the score matrix).
// The powers of two are undone where a result leaves its accumulator: one multiply per value that was a conversion's input anyway.
#include "imx_kernels.h"
#include "gnn_tail_pack.h"

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GT_STAGE_RS = 36;
#ifndef GT_RING
#define GT_RING 3      // weight images resident per workgroup: the stream runs two images ahead (round 6: with two, one ahead, 2.69 -> 2.63 ms
                       // per C3 step on one box; 70 KB of LDS per workgroup, two workgroups per CU)
#endif
constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};      // the three plane products, smallest first: planes (A, B)

// eight fp32 values (already scaled) -> the two fp16 planes of one B operand (attention_x3.hip: FmtH2::split)
__device__ __forceinline__ void split8(const float (&v)[8], f16x8 (&pl)[2]) {
  unsigned lo_u, hi_u;
  asm("s_mov_b32 %0, 0x0000bc00" : "=s"(lo_u));
  asm("s_mov_b32 %0, 0xbc000000" : "=s"(hi_u));
  const f16x2 lo = __builtin_bit_cast(f16x2, lo_u), hi = __builtin_bit_cast(f16x2, hi_u);
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f16x2 h, m;
    h[0] = (_Float16)v[j]; h[1] = (_Float16)v[j + 1];
    const float r0 = __builtin_amdgcn_fdot2(h, lo, v[j], false);
    const float r1 = __builtin_amdgcn_fdot2(h, hi, v[j + 1], false);
    m[0] = (_Float16)r0; m[1] = (_Float16)r1;
    pl[0][j] = h[0]; pl[0][j + 1] = h[1];
    pl[1][j] = m[0]; pl[1][j + 1] = m[1];
  }
}
// the power of two that brings `bound` (> 0) to [2^13, 2^14) (exponents clamped: attention_x3.hip's pow2_scale)
__device__ __forceinline__ float pow2_of_bound(float bound) {
  unsigned e = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;
  e = e < 90u ? 90u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (267u - e) << 23);
}

// The workgroup barrier between weight images.  __syncthreads() is a fence: hipcc puts s_waitcnt vmcnt(0) lgkmcnt(0) in front of the
// s_barrier, which would drain the wave's activation prefetch (an HBM latency) at EVERY image.  Only this wave's LDS stores of the
// next image have to be complete before the others may read them: lgkmcnt(0).
// The weight images arrive by LDS-DMA (global_load_lds: no staging registers, no ds_write pass); such data is ordered for the readers
// only by the ISSUING wave's counted vmcnt followed by a barrier.  VMEM operations retire in order, so vmcnt(N) with N = the loads
// this wave issued AFTER the image's four DMA pieces and has not consumed yet (the activation prefetch: two chunks = 8 loads) retires
// the image without draining that prefetch; lgkmcnt(0) retires this wave's LDS reads of the slot that is about to be refilled.
// n = 0, 4, 8 or 16 (compile-time constants after unrolling); exactly one s_barrier is executed whatever n is
__device__ __forceinline__ void image_barrier(int n) {
  if (n >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NW = 4 waves per workgroup, two workgroups of 128 rows per CU, 16-KB weight images of two k-steps (gnn_tail_x3.hip measured the grouping;
// round 6, this kernel: NW = 8 -- one 256-row workgroup per CU, half the weight stream per row -- 2.86 against 2.70 ms per C3 step, same box)
template <int NPASS, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void gnn_tail_h2_kernel(GnnTailArgs p) {
  constexpr int D = 128;
  constexpr int SPI = NW / 2;                              // k-steps per weight image (8 KB each: 4 blocks x 2 planes x 1 KB)
  constexpr int SLOT = SPI * 512;                          // 16-byte elements per image
  constexpr int NT = 64 * NW;
  constexpr int NSTEP = 2 * (16 + 8) + 8 * NPASS;          // k-steps of the whole tail
  constexpr int NIMG = NSTEP / SPI;
  extern __shared__ __attribute__((aligned(16))) u32x4 ring[];       // [GT_RING][SLOT] weight images, the biases (2 D + D + NPASS D floats), the transpose tiles
  float* lbias = reinterpret_cast<float*>(ring + GT_RING * SLOT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int row0 = blockIdx.x * (32 * NW) + 32 * wave;
  float* stage = lbias + (3 + NPASS) * D + wave * (32 * GT_STAGE_RS);      // this wave's 32 x 32 transpose tile (row stride 36 floats: conflict-free 16-byte accesses)
  const bool active = row0 < p.M;                                    // waves past M (a multiple of 32) compute on a clamped row and store nothing
  const int row = min(row0 + l31, p.M - 1);
  const u32x4* stream = reinterpret_cast<const u32x4*>(p.stream_h2);
  // this wave's (side, pair) -- its 32 rows belong to one: the padded counts are multiples of 32, the division is wave-uniform -- the
  // valid part of its rows, and the powers of two of its operands (header)
  int sp = 0, ksp = 0, i0 = 0, nvalid = 0;
  if (active) {
    const int r0 = __builtin_amdgcn_readfirstlane(row0), s1 = p.B * p.N0p;
    const int side = r0 >= s1 ? 1 : 0, Np = side ? p.N1p : p.N0p, rr = r0 - (side ? s1 : 0);
    const int b = rr / Np;
    i0 = rr - b * Np;
    nvalid = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
    sp = side * p.B + b;
    ksp = (p.cross ? 1 - side : side) * p.B + b;
  }
  const bool row_valid = i0 + l31 < nvalid;
  unsigned* amax_slot = (NPASS == 3 && p.amax && active) ? p.amax + (size_t)sp * 4 : nullptr;      // the next layer's q | k | v maxima of this (side, pair)
  const float ax = __builtin_bit_cast(float, p.amax_x_in[sp]), av = __builtin_bit_cast(float, p.amax_v[(size_t)ksp * 4 + 2]);
  const float bound_in = fmaxf(fmaxf(ax, av), 1e-30f);
  const float bound_h = fmaf(bound_in, p.l1_1, p.bmax_1);
  const float bound_x = ax + fmaf(bound_h, p.l1_2, p.bmax_2);
  const float s_in = pow2_of_bound(bound_in), s_h = pow2_of_bound(bound_h), s_x = pow2_of_bound(bound_x);
  const float k1 = s_in / p.w1_inv, k2 = s_h / p.w2_inv, k3 = s_x / p.w3_inv;      // what the three accumulators carry (powers of two)
  const float c1 = s_h / k1, c2inv = 1.0f / k2, c3inv = 1.0f / k3;
  const float sin_lane = row_valid ? s_in : 0.f;                                  // rows past the count: inputs zeroed (the maxima do not cover them)

  // ---- the weight stream: image i -> ring slot i % GT_RING by LDS-DMA, GT_RING - 1 images ahead (four pieces per thread: 16-byte elements j * NT + tid)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  int pend = 0;      // VMEM operations issued after the newest image's DMA pieces that may still be in flight at the next barrier
  auto fetch = [&](int i) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);            // the order of VMEM operations around the DMA pieces is what image_barrier counts on
#pragma unroll
    for (int j = 0; j < SLOT / NT; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)(stream + (size_t)i * SLOT + j * NT + tid),
                                       (lds_void*)(ring + (i % GT_RING) * SLOT + j * NT + 64 * wave), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    pend = 0;
  };
  // global k-step gs: at the first step of an image the workgroup meets (image gs / SPI is in its slot: its DMA was issued GT_RING - 1
  // images ago; the slot of the image just finished is free) and the image GT_RING - 1 ahead is requested
  auto enter_step = [&](int gs) __attribute__((always_inline)) {
    if (gs % SPI) return;
    const int img = gs / SPI;
    // (GT_RING = 3: image img was requested TWO boundaries ago -- the four DMA pieces of image img + 1 were issued after it and may
    // still be in flight)
    constexpr int AHEAD = GT_RING - 1, YOUNGER = 4 * (AHEAD - 1);
    if (img > 0) image_barrier((active ? pend : 0) + (img + 1 < NIMG ? YOUNGER : 0));
    if (img + AHEAD < NIMG) fetch(img + AHEAD);
  };
  // one k-step: 12 MFMAs.  The A operands (four output blocks x two planes from the image) are read two blocks at a time and the two
  // blocks' MFMAs alternate, so consecutive MFMAs never share an accumulator.
  auto step24 = [&](int gs, const f16x8 (&b)[2], f32x16 (&acc)[4]) __attribute__((always_inline)) {
    const int i = gs / SPI, t = gs % SPI;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f16x8 a[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int q = 0; q < 2; ++q) a[e][q] = __builtin_bit_cast(f16x8, ring[(i % GT_RING) * SLOT + ((t * 4 + 2 * pr + e) * 2 + q) * 64 + lane]);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[2 * pr + e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[e][PA[q]], b[PB[q]], acc[2 * pr + e], 0, 0, 0);
    }
  };
  // accumulators of four 32-channel blocks start from the bias of their channels: register 4 g + e <-> channel 32 blk + 8 g + 4 hi + e.
  // The biases sit in LDS: a global load here would queue behind the stores of the previous pass (VMEM returns in order).
  // (`k`: the power of two the product's accumulators carry)
  auto bias_init = [&](f32x16 (&acc)[4], int off, float k) __attribute__((always_inline)) {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lbias + off + 32 * blk + 8 * g + 4 * hi) * k;
        acc[blk][4 * g] = v[0]; acc[blk][4 * g + 1] = v[1]; acc[blk][4 * g + 2] = v[2]; acc[blk][4 * g + 3] = v[3];
      }
  };
  // the first product's activations: chunk c (32 k) of [x | att], sixteen contiguous floats of the lane's row
  const float* xrow = p.x + (size_t)row * D;
  const float* arow = p.att + (size_t)row * D;
  auto act_load = [&](int c, f32x4 (&a)[4]) __attribute__((always_inline)) {
    const float* src = (c < 4 ? xrow : arow) + 32 * (c & 3) + 16 * hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
    pend += 4;
  };

  // A 32-channel block of a transposed result -> 32 rows x 128 bytes of a row-major tensor, through this wave's LDS tile: a lane's
  // direct stores would put 32 bytes into each of 32 different rows per instruction (partial lines: measured 10-18 k cycles per
  // 128-channel burst); from the tile every store instruction writes eight full 128-byte row segments.
  auto store_block = [&](const f32x16& acc, float* dst, int ld, int col) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]} * c3inv;
      *reinterpret_cast<f32x4*>(stage + l31 * GT_STAGE_RS + 8 * g + 4 * hi) = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 8 * k + (lane >> 3);
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * GT_STAGE_RS + 4 * (lane & 7));
      *reinterpret_cast<f32x4*>(dst + (size_t)(row0 + r) * ld + col + 4 * (lane & 7)) = v;
    }
  };

  fetch(0);
  if constexpr (GT_RING > 2) { if (NIMG > 1) fetch(1); }
  for (int e = tid; e < (3 + NPASS) * D; e += NT) lbias[e] = e < 2 * D ? p.b1[e] : e < 3 * D ? p.b2[e - 2 * D] : p.b3[e - 3 * D];
  __syncthreads();                     // (a full fence: the biases, and image 0)
  f32x16 acc2[4];                      // x' (mlp.3's output, transposed)
  bias_init(acc2, 2 * D, k2);
  int gs = 0;                          // global k-step (a compile-time constant at every use: all loops are unrolled)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // ---- mlp.0', hidden channels 128 half .. +127: 16 k-steps = 8 chunks of 32 k
    f32x16 acc1[4];
    bias_init(acc1, D * half, k1);
    f32x4 act[3][4];                                       // ring of three chunks: loaded TWO chunks (four k-steps) ahead
    act_load(0, act[0]);
    act_load(1, act[1]);
#pragma unroll
    for (int s = 0; s < 16; ++s, ++gs) {
      enter_step(gs);
      const int c = s >> 1;
      if ((s & 1) == 0 && c + 2 < 8) act_load(c + 2, act[(c + 2) % 3]);
      const f32x4 (&cur)[4] = act[c % 3];
      const f32x4 lo = cur[2 * (s & 1)], hi4 = cur[2 * (s & 1) + 1];
      const float v[8] = {lo[0] * sin_lane, lo[1] * sin_lane, lo[2] * sin_lane, lo[3] * sin_lane, hi4[0] * sin_lane, hi4[1] * sin_lane, hi4[2] * sin_lane, hi4[3] * sin_lane};
      f16x8 bp[2];
      split8(v, bp);
      step24(gs, bp, acc1);
    }
    // ---- mlp.3 over this half's hidden channels: k-steps (b, h2) with the B operand straight from acc1 (ReLU here)
#pragma unroll
    for (int s = 0; s < 8; ++s, ++gs) {
      enter_step(gs);
      const int b = s / 2, h2 = s & 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc1[b][8 * h2 + j], 0.f) * c1;      // hidden s_h
      f16x8 bp[2];
      split8(v, bp);
      step24(gs, bp, acc2);
    }
  }
  float xmax = 0.f;
  // ---- x' = x + (hidden W2 + b2): through the transpose tile, where a lane sees 16 contiguous bytes of a row -- the residual is read
  //      and x' written as full 128-byte row segments (16 loads + 16 stores; the next image_barrier counts the stores), and the sum
  //      comes back from the tile in the transposed layout for the next product
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc2[blk][4 * g], acc2[blk][4 * g + 1], acc2[blk][4 * g + 2], acc2[blk][4 * g + 3]};
      *reinterpret_cast<f32x4*>(stage + l31 * GT_STAGE_RS + 8 * g + 4 * hi) = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 8 * k + (lane >> 3);
      float* xa = p.x + (size_t)min(row0 + r, p.M - 1) * D + 32 * blk + 4 * (lane & 7);
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * GT_STAGE_RS + 4 * (lane & 7)) * c2inv + *reinterpret_cast<const f32x4*>(xa);
      if (active) *reinterpret_cast<f32x4*>(xa) = v;
      *reinterpret_cast<f32x4*>(stage + r * GT_STAGE_RS + 4 * (lane & 7)) = v;
      if (i0 + r < nvalid) xmax = fmaxf(xmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));      // x' of the valid rows
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + l31 * GT_STAGE_RS + 8 * g + 4 * hi);
      acc2[blk][4 * g] = v[0]; acc2[blk][4 * g + 1] = v[1]; acc2[blk][4 * g + 2] = v[2]; acc2[blk][4 * g + 3] = v[3];
    }
  }
  pend = 16;                           // the sixteen stores of x' (their loads are older and were consumed)
  if (p.amax_x_out && active) {        // the largest |x'| of this (side, pair): the NEXT layer's amax_x (one atomic without return per wave)
    unsigned mb = __builtin_bit_cast(unsigned, xmax);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if (lane == 0 && mb) atomicMax(p.amax_x_out + sp, mb);
    pend += 1;
  }
  f16x8 xp[8][2];                      // x' s_x as the B operands of the next product's k-steps (ob, h2): split once, used by every pass
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = acc2[ob][8 * h2 + j] * s_x;
      split8(v, xp[2 * ob + h2]);
    }
  // ---- the next product: NPASS passes of 128 output channels, 8 k-steps each
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __builtin_amdgcn_sched_barrier(0);           // (keeps the next pass's accumulators from being initialised before this pass's are stored: registers)
    f32x16 acc3[4];
    bias_init(acc3, 3 * D + D * pass, k3);
#pragma unroll
    for (int s = 0; s < 8; ++s, ++gs) {
      enter_step(gs);
      step24(gs, xp[s], acc3);
    }
    if (active) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) store_block(acc3[blk], p.out, p.n3, D * pass + 32 * blk);
    }
    pend = 16;
    if constexpr (NPASS == 3) {
      // max |q| / |k| / |v| (pass 0 / 1 / 2) over the valid rows of this wave's (side, pair), for the next layer's attention
      // (GnnTailArgs::amax): a lane holds 64 values of ITS row -- 32 v_max3 with |.| modifiers, one select, a wave reduction, one
      // atomic without return per wave and pass (32 waves share a word: the atomics sit in the wave's in-order VMEM queue like a store)
      if (amax_slot) {
        float mx = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(acc3[blk][r]), __builtin_fabsf(acc3[blk][r + 1])), mx);
        unsigned mb = row_valid ? __builtin_bit_cast(unsigned, mx * c3inv) : 0u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
        if (lane == 0 && mb) atomicMax(amax_slot + pass, mb);
        pend += 1;
      }
    }
  }
}

}  // namespace

bool gnn_tail_h2_supported(const GnnTailArgs& a) {
  if (!a.stream_h2 || !a.amax_x_in || !a.amax_v || !(a.w1_inv > 0.f) || !(a.w2_inv > 0.f) || !(a.w3_inv > 0.f)) return false;
  if (a.B <= 0 || a.N0p % 32 || a.N1p % 32 || a.M != a.B * (a.N0p + a.N1p)) return false;
  if (a.amax && a.n3 != 384) return false;
  return a.d == 128 && (a.n3 == 384 || a.n3 == 128) && a.M > 0 && a.b1 && a.b2 && a.b3;
}

hipError_t launch_gnn_tail_h2(const GnnTailArgs& a, hipStream_t s) {
  if (!gnn_tail_h2_supported(a)) return hipErrorInvalidValue;
  last_form = "gnn_tail_h2:f16x2";
  static unsigned long long attr[2] = {0, 0};
  auto go = [&](auto kern, int nw, int which) {
    const size_t lds = GT_RING * (size_t)(nw / 2) * 8192 + (size_t)(3 * a.d + a.n3 + nw * 32 * GT_STAGE_RS) * sizeof(float);
    raise_lds_limit(reinterpret_cast<const void*>(kern), (int)lds, attr[which]);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 32 * nw - 1) / (32 * nw))), dim3(64 * nw), lds, s, a);
  };
  if (a.n3 == 384) go(gnn_tail_h2_kernel<3, 4>, 4, 0); else go(gnn_tail_h2_kernel<1, 4>, 4, 1);
  return hipGetLastError();
}

}  // namespace imx
