// gnn_tail_x3.hip -- the tail of one GNN layer of the THROUGHPUT path in one launch (round 4; superglue_test.py:110-119,134-137 and
// the next layer's projections :99-100, or final_proj :256):
//     hidden = relu([x | att] W1' + b1)      (mlp.0 with BatchNorm and attn.merge folded, 2d -> 2d)
//     x     += hidden W2 + b2                 (mlp.3, 2d -> d, residual)
//     out    = x W3 + b3                      (the NEXT layer's q|k|v, d -> 3d, or final_proj, d -> d)
// with every fp32 product carried as six bf16 term products on v_mfma_f32_32x32x16_bf16 (the arithmetic of gemm_x3.hip).
//
// Why one kernel: the three gemm_x3 launches it replaces run at 0.32-0.39 of the bf16 pipe -- K is 128 / 256, so a 128 x 128 tile is
// 4-8 chunks long and its prologue (first loads), its LDS staging of the split activations (a barrier per chunk) and its epilogue
// (64 KB of stores per tile) weigh as much as its MFMAs; between the launches 268 MB of hidden activations and 67 MB of x' go to HBM
// and come back.  Here a WAVE owns 32 rows from the first product to the last store:
//   * every product is TRANSPOSED, D[channel][row] = W^T . act^T: the weights are the MFMA's A operand, the wave's activations its B
//     operand.  A lane of the result holds sixteen channels of ITS row -- which is a B operand of the next product when that
//     product's k index is mapped onto those channels (gnn_tail_pack.h permutes the weights accordingly).  The hidden activations
//     and x' stay in the registers of the wave: no LDS staging, no barrier, no HBM round trip for them;
//   * the first product's B operand is loaded straight from global memory (a lane reads 64 contiguous bytes of its row per 32-k
//     chunk) and split in registers (split3.h);
//   * the only thing in LDS is the weight stream: images of 2 k-steps x 4 output blocks x 3 planes (24 KB; gnn_tail_pack.h writes them
//     in consumption order) in a ring of two, copied by LDS-DMA one image ahead (global_load_lds: no staging registers, no ds_write
//     pass), ONE barrier per image (48 MFMAs per wave) behind a COUNTED vmcnt; the four waves of a workgroup share every image;
//   * mlp.0's 256 hidden channels are produced in two halves (64 accumulator registers each), each half consumed by mlp.3 right away
//     (the first product's activations are read and split twice: +7 % VALU work, -64 registers);
//   * results and the residual go through a per-wave 32 x 32 transpose tile in LDS, so that every global access is a full 128-byte
//     row segment.
// Work per wave and 32 rows: 768 + 384 + 576 MFMAs (q|k|v) against ~1700 VALU instructions.
#include "imx_kernels.h"
#include "split3.h"
#include "gnn_tail_pack.h"

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GT_STAGE_RS = 36;
constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};      // the six term products, smallest first: planes (A, B)

// eight fp32 values -> the three bf16 planes of one B operand
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&pl)[3]) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    split_bf16x2 h, m, l;
    split3_pair(v[j], v[j + 1], h, m, l);
    pl[0][j] = h[0]; pl[0][j + 1] = h[1];
    pl[1][j] = m[0]; pl[1][j + 1] = m[1];
    pl[2][j] = l[0]; pl[2][j + 1] = l[1];
  }
}

// The workgroup barrier between weight images.  __syncthreads() is a fence: hipcc puts s_waitcnt vmcnt(0) lgkmcnt(0) in front of the
// s_barrier, which would drain the wave's activation prefetch (an HBM latency) at EVERY image.  Only this wave's LDS stores of the
// next image have to be complete before the others may read them: lgkmcnt(0).
// The weight images arrive by LDS-DMA (global_load_lds: no staging registers, no ds_write pass); such data is ordered for the readers
// only by the ISSUING wave's counted vmcnt followed by a barrier.  VMEM operations retire in order, so vmcnt(N) with N = the loads
// this wave issued AFTER the image's six DMA pieces and has not consumed yet (the activation prefetch: two chunks = 8 loads) retires
// the image without draining that prefetch; lgkmcnt(0) retires this wave's LDS reads of the slot that is about to be refilled.
#ifdef GT_TRACE
__device__ long long gt_trace[64];
#define GT_STAMP(k) do { if (blockIdx.x == 300 && threadIdx.x == 64 && (k) < 64) gt_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GT_STAMP(k) do { } while (0)
#endif
// n = 0, 4, 8 or 16 (compile-time constants after unrolling); exactly one s_barrier is executed whatever n is
__device__ __forceinline__ void image_barrier(int n) {
  if (n >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NW waves per workgroup: 8 = one workgroup of 256 rows per CU and 48-KB weight images (4 k-steps); 4 = TWO workgroups of 128 rows per CU
// and 24-KB images (2 k-steps).  Two waves share a SIMD either way.  Measured (tools/ubench/gnn_tail_bench.cpp, -DGT_NW8): the same at
// 131072 rows (255 vs 256 us: a SIMD's time is the sum of its waves' MFMA and VALU issue whichever way they are grouped),
// NW = 4 ahead where workgroups are scarce (8224 rows: 40 vs 60 us; three gemm_x3 launches: 50) -- the library launches NW = 4.
template <int NPASS, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void gnn_tail_x3_kernel(GnnTailArgs p) {
  constexpr int D = 128;
  constexpr int SPI = NW / 2;                              // k-steps per weight image (12 KB each: 4 blocks x 3 planes x 1 KB)
  constexpr int SLOT = SPI * 768;                          // 16-byte elements per image
  constexpr int NT = 64 * NW;
  constexpr int NSTEP = 2 * (16 + 8) + 8 * NPASS;          // k-steps of the whole tail
  constexpr int NIMG = NSTEP / SPI;
  extern __shared__ __attribute__((aligned(16))) u32x4 ring[];       // [2][SLOT] weight images, the biases (2 D + D + NPASS D floats), the transpose tiles
  float* lbias = reinterpret_cast<float*>(ring + 2 * SLOT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int row0 = blockIdx.x * (32 * NW) + 32 * wave;
  float* stage = lbias + (3 + NPASS) * D + wave * (32 * GT_STAGE_RS);      // this wave's 32 x 32 transpose tile (row stride 36 floats: conflict-free 16-byte accesses)
  const bool active = row0 < p.M;                                    // waves past M (a multiple of 32) compute on a clamped row and store nothing
  const int row = min(row0 + l31, p.M - 1);
  const u32x4* stream = reinterpret_cast<const u32x4*>(p.stream);
  // (amax) is this lane's row inside its pair's keypoint count?  A wave's 32 rows belong to one (side, pair): the padded counts are
  // multiples of 32 -- the division is wave-uniform
  bool row_valid = false;
  unsigned* amax_slot = nullptr;       // the three words of this wave's (side, pair)
  if constexpr (NPASS == 3) {
    if (p.amax && active) {
      const int r0 = __builtin_amdgcn_readfirstlane(row0), s1 = p.B * p.N0p;
      const int side = r0 >= s1 ? 1 : 0, Np = side ? p.N1p : p.N0p, rr = r0 - (side ? s1 : 0);
      const int b = rr / Np, i0 = rr - b * Np;
      const int n = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
      row_valid = i0 + l31 < n;
      amax_slot = p.amax + (size_t)(side * p.B + b) * 4;
    }
  }

  // ---- the weight stream: image i -> ring slot i & 1 by LDS-DMA, one image ahead (six pieces per thread: 16-byte elements j * NT + tid)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  int pend = 0;      // VMEM operations issued after the newest image's DMA pieces that may still be in flight at the next barrier
  auto fetch = [&](int i) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);            // the order of VMEM operations around the DMA pieces is what image_barrier counts on
#pragma unroll
    for (int j = 0; j < SLOT / NT; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)(stream + (size_t)i * SLOT + j * NT + tid),
                                       (lds_void*)(ring + (i & 1) * SLOT + j * NT + 64 * wave), 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    pend = 0;
  };
  // global k-step gs: at the first step of an image the workgroup meets (image gs / SPI is in its slot: its DMA was issued an image ago;
  // the other slot is free) and the next image is requested
  auto enter_step = [&](int gs) __attribute__((always_inline)) {
    if (gs % SPI) return;
    const int img = gs / SPI;
    GT_STAMP(2 * img);
    if (img > 0) image_barrier(active ? pend : 0);
    GT_STAMP(2 * img + 1);
    if (img + 1 < NIMG) fetch(img + 1);
  };
  // one k-step: 24 MFMAs.  The A operands (four output blocks x three planes from the image) are read two blocks at a time -- 24
  // registers instead of 48 -- and the two blocks' MFMAs alternate, so consecutive MFMAs never share an accumulator.
  auto step24 = [&](int gs, const bf16x8 (&b)[3], f32x16 (&acc)[4]) __attribute__((always_inline)) {
    const int i = gs / SPI, t = gs % SPI;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      bf16x8 a[2][3];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[e][q] = __builtin_bit_cast(bf16x8, ring[(i & 1) * SLOT + ((t * 4 + 2 * pr + e) * 3 + q) * 64 + lane]);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[2 * pr + e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[e][PA[q]], b[PB[q]], acc[2 * pr + e], 0, 0, 0);
    }
  };
  // accumulators of four 32-channel blocks start from the bias of their channels: register 4 g + e <-> channel 32 blk + 8 g + 4 hi + e.
  // The biases sit in LDS: a global load here would queue behind the stores of the previous pass (VMEM returns in order).
  auto bias_init = [&](f32x16 (&acc)[4], int off) __attribute__((always_inline)) {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lbias + off + 32 * blk + 8 * g + 4 * hi);
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
      const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
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
  for (int e = tid; e < (3 + NPASS) * D; e += NT) lbias[e] = e < 2 * D ? p.b1[e] : e < 3 * D ? p.b2[e - 2 * D] : p.b3[e - 3 * D];
  __syncthreads();                     // (a full fence: the biases, and image 0)
  f32x16 acc2[4];                      // x' (mlp.3's output, transposed)
  bias_init(acc2, 2 * D);
  int gs = 0;                          // global k-step (a compile-time constant at every use: all loops are unrolled)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // ---- mlp.0', hidden channels 128 half .. +127: 16 k-steps = 8 chunks of 32 k
    f32x16 acc1[4];
    bias_init(acc1, D * half);
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
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      bf16x8 bp[3];
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
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc1[b][8 * h2 + j], 0.f);
      bf16x8 bp[3];
      split8(v, bp);
      step24(gs, bp, acc2);
    }
  }
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
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * GT_STAGE_RS + 4 * (lane & 7)) + *reinterpret_cast<const f32x4*>(xa);
      if (active) *reinterpret_cast<f32x4*>(xa) = v;
      *reinterpret_cast<f32x4*>(stage + r * GT_STAGE_RS + 4 * (lane & 7)) = v;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + l31 * GT_STAGE_RS + 8 * g + 4 * hi);
      acc2[blk][4 * g] = v[0]; acc2[blk][4 * g + 1] = v[1]; acc2[blk][4 * g + 2] = v[2]; acc2[blk][4 * g + 3] = v[3];
    }
  }
  pend = 16;                           // the sixteen stores of x' (their loads are older and were consumed)
  bf16x8 xp[8][3];                     // x' as the B operands of the next product's k-steps (ob, h2): split once, used by every pass
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = acc2[ob][8 * h2 + j];
      split8(v, xp[2 * ob + h2]);
    }
  // ---- the next product: NPASS passes of 128 output channels, 8 k-steps each
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __builtin_amdgcn_sched_barrier(0);           // (keeps the next pass's accumulators from being initialised before this pass's are stored: registers)
    f32x16 acc3[4];
    bias_init(acc3, 3 * D + D * pass);
#pragma unroll
    for (int s = 0; s < 8; ++s, ++gs) {
      enter_step(gs);
      step24(gs, xp[s], acc3);
    }
    if (pass == NPASS - 1) GT_STAMP(2 * NIMG);
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
        unsigned mb = row_valid ? __builtin_bit_cast(unsigned, mx) : 0u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
        if (lane == 0 && mb) atomicMax(amax_slot + pass, mb);      // (three words per LAUNCH instead of per (side, pair): +27 us)
        pend += 1;
      }
    }
  }
}

}  // namespace

#ifdef GT_TRACE
void gnn_tail_trace_dump() {
  long long t[64];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(gt_trace), sizeof(t));
  for (int i = 0; i < 18; ++i) printf("image %2d: barrier wait %6lld   body %6lld\n", i, t[2 * i + 1] - t[2 * i], t[2 * i + 2] - t[2 * i + 1]);
}
#endif

bool gnn_tail_x3_supported(const GnnTailArgs& a) {
  if (a.amax && (a.n3 != 384 || a.B <= 0 || a.N0p % 32 || a.N1p % 32 || a.M != a.B * (a.N0p + a.N1p))) return false;
  return a.d == 128 && (a.n3 == 384 || a.n3 == 128) && a.M > 0 && a.M % 32 == 0 && a.stream && a.b1 && a.b2 && a.b3;
}

hipError_t launch_gnn_tail_x3(const GnnTailArgs& a, hipStream_t s) {
  if (!gnn_tail_x3_supported(a)) return hipErrorInvalidValue;
  last_form = "gnn_tail_x3:bf16x3";
  static unsigned long long attr[4] = {0, 0, 0, 0};
  auto go = [&](auto kern, int nw, int which) {
    const size_t lds = 2 * (size_t)(nw / 2) * 12288 + (size_t)(3 * a.d + a.n3 + nw * 32 * GT_STAGE_RS) * sizeof(float);
    raise_lds_limit(reinterpret_cast<const void*>(kern), (int)lds, attr[which]);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.M + 32 * nw - 1) / (32 * nw))), dim3(64 * nw), lds, s, a);
  };
#ifdef GT_NW8
  if (a.n3 == 384) go(gnn_tail_x3_kernel<3, 8>, 8, 0); else go(gnn_tail_x3_kernel<1, 8>, 8, 1);
#else
  if (a.n3 == 384) go(gnn_tail_x3_kernel<3, 4>, 4, 2); else go(gnn_tail_x3_kernel<1, 4>, 4, 3);
#endif
  return hipGetLastError();
}

}  // namespace imx
