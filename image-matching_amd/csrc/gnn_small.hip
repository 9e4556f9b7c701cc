// gnn_small.hip — the tail of one AttentionalGNN layer for SMALL row counts (a single pair: M = 2 x 1024 keypoint rows), fused:
//   hidden = relu(mlp.0'([x | a]))            (superglue_test.py:110-119; attn.merge and the BatchNorm are folded into mlp.0' at load)
//   x     += mlp.3(hidden)                    (:134-137, the residual update)
//   next   = W_next x + b_next                (the NEXT layer's q|k|v projections :99-100, or final_proj :256 after the last layer)
// in one launch per layer instead of three (gemm_small x 3).  Measured at one pair (rocprofv3 kernel durations, d = 128): 15.8 us
// against 8.4 + 8.0 + 6.5 = 22.9 us; Matching.forward on one pair 1.80 -> 1.69 ms (BASELINE configs[2]).  The cycle trace
// (-DGNN_TRACE) has the three MFMA phases at 32-33 cycles per MFMA (8.5 k + 3.8 k + 5.7 k cycles); the rest is the cold start (the
// first weight group is an L2 miss: ~5 k cycles before the first MFMA) and the three epilogues (~1.4 k each).
//
// A workgroup owns 16 rows through all three products; its four waves split the output COLUMNS of each product, so a weight is
// read once per workgroup, straight from L2 into the MFMA's B operand registers (no reuse inside a workgroup -> no LDS staging;
// the 0.6-2.3 MB of a layer's weights stay L2 resident across the M/16 workgroups), from a copy in B-fragment order.  The 16 x K activation panel is the only
// thing in LDS: [x | a], then hidden, then the updated x.
//
// The arithmetic is gemm_small's, instruction for instruction: v_mfma_f32_16x16x4_f32 with k = 16 t + 4 kq + j, 128-k blocks
// folded into a running sum (two-level accumulation), bias, ReLU, residual in the same order -- the results are bit-identical
// to the three-launch form (tests/test_gpu_superglue.py::test_fused_small_layer_equals_the_three_launch_form).
#include "imx_kernels.h"
#include <type_traits>
#include <cstdio>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TR = 16;            // rows per workgroup

// one product for this wave's NB column blocks of 16: tot[blk] = sum_k A[row][k] W[k][col], A from LDS (row stride SA), W from
// global memory in B-fragment order ([K/16][4][N][4]: a lane's four values of a 16-k group are one 16-byte load), PF k-groups
// ahead in a register ring -- one wave per SIMD and 128 workgroups: nothing else hides the L2 latency
// Ring depth: as many k-groups (a power of two) as fit 192 registers (4 NB per group), at most the whole product.  The workgroups of a launch
// run in step and a layer's weights are not in L2 when it starts (18 layers x 0.6 MB against 4 MB per XCD), so every group is a
// ~2 us round trip that only distance covers: with 4 / 4 / 2 groups a layer took 18 us of kernel time for 8.8 us of MFMAs.
constexpr int ring_depth(int NB, int NG) {
  int pf = 1;
  while (pf * 2 <= NG && pf * 2 * 4 * NB <= 192) pf *= 2;
  return pf;
}
// workgroup barrier that publishes LDS writes only: __syncthreads() also drains vmcnt, i.e. waits for every weight group in
// flight (the ring refills and the next product's first groups: ~2 k cycles per barrier in the cycle trace)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NB, int NG>
struct Ring {
  static constexpr int PF = ring_depth(NB, NG);
  f32x4 r[PF][NB];
};
template <int NB, int N>
__device__ __forceinline__ const f32x4* frag_ptr(const float* __restrict__ Wf, int col0) {
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  return reinterpret_cast<const f32x4*>(Wf) + (size_t)kq * N + col0 + n;      // + t * 4 N + 16 b
}
// the first PF k-groups of a product: issued BEFORE the barrier that publishes its A panel (weights do not depend on it)
template <int NB, int NG, int N>
__device__ __forceinline__ void ring_fill(Ring<NB, NG>& ring, const f32x4* wp) {
#pragma unroll
  for (int q = 0; q < Ring<NB, NG>::PF; ++q)
#pragma unroll
    for (int b = 0; b < NB; ++b) ring.r[q][b] = wp[(size_t)q * 4 * N + 16 * b];
}
// `side(q)` runs after the MFMAs of group q of the LAST round: the kernel hangs the next product's first weight groups there, a
// few loads per group -- issued in one burst (48 x 1 KB per wave) they held the vector memory pipe, and with it the epilogue's
// LDS writes, for ~3 k cycles
// loads [q * L, q * L + L) of a ring's first fill, L = ceil(PF * NB / ROUNDS): call it for q = 0 .. ROUNDS-1
template <int NB, int NG, int N, int ROUNDS>
__device__ __forceinline__ void ring_fill_part(Ring<NB, NG>& ring, const f32x4* wp, int q) {
  constexpr int TOTAL = Ring<NB, NG>::PF * NB, L = (TOTAL + ROUNDS - 1) / ROUNDS;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int idx = q * L + i;
    if (idx < TOTAL) ring.r[idx / NB][idx % NB] = wp[(size_t)(idx / NB) * 4 * N + 16 * (idx % NB)];
  }
}
template <int NB, int K, int N, typename Side>
__device__ __forceinline__ void product(const float* __restrict__ As, int SA, const f32x4* wp, Ring<NB, K / 16>& ring, f32x4 (&tot)[NB], Side side) {
  constexpr int NG = K / 16, PF = Ring<NB, NG>::PF;
  static_assert(NG % PF == 0, "whole rounds of the ring");
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const float* ap = As + n * SA + 4 * kq;
  f32x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tot[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 an = *reinterpret_cast<const f32x4*>(ap);          // the A values one group ahead (LDS latency off the MFMA stream)
  // one round of the ring = PF k-groups; the LAST round refills nothing (a refill "past the end" is a load into registers the
  // next product's ring wants: the compiler then waits for it -- ~2 k cycles of a cold miss nobody needs)
  auto round = [&](int t0, auto refill_tag) __attribute__((always_inline)) {
    constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = t0 + q;
      const f32x4 a4 = an;
      an = *reinterpret_cast<const f32x4*>(ap + 16 * (t + 1 < NG ? t + 1 : t));
      f32x4 w4[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) w4[b] = ring.r[q][b];
      // j outer, block inner: consecutive MFMAs are independent; each accumulator still sees j = 0..3 in order
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], w4[b][j], acc[b], 0, 0, 0);
      if constexpr (REFILL) {
#pragma unroll
        for (int b = 0; b < NB; ++b) ring.r[q][b] = wp[(size_t)(t + PF) * 4 * N + 16 * b];
      } else {
        side(q);
      }
      if ((t & 7) == 7) {             // end of a 128-k block (two-level accumulation, as gemm_small)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          tot[b] += acc[b];
          acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  // the refilling rounds stay a LOOP (opaque bound): fully unrolled, the scheduler sinks every ring load down to its use to
  // save registers -- a vmcnt(0) before each block's MFMAs, 23 k cycles for product 1 instead of 9.6 k
  int ng = NG;
  asm volatile("" : "+s"(ng));
#pragma unroll 1
  for (int t0 = 0; t0 + PF < ng; t0 += PF) round(t0, std::true_type{});
  round(NG - PF, std::false_type{});
#pragma unroll
  for (int b = 0; b < NB; ++b) tot[b] += acc[b];
}

// D = descriptor_dim; NN = columns of the third product (3 D: the next layer's q|k|v; D: final_proj)
template <int D, int NN>
__global__ __launch_bounds__(256) void gnn_layer_small(GnnSmallArgs p) {
  constexpr int K1 = 2 * D, SA1 = K1 + 8, SA3 = D + 8;      // row strides: (K/4 + 2) sixteen-byte slots (conflict-free ds_read_b128)
  constexpr int NB1 = K1 / 64, NB2 = D / 64, NB3 = NN / 64;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* A1 = sm;                   // [16][SA1]  [x | a]
  float* Hs = A1 + TR * SA1;        // [16][SA1]  hidden
  float* Xs = Hs + TR * SA1;        // [16][SA3]  updated x
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const size_t r0 = (size_t)blockIdx.x * TR;
#ifdef GNN_TRACE
  unsigned long long ts[8]; int nts = 0;
#define GTS ts[nts++] = __builtin_readcyclecounter();
#else
#define GTS
#endif
  GTS

  // ---- every load the prologue can issue, oldest first (VMEM returns in order): the [x | a] panel (L2: the attention kernel
  //      just wrote it), then the first weight groups of product 1 (cold), then the biases
  constexpr int NV = TR * K1 / 4 / 256;       // float4 per thread: 2 / 4 / 8
  f32x4 av[NV];
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int e = tid + it * 256, row = e / (K1 / 4), k4 = (e % (K1 / 4)) * 4;
    const float* src = k4 < D ? p.x + (r0 + row) * D + k4 : p.att + (r0 + row) * D + (k4 - D);
    av[it] = *reinterpret_cast<const f32x4*>(src);
  }
  const f32x4* wp1 = frag_ptr<NB1, K1>(p.w1, wave * (NB1 * 16));
  const f32x4* wp2 = frag_ptr<NB2, D>(p.w2, wave * (NB2 * 16));
  const f32x4* wp3 = frag_ptr<NB3, NN>(p.w3, wave * (NB3 * 16));
  Ring<NB1, K1 / 16> ring1;
  Ring<NB2, K1 / 16> ring2;
  ring_fill<NB1, K1 / 16, K1>(ring1, wp1);
  float bias1[NB1], bias2[NB2], bias3[NB3];          // (an epilogue that fetched its own bias waited ~1.5 k cycles for it)
#pragma unroll
  for (int b = 0; b < NB1; ++b) bias1[b] = p.b1[wave * (NB1 * 16) + 16 * b + n];
#pragma unroll
  for (int b = 0; b < NB2; ++b) bias2[b] = p.b2[wave * (NB2 * 16) + 16 * b + n];
#pragma unroll
  for (int b = 0; b < NB3; ++b) bias3[b] = p.b3[wave * (NB3 * 16) + 16 * b + n];
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int e = tid + it * 256, row = e / (K1 / 4), k4 = (e % (K1 / 4)) * 4;
    *reinterpret_cast<f32x4*>(A1 + row * SA1 + k4) = av[it];
  }
  lds_barrier();
  GTS

  // ---- hidden = relu([x | a] W1 + b1)          D layout: lane (col = n, g) holds rows 4 g .. 4 g + 3 of its column
  Ring<NB3, D / 16> ring3;
  {
    f32x4 tot[NB1];
    const int col0 = wave * (NB1 * 16);
    product<NB1, K1, K1>(A1, SA1, wp1, ring1, tot,
                         [&](int q) __attribute__((always_inline)) { ring_fill_part<NB2, K1 / 16, D, Ring<NB1, K1 / 16>::PF>(ring2, wp2, q); });
    GTS
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      const int col = col0 + 16 * b + n;
      const float bias = bias1[b];
#pragma unroll
      for (int r = 0; r < 4; ++r) Hs[(4 * g + r) * SA1 + col] = fmaxf(tot[b][r] + bias, 0.f);
    }
  }
  lds_barrier();
  GTS

  // ---- x += hidden W2 + b2
  {
    f32x4 tot[NB2];
    const int col0 = wave * (NB2 * 16);
    product<NB2, K1, D>(Hs, SA1, wp2, ring2, tot,
                        [&](int q) __attribute__((always_inline)) { ring_fill_part<NB3, D / 16, NN, Ring<NB2, K1 / 16>::PF>(ring3, wp3, q); });
    GTS
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      const int col = col0 + 16 * b + n;
      const float bias = bias2[b];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        const float v = A1[row * SA1 + col] + (tot[b][r] + bias);      // res + v, as gemm_small
        Xs[row * SA3 + col] = v;
        p.x[(r0 + row) * D + col] = v;
      }
    }
  }
  lds_barrier();
  GTS

  // ---- next = x W3 + b3
  {
    f32x4 tot[NB3];
    const int col0 = wave * (NB3 * 16);
    product<NB3, D, NN>(Xs, SA3, wp3, ring3, tot, [](int) {});
    GTS
#pragma unroll
    for (int b = 0; b < NB3; ++b) {
      const int col = col0 + 16 * b + n;
      const float bias = bias3[b];
#pragma unroll
      for (int r = 0; r < 4; ++r) p.out[(r0 + 4 * g + r) * NN + col] = tot[b][r] + bias;
    }
  }
#ifdef GNN_TRACE
  GTS
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 77))
    printf("[gnn trace] wg %d: stage %llu  p1 %llu  ep1+bar %llu  p2 %llu  ep2+bar %llu  p3 %llu  ep3 %llu\n", (int)blockIdx.x, ts[1] - ts[0], ts[2] - ts[1],
           ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6]);
#endif
#undef GTS
}

template <int D, int NN>
hipError_t launch_t(const GnnSmallArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(2 * TR * (2 * D + 8) + TR * (D + 8)) * sizeof(float);       // D = 256: 83 KB
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(gnn_layer_small<D, NN>), 160 * 1024, attr);
  hipLaunchKernelGGL((gnn_layer_small<D, NN>), dim3((unsigned)(a.M / TR)), dim3(256), lds, s, a);
  return hipGetLastError();
}
}  // namespace

// descriptor_dim 64 / 128 / 256, rows a multiple of 16, unpadded weights (N == Npad), the third product 3 d or d columns wide
bool gnn_layer_small_supported(const GnnSmallArgs& a) {
  if (a.d != 64 && a.d != 128 && a.d != 256) return false;
  if (a.M <= 0 || a.M % TR || (a.n3 != 3 * a.d && a.n3 != a.d)) return false;
  return a.x && a.att && a.w1 && a.b1 && a.w2 && a.b2 && a.w3 && a.b3 && a.out;
}

hipError_t launch_gnn_layer_small(const GnnSmallArgs& a, hipStream_t s) {
  if (!gnn_layer_small_supported(a)) return hipErrorInvalidValue;
  last_form = "gnn_layer_small:f32";
  const bool qkv = a.n3 == 3 * a.d;
  switch (a.d) {
    case 64: return qkv ? launch_t<64, 192>(a, s) : launch_t<64, 64>(a, s);
    case 128: return qkv ? launch_t<128, 384>(a, s) : launch_t<128, 128>(a, s);
    default: return qkv ? launch_t<256, 768>(a, s) : launch_t<256, 256>(a, s);
  }
}

}  // namespace imx
