// gemm_x3.hip — the 1x1-conv / linear product (superglue_test.py:49-60,92-119,214-216; superpoint_test.py:140-146) with every fp32
// product carried by the bf16 matrix pipe as SIX bf16 term products.
//
// gfx950's fp32 MFMA runs at the vector rate (157 TFLOP/s, 1/16 of the bf16 MFMA) and shares the VALU datapath.  An fp32 value
// splits exactly into three bf16 terms x = h + m + l (8 significant bits each, round-to-nearest residuals); of the nine term
// products of a.b the six largest -- hh, hm, mh, hl, lh, mm -- leave an error below 2^-23 |a b|, every bf16 x bf16 product is exact
// in fp32, and v_mfma_f32_32x32x16_bf16 adds its sixteen products and the accumulator with ONE round-to-nearest
// (tools/ubench/mfma_bf16x3.hip, profiles/r02_mfma_bf16x3.txt: against a float64 product the six-product form has 0.7x the rms
// error of the fp32 MFMA chain at K = 32..1024, with no bias).  Six 32-cycle MFMAs replace eight 64-cycle ones per 32x32x16 block
// (3/8 of the cycles), and VALU instructions issue beneath them (about four per MFMA for free).
//
// Work split: a workgroup owns a 128 x BN output tile and walks K in chunks of 32; its four waves each own a 32-column strip
// (128 x 32 at BN = 128: four row blocks against one weight fragment set).
//   A rows are loaded as fp32 float4 (K0 | K1 concatenation like the other forms), split in registers (v_cvt_pk_bf16_f32 rounds
//   and packs) and written to three bf16 planes in LDS, double buffered: chunk c+1 is split and stored WHILE chunk c is multiplied
//   (the VALU work hides beneath the bf16 MFMAs), one barrier per chunk, and the loads of chunk c+2 are issued as soon as the
//   registers are free.  LDS rows are 80 bytes (5 sixteen-byte slots, odd): the 16 lanes of a ds_read_b128 group read rows that
//   are distinct mod 16 and land on 16 distinct slots.
//   W was split once on the host and stored in FRAGMENT order -- [column block of 32][16-k step][plane][lane][8 bf16]: a wave's
//   B operand for one step and plane is one coalesced 1 KB load straight into registers (no LDS, no barrier), prefetched one
//   step ahead; every workgroup of a column reads the same fragments (L2 / L1 resident: 3 x K x Npad x 2 bytes per layer).
#include "imx_kernels.h"
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
template <bool V>
struct BoolC { static constexpr bool value = V; };

constexpr int BM = 128, KC = 32, RS = 40;   // rows per tile, k per chunk, LDS row stride in bf16 elements (80 bytes)

template <int BN, bool RES, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_x3(GemmArgs p, const __bf16* __restrict__ wx, int exp, long long* trace) {
  constexpr int WC = BN / 32, WR = 4 / WC, WROWS = BM / WR, RB = WROWS / 32;   // BN = 128: 1 x 4 waves of 128 x 32; BN = 64: 2 x 2 of 64 x 32
  // two separate objects (not one [2] array): the compiler must see that the stores of chunk c+1 never alias the loads of chunk c
  __shared__ __attribute__((aligned(16))) __bf16 As0[3][BM * RS];
  __shared__ __attribute__((aligned(16))) __bf16 As1[3][BM * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kb = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int r0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int K = p.K0 + p.K1, nch = K / KC, nst = K / 16;

  int tslot = 0;
  auto stamp = [&]() { if (trace && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256 && tslot < 14) trace[blockIdx.x * 16 + tslot++] = __builtin_amdgcn_s_memtime(); };
  stamp();
  if (trace && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) trace[blockIdx.x * 16 + 14] = wall_clock64();
  f32x4 arega[4], aregb[4];      // two chunks of A in flight (a workgroup keeps 32 KB requested: the loop is latency bound otherwise)
  auto gload = [&](f32x4 (&areg)[4], int c) __attribute__((always_inline)) {
    const int k0 = c * KC;
    const float* src = k0 < p.K0 ? p.a0 + k0 : p.a1 + (k0 - p.K0);
    const int ld = k0 < p.K0 ? p.lda0 : p.lda1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int grow = min(r0 + (tid >> 3) + 32 * it, p.M - 1);       // rows past M re-read the last row (never stored)
      areg[it] = *reinterpret_cast<const f32x4*>(src + (size_t)grow * ld + (tid & 7) * 4);
    }
  };
  auto lstore = [&](__bf16 (&Ad)[3][BM * RS], const f32x4 (&areg)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      bf16x4 h, m, l;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = areg[it][t];
        h[t] = (__bf16)x;
        const float r1 = x - (float)h[t];
        m[t] = (__bf16)r1;
        l[t] = (__bf16)(r1 - (float)m[t]);
      }
      const int o = ((tid >> 3) + 32 * it) * RS + (tid & 7) * 4;
      *reinterpret_cast<bf16x4*>(&Ad[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&Ad[1][o]) = m;
      *reinterpret_cast<bf16x4*>(&Ad[2][o]) = l;
    }
  };
  // this wave's weight fragments: column block nb, step st, plane pl -> 512 elements at ((nb * nst + st) * 3 + pl) * 512
  const __bf16* wbase = wx + ((size_t)((n0 >> 5) + wc) * nst * 3) * 512 + lane * 8;
  auto wload = [&](bf16x8 (&wf)[3], int st) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(st * 3 + pl) * 512);
  };

  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

  // six term products per block, smallest first; the row blocks interleave so consecutive MFMAs use different accumulators
  auto step = [&](const __bf16 (&Ar)[3][BM * RS], int s, const bf16x8 (&wf)[3]) __attribute__((always_inline)) {
    bf16x8 af[RB][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        af[rb][pl] = *reinterpret_cast<const bf16x8*>(&Ar[pl][(wr * WROWS + rb * 32 + i) * RS + s * 16 + kb * 8]);
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        if (exp != 2 || t == 0) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][PA[t]], wf[PB[t]], acc[rb], 0, 0, 0);
  };

  bf16x8 wfa[2][3], wfb[2][3];       // the two steps' weight fragments of the current and of the next chunk
  gload(arega, 0);
  gload(aregb, min(1, nch - 1));
  wload(wfa[0], 0);
  wload(wfa[1], 1);
  lstore(As0, arega);
  gload(arega, min(2, nch - 1));
  __syncthreads();
  stamp();
  // one chunk: branch-free body (one scheduling region: the split of chunk c+1 issues beneath the MFMAs of chunk c, the weight
  // fragments of chunk c+1 are requested a whole chunk ahead, the A rows of chunk c+3 as soon as the registers of chunk c+1 are
  // free); past the end the last chunk / fragments are re-fetched and re-split into the idle buffer
  auto chunk = [&](int c, const __bf16 (&Ar)[3][BM * RS], __bf16 (&Ad)[3][BM * RS], const bf16x8 (&wcur)[2][3],
                   bf16x8 (&wnext)[2][3], f32x4 (&areg)[4]) __attribute__((always_inline)) {
    const int cn = min(c + 1, nch - 1);
    wload(wnext[0], 2 * cn);
    wload(wnext[1], 2 * cn + 1);
    lstore(Ad, areg);                        // chunk c+1
    gload(areg, min(c + 3, nch - 1));
    step(Ar, 0, wcur[0]);
    step(Ar, 1, wcur[1]);
#pragma unroll
    for (int g = 0; g < 12 * RB; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // three VALU beneath it
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // and at most one LDS store
    }
    __syncthreads();
  };
  for (int c = 0; c < nch; c += 2) {
    chunk(c, As0, As1, wfa, wfb, aregb);
    stamp();
    if (c + 1 < nch) { chunk(c + 1, As1, As0, wfb, wfa, arega); stamp(); }      // block-uniform
  }

  // epilogue from registers: lane (col = i, kb) holds rows (r & 3) + 8 (r >> 2) + 4 kb of its column; 32 lanes store 128
  // consecutive bytes of a row.  Tiles entirely inside M (block-uniform) skip the per-row bound checks.
  const int rbase = r0 + wr * WROWS + 4 * kb;
  const int col = n0 + wc * 32 + i;
  auto epilogue = [&](auto full) __attribute__((always_inline)) {
    if (col >= p.N) return;
    const float bias = p.bias ? p.bias[col] : 0.f;
    float* o = p.out + (size_t)rbase * p.ldo + col;
    const float* rs = RES ? p.res + (size_t)rbase * p.ldr + col : nullptr;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      // the residual may alias the output (x += ...): the sixteen loads of a row block are issued together, before its stores
      // (interleaved, every load would wait for the store before it: 64 serial round trips)
      float rv[16];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = rb * 32 + (r & 3) + 8 * (r >> 2);
          rv[r] = (decltype(full)::value || rbase + ro < p.M) ? rs[ro * p.ldr] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = rb * 32 + (r & 3) + 8 * (r >> 2);
        if (!decltype(full)::value && rbase + ro >= p.M) continue;
        if (exp == 1 && acc[rb][r] != 12345.f) continue;
        float v = acc[rb][r] + bias;
        if (RELU) v = fmaxf(v, 0.f);
        if (RES) v = rv[r] + v;
        o[ro * p.ldo] = v;
      }
    }
  };
  if (r0 + BM <= p.M) epilogue(BoolC<true>{}); else epilogue(BoolC<false>{});
  stamp();
  if (trace && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) trace[blockIdx.x * 16 + 15] = wall_clock64();
}
}  // namespace

long long* g_x3_trace = nullptr;   // tools/ubench/gemm_x3_bench.cpp: per-workgroup s_memtime stamps

bool gemm_x3_supported(const GemmArgs& a) {
  if (a.K0 % KC || a.K1 % KC || a.Npad % 64 || a.M <= 0) return false;
  if ((a.lda0 & 3) || (a.K1 && (!a.a1 || (a.lda1 & 3)))) return false;
  return true;
}

hipError_t launch_gemm_x3(const GemmArgs& a, const void* wx3, hipStream_t s) {
  if (!gemm_x3_supported(a) || !wx3) return hipErrorInvalidValue;
  const __bf16* wx = static_cast<const __bf16*>(wx3);
  const bool wide = a.Npad % 128 == 0;
  const char* ee = getenv("IMX_X3_EXP");
  const int exp = ee ? atoi(ee) : 0;
  const dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)(a.Npad / (wide ? 128 : 64)));
#define IMX_X3(BN_)                                                                                            \
  if (a.res) {                                                                                                 \
    if (a.relu) hipLaunchKernelGGL((gemm_x3<BN_, true, true>), grid, dim3(256), 0, s, a, wx, exp, g_x3_trace);                  \
    else hipLaunchKernelGGL((gemm_x3<BN_, true, false>), grid, dim3(256), 0, s, a, wx, exp, g_x3_trace);                        \
  } else {                                                                                                     \
    if (a.relu) hipLaunchKernelGGL((gemm_x3<BN_, false, true>), grid, dim3(256), 0, s, a, wx, exp, g_x3_trace);                 \
    else hipLaunchKernelGGL((gemm_x3<BN_, false, false>), grid, dim3(256), 0, s, a, wx, exp, g_x3_trace);                       \
  }
  if (wide) { IMX_X3(128) } else { IMX_X3(64) }
#undef IMX_X3
  return hipGetLastError();
}

}  // namespace imx
