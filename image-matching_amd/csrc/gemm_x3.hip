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
// Work split: PERSISTENT workgroups (two per CU) walk 128 x BN output tiles, column tiles of a row tile adjacent (and, through the
// XCD-aware start index, on one XCD: the row tile's A rows are fetched from HBM once); a tile's K is walked in chunks of 32 and the
// chunks of ALL of a workgroup's tiles form one continuous stream, so a tile's prologue (first loads) and epilogue (stores) overlap
// its neighbours' products.  The four waves each own a 32-column strip (128 x 32 at BN = 128: four row blocks, one weight fragment set).
//   A rows are loaded as fp32 float4 (K0 | K1 concatenation like the other forms), split in registers (v_cvt_pk_bf16_f32 rounds
//   and packs) and written to three bf16 planes in LDS, double buffered: chunk q+1 is split and stored WHILE chunk q is multiplied
//   (the VALU work hides beneath the bf16 MFMAs), one barrier per chunk, and the loads of chunk q+3 are issued as soon as the
//   registers of chunk q+1 are free.  LDS rows are 80 bytes (5 sixteen-byte slots, odd): the 16 lanes of a ds_read_b128 group read
//   rows that are distinct mod 16 and land on 16 distinct slots.
//   W was split once on the host and stored in FRAGMENT order -- [column block of 32][16-k step][plane][lane][8 bf16]: a wave's
//   B operand for one step and plane is one coalesced 1 KB load straight into registers (no LDS, no barrier), requested a chunk
//   ahead; every workgroup of a column reads the same fragments (L2 / L1 resident: 3 x K x Npad x 2 bytes per layer).
#include "imx_kernels.h"
#include "split3.h"
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

// position in a workgroup's stream of chunks: tile t (= row tile * column tiles + column tile), chunk c of its K
struct Cursor { int t, c, r0, n0; };

template <int BN, bool RES, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_x3(GemmArgs p, const __bf16* __restrict__ wx, int nct, int ntiles) {
  constexpr int WC = BN / 32, WR = 4 / WC, WROWS = BM / WR, RB = WROWS / 32;   // BN = 128: 1 x 4 waves of 128 x 32; BN = 64: 2 x 2 of 64 x 32
  // two separate objects (not one [2] array): the compiler must see that the stores of chunk q+1 never alias the loads of chunk q
  __shared__ __attribute__((aligned(16))) __bf16 As0[3][BM * RS];
  __shared__ __attribute__((aligned(16))) __bf16 As1[3][BM * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kb = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int K = p.K0 + p.K1, nch = K / KC, nst = K / 16;
  const int G = (int)gridDim.x;
  // XCD-aware start index: workgroups are dispatched round-robin over the 8 XCDs; consecutive tiles (the column tiles of a row
  // tile, neighbouring row tiles) go to workgroups of ONE XCD
  const int first = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G, Q = nmine * nch;

  auto place = [&](Cursor& cu) { cu.r0 = (cu.t / nct) * BM; cu.n0 = (cu.t % nct) * BN; };
  auto advance = [&](Cursor& cu) {          // block-uniform; past the end the cursor stays on the last chunk (harmless re-fetch)
    if (cu.c + 1 < nch) { ++cu.c; }
    else if (cu.t + G < ntiles) { cu.t += G; cu.c = 0; place(cu); }
  };

  f32x4 arega[4], aregb[4];      // two chunks of A in flight (a workgroup keeps 32 KB requested: the loop is latency bound otherwise)
  auto gload = [&](f32x4 (&areg)[4], const Cursor& cu) __attribute__((always_inline)) {
    const int k0 = cu.c * KC;
    const float* src = k0 < p.K0 ? p.a0 + k0 : p.a1 + (k0 - p.K0);
    const int ld = k0 < p.K0 ? p.lda0 : p.lda1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int grow = min(cu.r0 + (tid >> 3) + 32 * it, p.M - 1);       // rows past M re-read the last row (never stored)
      areg[it] = *reinterpret_cast<const f32x4*>(src + (size_t)grow * ld + (tid & 7) * 4);
    }
  };
  auto lstore = [&](__bf16 (&Ad)[3][BM * RS], const f32x4 (&areg)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      bf16x4 h, m, l;
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        split_bf16x2 h2, m2, l2;
        split3_pair(areg[it][t], areg[it][t + 1], h2, m2, l2);
        h[t] = h2[0]; h[t + 1] = h2[1]; m[t] = m2[0]; m[t + 1] = m2[1]; l[t] = l2[0]; l[t + 1] = l2[1];
      }
      const int o = ((tid >> 3) + 32 * it) * RS + (tid & 7) * 4;
      *reinterpret_cast<bf16x4*>(&Ad[0][o]) = h;
      *reinterpret_cast<bf16x4*>(&Ad[1][o]) = m;
      *reinterpret_cast<bf16x4*>(&Ad[2][o]) = l;
    }
  };
  // a wave's weight fragments: column block nb, step st, plane pl -> 512 elements at ((nb * nst + st) * 3 + pl) * 512
  auto wload = [&](bf16x8 (&wf)[2][3], const Cursor& cu) __attribute__((always_inline)) {
    const __bf16* wb = wx + ((size_t)((cu.n0 >> 5) + wc) * nst + 2 * cu.c) * (3 * 512) + lane * 8;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wf[st][pl] = *reinterpret_cast<const bf16x8*>(wb + (st * 3 + pl) * 512);
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
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][PA[t]], wf[PB[t]], acc[rb], 0, 0, 0);
  };

  // epilogue from registers: lane (col = i, kb) holds rows (r & 3) + 8 (r >> 2) + 4 kb of its column; 32 lanes store 128
  // consecutive bytes of a row.  Tiles entirely inside M (block-uniform) skip the per-row bound checks.
  auto epilogue = [&](const Cursor& cu, auto full) __attribute__((always_inline)) {
    const int rbase = cu.r0 + wr * WROWS + 4 * kb;
    const int col = cu.n0 + wc * 32 + i;
    if (col < p.N) {
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
          float v = acc[rb][r] + bias;
          if (RELU) v = fmaxf(v, 0.f);
          if (RES) v = rv[r] + v;
          o[ro * p.ldo] = v;
        }
      }
    }
    // the q|k|v projection: the maxima the two-plane fp16 attention scales its operands by, from the registers (a tile's 128 rows
    // belong to one (side, pair): aN0p, aN1p % 128 == 0; a wave's 32 columns to one of q | k | v)
    if (p.amax) {
      const int side = cu.r0 >= p.aB * p.aN0p ? 1 : 0;
      const int rel = cu.r0 - side * p.aB * p.aN0p, Np = side ? p.aN1p : p.aN0p;
      const int b = rel / Np;
      const int n = side ? (p.an1 ? p.an1[b] : p.aN1) : (p.an0 ? p.an0[b] : p.aN0);
      const int left = n - (rel - b * Np) - (wr * WROWS + 4 * kb);        // rows (local index ro) below this are valid
      unsigned mx = 0;
      if (col < p.N) {
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = rb * 32 + (r & 3) + 8 * (r >> 2);
            const unsigned bits = __builtin_bit_cast(unsigned, acc[rb][r] + bias) & 0x7fffffffu;
            mx = ro < left ? max(mx, bits) : mx;
          }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
      if (lane == 0 && mx) atomicMax(p.amax + (side * p.aB + b) * 4 + (cu.n0 + wc * 32) / (p.N / 3), mx);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  };

  Cursor comp{first, 0, 0, 0};
  place(comp);
  Cursor wcur = comp, acur = comp;
  bf16x8 wfa[2][3], wfb[2][3];       // the two steps' weight fragments of the current and of the next chunk
  gload(arega, acur);                // chunk 0
  advance(acur);
  gload(aregb, acur);                // chunk 1
  wload(wfa, wcur);
  lstore(As0, arega);
  advance(acur);
  gload(arega, acur);                // chunk 2
  __syncthreads();
  // one chunk: branch-free products (one scheduling region: the split of chunk q+1 issues beneath the MFMAs of chunk q, the weight
  // fragments of chunk q+1 are requested a whole chunk ahead, the A rows of chunk q+3 as soon as the registers of chunk q+1 are
  // free); a tile's last chunk is followed by its epilogue (block-uniform branch), while the next tile's operands are in flight
  auto chunk = [&](const __bf16 (&Ar)[3][BM * RS], __bf16 (&Ad)[3][BM * RS], const bf16x8 (&wcurf)[2][3], bf16x8 (&wnext)[2][3],
                   f32x4 (&areg)[4]) __attribute__((always_inline)) {
    advance(wcur);
    wload(wnext, wcur);
    lstore(Ad, areg);                        // chunk q+1
    advance(acur);
    gload(areg, acur);                       // chunk q+3
    step(Ar, 0, wcurf[0]);
    step(Ar, 1, wcurf[1]);
#pragma unroll
    for (int g = 0; g < 12 * RB; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // three VALU beneath it
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // and at most one LDS store
    }
    __syncthreads();
    if (comp.c == nch - 1) {                 // block-uniform
      if (comp.r0 + BM <= p.M) epilogue(comp, BoolC<true>{}); else epilogue(comp, BoolC<false>{});
    }
    advance(comp);
  };
  for (int q = 0; q < Q; q += 2) {
    chunk(As0, As1, wfa, wfb, aregb);
    if (q + 1 < Q) chunk(As1, As0, wfb, wfa, arega);       // block-uniform
  }
}
}  // namespace

bool gemm_x3_supported(const GemmArgs& a) {
  if (a.K0 % KC || a.K1 % KC || a.Npad % 64 || a.M <= 0) return false;
  if ((a.lda0 & 3) || (a.K1 && (!a.a1 || (a.lda1 & 3)))) return false;
  return true;
}

bool gemm_x3_amax_supported(const GemmArgs& a) {
  return gemm_x3_supported(a) && !a.res && !a.relu && a.N % 3 == 0 && (a.N / 3) % 32 == 0 && a.aB > 0 && a.aN0p % BM == 0 && a.aN1p % BM == 0 &&
         a.M == a.aB * (a.aN0p + a.aN1p);
}

hipError_t launch_gemm_x3(const GemmArgs& a, const void* wx3, hipStream_t s) {
  if (!gemm_x3_supported(a) || !wx3) return hipErrorInvalidValue;
  if (a.amax && !gemm_x3_amax_supported(a)) return hipErrorInvalidValue;
  const __bf16* wx = static_cast<const __bf16*>(wx3);
  const bool wide = a.Npad % 128 == 0;
  const int nct = a.Npad / (wide ? 128 : 64), ntiles = ((a.M + BM - 1) / BM) * nct;
  // persistent: two workgroups per CU; fewer tiles than that -> one workgroup per tile
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int want = 2 * cus;
  last_form = "gemm_x3:bf16x3";
  const dim3 grid((unsigned)(ntiles < want ? ntiles : want));
#define IMX_X3(BN_)                                                                                      \
  if (a.res) {                                                                                           \
    if (a.relu) hipLaunchKernelGGL((gemm_x3<BN_, true, true>), grid, dim3(256), 0, s, a, wx, nct, ntiles);  \
    else hipLaunchKernelGGL((gemm_x3<BN_, true, false>), grid, dim3(256), 0, s, a, wx, nct, ntiles);        \
  } else {                                                                                               \
    if (a.relu) hipLaunchKernelGGL((gemm_x3<BN_, false, true>), grid, dim3(256), 0, s, a, wx, nct, ntiles); \
    else hipLaunchKernelGGL((gemm_x3<BN_, false, false>), grid, dim3(256), 0, s, a, wx, nct, ntiles);       \
  }
  if (wide) { IMX_X3(128) } else { IMX_X3(64) }
#undef IMX_X3
  return hipGetLastError();
}

}  // namespace imx
