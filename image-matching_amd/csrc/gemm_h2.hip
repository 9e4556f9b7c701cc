// gemm_h2.hip -- gemm_x3.hip's linear product (superglue_test.py:49-60,92-119,214-216) with both operands as TWO fp16 planes and three
// plane products (h,m) (m,h) (h,h) on v_mfma_f32_32x32x16_f16: half the matrix work of the six bf16 products, and a split of 2.5
// instead of 3.5 VALU instructions per value (round 6; VERDICT r5 missing 3: at descriptor_dim 256 the layer tail is not fused and
// C5 ran q|k|v, mlp.0' and mlp.3 as three six-product launches per layer, 4.3 of its 21.3 ms).
//
// fp16 has five exponent bits, so both operands carry a power of two (the scheme of conv3x3_wino24h / attention FmtH2 / gnn_tail_h2):
//   * W: one power of two per matrix, chosen on the host so that max |W| lands in [2^13, 2^14) (GemmArgs::w_inv undoes it); planes
//     pre-split, in B-fragment order [column block of 32][16-k step][plane (2)][lane][8 halves];
//   * A: one power of two per 128-row tile from the ACTUAL maximum of its operand(s) over the valid rows of the tile's (side, pair) --
//     words written by the kernel that produced the operand (this kernel's epilogue: amax_row / the q|k|v thirds; rows_amax for the
//     first layer; for the attention output the key side's max |v|, of whose rows it is a convex combination).  A tile's rows belong
//     to one (side, pair) (padded counts are multiples of 128: gemm_h2_supported).  A value keeps 22 bits down to 2^-16 of that
//     maximum and is exact to 2^-38 of it below; rows past the pair's keypoint count are not covered by the maximum and are ZEROED
//     while they are staged (their outputs are the bias: finite, as every form leaves them unused).
// Work split, pipeline and epilogue are gemm_x3's: persistent workgroups walk 128 x BN tiles, the 32-k chunks of all their tiles one
// stream; A rows arrive as fp32 float4, are scaled and split in registers (v_cvt_pk_f16_f32, two v_dot2c_f32_f16 residuals,
// v_cvt_pk_f16_f32) and stored as two planes (80-byte rows) under the previous chunk's MFMAs.
#include "imx_kernels.h"
#include <cstdint>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {
template <bool V>
struct BoolC { static constexpr bool value = V; };

constexpr int BM = 128, KC = 32, RS = 40;   // rows per tile, k per chunk, LDS row stride in halves (80 bytes: an odd number of 16-byte slots)

// the power of two that brings `bound` (> 0) to [2^13, 2^14) (exponents clamped so that the scale and its reciprocal stay normal)
__device__ __forceinline__ float pow2_of_bound(float bound) {
  unsigned e = (__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu;
  e = e < 90u ? 90u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (267u - e) << 23);
}

// x = h + m in fp16, two values at a time (attention_x3.hip's FmtH2::split: constants through SGPRs behind an asm, hipcc 7.2 folds a
// packed {-1, 0} into the inline constant -1.0, which v_dot2c_f32_f16 does not read as a packed pair)
__device__ __forceinline__ void split_h2(float x0, float x1, f16x2& h, f16x2& m) {
  unsigned lo_u, hi_u;
  asm("s_mov_b32 %0, 0x0000bc00" : "=s"(lo_u));
  asm("s_mov_b32 %0, 0xbc000000" : "=s"(hi_u));
  const f16x2 lo = __builtin_bit_cast(f16x2, lo_u), hi = __builtin_bit_cast(f16x2, hi_u);
  h[0] = (_Float16)x0; h[1] = (_Float16)x1;
  const float r0 = __builtin_amdgcn_fdot2(h, lo, x0, false);
  const float r1 = __builtin_amdgcn_fdot2(h, hi, x1, false);
  m[0] = (_Float16)r0; m[1] = (_Float16)r1;
}

// position in a workgroup's stream of chunks: tile t (= row tile * column tiles + column tile), chunk c of its K; the tile's
// (side, pair) index, its valid rows (local index < left) and the power of two of its A operand
struct Cursor { int t, c, r0, n0, sp, left; float sA; };

template <int BN, bool RES, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_h2(GemmArgs p, const _Float16* __restrict__ wx, int nct, int ntiles, int stagger) {
  constexpr int WC = BN / 32, WR = 4 / WC, WROWS = BM / WR, RB = WROWS / 32;   // BN = 128: 1 x 4 waves of 128 x 32; BN = 64: 2 x 2 of 64 x 32
  // two separate objects (not one [2] array): the compiler must see that the stores of chunk q+1 never alias the loads of chunk q
  __shared__ __attribute__((aligned(16))) _Float16 As0[2][BM * RS];
  __shared__ __attribute__((aligned(16))) _Float16 As1[2][BM * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kb = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int K = p.K0 + p.K1, nch = K / KC, nst = K / 16;
  const int G = (int)gridDim.x;
  // XCD-aware start index (gemm_x3.hip)
  const int first = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nmine = (ntiles - first + G - 1) / G, Q = nmine * nch;
  // Staggered start (round 6): the workgroups run in lockstep -- same K, same start -- so every tile boundary was a chip-wide burst of
  // stores followed by a chip-wide wait for it (stamps, profiles/r06_gemm_h2_trace.txt: 18.7 k cycles of s_waitcnt vmcnt(0) after the first
  // tile of q|k|v at C5 against 1.2 - 4.5 k after the later, already spread ones).  Eight groups of workgroups start 64 x 31 cycles apart:
  // C5's three launches per layer 3.80 -> 3.59 ms per step (2 x 127, 4 x 63, 4 x 127, 2 x 254: 3.66 / 3.65 / 3.61 / 3.64).  Timing only.
  if (stagger)
    for (int k = 0; k < ((int)blockIdx.x * 8) / G; ++k) __builtin_amdgcn_s_sleep(31);

  auto place = [&](Cursor& cu) {
    cu.r0 = (cu.t / nct) * BM;
    cu.n0 = (cu.t % nct) * BN;
    const int side = cu.r0 >= p.aB * p.aN0p ? 1 : 0;
    const int rel = cu.r0 - side * p.aB * p.aN0p, Np = side ? p.aN1p : p.aN0p;
    const int b = __builtin_amdgcn_readfirstlane(rel / Np);
    // (uniform words through the SCALAR cache -- written by earlier launches, so it is clean: as vector loads they were waited for with
    // vmcnt(0), and any vector-memory access on a conditional path of the chunk loop makes hipcc's wait insertion fall back to
    // vmcnt(0) at the loop header: see the loop)
    typedef const int __attribute__((address_space(4)))* ci32p;
    typedef const unsigned __attribute__((address_space(4)))* cu32p;
    const ci32p cn = (ci32p)(uintptr_t)(side ? p.an1 : p.an0);
    const int n = cn ? cn[b] : (side ? p.aN1 : p.aN0);
    cu.sp = side * p.aB + b;
    cu.left = n - (rel - b * Np);
    float bound = __builtin_bit_cast(float, ((cu32p)(uintptr_t)p.sa0)[(size_t)cu.sp * p.sa0_stride + p.sa0_off]);
    if (p.sa1) {
      const int ksp = p.sa1_cross ? (1 - side) * p.aB + b : cu.sp;
      bound = fmaxf(bound, __builtin_bit_cast(float, ((cu32p)(uintptr_t)p.sa1)[(size_t)ksp * p.sa1_stride + p.sa1_off]));
    }
    cu.sA = pow2_of_bound(fmaxf(bound, 1e-30f));
  };
  auto advance = [&](Cursor& cu) {          // block-uniform; past the end the cursor stays on the last chunk (harmless re-fetch)
    if (cu.c + 1 < nch) { ++cu.c; }
    else if (cu.t + G < ntiles) { cu.t += G; cu.c = 0; place(cu); }
  };

  // two chunks of A in flight (gemm_x3.hip).  (Four -- 64 KB requested per workgroup -- measured the same, round 6: 1.64 / 1.48 / 1.14 ms
  // per C5 step for mlp.0' / q|k|v / mlp.3 either way; the registers went to the operand fragments instead, below.)
  f32x4 arega[4], aregb[4];
  float sca = 1.f, scb = 1.f;    // ... with the scale and the valid-row count of the tiles they belong to
  int lfa = BM, lfb = BM;
  auto gload = [&](f32x4 (&areg)[4], float& sc, int& lf, const Cursor& cu) __attribute__((always_inline)) {
    const int k0 = cu.c * KC;
    const float* src = k0 < p.K0 ? p.a0 + k0 : p.a1 + (k0 - p.K0);
    const int ld = k0 < p.K0 ? p.lda0 : p.lda1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int grow = min(cu.r0 + (tid >> 3) + 32 * it, p.M - 1);       // rows past M re-read the last row (never stored)
      areg[it] = *reinterpret_cast<const f32x4*>(src + (size_t)grow * ld + (tid & 7) * 4);
    }
    sc = cu.sA;
    lf = cu.left;
  };
  auto lstore = [&](_Float16 (&Ad)[2][BM * RS], const f32x4 (&areg)[4], float sc, int lf) __attribute__((always_inline)) {
    const f32x4 s4 = {sc, sc, sc, sc};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 v = areg[it] * s4;
      f16x4 h, m;
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        f16x2 h2, m2;
        split_h2(v[t], v[t + 1], h2, m2);
        h[t] = h2[0]; h[t + 1] = h2[1]; m[t] = m2[0]; m[t + 1] = m2[1];
      }
      const int o = ((tid >> 3) + 32 * it) * RS + (tid & 7) * 4;
      *reinterpret_cast<f16x4*>(&Ad[0][o]) = h;
      *reinterpret_cast<f16x4*>(&Ad[1][o]) = m;
    }
  };
  // rows past the pair's keypoint count (the tile that holds its end: block-uniform, rare): their planes are overwritten with zeros
  // AFTER the straight-line split above -- a mask inside it was a branch per row group in the middle of the region that has to
  // interleave with the MFMAs
  auto lzero = [&](_Float16 (&Ad)[2][BM * RS], int lf) __attribute__((always_inline)) {
    if (lf < BM) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if ((tid >> 3) + 32 * it >= lf) {
          const int o = ((tid >> 3) + 32 * it) * RS + (tid & 7) * 4;
          *reinterpret_cast<f16x4*>(&Ad[0][o]) = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          *reinterpret_cast<f16x4*>(&Ad[1][o]) = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        }
    }
  };
  // a wave's weight fragments: column block nb, step st, plane pl -> 512 halves at ((nb * nst + st) * 2 + pl) * 512
  auto wload = [&](f16x8 (&wf)[2][2], const Cursor& cu) __attribute__((always_inline)) {
    const _Float16* wb = wx + ((size_t)((cu.n0 >> 5) + wc) * nst + 2 * cu.c) * (2 * 512) + lane * 8;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wf[st][pl] = *reinterpret_cast<const f16x8*>(wb + (st * 2 + pl) * 512);
  };

  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

  // The A fragments of BOTH 16-k steps of a chunk are requested before its first MFMA (sixteen ds_read_b128, 64 registers): left to
  // the register allocator under this kernel's pressure each fragment was read into the same four registers right before its MFMA --
  // ds_read, s_waitcnt lgkmcnt(0), v_mfma, sixteen times per chunk, an LDS latency each (round 6: the ISA of the first build).
  auto frags = [&](const _Float16 (&Ar)[2][BM * RS], f16x8 (&af)[2][RB][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          af[s][rb][pl] = *reinterpret_cast<const f16x8*>(&Ar[pl][(wr * WROWS + rb * 32 + i) * RS + s * 16 + kb * 8]);
  };
  // three plane products per block, smallest first; the row blocks interleave so consecutive MFMAs use different accumulators
  auto step = [&](const f16x8 (&af)[RB][2], const f16x8 (&wf)[2]) __attribute__((always_inline)) {
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rb][PA[t]], wf[PB[t]], acc[rb], 0, 0, 0);
  };

  // Developer instrumentation (-DGH2_TRACE; tools/gemm_h2_trace.py, profiles/r06_gemm_h2_trace.txt): s_memtime of the first 64 workgroups at
  // every chunk's entry / before its barrier / after it, and at four points of every tile's epilogue (-DGH2_TRACE=2: also when a chunk's
  // fragments have arrived -- that stamp keeps the split from interleaving with the MFMAs)
#ifdef GH2_TRACE
#define GH2_STAMP(k) do { if (trw && qi < 24) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tid == 0) trw[qi * 4 + (k)] = t_; } } while (0)
#define GH2_ESTAMP(k) do { if (trw && ntile_done < 4) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tid == 0) trw[96 + ntile_done * 8 + (k)] = t_; } } while (0)
  unsigned long long* const trw = (p.trace && blockIdx.x < 64) ? p.trace + (size_t)blockIdx.x * 128 : nullptr;
  int ntile_done = 0;
#else
#define GH2_STAMP(k) do {} while (0)
#define GH2_ESTAMP(k) do {} while (0)
#endif
  int qi = 0;                        // chunks multiplied so far
  // epilogue from registers (gemm_x3.hip): lane (col = i, kb) holds rows (r & 3) + 8 (r >> 2) + 4 kb of its column; the accumulators
  // carry sA sW: one fma un-scales and adds the bias
  auto epilogue = [&](const Cursor& cu, auto full) __attribute__((always_inline)) {
    const int rbase = cu.r0 + wr * WROWS + 4 * kb;
    const int col = cu.n0 + wc * 32 + i;
    const float inv = p.w_inv / cu.sA;                    // powers of two: exact
    // rows (local index ro) below `left` are valid; nothing is when no maximum is wanted (branch-free: as a nested condition hipcc
    // compiled the tracking into a branch per element, 64 per tile and wave)
    const bool track = p.amax || p.amax_row;
    const int left = track ? cu.left - (wr * WROWS + 4 * kb) : -0x40000000;
    unsigned mx = 0;
    GH2_ESTAMP(0);
    if (col < p.N) {
      const float bias = p.bias ? p.bias[col] : 0.f;
      float* o = p.out + (size_t)rbase * p.ldo + col;
      const float* rs = RES ? p.res + (size_t)rbase * p.ldr + col : nullptr;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        // the residual may alias the output (x += ...): the sixteen loads of a row block are issued together, before its stores
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
          float v = fmaf(acc[rb][r], inv, bias);
          if (RELU) v = fmaxf(v, 0.f);
          if (RES) v = rv[r] + v;
          o[ro * p.ldo] = v;
          // what the NEXT kernel scales this tensor by: the largest |value| over the pair's valid rows
          mx = max(mx, ro < left ? __builtin_bit_cast(unsigned, v) & 0x7fffffffu : 0u);
        }
      }
    }
    GH2_ESTAMP(1);
    if (track) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
      if (lane == 0 && mx) {
        // the q|k|v projection: by column third (a wave's 32 columns lie inside one of q | k | v); else one word per (side, pair)
        if (p.amax) atomicMax(p.amax + (size_t)cu.sp * 4 + (cu.n0 + wc * 32) / (p.N / 3), mx);
        if (p.amax_row) atomicMax(p.amax_row + (size_t)cu.sp * p.amax_row_stride + p.amax_row_off, mx);
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    // This path (once per tile) rejoins the chunk loop with NOTHING outstanding: hipcc's s_waitcnt insertion merges the pending
    // vector-memory state of all predecessors of a block, and with this path's stores and bias load in the mix it gave up counting and
    // put s_waitcnt vmcnt(0) at the head of EVERY chunk -- the A rows and weight fragments requested one and two chunks ahead were
    // waited for one chunk later: a memory latency per chunk (round 6: 5.6 k cycles per chunk whatever the matrix work, both forms).
    GH2_ESTAMP(2);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    GH2_ESTAMP(3);
#ifdef GH2_TRACE
    ++ntile_done;
#endif
  };

  Cursor comp{first, 0, 0, 0, 0, BM, 1.f};
  place(comp);
  Cursor wcur = comp, acur = comp;
  f16x8 wfa[2][2], wfb[2][2];        // the two steps' weight fragments of the current and of the next chunk
  gload(arega, sca, lfa, acur);      // chunk 0
  advance(acur);
  gload(aregb, scb, lfb, acur);      // chunk 1
  wload(wfa, wcur);
  lstore(As0, arega, sca, lfa);
  lzero(As0, lfa);
  advance(acur);
  __syncthreads();
  // one chunk (gemm_x3.hip): the fragments of chunk q first, then its MFMAs with the split of chunk q+1 issuing beneath them; the
  // weight fragments of chunk q+1 are requested a whole chunk ahead, the A rows of chunk q+3 as soon as the registers of chunk q+1 are free
  auto chunk = [&](const _Float16 (&Ar)[2][BM * RS], _Float16 (&Ad)[2][BM * RS], const f16x8 (&wcurf)[2][2], f16x8 (&wnext)[2][2],
                   f32x4 (&areg)[4], float& sc, int& lf) __attribute__((always_inline)) {
    GH2_STAMP(0);
    f16x8 af[2][RB][2];
    frags(Ar, af);
    advance(wcur);
    wload(wnext, wcur);
    __builtin_amdgcn_sched_barrier(0);       // (the reads stay in front: see frags)
    lstore(Ad, areg, sc, lf);                // chunk q+1
    const int lf_stored = lf;
    advance(acur);
    gload(areg, sc, lf, acur);               // chunk q+3
#if defined(GH2_TRACE) && GH2_TRACE > 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the fragments have arrived: what the first MFMA waits for)
    GH2_STAMP(1);
#endif
    step(af[0], wcurf[0]);
    step(af[1], wcurf[1]);
#pragma unroll
    for (int g = 0; g < 6 * RB; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // three VALU beneath it
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // and at most one LDS store
    }
    lzero(Ad, lf_stored);
    GH2_STAMP(2);
    __syncthreads();
    GH2_STAMP(3);
    if (comp.c == nch - 1 && qi < Q) {       // block-uniform (qi >= Q: the padding chunk of an odd stream, below)
      if (comp.r0 + BM <= p.M) epilogue(comp, BoolC<true>{}); else epilogue(comp, BoolC<false>{});
    }
    advance(comp);
    ++qi;
  };
  // The two chunk bodies of an iteration are BOTH executed (an odd stream runs one padding chunk on re-fetched operands whose products
  // are never stored): a branch around the second body is a second predecessor with other loads in flight -- the same merge.
  __builtin_amdgcn_s_waitcnt(0x0f70);        // (and the loop is entered with nothing outstanding but what the prologue's last lines request)
  gload(arega, sca, lfa, acur);              // chunk 2
  for (int q = 0; q < Q; q += 2) {
    chunk(As0, As1, wfa, wfb, aregb, scb, lfb);
    chunk(As1, As0, wfb, wfa, arega, sca, lfa);
  }
}

// max |x| over the valid rows of every (side, pair) (rows of d floats, d % 4 == 0): grid (chunks, 2 B), 256 threads striding the
// float4 of the pair's valid rows; bit patterns through atomicMax (order-independent: reproducible bit for bit)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rows_amax_any_kernel(const float* x, int d4, int B, int N0p, int N1p, const int* n0, const int* n1, int N0, int N1,
                                                            unsigned* amax) {
  const int side = blockIdx.y / B, b = blockIdx.y % B;
  const int Np = side ? N1p : N0p;
  const int n = side ? (n1 ? n1[b] : N1) : (n0 ? n0[b] : N0);
  const size_t base = (side ? (size_t)B * N0p : 0) + (size_t)b * Np;
  const u32x4* src = reinterpret_cast<const u32x4*>(x) + base * d4;
  const int total = n * d4;
  unsigned mx = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const u32x4 v = src[e];
    mx = max(mx, max(max(v[0] & 0x7fffffffu, v[1] & 0x7fffffffu), max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu)));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(amax + blockIdx.y, mx);
}
}  // namespace

hipError_t launch_rows_amax_any(const float* x, int d, int B, int N0p, int N1p, const int* n0, const int* n1, int N0, int N1, unsigned* amax, hipStream_t s) {
  if (d <= 0 || (d & 3) || B <= 0) return hipErrorInvalidValue;
  const int nmax = N0p > N1p ? N0p : N1p;
  hipLaunchKernelGGL(rows_amax_any_kernel, dim3((unsigned)(nmax >= 512 ? 8 : 1), (unsigned)(2 * B)), dim3(256), 0, s, x, d / 4, B, N0p, N1p, n0, n1, N0, N1, amax);
  return hipGetLastError();
}

bool gemm_h2_supported(const GemmArgs& a) {
  if (!gemm_x3_supported(a)) return false;
  if (!a.sa0 || !(a.w_inv > 0.f) || a.aB <= 0 || a.aN0p % BM || a.aN1p % BM || a.M != a.aB * (a.aN0p + a.aN1p)) return false;
  if (a.amax && (a.res || a.relu || a.N % 3 || (a.N / 3) % 32)) return false;
  return true;
}

hipError_t launch_gemm_h2(const GemmArgs& a, const void* wh2, hipStream_t s) {
  if (!gemm_h2_supported(a) || !wh2) return hipErrorInvalidValue;
  const _Float16* wx = static_cast<const _Float16*>(wh2);
  // persistent: two workgroups per CU; fewer tiles than that -> one workgroup per tile
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int want = 2 * cus;
  // (64-wide tiles where the 128-wide ones leave a workgroup a single tile -- mlp.3 at C5 -- measured: 1.34 vs 1.09 ms per step, kept wide)
  const bool wide = a.Npad % 128 == 0;
  const int nct = a.Npad / (wide ? 128 : 64), ntiles = ((a.M + BM - 1) / BM) * nct;
  last_form = "gemm_h2:f16x2";
  const dim3 grid((unsigned)(ntiles < want ? ntiles : want));
  const int stagger = ntiles >= want ? 1 : 0;             // (a full chip: see the kernel)
#define IMX_H2(BN_)                                                                                      \
  if (a.res) {                                                                                           \
    if (a.relu) hipLaunchKernelGGL((gemm_h2<BN_, true, true>), grid, dim3(256), 0, s, a, wx, nct, ntiles, stagger);  \
    else hipLaunchKernelGGL((gemm_h2<BN_, true, false>), grid, dim3(256), 0, s, a, wx, nct, ntiles, stagger);        \
  } else {                                                                                               \
    if (a.relu) hipLaunchKernelGGL((gemm_h2<BN_, false, true>), grid, dim3(256), 0, s, a, wx, nct, ntiles, stagger); \
    else hipLaunchKernelGGL((gemm_h2<BN_, false, false>), grid, dim3(256), 0, s, a, wx, nct, ntiles, stagger);       \
  }
  if (wide) { IMX_H2(128) } else { IMX_H2(64) }
#undef IMX_H2
  return hipGetLastError();
}

}  // namespace imx
