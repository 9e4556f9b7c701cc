// conv3x3_f22x3.hip — 3x3 / pad 1 convolution + folded BatchNorm + ReLU (+ MaxPool2d(2)) of superpoint/models/unet_parts.py:10-48 and
// superpoint_test.py:113-123 as Winograd F(2x2, 3x3) whose 16 per-position GEMMs run on the bf16 matrix pipe, every fp32 product
// carried as six bf16 term products (gemm_x3.hip explains the split; tools/ubench/mfma_bf16x3.hip measures it).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 2 x 2 output tile ("wtile") and 4 x 4 input patch d
//   M_p[co][wtile] = sum_ci U_p[ci][co] V_p[wtile][ci],  p = (i, j) 16 positions, on v_mfma_f32_32x32x16_bf16.
// Why F(2x2) and not the F(2x4) of conv3x3_wino24.hip: 16 positions x a 32x32 accumulator block = 256 registers = the AGPR half
// of the register file at one wave per SIMD, so a wave keeps every position of its block and walks the input channels once
// (F(2x4): 384, tools/experimental/conv3x3_x3.hip); 4/3 the multiplies, on a pipe that needs 3/8 of the cycles.
//
// Operand delivery is what bounds a Winograd kernel on this pipe (six bf16 planes triple the bytes): BOTH operands go through
// LDS, one transformed row i (four positions) at a time.  Workgroup = 4 waves = 64 output channels x 64 wtiles (16 x 16 pixels),
// wave (wg, cg) = wtiles 32 wg.. x channels 32 cg..: every U fragment in LDS is read by two waves, every V fragment by two.
// The stream of steps (chunk of 16 input channels, row i) is software pipelined inside the wave; during the 24 MFMAs of step s
//   * U of step s+1 -- one contiguous 24 KB block of the host-made layout [3 planes][4 j][2 octets][64 co][8 ci] -- goes global ->
//     registers (requested at the start of the step) -> LDS (stored behind the MFMAs);
//   * V of step s+1 is transformed from the raw patch in LDS (thread = (wtile, channel quad): 8 ds_read_b128, the F(2,3) row and
//     column combinations, 16 values split into three bf16 terms, 12 ds_write_b64) -- each V value is produced once per workgroup;
//   * the next chunk's 18 x 18 x 16 patch goes global -> registers -> the other raw buffer.
// One barrier per step.  With U as the A operand a lane holds one wtile and four consecutive channels per register group for all
// 16 positions: output transform, bias, ReLU and the 2x2 max-pool are in-lane, every result leaves as one 16-byte store.
// LDS: 2 x (U 24 KB + V 24 KB) + 2 x 27 KB raw = 150 KB.
#include "imx_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TW = 8;                            // wtiles per tile side: 16 x 16 output pixels
constexpr int OT = 2 * TW, RP = OT + 2;          // 18 x 18 input patch (pad-1 halo)
constexpr int CK = 16;                           // input channels per chunk = one MFMA k step
// raw patch in LDS: [quad of 4 channels][column parity][row][12 slots: column / 2] x 16 bytes.  A lane of the transform is a wtile
// (twy, twx) reading row 2 twy + a, column 2 twx + b: slot 24 twy + twx + const -- the sixteen lanes of a ds_read_b128 group,
// (twy, twx) in {(0,0-3), (1,4-7), (2,4-7), (3,0-3)} or the complement, land on sixteen distinct slots (searched, tools/ notes)
constexpr int RSL = 12, RPST = RP * RSL;         // slots per row, per parity plane
constexpr int RAWF = 4 * 2 * RPST * 4;           // floats per raw buffer (27 KB)
constexpr int UVE = 3 * 4 * 2 * 64 * 8;          // bf16 elements of one step's U (or V): [3 planes][4 j][2 octets][64][8] = 24 KB
constexpr int NRAW = (RP * RP * 4 + 255) / 256;  // float4 pieces of the patch per thread (1296 -> 6)
constexpr int NU = UVE * 2 / 16 / 256;           // 16-byte pieces of a U block per thread (6)

__device__ __forceinline__ int roff(int q, int row, int col) { return ((q * 2 + (col & 1)) * RPST + row * RSL + (col >> 1)) * 4; }
__device__ __forceinline__ int uvoff(int pl, int j, int kb, int n) { return (((pl * 4 + j) * 2 + kb) * 64 + n) * 8; }

__device__ __forceinline__ void split2(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) {
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)(r0 - (float)m[0]); l[1] = (__bf16)(r1 - (float)m[1]);
}

// TRACE (IMX_X3_TRACE=1): s_memtime deltas per phase of a step, summed by thread 0 of every workgroup (bring-up instrumentation)
template <bool POOL, bool TRACE>
__global__ __launch_bounds__(256, 1) void conv3x3_f22x3(ConvArgs p, const __bf16* __restrict__ ux, int tiles_x, int tiles_y, int nitems, unsigned long long* trace) {
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define IMX_TS(i_)                                                                   \
  if constexpr (TRACE) {                                                             \
    const unsigned long long now = __builtin_amdgcn_s_memtime();                     \
    tph[i_] += now - tprev;                                                          \
    tprev = now;                                                                     \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw0 = smem;                                                    // [2][RAWF]
  __bf16* UV = reinterpret_cast<__bf16*>(smem + 2 * RAWF);               // [2 buffers][U | V][UVE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK, ncob = Cout / 64;
  // XCD-aware item order: consecutive items (the output blocks of one tile, neighbouring tiles) land on one XCD
  const int grid = (int)gridDim.x;
  const int item = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (item >= nitems) return;
  const int cob = item % ncob, tile = item / ncob;
  const int x0 = (tile % tiles_x) * OT, y0 = ((tile / tiles_x) % tiles_y) * OT, b = tile / (tiles_x * tiles_y);

  // ---- patch loader: piece e = (octet, row, px, half) in memory order of the channel-blocked input (B, Cin/8, H, W, 8)
  f32x4 lreg[NRAW];
  const float* inb = p.in + (size_t)b * Cin * H * W;
  auto gload = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NRAW; ++k) {
      const int e = tid + 256 * k;
      const int q = e % (RP * 2), row = (e / (RP * 2)) % RP, oct = e / (RP * 2 * RP);
      const int gy = y0 - 1 + row, gx = x0 - 1 + (q >> 1);
      const bool ok = e < RP * RP * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float* src = inb + (((size_t)(2 * c + oct) * H + gy) * W + gx) * 8 + 4 * (q & 1);
      lreg[k] = ok ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](float* raw) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NRAW; ++k) {
      const int e = tid + 256 * k;
      const int q = e % (RP * 2), row = (e / (RP * 2)) % RP, oct = e / (RP * 2 * RP);
      if (e < RP * RP * 4) *reinterpret_cast<f32x4*>(raw + roff(2 * oct + (q & 1), row, q >> 1)) = lreg[k];
    }
  };
  // ---- U of one step: a contiguous block of the host layout [Cout/64][Cin/16][4 rows i][UVE]
  u32x4 ureg[NU];
  const __bf16* ub = ux + (size_t)cob * nchunk * 4 * UVE;
  auto uload = [&](int step) __attribute__((always_inline)) {
    const u32x4* src = reinterpret_cast<const u32x4*>(ub + (size_t)step * UVE);
#pragma unroll
    for (int k = 0; k < NU; ++k) ureg[k] = src[tid + 256 * k];
  };
  auto ustore = [&](__bf16* U) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NU; ++k) reinterpret_cast<u32x4*>(U)[tid + 256 * k] = ureg[k];
  };

  // ---- transform role: wtile tw, channel quad tq (= the wave).  Row i of B^T d B for its 4 channels: four positions
  const int tw = tid & 63, tq = tid >> 6;
  const int twy = tw >> 3, twx = tw & 7;
  auto transform = [&](const float* raw, int i, __bf16* V) __attribute__((always_inline)) {
    const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;     // i0 = d0-d2, i1 = d1+d2, i2 = d2-d1, i3 = d1-d3
    const float sg = i == 1 ? 1.f : -1.f;
    f32x4 o[4];                                      // the row combination at columns 0..3, four channels
#pragma unroll
    for (int col = 0; col < 4; ++col) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(raw + roff(tq, 2 * twy + ra, 2 * twx + col));
      const f32x4 c = *reinterpret_cast<const f32x4*>(raw + roff(tq, 2 * twy + rb, 2 * twx + col));
      o[col] = a + sg * c;
    }
    const f32x4 t[4] = {o[0] - o[2], o[1] + o[2], o[2] - o[1], o[1] - o[3]};      // the same F(2,3) combination along the columns
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 pl[3];
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        bf16x2 h, m, l;
        split2(t[j][2 * cp], t[j][2 * cp + 1], h, m, l);
        pl[0][2 * cp] = h[0]; pl[0][2 * cp + 1] = h[1];
        pl[1][2 * cp] = m[0]; pl[1][2 * cp + 1] = m[1];
        pl[2][2 * cp] = l[0]; pl[2][2 * cp + 1] = l[1];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(V + uvoff(q, j, tq >> 1, tw) + (tq & 1) * 4) = pl[q];
    }
  };

  // ---- MFMA role: wave (wg, cg)
  const int wg = wave >> 1, cg = wave & 1;
  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // four positions (row I, j = 0..3) x six term products, smallest first; the positions interleave so consecutive MFMAs use
  // different accumulators
  auto products = [&](const __bf16* U, const __bf16* V, auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    bf16x8 uf[4][3], vf[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        uf[j][q] = *reinterpret_cast<const bf16x8*>(U + uvoff(q, j, hi, 32 * cg + l31));
        vf[j][q] = *reinterpret_cast<const bf16x8*>(V + uvoff(q, j, hi, 32 * wg + l31));
      }
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[j][PA[t]], vf[j][PB[t]], acc[j * 4 + i], 0, 0, 0);
  };
  auto hint = [&]() __attribute__((always_inline)) {      // scheduling region: one MFMA, five VALU, two LDS accesses, ...
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x300, 2, 0);
    }
  };

  __bf16* U0 = UV, *V0 = UV + UVE, *U1 = UV + 2 * UVE, *V1 = UV + 3 * UVE;
  float* raw1 = raw0 + RAWF;
  const int nsteps = nchunk * 4;

  // prologue: patch 0, U of step 0, V of step 0; patch 1 requested
  gload(0);
  uload(0);
  lstore(raw0);
  ustore(U0);
  __syncthreads();
  if (nchunk > 1) gload(1);
  transform(raw0, 0, V0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const float* rc = (c & 1) ? raw1 : raw0;            // this chunk's patch
    float* rn = (c & 1) ? raw0 : raw1;                  // the next chunk's
    const int s0 = 4 * c;
    // ---- step (c, 0): buffers 0 -> produce (c, 1) into buffers 1
    IMX_TS(5)
    uload(min(s0 + 1, nsteps - 1));
    products(U0, V0, std::integral_constant<int, 0>{});
    IMX_TS(0)
    transform(rc, 1, V1);
    IMX_TS(1)
    ustore(U1);
    IMX_TS(2)
    __syncthreads();
    IMX_TS(3)
    // ---- step (c, 1): the next chunk's patch (requested one chunk ago) moves to the other raw buffer
    uload(min(s0 + 2, nsteps - 1));
    products(U1, V1, std::integral_constant<int, 1>{});
    transform(rc, 2, V0);
    ustore(U0);
    if (c + 1 < nchunk) lstore(rn);                     // block-uniform
    hint();
    __syncthreads();
    // ---- step (c, 2)
    if (c + 2 < nchunk) gload(c + 2);                   // block-uniform
    uload(min(s0 + 3, nsteps - 1));
    products(U0, V0, std::integral_constant<int, 2>{});
    transform(rc, 3, V1);
    ustore(U1);
    hint();
    __syncthreads();
    // ---- step (c, 3): produce (c + 1, 0) from the next patch (past the end: a redundant transform of this one)
    uload(min(s0 + 4, nsteps - 1));
    products(U1, V1, std::integral_constant<int, 3>{});
    transform(c + 1 < nchunk ? rn : rc, 0, V0);
    ustore(U0);
    hint();
    __syncthreads();
  }

  IMX_TS(4)
  // ---- epilogue: lane (wtile, hi) holds channels (r & 3) + 8 (r >> 2) + 4 hi of its 32-channel block for all 16 positions
  //      rows:    s0[j] = m0j + m1j + m2j        s1[j] = m1j - m2j - m3j        (acc[j * 4 + i])
  //      columns: y[r][0] = s_r[0] + s_r[1] + s_r[2]     y[r][1] = s_r[1] - s_r[2] - s_r[3]
  const int w = 32 * wg + l31, wy = w >> 3, wx = w & 7;
  const int oy = y0 + 2 * wy, ox = x0 + 2 * wx;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = cob * 64 + 32 * cg + 8 * g + 4 * hi;
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 m[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = (f32x4){acc[j * 4 + i][4 * g], acc[j * 4 + i][4 * g + 1], acc[j * 4 + i][4 * g + 2], acc[j * 4 + i][4 * g + 3]};
      s0[j] = m[0] + m[1] + m[2];
      s1[j] = m[1] - m[2] - m[3];
    }
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + co);
    f32x4 y[2][2] = {{s0[0] + s0[1] + s0[2] + bias, s0[1] - s0[2] - s0[3] + bias}, {s1[0] + s1[1] + s1[2] + bias, s1[1] - s1[2] - s1[3] + bias}};
    if (p.relu) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int e = 0; e < 4; ++e) y[r][x][e] = fmaxf(y[r][x][e], 0.f);
    }
    if (POOL) {
      const int Ho = H >> 1, Wo = W >> 1, py = oy >> 1, px = ox >> 1;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(y[0][0][e], y[0][1][e]), fmaxf(y[1][0][e], y[1][1][e]));
      if (py < Ho && px < Wo) {
        float* dst = p.out_blocked ? p.out + ((((size_t)b * (Cout >> 3) + (co >> 3)) * Ho + py) * Wo + px) * 8 + (co & 7)
                                   : p.out + (((size_t)b * Ho + py) * Wo + px) * Cout + co;
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int gy = oy + r, gx = ox + x;
          if (gy < H && gx < W) {
            float* dst = p.out_blocked ? p.out + ((((size_t)b * (Cout >> 3) + (co >> 3)) * H + gy) * W + gx) * 8 + (co & 7)
                                       : p.out + (((size_t)b * H + gy) * W + gx) * Cout + co;
            *reinterpret_cast<f32x4*>(dst) = y[r][x];
          }
        }
    }
  }
  if constexpr (TRACE) {
    IMX_TS(5)
    if (tid == 0)
      for (int q = 0; q < 6; ++q) atomicAdd(trace + q, tph[q]);
  }
#undef IMX_TS
}
}  // namespace

// channel-blocked input, Cin % 16 == 0, Cout % 64 == 0, not the first layer; even H, W when pooling
bool conv3x3_f22x3_supported(const ConvArgs& a) {
  if (a.first || !a.in_blocked || a.Cin % CK || a.Cout % 64 || a.B <= 0 || a.H < 2 || a.W < 2) return false;
  if (a.pool && ((a.H | a.W) & 1)) return false;
  return true;
}

hipError_t launch_conv3x3_f22x3(const ConvArgs& a, const void* ux, hipStream_t s) {
  if (!conv3x3_f22x3_supported(a) || !ux) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OT - 1) / OT, tiles_y = (a.H + OT - 1) / OT;
  const int nitems = a.B * tiles_x * tiles_y * (a.Cout / 64);
  const int grid = (nitems + 7) / 8 * 8;
  const size_t lds = (size_t)2 * RAWF * 4 + (size_t)4 * UVE * 2;
  static unsigned long long* tr = nullptr;
  const char* te = getenv("IMX_X3_TRACE");
  const bool trace = te && atoi(te) != 0;
  if (trace && !tr) (void)hipMalloc(&tr, 64);
  if (trace) (void)hipMemsetAsync(tr, 0, 64, s);
  static bool attr[4] = {false, false, false, false};
  auto go = [&](auto kern, int id) {
    if (!attr[id]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr[id] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, static_cast<const __bf16*>(ux), tiles_x, tiles_y, nitems, tr);
  };
  if (trace) { if (a.pool) go(conv3x3_f22x3<true, true>, 3); else go(conv3x3_f22x3<false, true>, 2); }
  else { if (a.pool) go(conv3x3_f22x3<true, false>, 1); else go(conv3x3_f22x3<false, false>, 0); }
  if (trace) {
    unsigned long long t[6];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(t, tr, 48, hipMemcpyDeviceToHost);
    const double n0 = (double)nitems * (a.Cin / CK);          // traced steps: row 0 of every chunk
    fprintf(stderr, "[conv3x3_f22x3 %dx%d %d->%d%s] cycles of step (c, 0): U request + operand reads + 24 MFMAs issued %.0f | transform of the next row %.0f | U store (waits for the global loads) %.0f | barrier %.0f;  the other three steps of a chunk together %.0f;  prologue + epilogue per item %.0f\n",
            a.H, a.W, a.Cin, a.Cout, a.pool ? " pool" : "", t[0] / n0, t[1] / n0, t[2] / n0, t[3] / n0, (t[5] - 0.0) / n0, t[4] / (double)nitems);
  }
  return hipGetLastError();
}

}  // namespace imx
