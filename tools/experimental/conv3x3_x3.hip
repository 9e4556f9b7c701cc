// conv3x3_x3.hip — 3x3 / pad 1 convolution + folded BatchNorm + ReLU (+ MaxPool2d(2)) of superpoint/models/unet_parts.py:10-48 and
// superpoint_test.py:113-123 as Winograd F(2x4, 3x3) whose 24 per-position GEMMs run on the bf16 matrix pipe with every fp32
// product carried as six bf16 term products (gemm_x3.hip explains the split; tools/ubench/mfma_bf16x3.hip measures it).
//
//   Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4   per 2-row x 4-column output tile ("wtile") and 4x6 input patch d
//   M_p[co][wtile] = sum_ci U_p[ci][co] V_p[wtile][ci],  p = 24 positions, on v_mfma_f32_32x32x16_bf16: 16 input channels per step,
//   six MFMAs of 32 cycles per (32 co x 32 wtiles x 16 ci) block where the fp32 MFMA needs sixteen 16x16x4 of 32 cycles.
//
// The transforms stay in fp32 on the VALU (same formulas as conv3x3_wino24.hip); only the products change pipes.
// One workgroup (4 waves, ONE per CU: 512 registers per lane) owns 8 x 8 wtiles (16 x 32 output pixels) x 64 output channels:
// wave (wg, cg) = wtiles 32 wg .. + 31 x channels 32 cg .. + 31, all 24 positions: 24 accumulators of 16 registers (AGPRs).
// Per 16-channel chunk:  the raw 18 x 34 x 16 patch sits in LDS (fp32); two passes, one per pair of transformed rows i:
//   VALU  thread (wtile, channel octet, i) computes its six positions V[i][0..5] for eight channels from two raw rows, splits
//         them (v_cvt_pk_bf16_f32) and writes three bf16 planes to LDS -- each V value is produced ONCE per workgroup;
//   MFMA  12 positions x 6 term products per wave; the B operand (V: lane (wtile, kb), eight channels) is one ds_read_b128 per
//         plane, the A operand (U: lane (co, kb)) one coalesced 1 KB load per plane straight from the host-made fragment order,
//         requested a position ahead.
// With U as the A operand the accumulator layout gives a lane one wtile and four consecutive channels per register group, all 24
// positions: output transform, bias, ReLU and the 2x2 max-pool are in-lane and every result leaves as one 16-byte store.
// LDS: raw 54.3 KB + V 72 KB.  The next chunk's patch is requested before the current chunk's passes and stored after them.
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdio>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TWY = 8, TWX = 8;                  // wtiles per workgroup tile
constexpr int OHT = 2 * TWY, OWT = 4 * TWX;      // 16 x 32 output pixels
constexpr int RH = OHT + 2, RW = OWT + 2;        // 18 x 34 input patch (pad-1 halo)
constexpr int CK = 16;                           // input channels per chunk = one MFMA k step
// raw patch in LDS: [quad of 4 channels q][column mod 4][row][12 slots: column / 4] x 16 bytes, planes 217 slots apart.  A lane of
// the transform is a wtile (twy, twx): its reads land on slot 8 twy + twx + const (mod 16) -- the sixteen lanes of a ds_read_b128
// group, (twy, twx) in {(0,0-3), (1,4-7), (2,4-7), (3,0-3)} or the complement, cover all sixteen slots; the odd plane stride
// spreads the stores (eight consecutive lanes = four columns x two quads).
constexpr int RSL = 12, RPST = RH * RSL + 1;     // slots per row, slots per plane
constexpr int RAWF = 16 * RPST * 4;              // floats (55.5 KB)
__device__ __forceinline__ int roff(int q, int row, int col) { return (((q * 4 + (col & 3)) * RPST) + row * RSL + (col >> 2)) * 4; }
constexpr int VSET = 6 * 2 * 64 * 8;              // bf16 elements per V plane and set: [6 positions][2 octets][64 wtiles][8 ch]
constexpr int NLOAD = (RH * RW * 4 + 255) / 256; // float4 pieces of the patch per thread (2448 -> 10)

template <bool V>
struct BoolC { static constexpr bool value = V; };

__device__ __forceinline__ void split2(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) {
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)(r0 - (float)m[0]); l[1] = (__bf16)(r1 - (float)m[1]);
}

// TRACE (IMX_X3_TRACE=1): s_memtime deltas per phase, summed by wave 0 of every workgroup (bring-up instrumentation)
template <bool POOL, bool TRACE>
__global__ __launch_bounds__(256, 1) void conv3x3_x3(ConvArgs p, const __bf16* __restrict__ ux, int tiles_x, int tiles_y, int nitems, unsigned long long* trace) {
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define IMX_TS(i_)                                                                   \
  if constexpr (TRACE) {                                                             \
    const unsigned long long now = __builtin_amdgcn_s_memtime();                     \
    tph[i_] += now - tprev;                                                          \
    tprev = now;                                                                     \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                                             // [18][34][16] fp32
  __bf16* Vs = reinterpret_cast<__bf16*>(smem + RAWF);           // [3 planes][12][64][16] bf16

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK, ncob = Cout / 64;
  // XCD-aware item order: consecutive items (the output blocks of one tile, neighbouring tiles) land on one XCD
  const int grid = (int)gridDim.x;
  const int item = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (item >= nitems) return;
  const int cob = item % ncob, tile = item / ncob;
  const int x0 = (tile % tiles_x) * OWT, y0 = ((tile / tiles_x) % tiles_y) * OHT, b = tile / (tiles_x * tiles_y);

  // ---- patch loader: piece e = (octet, row, px, half) in memory order of the channel-blocked input (B, Cin/8, H, W, 8)
  f32x4 lreg[NLOAD];
  const float* inb = p.in + (size_t)b * Cin * H * W;
  auto gload = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = tid + 256 * k;
      const int q = e % (RW * 2), row = (e / (RW * 2)) % RH, oct = e / (RW * 2 * RH);
      const int gy = y0 - 1 + row, gx = x0 - 1 + (q >> 1);
      const bool ok = e < RH * RW * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float* src = inb + (((size_t)(2 * c + oct) * H + gy) * W + gx) * 8 + 4 * (q & 1);
      lreg[k] = ok ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = tid + 256 * k;
      const int q = e % (RW * 2), row = (e / (RW * 2)) % RH, oct = e / (RW * 2 * RH);
      if (e < RH * RW * 4) *reinterpret_cast<f32x4*>(raw + roff(2 * oct + (q & 1), row, q >> 1)) = lreg[k];
    }
  };

  // ---- transform role of this thread: wtile tw, channel quad tq (= the wave); every thread transforms ONE row i of its wtile
  const int tw = tid & 63, tq = tid >> 6;
  const int twy = tw >> 3, twx = tw & 7;
  const f32x2 m5 = {-5.f, -5.f};
  // ---- MFMA role: wave (wg, cg)
  const int wg = wave >> 1, cg = wave & 1;
  const __bf16* ub = ux + ((size_t)cob * nchunk * 24 * 2 + cg) * 3 * 512 + lane * 8;    // + ((chunk * 24 + pos) * 2) * 3 * 512 + plane * 512

  // V of one "set" = the six positions of one transformed row i: planes [3][6 positions][2 octets][64 wtiles][8 channels] bf16,
  // double buffered (the MFMA's B operand: lane (wtile, octet) reads 16 contiguous bytes, consecutive lanes consecutive slots)
  auto vbuf = [&](int buf, int pl) { return Vs + (buf * 3 + pl) * VSET; };
  // transform row i of chunk patch -> V set buffer `buf`: 12 ds_read_b128, ~50 packed ops, 24 splits, 18 ds_write_b64
  auto transform = [&](int i, int buf) __attribute__((always_inline)) {
    const int ra = i == 0 ? 0 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;     // i0 = d0-d2, i1 = d1+d2, i2 = d2-d1, i3 = d1-d3
    const float sg = i == 1 ? 1.f : -1.f;
    f32x2 o[2][6];                                 // [channel pair][column]
#pragma unroll
    for (int col = 0; col < 6; ++col) {
      const int c4 = 4 * twx + col;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(raw + roff(tq, 2 * twy + ra, c4));
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(raw + roff(tq, 2 * twy + rb, c4));
      o[0][col] = (f32x2){a0[0], a0[1]} + sg * (f32x2){b0[0], b0[1]};
      o[1][col] = (f32x2){a0[2], a0[3]} + sg * (f32x2){b0[2], b0[3]};
    }
    f32x2 t[2][6];
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      const W24Half hh = w24_batch_a(o[cp], m5);
      w24_batch_b(o[cp], hh, t[cp]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      bf16x4 pl[3];
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        bf16x2 h, m, l;
        split2(t[cp][j][0], t[cp][j][1], h, m, l);
        pl[0][2 * cp] = h[0]; pl[0][2 * cp + 1] = h[1];
        pl[1][2 * cp] = m[0]; pl[1][2 * cp + 1] = m[1];
        pl[2][2 * cp] = l[0]; pl[2][2 * cp + 1] = l[1];
      }
      const int off = ((j * 2 + (tq >> 1)) * 64 + tw) * 8 + (tq & 1) * 4;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(vbuf(buf, q) + off) = pl[q];
    }
  };

  // Two passes over the input channels, one per pair of transformed rows (i = 0,1 then 2,3): 12 accumulators of 16 registers
  // live at a time (all 24 would need 384 of the 512 registers).  The output transform is linear, so pass 0 leaves its share of
  // the 2 x 4 outputs (A2^T M A4 over its rows) in registers and pass 1 adds its own before bias / ReLU / pool.
  // Inside a pass the stream of sets (chunk c, row ii) is software pipelined in the wave: the 36 MFMAs of a set issue together
  // with the VALU transform of the NEXT set (different V buffer), one barrier per set, plus one per chunk around the patch store.
  const int w = 32 * wg + l31, wy = w >> 3, wx = w & 7;
  const int oy = y0 + 2 * wy, ox = x0 + 4 * wx;
  const f32x2 k8 = {8.f, 8.f};
  w24_f32x4 ypart[4][2][4];            // [channel group g][row][column]: pass 0's share
  bf16x8 uf[6][3];                     // U fragments of the six positions of a set; refilled in place for the next set
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    auto uload = [&](bf16x8 (&dst)[3], int c, int ii, int j) __attribute__((always_inline)) {
      const __bf16* up = ub + ((size_t)c * 24 + j * 4 + 2 * ph + ii) * (2 * 3 * 512);       // position j * 4 + i
#pragma unroll
      for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(up + q * 512);
    };
    // six positions of set (row ii) from V buffer `buf`; uf[j] is refilled with the next set's position j behind its MFMAs
    auto products = [&](f32x16 (&acc)[12], int ii, int buf, int nc, int nii) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        bf16x8 vf[3];
        const int off = ((j * 2 + hi) * 64 + 32 * wg + l31) * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) vf[q] = *reinterpret_cast<const bf16x8*>(vbuf(buf, q) + off);
        const bf16x8 (&u)[3] = uf[j];
        f32x16 a = acc[ii * 6 + j];
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], vf[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], vf[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[2], vf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], vf[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[1], vf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[0], vf[0], a, 0, 0, 0);
        acc[ii * 6 + j] = a;
        uload(uf[j], nc, nii, j);
      }
    };
    auto interleave = [&]() __attribute__((always_inline)) {       // scheduling hint for the region: MFMA, 5 VALU, 1 LDS access ...
#pragma unroll
      for (int g = 0; g < 36; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);
      }
    };
    f32x16 acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if (ph) __syncthreads();            // pass 0's last readers of raw / V are done
    IMX_TS(5)
    gload(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) uload(uf[j], 0, 0, j);
    lstore();
    __syncthreads();
    if (nchunk > 1) gload(1);
    transform(2 * ph, 0);
    __syncthreads();
    IMX_TS(0)
    for (int c = 0; c < nchunk; ++c) {
      const int cn = min(c + 1, nchunk - 1);
      // ---- set (c, row 0) from buffer 0; meanwhile row 1 of the same chunk -> buffer 1
      products(acc, 0, 0, c, 1);
      transform(2 * ph + 1, 1);
      // interleave();
      IMX_TS(1)
      __syncthreads();
      IMX_TS(2)
      // ---- the next chunk's patch replaces this one's (its last reader was the transform above)
      if (c + 1 < nchunk) {               // block-uniform
        lstore();
        __syncthreads();
        if (c + 2 < nchunk) gload(c + 2);
      }
      IMX_TS(3)
      // ---- set (c, row 1) from buffer 1; meanwhile row 0 of the next chunk -> buffer 0 (past the end: a redundant transform)
      products(acc, 1, 1, cn, 0);
      transform(2 * ph, 0);
      // interleave();
      IMX_TS(4)
      __syncthreads();
    }
    // ---- this pass's share of the output transform.  Lane (wtile, hi) holds channels (r & 3) + 8 (r >> 2) + 4 hi of its
    //      32-channel block; acc[ii * 6 + j] = position (i = 2 ph + ii, j).  Rows: s0 = m0 + m1 + m2, s1 = m1 - m2 - m3.
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      w24_f32x4 m[24];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const w24_f32x4 ma = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};                    // ii = 0
        const w24_f32x4 mb = {acc[6 + j][4 * g], acc[6 + j][4 * g + 1], acc[6 + j][4 * g + 2], acc[6 + j][4 * g + 3]};    // ii = 1
        const w24_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        m[j * 4 + 0] = ph ? z : ma; m[j * 4 + 1] = ph ? z : mb; m[j * 4 + 2] = ph ? ma : z; m[j * 4 + 3] = ph ? mb : z;
      }
      w24_f32x4 y[2][4];
      w24_output_transform(m, k8, y);
      if (ph == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int x = 0; x < 4; ++x) ypart[g][r][x] = y[r][x];
        continue;
      }
      const int co = cob * 64 + 32 * cg + 8 * g + 4 * hi;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          y[r][x] += ypart[g][r][x] + bias;
          if (p.relu) { y[r][x][0] = fmaxf(y[r][x][0], 0.f); y[r][x][1] = fmaxf(y[r][x][1], 0.f); y[r][x][2] = fmaxf(y[r][x][2], 0.f); y[r][x][3] = fmaxf(y[r][x][3], 0.f); }
        }
      if (POOL) {
        const int Ho = H >> 1, Wo = W >> 1, py = oy >> 1;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(y[0][2 * x][e], y[0][2 * x + 1][e]), fmaxf(y[1][2 * x][e], y[1][2 * x + 1][e]));
          const int px = (ox >> 1) + x;
          if (py < Ho && px < Wo) {
            float* dst = p.out_blocked ? p.out + ((((size_t)b * (Cout >> 3) + (co >> 3)) * Ho + py) * Wo + px) * 8 + (co & 7)
                                       : p.out + (((size_t)b * Ho + py) * Wo + px) * Cout + co;
            *reinterpret_cast<f32x4*>(dst) = v;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int gy = oy + r, gx = ox + x;
            if (gy < H && gx < W) {
              float* dst = p.out_blocked ? p.out + ((((size_t)b * (Cout >> 3) + (co >> 3)) * H + gy) * W + gx) * 8 + (co & 7)
                                         : p.out + (((size_t)b * H + gy) * W + gx) * Cout + co;
              *reinterpret_cast<f32x4*>(dst) = y[r][x];
            }
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
bool conv3x3_x3_supported(const ConvArgs& a) {
  if (a.first || !a.in_blocked || a.Cin % CK || a.Cout % 64 || a.B <= 0 || a.H < 2 || a.W < 4) return false;
  if (a.pool && ((a.H | a.W) & 1)) return false;
  return true;
}

hipError_t launch_conv3x3_x3(const ConvArgs& a, const void* ux3, hipStream_t s) {
  if (!conv3x3_x3_supported(a) || !ux3) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OWT - 1) / OWT, tiles_y = (a.H + OHT - 1) / OHT;
  const int nitems = a.B * tiles_x * tiles_y * (a.Cout / 64);
  const int grid = (nitems + 7) / 8 * 8;
  const size_t lds = (size_t)RAWF * 4 + (size_t)2 * 3 * VSET * 2;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, static_cast<const __bf16*>(ux3), tiles_x, tiles_y, nitems, tr);
  };
  if (trace) { if (a.pool) go(conv3x3_x3<true, true>, 3); else go(conv3x3_x3<false, true>, 2); }
  else { if (a.pool) go(conv3x3_x3<true, false>, 1); else go(conv3x3_x3<false, false>, 0); }
  if (trace) {
    unsigned long long t[6];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(t, tr, 48, hipMemcpyDeviceToHost);
    const double nchp = (double)nitems * 2 * (a.Cin / CK);
    fprintf(stderr, "[conv3x3_x3 %dx%d %d->%d%s] cycles per chunk-pass: set A (products + next transform) %.0f | barrier %.0f | patch store %.0f | set B %.0f; per pass prologue %.0f; per item epilogue etc %.0f\n",
            a.H, a.W, a.Cin, a.Cout, a.pool ? " pool" : "", t[1] / nchp, t[2] / nchp, t[3] / nchp, t[4] / nchp, t[0] / (nitems * 2.0), t[5] / (double)nitems);
  }
  return hipGetLastError();
}

}  // namespace imx
