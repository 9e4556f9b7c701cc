// conv1ab_wino24.hip — the fused first layer (conv1a + conv1b + folded BN + ReLU + MaxPool2d(2);
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-114) with conv1b as Winograd F(2x4, 3x3):
//   Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4     per 2-row x 4-column output tile ("wtile") and 4x6 input patch d
// (F(2,3) down the rows, F(4,3) along the columns: 24 multiplies per 8 outputs, 3x fewer than the direct form and
// 1.33x fewer than F(2x2,3x3); fp32 throughout, error measured in the parity tests).  Summed over input channels as 24
// independent GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]  on v_mfma_f32_16x16x4_f32.
//
// Persistent, 2 workgroups per CU.  Workgroup = 256 threads (4 waves) -> 8x16 output pixels = 4x4 wtiles x 64 output channels;
// wave = 16 channels x the 16 wtiles x 24 positions = 24 accumulators of 4 VGPRs.  U is the MFMA's A operand, so the 16x16 D
// layout gives a lane one wtile and FOUR CONSECUTIVE CHANNELS with all 24 positions: output transform, bias, ReLU and the
// 2x2 max-pool (one wtile = 1x2 pooled pixels) are in-lane and every result leaves as one 16-byte store.
// conv1a (1 -> 64, K = 9 + bias tap) runs on the matrix cores once per tile for the 10x18 halo patch, into LDS.
// K loop: 8 input channels per chunk, ONE barrier per chunk:
//   the next chunk's input transform rides inside the MFMA phase, split over the four waves by transformed row: lane =
//     (wtile, channel PAIR), 12 ds_read_b64, every operation one packed (v_pk_*_f32) instruction over the pair (6 + 12 in three
//     dense batches), 6 ds_write_b64; V is double buffered;
//   V layout [12 quads][4 k][18][4]: a quad = two positions x the two channels of a pair.  The chunk's two MFMA k-steps
//     take the even / odd channels of the four pairs, so a lane's V operands of four MFMAs are ONE ds_read_b128
//     (12 LDS reads per chunk for 48 MFMAs), and its U operands (U = G2 g G4^T, transformed at weight load, laid out
//     [chunk][quad][co-block][lane][4]) ONE buffer_load_dwordx4, refilled in place right behind the MFMAs that consumed
//     them: U never touches LDS.
// LDS 79 KB (V 2 x 13.5, conv1a patch 49.5, image patch 1).
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per workgroup (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // conv1a patch (pad-1 halo)
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;  // image patch 12 x 20
constexpr int RS = 66;                         // conv1a patch pixel stride: wtile columns 4 px apart land 8 banks apart
constexpr int RAWSZ = 192 * RS;                // 10x18 = 180 pixels + 12 pad (the conv1a GEMM's twelfth pixel block stores unmasked)
// V: [12 quads][4 k rows of 16 slots, 4 slots of padding between rows 1 and 2][4 floats] -- the layout conv3x3_wino24.hip
// explains: ds_read_b128 lane groups mix k rows 2m / 2m+1, which must be a multiple of 16 slots apart (round 1's stride of
// 18 made every A-operand read a 2-way bank conflict)
constexpr int QSL = 68;                        // slots per quad
__host__ __device__ constexpr int vslot(int k, int n) { return 16 * k + 4 * (k >> 1) + n; }
constexpr int NQ = 12;                         // quads per chunk: 24 positions x 2 k-steps / 4
constexpr int VSZ = NQ * QSL * 4;              // 3456
constexpr int UCH = NQ * 4 * 64 * 4;           // 12288 floats of U per (64 co, 8 ci)
constexpr int CK = 8;

template <bool TRACE>
__global__ __launch_bounds__(256, 2) void conv1ab_wino24(ConvArgs p, int tiles_x, int tiles_y, int ntiles, unsigned* trace) {
  // TRACE: per-phase s_memtime deltas summed over chunks and tiles (bring-up instrumentation, IMX_WINO_TRACE=1)
  unsigned tph[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define IMX_TS(i_)                                                   \
  if constexpr (TRACE) {                                             \
    const unsigned long long now = __builtin_readcyclecounter();     \
    tph[i_] += (unsigned)(now - tprev);                              \
    tprev = now;                                                     \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;
  float* raw = V + 2 * VSZ;             // V is double buffered: chunk c+1 is transformed inside chunk c's MFMA phase
  float* img = raw + RAWSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;
  const int H = p.H, W = p.W, Cout = p.Cout;
  constexpr int nchunk = 64 / CK;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu24, 0, nchunk * UCH * 4, 0x00020000);
  const int voff = (cb * 64 + lane) * 16;

  // ---- per-lane constants of the conv1a GEMM (see the tile loop): A = weights, 12 registers, loaded once per workgroup
  const int n = lane & 15, kq = lane >> 4;
  const float c2 = kq == 0 ? 1.f : 0.f, a2 = kq == 1 ? 1.f : 0.f;      // third k-step: tap 8 | the bias "tap" (input 1) | zero padding
  float wa[4][3];
  int toff[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = 4 * ks + kq;
    toff[ks] = tap < 9 ? (tap / 3) * IMG_W + tap % 3 : 0;
#pragma unroll
    for (int cbk = 0; cbk < 4; ++cbk) {
      const float* src = tap < 9 ? p.w1 + tap * 64 + cbk * 16 + n : p.b1 + cbk * 16 + n;
      const float v = *(tap <= 9 ? src : p.b1);
      wa[cbk][ks] = tap <= 9 ? v : 0.f;
    }
  }
  // ---- per-lane constants of the input transform: lane = (channel pair tk, wtile tw), a quarter-wave (16 lanes) = 4 pairs x
  //      the 4 wtiles of one wtile row; transformed row i = wave:  i0 = d0 - d2, i1 = d1 + d2, i2 = d2 - d1, i3 = d1 - d3
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const float* rbase = raw + ((2 * twr) * RW + 4 * twc) * RS + 2 * tk;
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  const f32x2 sg2 = {sg, sg};
  const f32x2 m5 = {-5.f, -5.f};
  const float* rpa = rbase + ra * RW * RS;
  const float* rpb = rbase + rb * RW * RS;
  float* vwr = V + vslot(tk, tw) * 4 + (wave >> 1) * QSL * 4 + (wave & 1) * 2;
  int aoff = vslot(lane >> 4, lane & 15) * 4;
  asm volatile("" : "+v"(aoff));               // opaque: the 12 quad reads are immediate offsets from one base
  const float* vrd = V + aoff;

  // ---- persistent over a contiguous range of tiles (coordinates advance incrementally: no divisions in the loop); the
  //      NEXT tile's image patch element (12x20 patch: one float per thread) is fetched a whole tile ahead, so a tile's
  //      prologue never waits on HBM
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
  int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, b = t_begin / (tiles_x * tiles_y);
  const int ipy = tid / IMG_W - 2, ipx = tid % IMG_W - 2;
  auto fetch_px = [&](int ftx, int fty, int fb, bool live) -> float {
    const int gy = fty * OH + ipy, gx = ftx * OW + ipx;
    const bool ok = live && tid < IMG_H * IMG_W && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* im = (fb < p.split) ? p.in + (size_t)fb * H * W : p.in2 + (size_t)(fb - p.split) * H * W;
    const float* src = ok ? im + (size_t)gy * W + gx : p.in;   // unconditional load from a clamped address (past the last tile
    const float v = *src;                                      // `fb` is out of range): keeps the vmcnt bookkeeping exact
    return ok ? v : 0.f;
  };
  float pre = fetch_px(tx, ty, b, t_begin < t_end);
  // conv1a gather geometry of this lane's three pixel blocks (tile independent)
  int gpy[3], gpx[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pp = (3 * wave + j) * 16 + n;          // pixel of the 10x18 patch (180 real + 12 pad)
    const int pc = pp < RH * RW ? pp : RH * RW - 1;  // pad columns gather a valid address (their B column is zeroed)
    gpy[j] = pc / RW;
    gpx[j] = pc % RW;
  }
  int ntile_done = 0;
  // ConvArgs::amax_out: the largest output of every image (for the next layer's fp16-plane scale, conv3x3_wino24h.hip).  A workgroup walks
  // a contiguous tile range, i.e. one or two images: the running maximum is flushed (wave reduction + one atomicMax) when the image changes
  unsigned am_run = 0;
  int am_b = -1;
  auto am_flush = [&]() {
    unsigned mb = am_run;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if (lane == 0 && mb && am_b >= 0) atomicMax(p.amax_out + (am_b & 255), mb);      // image slot b % 256 (conv3x3_wino24h.hip: AMAX_SLOTS)
    am_run = 0;
  };
  const f32x4 bs4 = *reinterpret_cast<const f32x4*>(p.bias + cb * 16 + 4 * (lane >> 4));      // conv1b bias of this lane's four output channels
  f32x4 bf[NQ];                  // B operands of the coming chunk (chunk 0 here; refilled in place from then on)
#pragma unroll
  for (int g = 0; g < NQ; ++g) bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, g * 4096, 0));
  // drain every load once here: the compiler's waitcnt bookkeeping at the chunk-loop header then sees only the loop's own
  // in-order refills (vmcnt(11) per quad) instead of merging in this preamble's arbitrary order (it chose vmcnt(1))
  __builtin_amdgcn_s_waitcnt(0x0F70);

  for (int t = t_begin; t < t_end; ++t) {
    if constexpr (TRACE) tprev = __builtin_readcyclecounter();
    const int x0 = tx * OW, y0 = ty * OH, bcur = b;
    if (tid < IMG_H * IMG_W) img[tid] = pre;
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++b; } }      // -> next tile
    pre = fetch_px(tx, ty, b, t + 1 < t_end);
    __syncthreads();             // image patch visible
    IMX_TS(6)

    // ---- conv1a + folded BN + ReLU for all 64 channels of the 10x18 halo patch ON THE MATRIX CORES (the packed-FMA form
    //      cost ~470 instructions per wave, each ~20 cycles beside the co-resident workgroup's MFMA stream):
    //      D[channel][pixel] = W[channel][tap] * im2col[tap][pixel], K = 9 taps + the bias as a tenth "tap" whose input is
    //      1, padded to three k-steps of 4.  B = image gathers from LDS; a pixel outside the image gets an all-zero B
    //      column (taps AND bias), so its output is relu(0) = 0: conv1b's zero padding without a mask multiply.  A lane's
    //      four D registers are four consecutive channels of one pixel -> two ds_write_b64.
    //      wave w takes pixel blocks 3w..3w+2 (of twelve 16-pixel blocks) x all four 16-channel blocks: 36 MFMAs.
    {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pp = (3 * wave + j) * 16 + n;
        const int py = gpy[j], px = gpx[j];
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const float m = (pp < RH * RW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1.f : 0.f;   // branch-free: finite * 0
        const float* ip = img + py * IMG_W + px;
        float bv[3];
        bv[0] = ip[toff[0]] * m;
        bv[1] = ip[toff[1]] * m;
        bv[2] = (ip[toff[2]] * c2 + a2) * m;
#pragma unroll
        for (int cbk = 0; cbk < 4; ++cbk) {
          f32x4 d = zero4;
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cbk][ks], bv[ks], d, 0, 0, 0);
          // ReLU on the bit pattern (a negative float is a negative int): one v_max_i32 per value; the float maximum of an MFMA
          // result costs a canonicalising v_max x,x first (IEEE mode) -- 48 instructions per tile and wave
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          d = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, d), (i32x4){0, 0, 0, 0}));
          float* o = raw + pp * RS + cbk * 16 + 4 * kq;
          *reinterpret_cast<f32x2*>(o) = (f32x2){d[0], d[1]};
          *reinterpret_cast<f32x2*>(o + 2) = (f32x2){d[2], d[3]};
        }
      }
    }

    f32x4 acc[24];
    IMX_TS(4)

    // ---- transform of chunk 0 (the later ones ride inside the MFMA phases)
    __syncthreads();               // conv1a patch complete
    IMX_TS(0)
    {
      f32x2 va[6], vb[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) { va[bb] = *reinterpret_cast<const f32x2*>(rpa + bb * RS); vb[bb] = *reinterpret_cast<const f32x2*>(rpb + bb * RS); }
      f32x2 o[6];
#pragma unroll
      for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, vb[bb], va[bb]);
      const W24Half hb = w24_batch_a(o, m5);
      f32x2 T[6];
      w24_batch_b(o, hb, T);
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) *reinterpret_cast<f32x2*>(vwr + (2 * jj) * QSL * 4) = T[jj];
    }
    IMX_TS(1)

    const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};
    auto phase = [&](int ch, auto first_tag) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value;
      __syncthreads();               // V[ch & 1] complete; the other buffer's readers (chunk ch-1) are done
      IMX_TS(2)
      // ---- 24 positions x 2 k-steps of v_mfma_f32_16x16x4_f32 in 12 quads (A operands: one ds_read_b128 per quad, one
      //      quad ahead; B operands: registers, each quad's four refilled IN PLACE with the next chunk's right behind the
      //      MFMAs that consumed them -- chunk 7 fetches chunk 0 for the next tile: U is tile independent).
      //      The NEXT chunk's input transform rides along in the same instruction stream:  V = B2^T d B4 split over the
      //      four waves by transformed ROW i = wave (every wave handles all 64 (wtile, channel pair) items but only the two
      //      patch rows its row of B2^T touches: 12 ds_read_b64, 6 + 12 packed operations, 6 ds_write_b64), issued as the
      //      LDS reads at quad 0, three dense batches of six packed operations at quads 3 / 5 / 7 and the writes at quad 9
      //      (a VALU batch costs ~11 + 4.5 n cycles of matrix-pipe time, one instruction alone ~15, and a separate phase
      //      beside the other workgroup's MFMAs ~29 per instruction: tools/ubench/mfma_valu.hip and the cycle trace).
      //      After chunk 7 the transform reads channels 64.. of the padded patch and writes the idle V buffer: harmless.
      {
        const float* vr = vrd + (ch & 1) * VSZ;
        float* vw = vwr + ((ch + 1) & 1) * VSZ;
        const float* pa = rpa + (ch + 1) * CK;
        const float* pb = rpb + (ch + 1) * CK;
        const int uoff = __builtin_amdgcn_readfirstlane(((ch + 1) & (nchunk - 1)) * (UCH * 4));
        f32x2 va[6], vb[6], o[6], T[6];
        W24Half hb;
        f32x4 af[2];
        af[0] = *reinterpret_cast<const f32x4*>(vr);
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
          const int cur = g & 1, nxt = cur ^ 1;
          if (g + 1 < NQ) af[nxt] = *reinterpret_cast<const f32x4*>(vr + (g + 1) * QSL * 4);
          // the tile's first chunk starts from literal-zero accumulators (no 96-register clear in the prologue)
          acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][0], af[cur][0], FIRST ? zero4c : acc[2 * g], 0, 0, 0);
          acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][2], af[cur][2], FIRST ? zero4c : acc[2 * g + 1], 0, 0, 0);
          acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][1], af[cur][1], acc[2 * g], 0, 0, 0);
          acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][3], af[cur][3], acc[2 * g + 1], 0, 0, 0);
          bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, uoff + g * 4096, 0));
          if (g == 0) {
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) { va[bb] = *reinterpret_cast<const f32x2*>(pa + bb * RS); vb[bb] = *reinterpret_cast<const f32x2*>(pb + bb * RS); }
          }
          if (g == 3) {
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, vb[bb], va[bb]);   // down the rows: F(2,3), row i
          }
          if (g == 5) {                                                                          // along the columns: F(4,3)
            hb = w24_batch_a(o, m5);
          }
          if (g == 7) {
            w24_batch_b(o, hb, T);
          }
          if (g == 9) {
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) *reinterpret_cast<f32x2*>(vw + (2 * jj) * QSL * 4) = T[jj];   // position p = j*4 + i
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      IMX_TS(3)
    };
    phase(0, std::true_type{});
#pragma unroll 1
    for (int ch = 1; ch < nchunk; ++ch) phase(ch, std::false_type{});

    // ---- output transform Y = A2^T M A4, 2x2 max-pool, bias, ReLU (max-pool commutes with both), stores straight from
    //      registers.  The conv1b MFMAs take U as their A operand and V as B, so D is [channel][wtile]: acc[j*4 + i][r]
    //      belongs to wtile n = lane&15 (row n>>2, column n&3) and channel cb*16 + 4*(lane>>4) + r -- a lane's four registers
    //      are four consecutive channels of one pixel = one 16-byte store (no LDS staging tile, no barrier); buffer stores
    //      through a per-image descriptor drop the pixels outside the image.
    {
      f32x4 s0[6], s1[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        s0[j] = acc[j * 4 + 0] + acc[j * 4 + 1] + acc[j * 4 + 2];
        s1[j] = acc[j * 4 + 1] - acc[j * 4 + 2] - acc[j * 4 + 3];
      }
      f32x4 y[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x4* m = r ? s1 : s0;
        const f32x4 a12 = m[1] + m[2], b12 = m[1] - m[2], c34 = m[3] + m[4], d34 = m[3] - m[4];
        y[r][0] = m[0] + a12 + c34;
        y[r][1] = b12 + 2.f * d34;
        y[r][2] = a12 + 4.f * c34;
        y[r][3] = b12 + 8.f * d34 + m[5];
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int wr = (lane & 15) >> 2, wc = lane & 3;
      const int Ho = H >> 1, Wo = W >> 1;
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)bcur * Ho * Wo * Cout), 0, Ho * Wo * Cout * 4, 0x00020000);
      // NHWC, or channel-blocked (B, Cout/8, Ho, Wo, 8) for the next Winograd layer (conv3x3_wino24.hip: dense patch loads)
      const bool outb = p.out_blocked != 0;
      const int opx = outb ? 32 : Cout * 4;
      const int choff = outb ? (cb * 2 + (lane >> 5)) * (Ho * Wo * 32) + ((lane >> 4) & 1) * 16 : (cb * 16 + 4 * (lane >> 4)) * 4;
      typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      const int oy = (y0 >> 1) + wr;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 v = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]),
                                                                            __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1])) + bs4, zero4);
        const int ox = (x0 >> 1) + 2 * wc + hh;
        const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * opx + choff) : 0x7ffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)off, 0, 0);
        if (p.amax_out) {
          if (bcur != am_b) { am_flush(); am_b = bcur; }           // uniform
          am_run = max(am_run, __builtin_bit_cast(unsigned, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))));      // after ReLU: >= 0
        }
      }
    }
    IMX_TS(5)
    ++ntile_done;
  }
#undef IMX_TS
  if (p.amax_out) am_flush();
  if constexpr (TRACE) {
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned* o = trace + (blockIdx.x * 4 + wave) * 8;
      for (int i = 0; i < 6; ++i) o[i] = tph[i];
      o[6] = (unsigned)ntile_done;
      o[7] = tph[6];
    }
  }
}
}  // namespace

hipError_t launch_conv1ab_wino24(const ConvArgs& a, hipStream_t s) {
  if (!a.first || !a.pool || a.Cin != 64 || a.Cout != 64 || !a.wu24) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH, ntiles = tiles_x * tiles_y * a.B;
  const size_t lds = (size_t)(2 * VSZ + RAWSZ + IMG_H * IMG_W) * sizeof(float);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  const dim3 grid((unsigned)(ntiles < 2 * ncu ? ntiles : 2 * ncu));     // persistent: two workgroups per CU
  last_form = "conv1ab_wino24:f32";
  static const bool trace = getenv("IMX_WINO_TRACE") != nullptr;     // developer instrumentation, read once per process
  if (trace) {      // bring-up instrumentation: per-phase cycle counts per wave, averaged over tiles
    static unsigned* dbuf = nullptr;
    constexpr int NREC = 1024 * 4 * 8;
    if (!dbuf) (void)hipMalloc(&dbuf, NREC * sizeof(unsigned));
    (void)hipMemsetAsync(dbuf, 0, NREC * sizeof(unsigned), s);
    hipLaunchKernelGGL(conv1ab_wino24<true>, grid, dim3(256), lds, s, a, tiles_x, tiles_y, ntiles, dbuf);
    (void)hipStreamSynchronize(s);
    static unsigned host[NREC];
    (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
    const int n = grid.x < 1024 ? (int)grid.x : 1024;
    for (int w = 0; w < 4; ++w) {
      double sum[6] = {0}, nt = 0, b0 = 0;
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 6; ++j) sum[j] += host[(i * 4 + w) * 8 + j];
        nt += host[(i * 4 + w) * 8 + 6];
        b0 += host[(i * 4 + w) * 8 + 7];
      }
      fprintf(stderr, "[wino24 trace] wave %d | image patch + first barrier %.0f of the prologue\n", w, b0 / nt);
      fprintf(stderr, "[wino24 trace] wave %d | per tile: prologue %.0f  chunks 8 x (barrier1 %.0f  transform %.0f  barrier2 %.0f  mfma %.0f)  "
                      "epilogue %.0f cycles  (%.0f tiles per workgroup)\n", w, sum[4] / nt, sum[0] / nt / 8, sum[1] / nt / 8, sum[2] / nt / 8,
              sum[3] / nt / 8, sum[5] / nt, nt / n);
    }
    return hipGetLastError();
  }
  hipLaunchKernelGGL(conv1ab_wino24<false>, grid, dim3(256), lds, s, a, tiles_x, tiles_y, ntiles, (unsigned*)nullptr);
  return hipGetLastError();
}

}  // namespace imx
