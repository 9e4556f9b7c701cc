// conv3x3_wino.hip — 3x3 / pad 1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores.
//
// Same contract as conv3x3.hip (replaces F.conv2d + folded BatchNorm + ReLU (+ MaxPool2d(2)) of
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123), 2.25x fewer multiplies:
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile ("wtile") and 4x4 input patch d,
// summed over input channels as 16 independent GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]
// (p = 4x4 transform position).  U = G g G^T is computed once at weight load (imx_api.cpp).
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32); the transforms only use +,- and (in U) 1/2,
// so the result differs from the direct form by fp32 rounding only (measured in the parity tests).
//
// Workgroup = 256 threads (4 waves) -> 4x8 wtiles (8x16 output pixels) x 64 output channels.
//   wave: one 16-channel column block x both 16-wtile row blocks x all 16 positions -> 32 accumulators
//   of 4 VGPRs (B fragments shared by the two row blocks: half the L1 traffic of a 16x32 wave tile).  The 16x16 MFMA's D layout gives each lane one output
//   channel and 4 wtiles with all 16 positions: the output transform, bias, ReLU and the 2x2
//   max-pool (one wtile = one pooled pixel) are in-lane, no LDS exchange.
// K loop, 8 input channels per chunk, two barriers per chunk:
//   raw 10x18x8 input patch (fetched + written to LDS inside the previous chunk's MFMA loop)
//   -> input transform (one (channel, wtile) per thread, 32 adds) -> V [16 pos][4 ch pairs][72]
//   -> 64 MFMAs per wave in 8 groups; A operands (V) from LDS one group ahead, B operands (U) read
//   straight from global/L2 into 64 registers while the transform runs (layout
//   [pos][k-step][co-block][4 k][16 co]: one wave load = 256 contiguous bytes; U never touches LDS).
//   LDS 27 KB, <=224 VGPRs -> 2 workgroups per CU.
// FIRST mode fuses conv1a (1->64, K=9, weights broadcast from LDS) into the raw-patch staging.
// Measured (round 1, PMC + in-kernel cycle trace, IMX_WINO_TRACE=1): MFMA phase 2.5-2.7k cycles of a
// 5.6k-cycle chunk period with two co-resident groups; the limiter is operand delivery (a 16x16x4 fp32
// MFMA consumes 512 B of operands per 32 cycles) - see DESIGN.md for the planned 32x32x2 / 256-accumulator
// variant.
#include "imx_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TR = 4, TC = 8, OH = 2 * TR, OW = 2 * TC;   // wtiles / output pixels per workgroup
constexpr int RH = OH + 2, RW = OW + 2;                   // raw patch (pad 1 halo)
constexpr int RPITCH = RW, RS = 12;                       // FIRST: raw conv1a patch [10][18][12]: (4 wtiles x 8 ch) reads conflict-free
constexpr int CK = 8, NT = 64;
constexpr int RAW = RH * RPITCH * RS;                     // 2160 (FIRST only)
constexpr int VK = 72;                                    // V channel-pair stride (64 + 8 pad: conflict-free transform writes)
constexpr int QS = 320;                                   // V position stride = 5 x 64 dwords: reads of different positions from one base
                                                          // fuse into ds_read2st64_b32 (no per-read address VALU beside the MFMAs)
constexpr int VSZ = 16 * QS;                              // 5120
constexpr int USZ = 16 * 2 * 4 * 4 * 16;                  // 8192 = [16 pos][2 k-steps][4 co-blocks][4 k][16 co] per (64 co, 8 ci)
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;             // FIRST: image patch 12 x 20
constexpr int RSF = 68;                                   // FIRST: conv1a patch [10*18 px][64 ch], pixel stride 68 (conflict-free)
constexpr int RAWF = RH * RW * RSF;                       // 12240

template <bool POOL, bool RELU, bool FIRST, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wino(ConvArgs p, int tiles_x, int tiles_y, unsigned* trace = nullptr) {
  unsigned long long t_start = 0, t_loop = 0, t_epi = 0;
  if constexpr (TRACE) t_start = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;
  float* raw = V + VSZ;                 // raw input patch [10][18][12]
  float* img = raw + (FIRST ? RAWF : RAW);   // FIRST only: image patch 12 x 20

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;          // wave = one 16-channel column block x BOTH 16-wtile row blocks (halves the B traffic)
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * OW, y0 = ty * OH;
  const int n0 = blockIdx.y * NT;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu6, 0, (Cout / NT) * nchunk * USZ * 4, 0x00020000);
  const int voff = (cb * 64 + lane) * 16;

  if constexpr (FIRST) {
    const float* im = (b < p.split) ? p.in + (size_t)b * H * W : p.in2 + (size_t)(b - p.split) * H * W;
    // conv1a weights of this lane's channel PAIR, issued together with the image patch loads: one latency, not two
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int c1_cp = tid & 31, c1_g = tid >> 5;
    f32x2 wr[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wr[tp] = *reinterpret_cast<const f32x2*>(p.w1 + tp * 64 + 2 * c1_cp);
    const f32x2 bias = *reinterpret_cast<const f32x2*>(p.b1 + 2 * c1_cp);
    for (int e = tid; e < IMG_H * IMG_W; e += 256) {
      const int py = e / IMG_W, px = e % IMG_W;
      const int gy = y0 + py - 2, gx = x0 + px - 2;
      img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    // conv1a + folded BN + ReLU for ALL 64 channels of the 10x18 halo patch, once per workgroup, packed over channel
    // pairs: lane = (pair, one of 8 pixel groups); a group takes runs g, g+8, .. of the 30 six-pixel runs.  Both channels
    // share every tap, so each multiply-add is one v_pk_fma_f32 with the tap broadcast -- 216 packed FMAs per lane instead
    // of 405 scalar ones (the prologue runs beside the co-resident workgroup's MFMAs, where instructions are expensive).
    // Positions outside the image are conv1b's zero padding (mask multiply, no branches).
    {
      const f32x2 zero2 = {0.f, 0.f};
#pragma unroll 1
      for (int run = c1_g; run < 30; run += 8) {
        const int py = run / 3, xr = (run % 3) * 6;
        float tap[3][8];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float2 tv = *reinterpret_cast<const float2*>(img + (py + dy) * IMG_W + xr + 2 * j);
            tap[dy][2 * j] = tv.x;
            tap[dy][2 * j + 1] = tv.y;
          }
        const int gy = y0 + py - 1;
        const float rowmask = (gy >= 0 && gy < H) ? 1.f : 0.f;
#pragma unroll
        for (int px = 0; px < 6; ++px) {
          f32x2 v = bias;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float t = tap[dy][px + dx];
              v = __builtin_elementwise_fma((f32x2){t, t}, wr[dy * 3 + dx], v);
            }
          const int gx = x0 + xr + px - 1;
          const float mask = (gx >= 0 && gx < W) ? rowmask : 0.f;
          *reinterpret_cast<f32x2*>(raw + (py * RW + xr + px) * RSF + 2 * c1_cp) = __builtin_elementwise_max(v, zero2) * mask;
        }
      }
    }
  }

  f32x4 acc[16][2];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][j][r] = 0.f;

  // ---- staging pipeline.  Per chunk c, inside the MFMA loop of chunk c-1 (so their latency and issue
  //      cost hide behind matrix work): the raw 10x18x8 input patch of chunk c is fetched (2 float4 per
  //      thread) and written to LDS `raw`, and the U block of chunk c is fetched into registers.
  //      After the barrier that ends MFMA(c-1): U registers -> LDS, input transform raw -> V (thread =
  //      (channel tc, wtile tw); raw pixel stride 12 makes the 4 wtiles x 8 channels of a half-wave
  //      conflict free), second barrier, MFMA(c).  Two barriers per chunk.
  const int tc = tid & 7, tw = tid >> 3, twr = tw >> 3, twc = tw & 7;
  float4 r0, r1;
  // raw patch items: e = tid + 256*it, e < 360: pixel e>>1 (py = pix / 18, px = pix % 18), channel half e&1
  const float* rsrc0 = nullptr; const float* rsrc1 = nullptr;   // null = zero padding / no item
  int rdst0 = -1, rdst1 = -1;
  if constexpr (!FIRST) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 256;
      if (e < RH * RW * 2) {
        const int pix = e >> 1, half = e & 1, py = pix / RW, px = pix % RW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const float* src = (gy >= 0 && gy < H && gx >= 0 && gx < W)
                               ? p.in + ((size_t)(b * H + gy) * W + gx) * Cin + 4 * half : nullptr;
        const int dst = (py * RPITCH + px) * RS + 4 * half;
        if (it == 0) { rsrc0 = src; rdst0 = dst; } else { rsrc1 = src; rdst1 = dst; }
      }
    }
  }
#define IMX_GRAW(cc_)                                                                                  \
  if constexpr (!FIRST) {                                                                              \
    r0 = rsrc0 ? *reinterpret_cast<const float4*>(rsrc0 + (cc_)) : make_float4(0.f, 0.f, 0.f, 0.f);    \
    r1 = rsrc1 ? *reinterpret_cast<const float4*>(rsrc1 + (cc_)) : make_float4(0.f, 0.f, 0.f, 0.f);    \
  }
#define IMX_SRAW()                                                                                     \
  if constexpr (!FIRST) {                                                                              \
    if (rdst0 >= 0) *reinterpret_cast<float4*>(raw + rdst0) = r0;                                      \
    if (rdst1 >= 0) *reinterpret_cast<float4*>(raw + rdst1) = r1;                                      \
  }

  // TRACE: per-phase s_memtime deltas summed over chunks (bring-up instrumentation, off in the product build)
  unsigned tph[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define IMX_TS(i_)                                                   \
  if constexpr (TRACE) {                                             \
    const unsigned long long now = __builtin_readcyclecounter();     \
    tph[i_] += (unsigned)(now - tprev);                              \
    tprev = now;                                                     \
  }
  IMX_GRAW(0)
  IMX_SRAW()
  if constexpr (TRACE) { tprev = __builtin_readcyclecounter(); t_loop = tprev; }
  for (int ch = 0; ch < nchunk; ++ch) {
    // ---- this chunk's B operands (U = G g G^T) straight from global/L2 into 64 registers: the layout
    //      [pos][k-step][co-block][4 k][16 co] makes every wave load 256 contiguous bytes; they are issued BEFORE the
    //      barrier (the registers are dead once the wave leaves its MFMA loop) and land while the wave waits and the
    //      input transform runs, so the MFMA loop below touches no global memory except the
    //      next raw patch.  U never goes through LDS.
    // wu6 layout [k-step][pos group][co-block][lane][4 pos]: group g's four B registers are ONE buffer_load_dwordx4 with
    // SGPR descriptor / offset -- 8 vector-memory instructions and no address VALU per chunk (was 32 loads + 64 adds)
    f32x4 bf[8];
    {
      const int uoff = __builtin_amdgcn_readfirstlane(((int)blockIdx.y * nchunk + ch) * (USZ * 4));
#pragma unroll
      for (int g = 0; g < 8; ++g)
        bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, uoff + g * 4096, 0));
    }
    __syncthreads();               // previous chunk's MFMA phase is done with raw / V
    IMX_TS(0)
    float d[16];
    IMX_TS(1)
    IMX_TS(2)
    // ---- input transform  V = B^T d B  (LDS raw -> registers -> LDS V), thread = (channel tc, wtile tw)
    {
      constexpr int RSX = FIRST ? RSF : RS;
      const float* rp = raw + ((2 * twr) * RPITCH + 2 * twc) * RSX + tc + (FIRST ? ch * CK : 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) d[a * 4 + bb] = rp[(a * RPITCH + bb) * RSX];
      float tt[4][4];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {       // rows: B^T d
        tt[0][bb] = d[0 * 4 + bb] - d[2 * 4 + bb];
        tt[1][bb] = d[1 * 4 + bb] + d[2 * 4 + bb];
        tt[2][bb] = d[2 * 4 + bb] - d[1 * 4 + bb];
        tt[3][bb] = d[1 * 4 + bb] - d[3 * 4 + bb];
      }
      float* vp = V + (tc >> 1) * VK + tw * 2 + (tc & 1);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {       // columns: (B^T d) B ; position p = xi*4 + nu
        vp[(xi * 4 + 0) * QS] = tt[xi][0] - tt[xi][2];
        vp[(xi * 4 + 1) * QS] = tt[xi][1] + tt[xi][2];
        vp[(xi * 4 + 2) * QS] = tt[xi][2] - tt[xi][1];
        vp[(xi * 4 + 3) * QS] = tt[xi][1] - tt[xi][3];
      }
    }
    IMX_TS(3)
    __syncthreads();
    IMX_TS(4)
    IMX_TS(5)
    const int nch = ch + 1 < nchunk ? ch + 1 : ch;   // next chunk (branch-free: the last iteration re-fetches its own, unused)
    // ---- 16 positions x 2 k-steps x 2 column blocks of v_mfma_f32_16x16x4_f32 in 8 groups of 4
    //      positions (8 MFMAs = 256 cycles).  A operands (V) come from LDS one group ahead, B operands
    //      are already in registers; the next chunk's raw patch load/store rides along.
    {
      // four opaque element offsets (k-step 0/1 x row block 0/1): positions q, q' from one base are q*QS apart -> ds_read2st64
      const int vlane = (lane >> 5) * VK + (lane & 15) * 2 + ((lane >> 4) & 1);
      int o00 = vlane, o01 = vlane + 32, o10 = vlane + 2 * VK, o11 = vlane + 2 * VK + 32;
      asm volatile("" : "+v"(o00), "+v"(o01), "+v"(o10), "+v"(o11));
      const float* b00 = V + o00;
      const float* b01 = V + o01;
      const float* b10 = V + o10;
      const float* b11 = V + o11;
      float af[2][4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i][0] = b00[i * QS];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i][1] = b01[i * QS];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g == 0) { IMX_GRAW(nch * CK) }
        if (g == 6) { IMX_SRAW() }
        if (g + 1 < 8) {
          const int s1 = (g + 1) >> 2, qb = ((g + 1) & 3) * 4;
          const float* r0 = s1 ? b10 : b00;
          const float* r1 = s1 ? b11 : b01;
#pragma unroll
          for (int i = 0; i < 4; ++i) af[nxt][i][0] = r0[(qb + i) * QS];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[nxt][i][1] = r1[(qb + i) * QS];
        }
        const int q0 = (g & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[q0 + i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i][0], bf[g][i], acc[q0 + i][0], 0, 0, 0);
          acc[q0 + i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i][1], bf[g][i], acc[q0 + i][1], 0, 0, 0);
        }
        // one MFMA, then one LDS read (next group's A operand) / one other instruction in its shadow
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
          __builtin_amdgcn_sched_group_barrier(0x026, 1, 0);     // VALU / SALU / VMEM read (raw patch fetch, addressing)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    IMX_TS(6)
  }
  if constexpr (TRACE) t_epi = __builtin_readcyclecounter();
#undef IMX_TS
#undef IMX_GRAW
#undef IMX_SRAW


  // ---- output transform Y = A^T M A, bias, ReLU, (2x2 max-pool), store.
  //      acc[p][ibx][r]: wtile ibx*16 + 4*(lane>>4) + r, channel n0 + cb*16 + (lane&15).
  //      The full-resolution tile goes through LDS (free after the loop) so HBM sees whole 256-byte
  //      channel rows written as float4: per-lane dword stores at a pixel stride are store-issue bound
  //      (measured on the wino4 variant: 19k -> 7k cycles).  The pooled tile is 4x smaller: direct stores.
  constexpr int OS = NT + 4;
  float* Ot = smem;
  __syncthreads();          // every wave is done with V / raw (the staging tile aliases them)
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {          // jb = wtile row block
    const int col = cb * 16 + (lane & 15);
    const float bs = p.bias[n0 + col];
#define ACC(q_) acc[q_][jb]
    // The four accumulator registers of a (position, row block) are the same lane's four wtiles r = 0..3: the whole
    // transform runs on them as f32x4 values, i.e. as v_pk_add_f32 pairs -- half the VALU instructions of a scalar loop
    // over r, with no swizzles.
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t0[nu] = ACC(0 * 4 + nu) + ACC(1 * 4 + nu) + ACC(2 * 4 + nu);
      t1[nu] = ACC(1 * 4 + nu) - ACC(2 * 4 + nu) - ACC(3 * 4 + nu);
    }
    const f32x4 bs4 = {bs, bs, bs, bs}, zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 y00 = t0[0] + t0[1] + t0[2] + bs4, y01 = t0[1] - t0[2] - t0[3] + bs4;
    f32x4 y10 = t1[0] + t1[1] + t1[2] + bs4, y11 = t1[1] - t1[2] - t1[3] + bs4;
    if (RELU) {
      y00 = __builtin_elementwise_max(y00, zero4); y01 = __builtin_elementwise_max(y01, zero4);
      y10 = __builtin_elementwise_max(y10, zero4); y11 = __builtin_elementwise_max(y11, zero4);
    }
    f32x4 pooled = zero4;
    if constexpr (POOL) pooled = __builtin_elementwise_max(__builtin_elementwise_max(y00, y01), __builtin_elementwise_max(y10, y11));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = jb * 16 + 4 * (lane >> 4) + r;
      const int wr = w >> 3, wc = w & 7;
      if constexpr (POOL) {
        Ot[(wr * TC + wc) * OS + col] = pooled[r];
      } else {
        float* o = Ot + ((2 * wr) * OW + 2 * wc) * OS + col;
        o[0] = y00[r];
        o[OS] = y01[r];
        o[OW * OS] = y10[r];
        o[OW * OS + OS] = y11[r];
      }
    }
#undef ACC
  }
  if constexpr (POOL) {
    __syncthreads();
    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
    for (int it = 0; it < TR * TC * (NT / 4) / 256; ++it) {
      const int e = tid + it * 256;
      const int pix = e / (NT / 4), v4 = e % (NT / 4);
      const int oy = (y0 >> 1) + pix / TC, ox = (x0 >> 1) + pix % TC;
      if (oy < Ho && ox < Wo)
        *reinterpret_cast<float4*>(p.out + ((size_t)(b * Ho + oy) * Wo + ox) * Cout + n0 + 4 * v4) =
            *reinterpret_cast<const float4*>(Ot + pix * OS + 4 * v4);
    }
  } else {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < OH * OW * (NT / 4) / 256; ++it) {
      const int e = tid + it * 256;
      const int pix = e / (NT / 4), v4 = e % (NT / 4);
      const int oy = y0 + pix / OW, ox = x0 + pix % OW;
      if (oy < H && ox < W)
        *reinterpret_cast<float4*>(p.out + ((size_t)(b * H + oy) * W + ox) * Cout + n0 + 4 * v4) =
            *reinterpret_cast<const float4*>(Ot + pix * OS + 4 * v4);
    }
  }
  if constexpr (TRACE) {
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
      for (int i = 0; i < 4; ++i) trace[blockIdx.x * 8 + i] = tph[i * 2] + (i < 3 ? tph[i * 2 + 1] : 0);
      trace[blockIdx.x * 8 + 4] = (unsigned)(t_loop - t_start);
      trace[blockIdx.x * 8 + 5] = (unsigned)(t_epi - t_loop);
      trace[blockIdx.x * 8 + 6] = (unsigned)(t_end - t_epi);
    }
  }
}

template <bool POOL, bool RELU, bool FIRST>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  dim3 grid((unsigned)(tiles_x * tiles_y * a.B), (unsigned)(a.Cout / NT));
  size_t lds = (size_t)(VSZ + (FIRST ? RAWF + IMG_H * IMG_W : RAW)) * sizeof(float);
  if (!POOL && lds < (size_t)OH * OW * (NT + 4) * sizeof(float)) lds = (size_t)OH * OW * (NT + 4) * sizeof(float);   // epilogue staging tile
  if (getenv("IMX_WINO_TRACE")) {      // bring-up instrumentation: per-phase cycle counts of the first 4096 workgroups
    static unsigned* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 8 * sizeof(unsigned));
    (void)hipMemsetAsync(dbuf, 0, 4096 * 8 * sizeof(unsigned), s);
    auto kt = conv3x3_wino<POOL, RELU, FIRST, true>;
    hipLaunchKernelGGL(kt, grid, dim3(256), lds, s, a, tiles_x, tiles_y, dbuf);
    (void)hipStreamSynchronize(s);
    static unsigned host[4096 * 8];
    (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
    const int n = grid.x < 4096 ? (int)grid.x : 4096;
    double sum[7] = {0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 7; ++j) sum[j] += host[i * 8 + j];
    const int nchunk = a.Cin / CK;
    fprintf(stderr, "[wino trace] H=%d W=%d Cin=%d Cout=%d pool=%d first=%d grid=%u | prologue %.0f  loop %.0f (%.0f / chunk: barrier+ %.0f  "
                    "transform+B %.0f  barrier %.0f  mfma %.0f)  epilogue %.0f cycles\n", a.H, a.W, a.Cin, a.Cout, (int)POOL, (int)FIRST, grid.x,
            sum[4] / n, sum[5] / n, sum[5] / n / nchunk, sum[0] / n / nchunk, sum[1] / n / nchunk, sum[2] / n / nchunk,
            sum[3] / n / nchunk, sum[6] / n);
    return hipGetLastError();
  }
  auto k = conv3x3_wino<POOL, RELU, FIRST>;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y, (unsigned*)nullptr);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % CK || a.Cout % NT || (a.first && a.Cin != 64) || !a.wu6) return hipErrorInvalidValue;
  if (a.first) return a.pool ? launch_t<true, true, true>(a, s) : launch_t<false, true, true>(a, s);
  if (a.pool) return a.relu ? launch_t<true, true, false>(a, s) : launch_t<true, false, false>(a, s);
  return a.relu ? launch_t<false, true, false>(a, s) : launch_t<false, false, false>(a, s);
}

}  // namespace imx
