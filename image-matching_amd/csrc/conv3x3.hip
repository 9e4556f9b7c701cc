// conv3x3.hip — 3x3 / pad 1 convolution as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain), channels-last activations.
//
// Replaces F.conv2d + folded BatchNorm + ReLU (+ the following MaxPool2d(2)) of
//   superpoint/models/unet_parts.py:10-48 (double_conv / down), superpoint_test.py:113-123.
//
// Workgroup = 256 threads (4 waves) -> output tile 8 rows x 32 cols x 64 output channels.
//   wave w owns rows {2w, 2w+1} (two 32-pixel M-blocks) x two 32-channel N-blocks: 64 acc VGPRs.
// K loop: input channels in chunks of 16; per chunk the (8+2)x(32+2) halo tile [pixel][16(+1 pad)]
// and the weights [9][16][64] are staged in LDS (60 KB -> 2 workgroups per CU overlap staging with
// MFMA).  A operand: lane l reads pixel x=(l&31)+dx, channel 2kk+(l>>5) (pixel stride 17 floats ->
// conflict free); B operand: lane l reads channel-row 2kk+(l>>5), output channel (l&31).
// FIRST mode fuses conv1a (1->64, K=9, VALU) into the staging step: the 64-channel input tile is
// never written to HBM (saves 2 x 78.6 MB per 480x640 image).  POOL mode fuses the 2x2 max pool
// into the epilogue (both vertical neighbours live in the same lane's two M-blocks, horizontal
// neighbours in adjacent accumulator registers).
#include "imx_kernels.h"

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int CK = 16, S = CK + 1, NT = 64, NB = NT / 32;
constexpr int IN_TILE = PH * PW * S;                 // 5780 floats
constexpr int IN_TILE_PAD = (IN_TILE + 3) & ~3;      // 16-B aligned start for the weight tile
constexpr int W_TILE = 9 * CK * NT;                  // 9216 floats
constexpr int IMG_H = TH + 4, IMG_W = TW + 4;        // FIRST: image patch 12 x 36
constexpr int FIRST_EXTRA = IMG_H * IMG_W;

template <bool POOL, bool RELU, bool FIRST>
__global__ __launch_bounds__(256) void conv3x3_mfma(ConvArgs p, int tiles_x, int tiles_y) {
  extern __shared__ float smem[];
  float* in_tile = smem;
  float* w_tile = smem + IN_TILE_PAD;
  float* img = w_tile + W_TILE;          // FIRST only: (8+4) x (32+4) image patch

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * NT;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;

  if constexpr (FIRST) {
    const float* im = (b < p.split) ? p.in + (size_t)b * H * W : p.in2 + (size_t)(b - p.split) * H * W;
    for (int e = tid; e < IMG_H * IMG_W; e += 256) {
      int py = e / IMG_W, px = e % IMG_W;
      int gy = y0 + py - 2, gx = x0 + px - 2;
      img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    }
  }

  f32x16 acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // Global -> register -> LDS staging.  All of a chunk's loads (6 float4 of the input halo tile +
  // 9 float4 of weights per thread) are issued back to back so one memory latency covers the lot,
  // and they are issued BEFORE the previous chunk's MFMA loop so that latency hides behind ~18k
  // cycles of matrix work; the LDS write happens after the loop (barrier, write, barrier).
  // Named float4 registers (not arrays): see gemm.hip.
  float4 i0, i1, i2, i3, i4, i5, w0, w1r, w2, w3, w4, w5, w6, w7, w8;
  constexpr int V = CK / 4;
#define IMX_GI(reg_, it_)                                                                              \
  {                                                                                                    \
    const int e = tid + (it_) * 256;                                                                   \
    reg_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                            \
    if (e < PH * PW * V) {                                                                             \
      const int pix = e / V, v4 = e % V, py = pix / PW, px = pix % PW;                                 \
      const int gy = y0 + py - 1, gx = x0 + px - 1;                                                    \
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)                                                      \
        reg_ = *reinterpret_cast<const float4*>(p.in + ((size_t)(b * H + gy) * W + gx) * Cin + cc + 4 * v4); \
    }                                                                                                  \
  }
#define IMX_SI(reg_, it_)                                                                              \
  {                                                                                                    \
    const int e = tid + (it_) * 256;                                                                   \
    if (e < PH * PW * V) {                                                                             \
      float* d = in_tile + (e / V) * S + 4 * (e % V);                                                  \
      d[0] = reg_.x; d[1] = reg_.y; d[2] = reg_.z; d[3] = reg_.w;                                      \
    }                                                                                                  \
  }
#define IMX_GWT(reg_, it_)                                                                             \
  {                                                                                                    \
    const int idx = (tid + (it_) * 256) * 4, col = idx % NT, row = idx / NT, tap = row / CK, k = row % CK; \
    reg_ = *reinterpret_cast<const float4*>(p.w + ((size_t)(tap * Cin + cc + k)) * Cout + n0 + col);   \
  }
#define IMX_SWT(reg_, it_) *reinterpret_cast<float4*>(w_tile + (tid + (it_) * 256) * 4) = reg_;
#define IMX_GLOAD(c0_)                                                                                 \
  {                                                                                                    \
    const int cc = (c0_);                                                                              \
    if constexpr (!FIRST) { IMX_GI(i0, 0) IMX_GI(i1, 1) IMX_GI(i2, 2) IMX_GI(i3, 3) IMX_GI(i4, 4) IMX_GI(i5, 5) } \
    IMX_GWT(w0, 0) IMX_GWT(w1r, 1) IMX_GWT(w2, 2) IMX_GWT(w3, 3) IMX_GWT(w4, 4)                        \
    IMX_GWT(w5, 5) IMX_GWT(w6, 6) IMX_GWT(w7, 7) IMX_GWT(w8, 8)                                        \
  }
#define IMX_LSTORE()                                                                                   \
  {                                                                                                    \
    if constexpr (!FIRST) { IMX_SI(i0, 0) IMX_SI(i1, 1) IMX_SI(i2, 2) IMX_SI(i3, 3) IMX_SI(i4, 4) IMX_SI(i5, 5) } \
    IMX_SWT(w0, 0) IMX_SWT(w1r, 1) IMX_SWT(w2, 2) IMX_SWT(w3, 3) IMX_SWT(w4, 4)                        \
    IMX_SWT(w5, 5) IMX_SWT(w6, 6) IMX_SWT(w7, 7) IMX_SWT(w8, 8)                                        \
  }
  static_assert(W_TILE / 4 == 9 * 256 && PH * PW * V <= 6 * 256, "staging map");

  IMX_GLOAD(0)
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    __syncthreads();          // every wave is done reading the previous chunk's tiles
    // ---- stage the input halo tile for channels [c0, c0+CK)
    if constexpr (FIRST) {
      // conv1a + folded BN + ReLU evaluated in place; positions outside the image are conv1b's
      // zero padding (NOT conv1a evaluated out of range).  One work item = one halo pixel x 8
      // channels: 9 patch reads feed 72 FMAs whose weights are wave-uniform (scalar loads from
      // the 2.3 KB conv1a table), so the staging costs ~4x fewer instructions than one
      // (pixel, channel) per thread.  Slots: [half 0: 384][half 1: 384], 340 valid per half.
#pragma unroll 1
      for (int it = 0; it < 3; ++it) {
        const int slot = it * 256 + tid;
        const int half = __builtin_amdgcn_readfirstlane(slot / 384);
        const int pix = slot - half * 384;
        if (pix < PH * PW) {
          const int py = pix / PW, px = pix - py * PW;
          const int gy = y0 + py - 1, gx = x0 + px - 1;
          const int cb = c0 + 8 * half;
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float im[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) im[tp] = img[(py + tp / 3) * IMG_W + px + tp % 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p.b1[cb + j];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaf(im[tp], p.w1[tp * 64 + cb + j], v[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          float* d = in_tile + pix * S + 8 * half;
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = v[j];
        }
      }
    }
    IMX_LSTORE()
    __syncthreads();
    // prefetch the next chunk (branch-free: the last iteration re-fetches its own chunk, unused)
    IMX_GLOAD(c0 + CK < Cin ? c0 + CK : c0)
    // ---- 9 taps x 8 k-steps x 4 MFMAs, fully unrolled (all LDS offsets are immediates); the operand
    //      fragments of step s+1 are fetched from LDS before the MFMAs of step s are issued
    //      (sched_barrier pins that order: hipcc otherwise sinks each ds_read next to its use and
    //      exposes the LDS latency every two MFMAs).
    {
      const float* a0p = in_tile + ((2 * wave) * PW + (lane & 31)) * S + (lane >> 5);
      const float* bp = w_tile + (lane >> 5) * NT + (lane & 31);
      constexpr int NSTEP = 9 * (CK / 2);
      float af[2][2], bf[2][NB];
      af[0][0] = a0p[0];
      af[0][1] = a0p[PW * S];
#pragma unroll
      for (int n = 0; n < NB; ++n) bf[0][n] = bp[n * 32];
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        if (st + 1 < NSTEP) {
          const int tap = (st + 1) / (CK / 2), kk = (st + 1) % (CK / 2);
          const int dy = tap / 3, dx = tap % 3;
          const int aoff = (dy * PW + dx) * S + 2 * kk;
          af[nxt][0] = a0p[aoff];
          af[nxt][1] = a0p[aoff + PW * S];
#pragma unroll
          for (int n = 0; n < NB; ++n) bf[nxt][n] = bp[(tap * CK + 2 * kk) * NT + n * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][0], bf[cur][n], acc[0][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][1], bf[cur][n], acc[1][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

#undef IMX_GI
#undef IMX_SI
#undef IMX_GWT
#undef IMX_SWT
#undef IMX_GLOAD
#undef IMX_LSTORE

  // ---- epilogue.  acc[m][n][r]: pixel row y0+2*wave+m, pixel col x0 + (r&3)+8*(r>>2)+4*(lane>>5),
  //      output channel n0 + 32n + (lane&31).
  const int hi = lane >> 5;
  if constexpr (POOL) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int oy = (y0 >> 1) + wave;
    if (oy < Ho) {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int co = n0 + n * 32 + (lane & 31);
        const float bs = p.bias[co];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float v = fmaxf(fmaxf(acc[0][n][r], acc[0][n][r + 1]), fmaxf(acc[1][n][r], acc[1][n][r + 1])) + bs;
          if (RELU) v = fmaxf(v, 0.f);
          const int xi = (r & 3) + 8 * (r >> 2) + 4 * hi;   // even
          const int ox = (x0 + xi) >> 1;
          if (ox < Wo) p.out[((size_t)(b * Ho + oy) * Wo + ox) * Cout + co] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = y0 + 2 * wave + m;
      if (y >= H) continue;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int co = n0 + n * 32 + (lane & 31);
        const float bs = p.bias[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[m][n][r] + bs;
          if (RELU) v = fmaxf(v, 0.f);
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (x < W) p.out[((size_t)(b * H + y) * W + x) * Cout + co] = v;
        }
      }
    }
  }
}

template <bool POOL, bool RELU, bool FIRST>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  dim3 grid((unsigned)(tiles_x * tiles_y * a.B), (unsigned)(a.Cout / NT));
  size_t lds = (size_t)(IN_TILE_PAD + W_TILE + (FIRST ? FIRST_EXTRA : 0)) * sizeof(float);
  auto k = conv3x3_mfma<POOL, RELU, FIRST>;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % CK || a.Cout % NT || (a.first && a.Cin != 64)) return hipErrorInvalidValue;
  last_form = "conv3x3_direct:f32";
  if (a.first) return a.pool ? launch_t<true, true, true>(a, s) : launch_t<false, true, true>(a, s);
  if (a.pool) return a.relu ? launch_t<true, true, false>(a, s) : launch_t<true, false, false>(a, s);
  return a.relu ? launch_t<false, true, false>(a, s) : launch_t<false, false, false>(a, s);
}

}  // namespace imx
