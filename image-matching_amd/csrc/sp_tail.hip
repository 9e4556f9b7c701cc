// sp_tail.hip — SuperPoint post-network stages (HBM/LDS-bound, no matrix work):
//   softmax over 65 channels + pixel shuffle      superpoint_test.py:128-131
//   simple_nms (all 5 max-pools fused in LDS)     superpoint_test.py:7-22
//   threshold / remove_borders / top-k            superpoint_test.py:135-149, :25-37
//   flip + sample_descriptors                     superpoint_test.py:151-155, :40-52
#include "imx_kernels.h"
#include <math.h>

namespace imx {
namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------ softmax + pixel shuffle
// One wave per 8x8 cell: lane c holds channel c (c<64), channel 64 (dustbin) is read by all lanes.  A workgroup of eight
// waves takes eight consecutive cells of a cell row and stages their 8 x 64 output pixels through LDS, so the score map
// leaves as 256-byte row segments (one wave per cell writing its own 8 x 8 block put 32-byte pieces on eight different
// rows: 1.8 TB/s; round 2).  Same per-cell arithmetic.
__global__ __launch_bounds__(512) void softmax_shuffle_kernel(const float* __restrict__ semi, int ld,
                                                              float* __restrict__ scores, int B, int Hc, int Wc) {
  __shared__ float tile[8][64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = blockIdx.y * 8 + wave;
  const long rowc = blockIdx.x;                      // b * Hc + y (grid.x: no 65535 limit)
  if (x < Wc) {
    const float* s = semi + (rowc * Wc + x) * ld;
    const float v = s[lane], vd = s[64];
    const float m = fmaxf(wave_max(v), vd);
    const float e = expf(v - m), ed = expf(vd - m);
    const float sum = wave_sum(e) + ed;
    tile[lane >> 3][wave * 8 + (lane & 7)] = e / sum;
  }
  __syncthreads();
  const int W8 = Wc * 8, px = blockIdx.y * 64 + lane;      // thread -> output row `wave` of the cell row, pixel `lane` of the 64
  if (px < W8) scores[(rowc * 8 + wave) * W8 + px] = tile[wave][lane];
}

// ------------------------------------------------------------------ dense export (label-export / training forward)
// One wave per 8x8 cell: channels-last semi (B,Hc,Wc,ld) and raw descriptors (B,Hc,Wc,d) ->
// reference layout semi (B,65,Hc,Wc), desc (B,d,Hc,Wc) divided by its channel norm
// (superpoint/models/superpoint_train.py:46-55; same arithmetic as superpoint_test.py:119-126).
__global__ __launch_bounds__(256) void dense_export_kernel(const float* __restrict__ semi, int ld, const float* __restrict__ dense,
                                                           int d, float* __restrict__ semi_out, float* __restrict__ desc_out,
                                                           int B, int Hc, int Wc, int eps_mode) {
  const int lane = threadIdx.x & 63;
  const long cell = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)B * Hc * Wc;
  if (cell >= total) return;
  const long hw = (long)Hc * Wc;
  const long b = cell / hw, yx = cell % hw;
  for (int c = lane; c < 65; c += 64) semi_out[(b * 65 + c) * hw + yx] = semi[cell * ld + c];
  float s2 = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = dense[cell * d + c]; s2 += v * v; }
  float dn = sqrtf(wave_sum(s2));
  if (eps_mode) dn = fmaxf(dn, 1e-12f);
  for (int c = lane; c < d; c += 64) desc_out[(b * d + c) * hw + yx] = dense[cell * d + c] / dn;
}

// ------------------------------------------------------------------ simple_nms
// Two forms:
//   staged   radius 1..4 (the reference's default is 4): three LDS-tiled kernels, one per round, masks as bit rows -- the fast one;
//   generic  any other radius: the five max-pools as separable global-memory passes (row maximum, then column maximum with the
//            round's elementwise step fused into its epilogue).  Not tuned: it exists so that no nms_radius is refused.
// Both are compare-only arithmetic on the caller's score map: bit-identical to the reference given the same map.
// Suppressed pixels are encoded as -1 in the supp_scores copy (scores are >= 0): for an un-suppressed pixel the window maximum
// is its own score either way, so `supp_scores == max_pool(supp_scores) & ~supp` is unchanged.
__global__ __launch_bounds__(256) void nms_pool_rows(const float* __restrict__ in, float* __restrict__ out, int H, int W, int r) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float* row = in + ((size_t)blockIdx.z * H + y) * W;
  const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;     // max_pool2d pads with -inf: clamp the window
  float m = row[x0];
  for (int xx = x0 + 1; xx <= x1; ++xx) m = fmaxf(m, row[xx]);
  out[((size_t)blockIdx.z * H + y) * W + x] = m;
}

enum { NMS_MASK0 = 0, NMS_SUPP = 1, NMS_ROUND = 2, NMS_FINAL = 3 };
// column maximum of the row maxima `rp` = the (2r+1)^2 pool, then (superpoint_test.py:7-22)
//   MASK0  mask = scores == pool(scores)                                           (:13)
//   SUPP   rp = rows of pool(mask): ss = pool(mask) > 0 ? -1 : scores              (:16-17)
//   ROUND  rp = rows of pool(ss):   mask |= ss >= 0 && ss == pool(ss)              (:18-19)
//   FINAL  the last ROUND, writing out = mask ? scores : 0                         (:20)
__global__ __launch_bounds__(256) void nms_pool_cols(const float* __restrict__ rp, const float* __restrict__ scores,
                                                     float* __restrict__ ss, float* __restrict__ mask, float* __restrict__ out,
                                                     int H, int W, int r, int mode) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t img = (size_t)blockIdx.z * H * W, at = img + (size_t)y * W + x;
  const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
  float m = rp[img + (size_t)y0 * W + x];
  for (int yy = y0 + 1; yy <= y1; ++yy) m = fmaxf(m, rp[img + (size_t)yy * W + x]);
  if (mode == NMS_MASK0) {
    mask[at] = scores[at] == m ? 1.f : 0.f;
  } else if (mode == NMS_SUPP) {
    ss[at] = m > 0.f ? -1.f : scores[at];
  } else {
    const float v = ss[at];
    const bool keep = mask[at] != 0.f || (v >= 0.f && v == m);
    if (mode == NMS_ROUND) mask[at] = keep ? 1.f : 0.f;
    else out[at] = keep ? scores[at] : 0.f;
  }
}

// Staged form (radius 1..4, needs 2 x B x H x ceil(W/32) words of scratch): simple_nms as THREE kernels, one per round.
// A single fused kernel (round 2, removed) pays for its five dependent pools with a 5R halo (3.7x the tile area at R = 4) and
// ~20 block-wide barriers of 16 waves: 0.81 ms per C3 step against 0.36 here, where a round only needs the halo of its own pools:
//   stage 0   max_mask = scores == max_pool(scores)                                     halo R, mask written as BIT rows
//   stage 1/2 supp = dilate(max_mask) (bit rows: shifts + an OR over 2R+1 rows), supp_scores = supp ? -1 : scores,
//             max_mask |= ~supp & (supp_scores == max_pool(supp_scores))                halo 2R for the bits, R for the scores
// stage 2 writes where(max_mask, scores, 0) instead of the bits.  Same compare-only arithmetic: bit-exact.
// stage 3 (round 6) = stage 2 for the detector's own use: it writes the KEYPOINT CANDIDATE bits -- max_mask & score > threshold & inside
// the border (superpoint_test.py:135-141) -- as bit rows again, and the keypoint kernels count / scatter from those 20 words per row
// instead of three passes over the dense map (its 157-MB write here and two reads there, at C3).
template <int R, int STAGE>
__global__ __launch_bounds__(256) void nms_stage_kernel(const float* __restrict__ scores, const unsigned* __restrict__ min_,
                                                         unsigned* __restrict__ mout, float* __restrict__ out, int H, int W, int WW,
                                                         float thr, int border) {
  constexpr int TY = 32, TX = 64, SY = TY + 2 * R, SX = TX + 2 * R, PITCH = SX | 1, NV = 8 + 2 * R;
  constexpr int MR = SY + 2 * R;                     // bit rows of the previous mask (halo 2R)
  __shared__ float P[SY * PITCH];                    // scores / supp_scores on the region
  __shared__ float Hb[SY * (TX + 1)];                // horizontal pass
  __shared__ unsigned Mb[MR * 4], Hd[MR * 4], Sp[SY * 4];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const float* img = scores + (size_t)b * H * W;
  const float NEG = -INFINITY;

  if constexpr (STAGE > 0) {
    // previous mask: bit rows y0 - 2R .. , the 128-bit window of columns x0 - 32 .. x0 + 95 (x0 is a multiple of 64)
    const unsigned* mb = min_ + (size_t)b * H * WW;
    for (int e = tid; e < MR * 4; e += 256) {
      const int gy = y0 - 2 * R + e / 4, wi = (x0 >> 5) - 1 + (e & 3);
      Mb[e] = (gy >= 0 && gy < H && wi >= 0 && wi < WW) ? mb[(size_t)gy * WW + wi] : 0u;
    }
    __syncthreads();
    for (int e = tid; e < MR * 4; e += 256) {        // horizontal dilation inside the window
      const int wi = e & 3;
      const unsigned w = Mb[e], wl = wi > 0 ? Mb[e - 1] : 0u, wr = wi < 3 ? Mb[e + 1] : 0u;
      unsigned d = w;
#pragma unroll
      for (int sft = 1; sft <= R; ++sft) d |= (w << sft) | (wl >> (32 - sft)) | (w >> sft) | (wr << (32 - sft));
      Hd[e] = d;
    }
    __syncthreads();
    for (int e = tid; e < SY * 4; e += 256) {        // vertical: region row ry <-> mask rows ry .. ry + 2R
      unsigned d = 0u;
#pragma unroll
      for (int k = 0; k <= 2 * R; ++k) d |= Hd[e + 4 * k];
      Sp[e] = d;
    }
    __syncthreads();
  }
  // region of (supp_)scores: image rows y0 - R .., columns x0 - R ..; outside the image -inf (max_pool padding)
  if (R == 4 && (W & 3) == 0) {
    // (round 6) the default radius on a width that is a multiple of four: the region's rows start 16-byte aligned (x0 - 4, x0 a multiple
    // of 64) and every aligned group of four pixels lies wholly inside or outside the image -- 720 float4 loads instead of 2880 dword
    // loads, a quarter of the index arithmetic (these kernels are VALU-bound: ~600 instructions per thread), the four suppression bits
    // of a group in one word.  Same values into the same LDS cells.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    constexpr int Q = SX / 4;                          // 18
    for (int e = tid; e < SY * Q; e += 256) {
      const int ry = e / Q, c4 = e - ry * Q, gy = y0 - R + ry, gx = x0 - R + 4 * c4;
      f32x4_ v = {NEG, NEG, NEG, NEG};
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4_*>(img + (size_t)gy * W + gx);
      if constexpr (STAGE > 0) {
        const int bit = 4 * c4 + 32 - R;               // (a multiple of four: the group's bits share a word)
        const unsigned nib = (Sp[ry * 4 + (bit >> 5)] >> (bit & 31)) & 0xfu;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] > NEG && ((nib >> k) & 1u)) v[k] = -1.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) P[ry * PITCH + 4 * c4 + k] = v[k];
    }
  } else {
    for (int e = tid; e < SY * SX; e += 256) {
      const int ry = e / SX, rx = e - ry * SX, gy = y0 - R + ry, gx = x0 - R + rx;
      float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(size_t)gy * W + gx] : NEG;
      if constexpr (STAGE > 0) {
        const int bit = rx + 32 - R;                   // column x0 - R + rx in the window that starts at x0 - 32
        if (v > NEG && ((Sp[ry * 4 + (bit >> 5)] >> (bit & 31)) & 1u)) v = -1.f;
      }
      P[ry * PITCH + rx] = v;
    }
  }
  __syncthreads();
  // separable max-pool through 8-output register strips: rows SY x columns TX, then rows TY x columns TX
  for (int sidx = tid; sidx < SY * (TX / 8); sidx += 256) {
    const int row = sidx % SY, st = sidx / SY;
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = P[row * PITCH + 8 * st + k];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float m = v[o];
#pragma unroll
      for (int k = 1; k <= 2 * R; ++k) m = fmaxf(m, v[o + k]);
      Hb[row * (TX + 1) + 8 * st + o] = m;
    }
  }
  __syncthreads();
  {
    const int tx = tid & 63, st = tid >> 6;          // a wave = 64 consecutive columns of rows 8 st .. 8 st + 7
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = Hb[(8 * st + k) * (TX + 1) + tx];
    const int gx = x0 + tx;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float m = v[o];
#pragma unroll
      for (int k = 1; k <= 2 * R; ++k) m = fmaxf(m, v[o + k]);
      const int ty = 8 * st + o, gy = y0 + ty;
      const float x = P[(ty + R) * PITCH + tx + R];  // this pixel's own (supp_)score
      const bool in = gy < H && gx < W;
      bool mx;
      if constexpr (STAGE == 0) {
        mx = in && x == m;                                                                // (:16)
      } else {
        const int bit = tx + 32;
        const bool was = (Mb[(ty + 2 * R) * 4 + (bit >> 5)] >> (bit & 31)) & 1u;
        mx = in && (was || (x >= 0.f && x == m));                                         // (:19-21)
      }
      if constexpr (STAGE == 3) {
        mx = mx && gy >= border && gy < H - border && gx >= border && gx < W - border;
        if (mx) mx = img[(size_t)gy * W + gx] > thr;         // (the ORIGINAL score: P holds -1 where the pixel was suppressed)
      }
      if constexpr (STAGE != 2) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mx);
        if (tx == 0 && gy < H) {
          unsigned* mo = mout + ((size_t)b * H + gy) * WW + (x0 >> 5);
          mo[0] = (unsigned)bal;
          if ((x0 >> 5) + 1 < WW) mo[1] = (unsigned)(bal >> 32);
        }
      } else {
        if (in) out[((size_t)b * H + gy) * W + gx] = mx ? img[(size_t)gy * W + gx] : 0.f;   // (:22)
      }
    }
  }
}

template <int R>
hipError_t launch_nms_staged(const float* scores, float* out, unsigned* scratch, int B, int H, int W, hipStream_t s, float thr = 0.f, int border = 0) {
  const int WW = (W + 31) / 32;
  unsigned* m0 = scratch;
  unsigned* m1 = scratch + (size_t)B * H * WW;
  dim3 grid((W + 63) / 64, (H + 31) / 32, B);
  hipLaunchKernelGGL((nms_stage_kernel<R, 0>), grid, dim3(256), 0, s, scores, (const unsigned*)nullptr, m0, (float*)nullptr, H, W, WW, 0.f, 0);
  hipLaunchKernelGGL((nms_stage_kernel<R, 1>), grid, dim3(256), 0, s, scores, (const unsigned*)m0, m1, (float*)nullptr, H, W, WW, 0.f, 0);
  if (out) hipLaunchKernelGGL((nms_stage_kernel<R, 2>), grid, dim3(256), 0, s, scores, (const unsigned*)m1, (unsigned*)nullptr, out, H, W, WW, 0.f, 0);
  else hipLaunchKernelGGL((nms_stage_kernel<R, 3>), grid, dim3(256), 0, s, scores, (const unsigned*)m1, m0, (float*)nullptr, H, W, WW, thr, border);   // candidate bits over mask 0
  return hipGetLastError();
}

// ------------------------------------------------------------------ keypoint extraction
__device__ __forceinline__ bool kp_flag(const float* nms, int H, int W, int y, int x, float thr, int border) {
  return y >= border && y < H - border && x >= border && x < W - border && nms[(size_t)y * W + x] > thr;
}

// one wave per image row
__global__ __launch_bounds__(256) void kp_count_rows(KeypointArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * a.H) return;
  const int b = (int)(row / a.H), y = (int)(row % a.H);
  const float* nms = a.nms + (size_t)b * a.H * a.W;
  int cnt = 0;
  for (int x0 = 0; x0 < a.W; x0 += 64) {
    int x = x0 + lane;
    bool f = x < a.W && kp_flag(nms, a.H, a.W, y, x, a.threshold, a.border);
    cnt += __popcll(__ballot(f));
  }
  if (lane == 0) a.row_count[row] = cnt;
}

// one block per image: exclusive scan of the row counts
__global__ __launch_bounds__(256) void kp_scan_rows(KeypointArgs a) {
  __shared__ int part[256];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int y0 = 0; y0 < a.H; y0 += 256) {
    int y = y0 + tid;
    int v = y < a.H ? a.row_count[(size_t)b * a.H + y] : 0;
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int t = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (y < a.H) a.row_off[(size_t)b * a.H + y] = carry + part[tid] - v;
    __syncthreads();
    if (tid == 255) carry += part[255];
    __syncthreads();
  }
  if (tid == 0) a.cand_count[b] = carry;
}

__global__ __launch_bounds__(256) void kp_scatter_rows(KeypointArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * a.H) return;
  const int b = (int)(row / a.H), y = (int)(row % a.H);
  const float* nms = a.nms + (size_t)b * a.H * a.W;
  int off = a.row_off[row];
  int* ci = a.cand_idx + (size_t)b * a.H * a.W;
  float* cs = a.cand_score + (size_t)b * a.H * a.W;
  for (int x0 = 0; x0 < a.W; x0 += 64) {
    int x = x0 + lane;
    bool f = x < a.W && kp_flag(nms, a.H, a.W, y, x, a.threshold, a.border);
    unsigned long long m = __ballot(f);
    if (f) {
      int pos = off + __popcll(m & ((1ull << lane) - 1ull));
      ci[pos] = y * a.W + x;
      cs[pos] = nms[(size_t)y * a.W + x];
    }
    off += __popcll(m);
  }
}

// ---- the same three steps from the CANDIDATE BIT rows of nms_stage_kernel<R, 3> (round 6: a.cand_bits, a.scores = the un-suppressed score
// map): a row is ceil(W/32) words, so counting is a popcount per word inside the per-image scan and the scatter touches the score map only
// at the candidates.  Row-major order as before (words ascending, bits ascending).
__global__ __launch_bounds__(256) void kp_scan_rows_bits(KeypointArgs a) {
  __shared__ int part[256];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x, WW = (a.W + 31) / 32;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int y0 = 0; y0 < a.H; y0 += 256) {
    const int y = y0 + tid;
    int v = 0;
    if (y < a.H) {
      const unsigned* wr = a.cand_bits + ((size_t)b * a.H + y) * WW;
      for (int w = 0; w < WW; ++w) v += __popc(wr[w]);
    }
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int t = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (y < a.H) a.row_off[(size_t)b * a.H + y] = carry + part[tid] - v;
    __syncthreads();
    if (tid == 255) carry += part[255];
    __syncthreads();
  }
  if (tid == 0) a.cand_count[b] = carry;
}

// one wave per image row: lane l owns word l (+ 64 k) of the row
__global__ __launch_bounds__(256) void kp_scatter_rows_bits(KeypointArgs a) {
  const int lane = threadIdx.x & 63, WW = (a.W + 31) / 32;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * a.H) return;
  const int b = (int)(row / a.H), y = (int)(row % a.H);
  const unsigned* wr = a.cand_bits + (size_t)row * WW;
  const float* sc = a.scores + ((size_t)b * a.H + y) * a.W;
  int off = a.row_off[row];
  int* ci = a.cand_idx + (size_t)b * a.H * a.W;
  float* cs = a.cand_score + (size_t)b * a.H * a.W;
  for (int w0 = 0; w0 < WW; w0 += 64) {
    unsigned bits = w0 + lane < WW ? wr[w0 + lane] : 0u;
    const int n = __popc(bits);
    int inc = n;                                      // inclusive prefix of the lanes' counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    int pos = off + inc - n;
    while (bits) {
      const int x = (w0 + lane) * 32 + __ffs(bits) - 1;
      bits &= bits - 1;
      ci[pos] = y * a.W + x;
      cs[pos] = sc[x];
      ++pos;
    }
    off += __shfl(inc, 63);
  }
}

// block-wide exclusive scan of one int per thread (1024 threads); returns exclusive prefix, total in *tot
__device__ int block_scan_1024(int v, int* wsum /*[17]*/, int* tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < 16; ++w) { int t = wsum[w]; wsum[w] = c; c += t; }
    wsum[16] = c;
  }
  __syncthreads();
  int res = wsum[wave] + inc - v;
  *tot = wsum[16];
  __syncthreads();
  return res;
}

__device__ __forceinline__ void kp_store_count(const KeypointArgs& a, int b, int n) {
  a.sel_count[b] = n;
  if (a.counts_out[0]) a.counts_out[0][b] = n;
  if (b < a.counts_split) { if (a.counts_out[1]) a.counts_out[1][b] = n; }
  else if (a.counts_out[2]) a.counts_out[2][b - a.counts_split] = n;
}

// top-k per image: radix select of the k-th largest score, ordered tie handling (lowest index
// first), then bitonic sort of the k survivors by (score desc, index asc) = torch.topk's sorted
// output (:33-37) with a deterministic tie rule.  If count <= k (or k < 0) the row-major
// candidate order is kept unchanged, as the reference does.
// The P sort slots live in LDS up to 16384 of them (128 KB); above that (max_keypoints > 16384 with more candidates than that)
// they live in a.sort_scratch -- the same code, one block per image walking global memory: slow, but no max_keypoints is refused.
__global__ __launch_bounds__(1024) void kp_topk(KeypointArgs a, int P /* pow2 >= k */) {
  extern __shared__ unsigned long long lds_keys[];   // P entries when P <= 16384
  unsigned long long* keys = P <= 16384 ? lds_keys : a.sort_scratch + (size_t)blockIdx.x * P;
  // round 3 (single-pair latency: this kernel was 76 us of a 1.7 ms pair): one histogram per wave (the top byte of a score in
  // (0.005, 1) takes two or three values: 7000 atomics on three LDS words), the digit picked by a 256-thread scan instead of a
  // serial walk, survivors placed with one atomic per wave, and the sort's 45 in-wave stages (partner distance < 64) as shuffles
  __shared__ int hist[16][256];
  __shared__ int wsum[17];
  __shared__ unsigned sh_prefix;
  __shared__ int sh_kth, sh_npos;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.cand_count[b];
  const int K = a.max_keypoints;
  const int* ci = a.cand_idx + (size_t)b * a.H * a.W;
  const float* cs = a.cand_score + (size_t)b * a.H * a.W;
  int* si = a.sel_idx + (size_t)b * a.Ksel;
  float* ss = a.sel_score + (size_t)b * a.Ksel;

  if (K == 0) {
    if (tid == 0) kp_store_count(a, b, 0);
    return;
  }
  if (K < 0 || n <= K) {
    for (int i = tid; i < n; i += 1024) { si[i] = ci[i]; ss[i] = cs[i]; }
    if (tid == 0) kp_store_count(a, b, n);
    return;
  }
  // ---- radix select (scores are positive floats: bit pattern is monotone)
  unsigned prefix = 0, mask = 0;
  int kth = K;
  for (int pass = 3; pass >= 0; --pass) {
#pragma unroll
    for (int e = 0; e < 4; ++e) (&hist[0][0])[tid + 1024 * e] = 0;
    __syncthreads();
    const int sh = 8 * pass;
    for (int i = tid; i < n; i += 1024) {
      unsigned key = __float_as_uint(cs[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[wave][(key >> sh) & 255], 1);
    }
    __syncthreads();
    // digit d = 255 - tid (descending): inclusive count of keys with a digit >= d; the digit of the k-th largest is the one
    // whose count reaches kth
    int h = 0, inc = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < 16; ++w) h += hist[w][255 - tid];
      inc = h;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wsum[wave] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      int base = 0;
      for (int w = 0; w < wave; ++w) base += wsum[w];
      inc += base;
      if (inc >= kth && inc - h < kth) {        // exactly one thread: the counts are non-decreasing and the last one is >= kth
        sh_kth = kth - (inc - h);
        sh_prefix = prefix | ((unsigned)(255 - tid) << sh);
      }
    }
    __syncthreads();
    kth = sh_kth;
    prefix = sh_prefix;
    mask |= 0xFFu << sh;
  }
  const unsigned pivot = prefix;       // bit pattern of the k-th largest score
  const int take_eq = kth;             // how many scores == pivot to keep (lowest indices first)
  // ---- gather survivors into LDS
  for (int i = tid; i < P; i += 1024) keys[i] = 0ull;
  if (tid == 0) sh_npos = 0;
  __syncthreads();
  int eq_base = 0;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    int i = i0 + tid;
    unsigned key = i < n ? __float_as_uint(cs[i]) : 0u;
    int idx = i < n ? ci[i] : 0;
    bool gt = i < n && key > pivot, eq = i < n && key == pivot;
    // ties at the pivot are kept lowest index first: their rank needs a block scan only in a chunk that holds some of them and
    // cannot take them all (scores are almost always distinct: one barrier per chunk instead of three)
    const int ne = __syncthreads_count(eq);
    bool take = gt;
    if (ne > 0) {
      if (eq_base + ne <= take_eq) {
        take = gt || eq;
      } else {
        int tot;
        const int er = block_scan_1024(eq ? 1 : 0, wsum, &tot);
        take = gt || (eq && (eq_base + er) < take_eq);
      }
      eq_base += ne;
    }
    const unsigned long long bal = __ballot(take);
    if (bal) {
      int base = 0;
      const int first = __ffsll((long long)bal) - 1;
      if (lane == first) base = atomicAdd(&sh_npos, __popcll(bal));
      base = __shfl(base, first);
      if (take) keys[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)idx);
    }
  }
  __syncthreads();
  // ---- bitonic sort, descending
  if (P <= 1024) {
    // one key per thread in a register; partner distances below 64 are in-wave shuffles, the others go through LDS
    unsigned long long x = tid < P ? keys[tid] : 0ull;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
      const bool desc = (tid & k2) == 0;
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        unsigned long long y;
        if (j >= 64) {
          __syncthreads();                       // every thread has read its partner of the previous LDS stage
          if (tid < P) keys[tid] = x;
          __syncthreads();
          y = tid < P ? keys[tid ^ j] : 0ull;
        } else {
          const unsigned lo = __shfl_xor((unsigned)(x & 0xFFFFFFFFull), j), hi = __shfl_xor((unsigned)(x >> 32), j);
          y = ((unsigned long long)hi << 32) | lo;
        }
        const bool lower = (tid & j) == 0;       // the lower index of the pair keeps the larger key in a descending run
        const bool want_max = desc == lower;
        x = want_max ? (x > y ? x : y) : (x < y ? x : y);
      }
    }
    if (tid < K) {
      si[tid] = (int)(~(unsigned)(x & 0xFFFFFFFFull));
      ss[tid] = __uint_as_float((unsigned)(x >> 32));
    }
    if (tid == 0) kp_store_count(a, b, K);
    return;
  }
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 1024) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long x = keys[i], y = keys[ixj];
          bool desc = (i & k2) == 0;
          if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < K; i += 1024) {
    unsigned long long kk = keys[i];
    si[i] = (int)(~(unsigned)(kk & 0xFFFFFFFFull));
    ss[i] = __uint_as_float((unsigned)(kk >> 32));
  }
  if (tid == 0) kp_store_count(a, b, K);
}

// ------------------------------------------------------------------ descriptors
// One wave per keypoint slot.  Channel c of lane handles c = lane, lane+64, ...
template <int MAXC>
__global__ __launch_bounds__(256) void describe_kernel(DescribeArgs a) {
  const int lane = threadIdx.x & 63;
  const long slot = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= (long)a.B * a.Kcap) return;
  const int bl = (int)(slot / a.Kcap), i = (int)(slot % a.Kcap);
  const int b = a.b0 + bl;                     // image index into dense / sel_* arrays
  const int cnt = a.sel_count[b];
  const bool second = bl >= a.split;
  const long oslot = second ? slot - (long)a.split * a.Kcap : slot;
  float* kp = (second ? a.kpts2 : a.kpts) + oslot * 2;
  float* dsc = (second ? a.desc2 : a.desc) + oslot * a.d;
  float* scp = (second ? a.scores2 : a.scores) + oslot;
  if (i >= cnt) {
    if (lane < 2) kp[lane] = 0.f;
    if (lane == 0) *scp = 0.f;
    for (int c = lane; c < a.d; c += 64) dsc[c] = 0.f;
    return;
  }
  const int idx = a.sel_idx[(size_t)b * a.Ksel + i];
  const int py = idx / a.W8, px = idx - py * a.W8;
  const float kx = (float)px, ky = (float)py;                                   // flip (y,x)->(x,y) (:151)
  if (lane == 0) { kp[0] = kx; kp[1] = ky; *scp = a.sel_score[(size_t)b * a.Ksel + i]; }

  const int w = a.Wc, h = a.Hc;
  // sample_descriptors (:43-46), s = 8
  float xn = ((kx - 4.0f) + 0.5f) / ((float)(w * 8) - 4.0f - 0.5f);
  float yn = ((ky - 4.0f) + 0.5f) / ((float)(h * 8) - 4.0f - 0.5f);
  xn = xn * 2.0f - 1.0f;
  yn = yn * 2.0f - 1.0f;
  // grid_sample un-normalisation (bilinear, zeros padding)
  float ix, iy;
  if (a.align_corners) {
    ix = ((xn + 1.0f) / 2.0f) * (float)(w - 1);
    iy = ((yn + 1.0f) / 2.0f) * (float)(h - 1);
  } else {
    ix = ((xn + 1.0f) * (float)w - 1.0f) / 2.0f;
    iy = ((yn + 1.0f) * (float)h - 1.0f) / 2.0f;
  }
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
  const float cw[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};            // nw, ne, sw, se
  const int cx[4] = {x0, x0 + 1, x0, x0 + 1}, cy[4] = {y0, y0, y0 + 1, y0 + 1};

  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (cx[q] < 0 || cx[q] >= w || cy[q] < 0 || cy[q] >= h) continue;          // zeros padding
    const float* cell = a.dense + ((size_t)(b * h + cy[q]) * w + cx[q]) * a.ld;
    float v[MAXC];
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      int c = lane + 64 * j;
      v[j] = c < a.d ? cell[c] : 0.f;
      s2 += v[j] * v[j];
    }
    float dn = sqrtf(wave_sum(s2));                                             // torch.norm(desc, dim=1) (:125)
    if (a.dense_eps) dn = fmaxf(dn, 1e-12f);                                    // official: F.normalize
#pragma unroll
    for (int j = 0; j < MAXC; ++j) acc[j] += (v[j] / dn) * cw[q];
  }
  float n2 = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) n2 += acc[j] * acc[j];
  const float denom = fmaxf(sqrtf(wave_sum(n2)), 1e-12f);                       // F.normalize (:50-51)
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    int c = lane + 64 * j;
    if (c < a.d) dsc[c] = acc[j] / denom;
  }
}

}  // namespace

hipError_t launch_dense_export(const float* semi, int ld, const float* dense, int d, float* semi_out, float* desc_out,
                               int B, int Hc, int Wc, int eps_mode, hipStream_t s) {
  long cells = (long)B * Hc * Wc;
  hipLaunchKernelGGL(dense_export_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, semi, ld, dense, d, semi_out,
                     desc_out, B, Hc, Wc, eps_mode);
  return hipGetLastError();
}

hipError_t launch_softmax_shuffle(const float* semi, int ld, float* scores, int B, int Hc, int Wc, hipStream_t s) {
  if ((Wc + 7) / 8 > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_shuffle_kernel, dim3((unsigned)(B * Hc), (unsigned)((Wc + 7) / 8)), dim3(512), 0, s, semi, ld, scores, B, Hc, Wc);
  return hipGetLastError();
}

size_t nms_scratch_bytes(int B, int H, int W, int radius) {
  if (radius >= 1 && radius <= 4) return (size_t)2 * B * H * ((W + 31) / 32) * sizeof(unsigned);   // two bit-row masks
  return (size_t)3 * B * H * W * sizeof(float);                                                    // row maxima, mask, supp_scores
}

bool nms_candidate_bits_supported(int radius, float threshold) { return radius >= 1 && radius <= 4 && threshold >= 0.f; }

// simple_nms + threshold + remove_borders as candidate bit rows (B, H, ceil(W/32) words at the head of `scratch`): kp_*_bits' input
hipError_t launch_nms_candidate_bits(const float* scores, int B, int H, int W, int radius, float threshold, int border, hipStream_t s, void* scratch) {
  if (!nms_candidate_bits_supported(radius, threshold) || !scratch) return hipErrorInvalidValue;
  last_form = "nms_staged_bits:hbm";
  unsigned* bits = static_cast<unsigned*>(scratch);
  switch (radius) {
    case 1: return launch_nms_staged<1>(scores, nullptr, bits, B, H, W, s, threshold, border);
    case 2: return launch_nms_staged<2>(scores, nullptr, bits, B, H, W, s, threshold, border);
    case 3: return launch_nms_staged<3>(scores, nullptr, bits, B, H, W, s, threshold, border);
    default: return launch_nms_staged<4>(scores, nullptr, bits, B, H, W, s, threshold, border);
  }
}

hipError_t launch_nms(const float* scores, float* out, int B, int H, int W, int radius, hipStream_t s, void* scratch) {
  if (!out) return hipErrorInvalidValue;
  if (radius < 0 || !scratch) return hipErrorInvalidValue;
  if (radius == 0) return hipMemcpyAsync(out, scores, (size_t)B * H * W * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (radius <= 4) {
    last_form = "nms_staged:hbm";
    unsigned* bits = static_cast<unsigned*>(scratch);
    switch (radius) {
      case 1: return launch_nms_staged<1>(scores, out, bits, B, H, W, s);
      case 2: return launch_nms_staged<2>(scores, out, bits, B, H, W, s);
      case 3: return launch_nms_staged<3>(scores, out, bits, B, H, W, s);
      default: return launch_nms_staged<4>(scores, out, bits, B, H, W, s);
    }
  }
  last_form = "nms_generic:hbm";
  const size_t n = (size_t)B * H * W;
  float* rp = static_cast<float*>(scratch);
  float* mask = rp + n;
  float* ss = mask + n;
  const dim3 grid((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)B), blk(256);
  auto pool = [&](const float* in, int mode) {
    hipLaunchKernelGGL(nms_pool_rows, grid, blk, 0, s, in, rp, H, W, radius);
    hipLaunchKernelGGL(nms_pool_cols, grid, blk, 0, s, (const float*)rp, scores, ss, mask, out, H, W, radius, mode);
  };
  pool(scores, NMS_MASK0);
  for (int round = 0; round < 2; ++round) {
    pool(mask, NMS_SUPP);
    pool(ss, round == 1 ? NMS_FINAL : NMS_ROUND);
  }
  return hipGetLastError();
}

hipError_t launch_keypoints(const KeypointArgs& a, hipStream_t s) {
  long rows = (long)a.B * a.H;
  if (a.cand_bits) {
    if (!a.scores) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kp_scan_rows_bits, dim3(a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(kp_scatter_rows_bits, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(kp_count_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(kp_scan_rows, dim3(a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(kp_scatter_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
  }
  int P = 1;
  if (a.max_keypoints > 0) { while (P < a.max_keypoints) P <<= 1; }
  if (P > 16384 && !a.sort_scratch) return hipErrorInvalidValue;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(kp_topk), 128 * 1024, attr);
  hipLaunchKernelGGL(kp_topk, dim3(a.B), dim3(1024), P <= 16384 ? (size_t)P * 8 : 0, s, a, P);
  return hipGetLastError();
}

hipError_t launch_describe(const DescribeArgs& a, hipStream_t s) {
  long slots = (long)a.B * a.Kcap;
  if (slots == 0) return hipSuccess;
  dim3 grid((unsigned)((slots + 3) / 4));
  if (a.d <= 128) hipLaunchKernelGGL(describe_kernel<2>, grid, dim3(256), 0, s, a);
  else if (a.d <= 256) hipLaunchKernelGGL(describe_kernel<4>, grid, dim3(256), 0, s, a);
  else if (a.d <= 512) hipLaunchKernelGGL(describe_kernel<8>, grid, dim3(256), 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace imx
