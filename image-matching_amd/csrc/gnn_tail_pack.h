// gnn_tail_pack.h -- host side of gnn_tail_x3.hip: the weights of one GNN layer tail (mlp.0' -> ReLU -> mlp.3 + residual -> the next
// layer's q|k|v, or final_proj) as ONE stream of LDS images, in the order the kernel consumes them (the kernel reads it as images of
// two k-steps = 24 KB, or of four = 48 KB with 8-wave workgroups: the stream is step-major, so both views are the same bytes).
//
// Every product of the kernel is TRANSPOSED: D[channel][row] = sum_k W^T[channel][k] . act^T[k][row] on v_mfma_f32_32x32x16_bf16, the
// weights as the A operand (lane (c, kb) holds eight k values of output channel c), the activations as the B operand (lane (row, kb)
// holds eight k values of its row).  A lane of the result holds, for ITS row, sixteen channels of a 32-channel block (register r <->
// channel (r & 3) + 8 (r >> 2) + 4 hi) -- which is exactly a B operand of the NEXT product if that product's k index is mapped to
// those channels: k-step (block b, half h2), lane half kb, element j  <->  channel 32 b + 16 h2 + (j & 3) + 8 (j >> 2) + 4 kb.
// So the hidden activations (and x') never leave the registers of the wave that owns the rows; only the weights are permuted, here.
//
// Stream of a layer (d = 128; every image 48 KB = [step][block of 32 output channels][plane (3)][lane (64)][8 bf16]):
//   for half in 0, 1:                               hidden channels 128 half .. 128 half + 127 (blocks 4 half .. 4 half + 3)
//     4 images  mlp.0', 4 k-steps each              k-step s = 4 i + t covers input k = 32 (s / 2) + 16 kb + 8 (s & 1) + j  ([x | att] order)
//     2 images  mlp.3,  4 k-steps each              k-steps (b, h2) of this half's hidden blocks, 4 output blocks
//   6 images    next product (q|k|v: 3 passes of 128 output channels; final_proj: 1 pass = 2 images), 4 k-steps (ob, h2) each
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace imx {

constexpr int GT_IMAGE_BYTES = 49152;     // 4 steps x 4 blocks x 3 planes x 1 KB

inline uint16_t gt_bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float gt_bf16_f(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
inline void gt_split(float x, uint16_t (&t)[3]) {
  t[0] = gt_bf16_rne(x);
  const float r1 = x - gt_bf16_f(t[0]);
  t[1] = gt_bf16_rne(r1);
  t[2] = gt_bf16_rne(r1 - gt_bf16_f(t[1]));
}

// w1 [2d][ld1] (k-major: row k = input channel of [x | att], merge folded), w2 [2d][ld2], w3 [d][ld3] with n3 = 3d or d output columns.
// Returns the stream as 16-bit patterns (images of GT_IMAGE_BYTES each).
inline std::vector<uint16_t> gnn_tail_pack(const float* w1, int ld1, const float* w2, int ld2, const float* w3, int ld3, int d, int n3) {
  const int per_image = GT_IMAGE_BYTES / 2;
  const int n_img = 2 * (4 + 2) + 2 * (n3 / d);
  std::vector<uint16_t> out((size_t)n_img * per_image, 0);
  size_t img = 0;
  auto put = [&](size_t image, int step, int block, int lane, int j, float v) {
    uint16_t t[3];
    gt_split(v, t);
    for (int q = 0; q < 3; ++q) out[image * per_image + ((((size_t)step * 4 + block) * 3 + q) * 64 + lane) * 8 + j] = t[q];
  };
  for (int half = 0; half < 2; ++half) {
    for (int i = 0; i < 4; ++i, ++img)               // mlp.0'
      for (int t = 0; t < 4; ++t) {
        const int s = 4 * i + t;
        for (int blk = 0; blk < 4; ++blk)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int k = 32 * (s / 2) + 16 * (lane >> 5) + 8 * (s & 1) + j, c = 32 * (4 * half + blk) + (lane & 31);
              put(img, t, blk, lane, j, w1[(size_t)k * ld1 + c]);
            }
      }
    for (int i = 0; i < 2; ++i, ++img)               // mlp.3: k-steps (b, h2) = (4 half + 2 i + t / 2, t & 1)
      for (int t = 0; t < 4; ++t) {
        const int b = 4 * half + 2 * i + t / 2, h2 = t & 1;
        for (int blk = 0; blk < 4; ++blk)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int k = 32 * b + 16 * h2 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), c = 32 * blk + (lane & 31);
              put(img, t, blk, lane, j, w2[(size_t)k * ld2 + c]);
            }
      }
  }
  for (int pass = 0; pass < n3 / d; ++pass)
    for (int i = 0; i < 2; ++i, ++img)               // next product: k-steps (ob, h2) = (2 i + t / 2, t & 1) over x' channels
      for (int t = 0; t < 4; ++t) {
        const int ob = 2 * i + t / 2, h2 = t & 1;
        for (int blk = 0; blk < 4; ++blk)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int k = 32 * ob + 16 * h2 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), c = d * pass + 32 * blk + (lane & 31);
              put(img, t, blk, lane, j, w3[(size_t)k * ld3 + c]);
            }
      }
  return out;
}

// gnn_tail_h2.hip: the same stream with every weight as TWO fp16 planes of w s, s = the power of two that brings its matrix's largest
// |value| to [2^13, 2^14); images of [step][block][plane (2)][lane][8 halves] (8 KB per k-step).  Also returns what the kernel's
// bounds need: the reciprocals of the three scales and the largest column L1 norm of w1 / w2 (a column = one output channel).
// loose_h / loose_x: estimates of how far gnn_tail_h2's BOUNDS of the hidden activations h and of x' sit above typical values --
// (largest column L1 norm) / (median column L2 norm) of mlp.0' (times the same ratio of mlp.3 for x'), times an activation crest
// factor of 2^4.  The bound is brought to 2^13 (pow2_of_bound); a value keeps both fp16 planes down to 2^-3, so beyond 2^16 the
// typical operand starts to lose its low plane and imx_api.cpp runs that layer's tail on three bf16 planes instead.
// w_spread (round 6): over the three matrices, (largest |w|) / (median over output columns of their largest |w|): each matrix carries ONE power of
// two, so a runaway column pushes the typical one towards fp16's low end (2^12: the typical weight still keeps the scheme's 22 bits)
struct GnnTailH2Consts { float w1_inv, w2_inv, w3_inv, l1_1, l1_2, loose_h, loose_x, w_spread; };
inline uint16_t gt_f16_rne(float x) {      // fp32 -> fp16 bit pattern, round to nearest even (normal range and subnormals; the scaled weights never overflow)
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (u < 0x38800000u) {
    if (u < 0x33000000u) return (uint16_t)sign;
    const int e = (int)(u >> 23);
    const uint32_t m = (u & 0x7fffffu) | 0x800000u;
    const int sh = 126 - e;
    const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t r = u + 0xfffu + ((u >> 13) & 1u);
  return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
inline float gt_f16_f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  float f;
  if (e == 0) { f = (float)m * 5.9604644775390625e-8f; memcpy(&u, &f, 4); u |= sign; }
  else u = sign | ((e + 112u) << 23) | (m << 13);
  memcpy(&f, &u, 4);
  return f;
}
inline std::vector<uint16_t> gnn_tail_pack_h2(const float* w1, int ld1, const float* w2, int ld2, const float* w3, int ld3, int d, int n3,
                                              GnnTailH2Consts* consts) {
  double spread = 1.0;
  auto scale_of = [&spread](const float* w, int rows, int cols, int ld, float* l1, double* gain_ratio = nullptr) {
    double mx = 0.0, best = 0.0;
    std::vector<double> l2(cols, 0.0), cmax(cols, 0.0);
    for (int c = 0; c < cols; ++c) {
      double acc = 0.0, sq = 0.0;
      for (int k = 0; k < rows; ++k) { const double a = std::fabs((double)w[(size_t)k * ld + c]); acc += a; sq += a * a; if (a > mx) mx = a; if (a > cmax[c]) cmax[c] = a; }
      if (acc > best) best = acc;
      l2[c] = std::sqrt(sq);
    }
    std::nth_element(cmax.begin(), cmax.begin() + cols / 2, cmax.end());
    spread = std::max(spread, cmax[cols / 2] > 0 ? mx / cmax[cols / 2] : (mx > 0 ? 1e30 : 1.0));
    if (l1) *l1 = (float)best;
    if (gain_ratio) {
      std::nth_element(l2.begin(), l2.begin() + cols / 2, l2.end());
      *gain_ratio = l2[cols / 2] > 0 ? best / l2[cols / 2] : 1e30;
    }
    int e = 0;
    if (mx > 0) std::frexp(mx, &e);
    return std::ldexp(1.0, 14 - e);
  };
  double g1 = 1.0, g2 = 1.0;
  const double s1 = scale_of(w1, 2 * d, 2 * d, ld1, &consts->l1_1, &g1), s2 = scale_of(w2, 2 * d, d, ld2, &consts->l1_2, &g2), s3 = scale_of(w3, d, n3, ld3, nullptr);
  consts->loose_h = (float)std::min(1e30, 16.0 * g1);
  consts->loose_x = (float)std::min(1e30, 16.0 * g1 * g2);
  consts->w1_inv = (float)(1.0 / s1); consts->w2_inv = (float)(1.0 / s2); consts->w3_inv = (float)(1.0 / s3);
  consts->w_spread = (float)std::min(1e30, spread);
  const int per_step = 4 * 2 * 64 * 8;                                  // 16-bit values per k-step
  const int n_step = 2 * (16 + 8) + 8 * (n3 / d);
  std::vector<uint16_t> out((size_t)n_step * per_step, 0);
  auto put = [&](size_t step, int block, int lane, int j, double v) {
    const float x = (float)v;
    const uint16_t h = gt_f16_rne(x), m = gt_f16_rne(x - gt_f16_f(h));
    out[step * per_step + (((size_t)block * 2 + 0) * 64 + lane) * 8 + j] = h;
    out[step * per_step + (((size_t)block * 2 + 1) * 64 + lane) * 8 + j] = m;
  };
  size_t step = 0;
  for (int half = 0; half < 2; ++half) {
    for (int s = 0; s < 16; ++s, ++step)               // mlp.0': k-step s covers input k = 32 (s / 2) + 16 kb + 8 (s & 1) + j  ([x | att] order)
      for (int blk = 0; blk < 4; ++blk)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * (s / 2) + 16 * (lane >> 5) + 8 * (s & 1) + j, c = 32 * (4 * half + blk) + (lane & 31);
            put(step, blk, lane, j, (double)w1[(size_t)k * ld1 + c] * s1);
          }
    for (int s = 0; s < 8; ++s, ++step) {              // mlp.3: k-steps (b, h2) of this half's hidden blocks
      const int b = 4 * half + s / 2, h2 = s & 1;
      for (int blk = 0; blk < 4; ++blk)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * b + 16 * h2 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), c = 32 * blk + (lane & 31);
            put(step, blk, lane, j, (double)w2[(size_t)k * ld2 + c] * s2);
          }
    }
  }
  for (int pass = 0; pass < n3 / d; ++pass)
    for (int s = 0; s < 8; ++s, ++step) {              // next product: k-steps (ob, h2) over x' channels
      const int ob = s / 2, h2 = s & 1;
      for (int blk = 0; blk < 4; ++blk)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * ob + 16 * h2 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), c = d * pass + 32 * blk + (lane & 31);
            put(step, blk, lane, j, (double)w3[(size_t)k * ld3 + c] * s3);
          }
    }
  return out;
}

}  // namespace imx
