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

}  // namespace imx
