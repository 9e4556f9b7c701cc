// imx_kernels.h — internal launch interface between the C-ABI host code (imx_api.cpp) and the
// gfx950 kernels.  Activations are fp32, channels-last: images (B,H,W,C) "NHWC", keypoint
// features (rows, C).  See DESIGN.md for the data layout in HBM and per-kernel rooflines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace imx {

// Kernel-form options of a handle (imx_set_option; environment variables only seed the defaults at imx_create).  Launchers take
// what they need from here instead of reading the environment on every launch.
struct Options {
  int mfma_f32 = 0;        // "mfma": 0 = "x3" (fp32 products as six bf16 term products on the bf16 matrix pipe where a kernel has
                           //         that form), 1 = "f32" (fp32 MFMA everywhere: the A/B reference of the parity tests)
  int latency_forms = -1;  // "latency_forms": -1 = "auto" (gemm_small for M <= 4096 rows, key-split attention for grids of
                           //         <= 256 workgroups, one launch per GNN layer tail), 0 = "off" (results do not depend on the batch size
                           //         bit for bit), 1 = "on", 2 = "unfused" (on, with the layer tail as three gemm_small launches: the A/B of the fusion)
  int gnn_tail = -1;       // "gnn_tail": -1 = "auto" = 1 = "fused" (one launch per layer tail wherever the throughput forms run, d = 128: gnn_tail_h2 --
                           //         three fp16 plane products -- beside the two-plane attention, else gnn_tail_x3), 2 = "bf16x3" (gnn_tail_x3: six bf16
                           //         plane products), 0 = "unfused" (three gemm_x3 launches: the A/B of the fusion)
  int attention = -1;      // "attention": -1 = "auto" = 1 = "f16x2" (attention_x3.hip with two fp16 planes per operand, three term products: needs
                           //         the q / k / v maxima, which the fused layer tail writes), 0 = "bf16x3" (three bf16 planes, six term products)
  int conv_direct = 0;     // "conv": 0 = "wino" / "wino32" (Winograd F(2x4,3x3); direct only for shapes it rejects), 1 = "direct"
  int conv_f16 = 1;        //         (2 = "wino_h": conv3x3_wino24h for every layer, never the tile-pair form conv3x3_wino24p)  "wino" (default): the layers after the first run the Winograd products on the fp16 matrix pipe (two planes per
                           //         transformed operand, conv3x3_wino24h.hip; needs "mfma" = "x3"); "wino32": every product on the fp32 MFMA
  // A/B switches of the bit-identity tests and of tools/ (round 6: handle options; environment variables of these names are not read)
  int conv_swizzle = 1;    // "conv_swizzle": 1 = "on" (the tensor between two pair-form layers without a pool is tile-swizzled), 0 = "off" (blocked)
  int qkv_amax = 0;        // "qkv_amax": 0 = "epilogue" (a plain q|k|v projection's epilogue writes the (side, pair) maxima), 1 = "kernel" (the separate pass)
  int sinkhorn_group = 0;  // "sinkhorn_group": 0 = "auto" (2 slabs per workgroup up to 1024 columns, 4 above, 1 below 64 slabs), 1 | 2 | 4
  int sinkhorn_prefetch = -1;  // "sinkhorn_prefetch": -1 = "auto" (= off since round 6), 0 = "off", 1 = "on"
  int sinkhorn_merge = -1;     // "sinkhorn_merge": -1 = "auto" (= kernel), 0 = "kernel" (sinkhorn_vmerge, a second launch per iteration), 1 = "fused" (the last-arriving slab workgroups of a pair merge its column partials)
  int keypoints = -1;          // "keypoints": -1 = "auto" (candidate bit rows where the NMS is the staged form and the threshold >= 0), 0 = "dense" (the NMS score map, three passes), 1 = "bits"
  int attention_qblocks = -1;  // "attention_qblocks": -1 = "auto" (2 where the padded keypoint count is a multiple of 256 and one block per wave would
                               //         still leave >= 1024 workgroups -- two per slot of the chip --, else 1) | 1 | 2: 32-query blocks per wave of the
                               //         two-plane attention at head dim 32 (attention_h2q2_kernel; bit-identical results)
  int linear = -1;         // "linear": -1 = "auto" = 1 = "f16x2" (gemm_h2: the GNN's plain linear layers -- q|k|v, mlp.0', mlp.3, final_proj where the
                           //         layer tail is not fused -- as three fp16 plane products, scaled by the operands' actual (side, pair) maxima; needs
                           //         "attention" = f16x2 and weights inside the spread guard), 0 = "bf16x3" (gemm_x3: six bf16 plane products)
};

// The form a launcher picked ("gemm_x3:bf16x3", "conv3x3_wino24:f32", ...: kernel family, then the matrix pipe it runs on or
// "hbm" for streaming kernels).  Set by every launch_* that has more than one form, read by imx_api.cpp's per-launch timing so
// that imx_timing_form reports what actually ran.
extern thread_local const char* last_form;

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: `done_mask` (one static per kernel
// instantiation at the call site) remembers the devices it has been raised on, so a process holding handles on two devices
// raises it on both.
// Two host threads driving handles on different devices may arrive here together: the mask is read and updated atomically, and
// the bit is set only AFTER the attribute call, so a racing thread at worst repeats the (idempotent) call; devices past 63 have no
// bit and set the attribute on every launch.
inline void raise_lds_limit(const void* kern, int bytes, unsigned long long& done_mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
  if (bit && (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit)) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (bit) __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
}

// ---------------------------------------------------------------- conv3x3 (MFMA fp32, implicit GEMM)
struct ConvArgs {
  const float* in;    // NHWC (B,H,W,Cin); FIRST mode: grayscale images (B,H,W), images >= split come from in2
  const float* in2;
  int split;
  const float* w;     // [9][Cin][Cout]  (BN folded)
  const float* wu24;  // Winograd F(2x4,3x3) weights G2 g G4^T: [Cout/64][Cin/8][12 quads][4 co-blocks][64 lanes][4] (conv1ab_wino24.hip, conv3x3_wino24.hip)
  const float* bias;  // [Cout]
  const float* w1;    // FIRST mode: conv1a weights [9][64] and bias [64] (BN folded), Cin == 64
  const float* b1;
  float* out;         // NHWC (B,Ho,Wo,Cout); Ho,Wo = H/2,W/2 (floor) when pool
  int B, H, W, Cin, Cout;
  int relu, pool, first;
  // Activation layout between the Winograd layers: 0 = NHWC (B,H,W,C); 1 = channel-blocked (B, C/8, H, W, 8) -- a chunk of eight
  // channels of consecutive pixels is contiguous, which is what the Winograd kernels' per-chunk patch loads read (conv3x3_wino24.hip).
  // The direct kernel (conv3x3.hip) and the 1x1-conv GEMMs take NHWC only.
  int in_blocked, out_blocked;
  // conv3x3_wino24h.hip (Winograd with both transformed operands as two fp16 planes): U planes [Cout/64][Cin/32][24][2][4][64][8] halves,
  // 1 / (the power of two U was scaled by), and one word per image with the bit pattern of an upper bound of |input| (amax_in; written
  // by the producing layer's epilogue).  amax_out (any Winograd kernel, optional): [B] zeroed words that receive this layer's maxima.
  const void* wuh;
  float u_scale_inv;
  const unsigned* amax_in;
  unsigned* amax_out;
  // conv1ab_wino24h.hip: the largest column L1 norm of the folded conv1a weights and the largest |bias| (bound of conv1a's outputs per unit of |image|)
  float c1a_l1, c1a_bmax;
};
// Cin % 16 == 0, Cout % 64 == 0.
hipError_t launch_conv3x3(const ConvArgs& a, hipStream_t s);       // direct form (conv3x3.hip)
// first && pool, Cin == Cout == 64.
hipError_t launch_conv1ab_wino24(const ConvArgs& a, hipStream_t s); // fused conv1a + conv1b (Winograd F(2x4,3x3)) + pool
bool conv1ab_wino24h_supported(const ConvArgs& a);                   // + wuh, c1a_l1
hipError_t launch_conv1ab_wino24h(const ConvArgs& a, hipStream_t s); // the same with conv1b's products on the fp16 matrix pipe
bool conv1ab_wino24p_supported(const ConvArgs& a);                   // = conv1ab_wino24h_supported
bool conv1ab_wino24p_preferred(const ConvArgs& a);                   // supported and at least two tile pairs per CU
hipError_t launch_conv1ab_wino24p(const ConvArgs& a, hipStream_t s); // conv1ab_wino24h's arithmetic on tile pairs, positions split over two waves (8 waves per CU)
// Cin % 64 == 0, Cout % 64 == 0, not first.
bool conv3x3_wino24_supported(const ConvArgs& a);                   // false (e.g. an image of >= 2 GB per layer): callers fall back to the direct form
hipError_t launch_conv3x3_wino24(const ConvArgs& a, hipStream_t s); // Winograd F(2x4,3x3), persistent, in-stream transform
bool conv3x3_wino24h_supported(const ConvArgs& a);                  // + wuh, amax_in
hipError_t launch_conv3x3_wino24h(const ConvArgs& a, hipStream_t s); // the same on the fp16 matrix pipe: two planes per operand, three plane products
bool conv3x3_wino24p_supported(const ConvArgs& a);                  // = conv3x3_wino24h_supported
bool conv3x3_wino24p_preferred(const ConvArgs& a);                  // supported and at least one item (tile pair x 64 channels) per CU
hipError_t launch_conv3x3_wino24p(const ConvArgs& a, hipStream_t s); // conv3x3_wino24h's arithmetic: tile pairs, positions split over two waves (8 waves per CU)


// ---------------------------------------------------------------- GEMM (MFMA fp32): 1x1 conv / linear
// out[r][n] = act( sum_k A[r][k] * W[k][n] + bias[n] ) (+ res[r][n]);  A = [a0 | a1] column concat.
struct GemmArgs {
  const float* a0; int lda0; int K0;   // rows x K0
  const float* a1; int lda1; int K1;   // rows x K1 (may be null / 0)
  const float* w;                      // [K0+K1][Npad] row-major, Npad = ceil(N/64)*64, zero padded
  const float* bias;                   // [Npad]
  const float* res; int ldr;           // optional residual (may alias out)
  float* out; int ldo;
  int M, N, Npad;
  int relu;
  // optional (gemm_x3 only; the q|k|v projection): max |out| over the VALID rows of every (side, pair), by column third (q | k | v),
  // as bit patterns through atomicMax into amax[(side B + pair) 4 + third] -- what launch_qkv_amax computes from the stored rows, out
  // of the epilogue's registers.  Rows = [side 0: aB pairs x aN0p][side 1: aB x aN1p], counts an0 / an1 (null: aN0 / aN1);
  // aN0p, aN1p % 128 == 0 and N == 3 d with d % 32 == 0 (gemm_x3_amax_supported)
  unsigned* amax;
  const int* an0; const int* an1;
  int aB, aN0p, aN1p, aN0, aN1;
  // gemm_h2 only (two fp16 planes per operand; gemm_h2.hip).  The A operand of a 128-row tile is scaled by the power of two that brings
  // max(sa0[sp sa0_stride + sa0_off], sa1[sp' sa1_stride + sa1_off]) to 2^13: words (bit patterns of max |value| over the valid rows
  // of (side, pair) sp = side aB + pair) written by the kernels that produced a0 / a1 (sa1 null: a0 alone; sa1_cross: sp' is the pair
  // on the OTHER side -- the attention output of a cross layer is bounded by the key side's max |v|).  amax_row (optional): max |out|
  // over the valid rows of each (side, pair), all columns, into amax_row[sp amax_row_stride + amax_row_off].  w_inv: 1 / (the power of
  // two the fp16 weight planes were scaled by).  Rows as for `amax`: aB, aN0p, aN1p (% 128 == 0), an0 / an1, aN0 / aN1.
  const unsigned* sa0; int sa0_stride, sa0_off;
  const unsigned* sa1; int sa1_stride, sa1_off, sa1_cross;
  unsigned* amax_row; int amax_row_stride, amax_row_off;
  float w_inv;
  unsigned long long* trace = nullptr;   // developer instrumentation (a -DGH2_TRACE build of gemm_h2.hip only): s_memtime stamps of the first workgroups' chunks
};
bool gemm_x3_amax_supported(const GemmArgs& a);
// wh2 = the weights as two fp16 planes of w s in B-fragment order [Npad/32][K/16][2][64][8] (imx_api.cpp: split_f16x2)
bool gemm_h2_supported(const GemmArgs& a);
hipError_t launch_gemm_h2(const GemmArgs& a, const void* wh2, hipStream_t s);
// max |x| over the valid rows of every (side, pair), rows of d floats (d % 4 == 0): what gnn_tail_h2 / gemm_h2 scale layer 0's x by
hipError_t launch_rows_amax_any(const float* x, int d, int B, int N0p, int N1p, const int* n0, const int* n1, int N0, int N1, unsigned* amax, hipStream_t s);
// (K0, K1) % 32 == 0.
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
// The tail of one GNN layer for small row counts, fused (gnn_small.hip): hidden = relu([x | att] w1 + b1); x += hidden w2 + b2;
// out = x w3 + b3 (the next layer's q|k|v, n3 = 3 d, or final_proj, n3 = d).  Weights unpadded, in B-fragment order
// [K/16][4][N][4] (imx_api.cpp:fragment_order).
struct GnnSmallArgs {
  float* x;                             // [M][d], updated in place
  const float* att;                     // [M][d]
  const float* w1; const float* b1;     // [2d][2d]
  const float* w2; const float* b2;     // [2d][d]
  const float* w3; const float* b3;     // [d][n3]
  float* out;                           // [M][n3]
  int M, d, n3;
};
bool gnn_layer_small_supported(const GnnSmallArgs& a);
hipError_t launch_gnn_layer_small(const GnnSmallArgs& a, hipStream_t s);
// fp32 products as six bf16 term products on the bf16 matrix pipe (gemm_x3.hip): wx3 = the weights as three bf16 terms per value in
// B-fragment order [Npad/32][K/16][3][64][8]; (K0, K1) % 32 == 0, Npad % 64 == 0, float4-aligned leading dimensions
bool gemm_x3_supported(const GemmArgs& a);
hipError_t launch_gemm_x3(const GemmArgs& a, const void* wx3, hipStream_t s);
// latency form for small row counts (gemm_small.hip): 32 x 32 output tiles, whole-K panels staged in LDS; K % 64 == 0, K <= 512
bool gemm_small_supported(const GemmArgs& a);
hipError_t launch_gemm_small(const GemmArgs& a, hipStream_t s);

// scores[b][i][j] = scale * sum_k m0[b][i][k] * m1[b][j][k]   ("NT" GEMM, per pair)
struct ScoreArgs {
  const float* m0; const float* m1;    // rows (b*N0p + i) / (b*N1p + j), leading dim = d
  float* out;                          // (B, N0p, N1p)
  int B, N0p, N1p, d;
  float scale;
};
hipError_t launch_score_gemm(const ScoreArgs& a, hipStream_t s);

// ---------------------------------------------------------------- SuperPoint tail
// softmax over 65 channels + drop dustbin + 8x8 pixel shuffle.  semi: (B,Hc,Wc,ld) -> scores (B,8Hc,8Wc)
hipError_t launch_softmax_shuffle(const float* semi, int ld, float* scores, int B, int Hc, int Wc, hipStream_t s);
// channels-last semi / raw descriptors -> reference (B,C,Hc,Wc) tensors, descriptors divided by their channel norm
hipError_t launch_dense_export(const float* semi, int ld, const float* dense, int d, float* semi_out, float* desc_out,
                               int B, int Hc, int Wc, int eps_mode, hipStream_t s);
// simple_nms (three rounds, any radius >= 0); out = where(max_mask, scores, 0).  scratch: nms_scratch_bytes(...) bytes
// (radius 1..4: two bit-row masks for the staged three-kernel form; otherwise three float maps for the generic passes)
size_t nms_scratch_bytes(int B, int H, int W, int radius);
hipError_t launch_nms(const float* scores, float* out, int B, int H, int W, int radius, hipStream_t s, void* scratch);
// the detector's form (radius 1..4, threshold >= 0): NMS + threshold + remove_borders as bit rows at the head of `scratch` (KeypointArgs::cand_bits)
bool nms_candidate_bits_supported(int radius, float threshold);
hipError_t launch_nms_candidate_bits(const float* scores, int B, int H, int W, int radius, float threshold, int border, hipStream_t s, void* scratch);
// threshold + border removal + row-major compaction, then top-k (score desc, index asc on ties).
struct KeypointArgs {
  const float* nms;          // (B,H,W) where(max_mask, scores, 0); or null with:
  const unsigned* cand_bits; // (B,H,ceil(W/32)) candidate bit rows of launch_nms_candidate_bits (NMS + threshold + border), and
  const float* scores;       // (B,H,W) the score map they index (round 6)
  int B, H, W;
  float threshold; int border; int max_keypoints;   // -1 = keep all
  int* row_count;            // (B,H) scratch
  int* row_off;              // (B,H) scratch
  int* cand_count;           // (B)   number of candidates after threshold+border
  int* cand_idx;             // (B,H*W) linear index y*W+x, row-major order
  float* cand_score;         // (B,H*W)
  int* sel_count;            // (B)   kept keypoints
  int* counts_out[3];        // optional copies of sel_count written by the same kernel (a 4-byte device-to-device copy is a ~4 us
  int counts_split;          //   stream operation): [0] all B; [1] images b < counts_split; [2] images b >= counts_split, rebased
  int* sel_idx;              // (B,Ksel)  Ksel = max_keypoints >=0 ? max_keypoints : H*W
  float* sel_score;          // (B,Ksel)
  int Ksel;
  unsigned long long* sort_scratch;   // (B, pow2 >= max_keypoints) sort slots when max_keypoints > 16384, else may be null
};
hipError_t launch_keypoints(const KeypointArgs& a, hipStream_t s);
// flip to (x,y), bilinear descriptor sampling from the raw dense descriptor map (B,Hc,Wc,ld),
// dense channel-L2 normalisation applied on the fly, final L2 normalise (eps 1e-12).
struct DescribeArgs {
  const float* dense; int ld; int d; int Hc, Wc; int W8;   // W8 = 8*Wc (score-map width for idx decode)
  const int* sel_count; const int* sel_idx; const float* sel_score; int Ksel;
  int b0;                                    // first image (index into dense / sel_*) of this call
  int B, Kcap;
  float* kpts; float* scores; float* desc;   // (B,Kcap,2) (B,Kcap) (B,Kcap,d); rows >= count zeroed
  float* kpts2; float* scores2; float* desc2; int split;   // images >= split (rebased) go to this second set: both sides of
                                             // imx_match_pairs in one launch; split = B and nulls otherwise
  int align_corners; int dense_eps;          // dense_eps: 1 = F.normalize(eps 1e-12), 0 = plain division
};
hipError_t launch_describe(const DescribeArgs& a, hipStream_t s);

// ---------------------------------------------------------------- SuperGlue pieces
// normalize_keypoints + first kenc layer (3 -> C1, BN folded, ReLU); rows laid out side-major.
struct Kenc0Args {
  const float* kpts; const float* scores;  // (B,N,2), (B,N)
  int B, N, Np;                            // Np = padded rows per image in the internal layout
  float cx, cy, scaling;                   // center and scaling (superglue_test.py:63-70)
  const float* w; const float* bias;       // [3][C1], [C1]
  int C1;
  float* out;                              // rows (b*Np + i), ld = C1; rows >= N zeroed
};
// SuperGlue's prologue for BOTH sides in one launch: the descriptor gather and the first keypoint-encoder layer (four launches of
// ~4.5 us each on the single-pair path).  Same per-element arithmetic as launch_gather_desc / launch_kenc0.
struct SgPrologueArgs {
  const float* desc[2]; long sb[2], sc[2], sn[2];   // descriptors (arbitrary strides) per side
  float* xrow[2];                                    // rows (b*Np + i), ld = d
  Kenc0Args k[2];                                    // B, N, Np per side in here
  int d;
};
hipError_t launch_sg_prologue(const SgPrologueArgs& a, hipStream_t s);
// copy descriptors (arbitrary strides) into rows (b*Np+i), ld=d; rows >= N zero.
hipError_t launch_gather_desc(const float* src, int64_t sb, int64_t sc, int64_t sn, int B, int N, int Np, int d,
                              float* out, hipStream_t s);

// flash-style multi-head attention, fp32 MFMA.  qkv rows: [side0: B*N0p rows][side1: B*N1p rows],
// ld = 3*d, columns [q | k | v], each head-major (head*32 + dim).  out same rows, ld = d.
struct AttnArgs {
  const float* qkv; float* out;
  int B, N0p, N1p, d, heads;     // head dim fixed at 32
  const int* n0; const int* n1;  // valid counts per pair (device), may be null => N0/N1
  int N0, N1;
  int cross;
  int mfma_f32;                  // Options::mfma_f32
  int latency_forms;             // Options::latency_forms
  const unsigned* amax;          // [2 B][4]: bit patterns of max |q|, |k|, |v| over the valid rows of (side, pair) = (s, b) at [s B + b] (launch_qkv_amax,
                                 // or the producing gnn_tail_x3); non-null selects the two-plane fp16 form of attention_x3.hip (three term
                                 // products instead of six), null the bf16 x 3 form
  int qblocks;                   // Options::attention_qblocks: 2 = two 32-query blocks per wave in the two-plane head-dim-32 kernel, 1 = one, -1 / 0 = auto
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
hipError_t launch_qkv_amax(const AttnArgs& a, unsigned* amax, hipStream_t s);
bool attention_takes_x3(const AttnArgs& a);      // launch_attention would run attention_x3.hip's kernels for these arguments
// The tail of one GNN layer of the throughput path in one launch (gnn_tail_x3.hip): hidden = relu([x | att] W1' + b1); x += hidden W2 + b2;
// out = x W3 + b3 (the next layer's q|k|v, n3 = 3 d, or final_proj, n3 = d) -- six bf16 term products per fp32 product; d = 128.
// `stream` = gnn_tail_pack() of the three weight matrices (gnn_tail_pack.h).
struct GnnTailArgs {
  float* x;                  // [M][d], updated in place
  const float* att;          // [M][d]
  const void* stream;        // weight images, consumption order
  const float* b1; const float* b2; const float* b3;     // [2d], [d], [n3]
  float* out;                // [M][n3]
  int M, d, n3;
  // n3 = 3 d only, optional: the maxima of |q|, |k|, |v| over the VALID rows of every (side, pair) of `out`, for the next layer's
  // two-plane fp16 attention (AttnArgs::amax: [2 B][4] zeroed words, bit patterns through atomicMax).  Row layout of the SuperGlue
  // workspace: side 0 pairs b = 0 .. B-1 (N0p rows each, n0[b] or N0 valid), then side 1 (N1p rows each).
  unsigned* amax;
  const int* n0; const int* n1;
  int B, N0p, N1p, N0, N1;
  // gnn_tail_h2.hip (three fp16 plane products of two-plane operands): the weights as gnn_tail_pack_h2() wrote them, the reciprocals of the
  // powers of two they were scaled by, the largest column L1 norms / |bias| of mlp.0' and mlp.3 (bounds of the hidden activations and
  // of x'), the (side, pair) maxima of this layer's x (amax_x_in, [2 B] bit patterns) and of its attention's v (amax_v: the [2 B][4]
  // table AttnArgs::amax of the same layer), whether that attention was a cross layer, and where the maxima of x' go (amax_x_out,
  // [2 B] zeroed words, or null)
  const void* stream_h2;
  float w1_inv, w2_inv, w3_inv, l1_1, bmax_1, l1_2, bmax_2;
  const unsigned* amax_x_in; const unsigned* amax_v; unsigned* amax_x_out;
  int cross;
};
bool gnn_tail_x3_supported(const GnnTailArgs& a);
hipError_t launch_gnn_tail_x3(const GnnTailArgs& a, hipStream_t s);
bool gnn_tail_h2_supported(const GnnTailArgs& a);
hipError_t launch_gnn_tail_h2(const GnnTailArgs& a, hipStream_t s);
// max |x| over the valid rows of every (side, pair) of a [B (N0p + N1p)][d] tensor -> amax[2 B] (zeroed words), d = 128: layer 0's amax_x_in

// both products as six bf16 term products on the bf16 matrix pipe (attention_x3.hip): head dim 32 or 64
bool attention_x3_supported(const AttnArgs& a);
hipError_t launch_attention_x3(const AttnArgs& a, hipStream_t s);

// log-domain Sinkhorn with implicit dustbins (superglue_test.py:141-170).
struct SinkhornArgs {
  const float* S;            // (B,N0p,N1p) scores
  float* u; float* v;        // (B,N0p+1) / (B,N1p+1); entries [n0] / [n1] hold the dustbin terms
  int B, N0p, N1p;
  const int* n0; const int* n1; int N0, N1;
  float alpha;               // bin_score
  int iters;
  float* part;               // scratch (B, N0p/R + 1, N1p + 1, 2) for the slab form, R = sinkhorn_slab_rows(N1p); may be null
  int group = 0;             // Options::sinkhorn_group (0 = auto)
  int prefetch = -1;         // Options::sinkhorn_prefetch (-1 = auto)
  unsigned* merge_cnt = nullptr;   // (B + 1) words, or null: with it the slab kernel merges its own partials ("sinkhorn_merge" = fused, round 6) --
                                   // per-pair arrival counters over the iterations of one launch_sinkhorn (zeroed by it), word [B] = a spin that gave up
  int it = 0;                      // (set by launch_sinkhorn: the iteration a launch belongs to)
  unsigned long long* trace = nullptr;   // developer instrumentation (a -DSK_TRACE build of sg_misc.hip only; tools/sinkhorn_trace.py): eight 64-bit words per
                                         // slab-kernel workgroup of the LAST iteration -- s_memrealtime at entry / after each slab / at exit, HW_ID, XCC_ID
};
int sinkhorn_slab_rows(int N1p);
hipError_t launch_sinkhorn(const SinkhornArgs& a, hipStream_t s);

struct MatchArgs {
  const float* S; const float* u; const float* v;
  int B, N0p, N1p; const int* n0; const int* n1; int N0, N1;
  float alpha; float threshold;
  float* max0; int* idx0; float* max1; int* idx1;   // scratch (B,N0p) / (B,N1p)
  int64_t* matches0; int64_t* matches1; float* ms0; float* ms1;  // (B,N0) / (B,N1)
};
hipError_t launch_matches(const MatchArgs& a, hipStream_t s);

// fixed-size match records of the multi-GPU gather (one row of 3 + 8K 32-bit words per pair; image-matching_amd/shard.py)
struct PackArgs {
  const int* pair_ids;                      // (B) global pair ids
  const float* kpts0; const float* kpts1;   // (B,K,2)
  const int* counts0; const int* counts1;   // (B)
  const long long* matches0; const long long* matches1;   // (B,K)
  const float* ms0; const float* ms1;       // (B,K)
  int* rec;                                 // (rows, 3 + 8K); rows >= B, rows past B are padding (pair id -1)
  int B, K, rows;
};
hipError_t launch_pack_records(const PackArgs& a, hipStream_t s);

// ---------------------------------------------------------------- registration post-step
// RANSAC 4-DoF similarity ("partial affine") on the matched keypoints of each pair.
struct RansacArgs {
  const float* kpts0; const float* kpts1;   // (B,K,2) (x,y)
  const long long* matches0;                // (B,K) index into kpts1 or -1
  const int* counts0;                       // (B) valid keypoints per pair, may be null
  int B, K;
  float threshold; int hypotheses; unsigned seed;
  float* M;                                 // (B,2,3)
  unsigned char* inlier;                    // (B,K) 1 = matched and inlier of the RANSAC model
  int* n_inliers;                           // (B) 0 = no fit (fewer than 4 matches)
  float* scratch;                           // (B,4,K) compacted coordinates when K > 8192 (they live in LDS up to that), else may be null
};
hipError_t launch_ransac(const RansacArgs& a, hipStream_t s);

// ingest / warp byte kernels (ingest.hip)
hipError_t launch_resize_u8_unit(const uint8_t* src, long sstride, int B, int Hs, int Ws, float* dst, int H, int W, hipStream_t s);
hipError_t launch_warp_affine_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int H, int W, const double* Minv, hipStream_t s);

// exact 2-NN + ratio test on descriptor rows (dots from launch_score_gemm with scale 1)
struct KnnArgs {
  const float* dots;                 // (B,N0p,N1p)
  const float* norm0; const float* norm1;   // squared row norms (B,N0p) / (B,N1p)
  int B, N0, N1, N0p, N1p;
  const int* n0; const int* n1;      // valid counts per pair, may be null
  float ratio;
  long long* matches;                // (B,N0) nearest index if dist1 < ratio*dist2 else -1
  float* dist1; float* dist2;        // (B,N0)
};
hipError_t launch_rownorm2(const float* x, int d, long rows, float* out, hipStream_t s);
hipError_t launch_knn2(const KnnArgs& a, hipStream_t s);

}  // namespace imx
