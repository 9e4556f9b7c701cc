/* imx.h — C ABI of libimx.so: MI355X-native SuperPoint + SuperGlue inference hot path.
 *
 * The reference (PH8411/image-matching) is pure Python and has no FFI of its own; the
 * boundary it exposes for this path is the nn.Module API
 *     SuperPoint.forward   superpoint/models/superpoint_test.py:103-161
 *                          (official variant superglue/models/superpoint.py:145-202)
 *     SuperGlue.forward    superglue/models/superglue_test.py:230-285
 *     Matching.forward     superglue/models/matching_test.py:54-82
 * Each entry point below names the reference call it replaces.  The Python drop-in classes
 * (image-matching_amd/superpoint, image-matching_amd/superglue) bind these through ctypes;
 * INTEGRATION.md shows the binding a maintainer would add to the reference.
 *
 * Conventions
 *   - every call returns int: 0 = ok, <0 = error (imx_last_error gives the text); nothing
 *     throws or aborts across the ABI;
 *   - `*_dev` pointers are device (HBM) pointers owned by the CALLER (e.g. torch tensors);
 *     the library borrows them for the duration of the call only;
 *   - work is enqueued asynchronously on the caller's HIP stream (`stream`, a hipStream_t
 *     passed as void*; NULL = the default stream); results are ready when the stream is;
 *   - workspace and weights are owned by the handle; a handle is bound to one device and is
 *     not thread-safe (one host thread per handle/GPU); handles are independent;
 *   - all tensors are fp32 unless stated; images are (B,1,H,W) contiguous in [0,1].
 */
#ifndef IMX_H
#define IMX_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols (tests/test_host.py checks nm -D). */
#if defined(__GNUC__) || defined(__clang__)
#define IMX_API __attribute__((visibility("default")))
#else
#define IMX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define IMX_MAX_GNN_LAYERS 64
#define IMX_MAX_KENC 8

#define IMX_NET_SUPERPOINT 0
#define IMX_NET_SUPERGLUE 1

#define IMX_SP_VARIANT_BN 0        /* superpoint/models/superpoint_test.py (BatchNorm, desc/‖desc‖ no eps) */
#define IMX_SP_VARIANT_OFFICIAL 1  /* superglue/models/superpoint.py (no BN, F.normalize eps 1e-12)       */

typedef struct imx_handle_s* imx_handle_t;

/* Mirrors SuperPoint.default_config (superpoint_test.py:57-63) and SuperGlue.default_config
 * (superglue_test.py:195-202), merged with the user's config by the Python classes. */
typedef struct imx_config {
  /* SuperPoint */
  int32_t descriptor_dim;      /* 'descriptor_dim': SuperPoint alone: any multiple of 4 up to 512; with SuperGlue 64, 128 or 256
                                  (4 heads of 16/32/64 dims) -- checked when the SuperGlue weights are finalized             */
  int32_t nms_radius;          /* 'nms_radius' (any >= 0; 1..4 take the fast staged kernels)                                  */
  float keypoint_threshold;    /* 'keypoint_threshold'                                  */
  int32_t max_keypoints;       /* 'max_keypoints' (-1 = keep all; any value, above 16384 the top-k sort runs out of HBM)      */
  int32_t remove_borders;      /* 'remove_borders'                                      */
  int32_t align_corners;       /* grid_sample mode chosen at superpoint_test.py:47      */
  int32_t sp_variant;          /* IMX_SP_VARIANT_*                                      */
  /* SuperGlue */
  int32_t num_gnn_layers;                       /* len('GNN_layers')                    */
  int32_t gnn_layer_is_cross[IMX_MAX_GNN_LAYERS]; /* 0 = 'self', 1 = 'cross'            */
  int32_t kenc_n;                               /* len('keypoint_encoder')              */
  int32_t kenc_channels[IMX_MAX_KENC];          /* 'keypoint_encoder'                   */
  int32_t sinkhorn_iterations;                  /* 'sinkhorn_iterations'                */
  float match_threshold;                        /* 'match_threshold'                    */
} imx_config_t;

/* Replaces Matching.__init__ / `.to(device)` (matching_test.py:49-52, superpoint_glue_test.py:69). */
IMX_API int imx_create(int device_id, const imx_config_t* cfg, imx_handle_t* out);
IMX_API int imx_destroy(imx_handle_t h);
/* Text of the last error on this handle (h == NULL: last error of a failed imx_create). */
IMX_API const char* imx_last_error(imx_handle_t h);

/* Replaces load_state_dict (superpoint_test.py:87-99, superglue_test.py:221-227,
 * superglue/models/superpoint.py:136-137).  `name` is the reference state-dict key, `host` a
 * HOST pointer to the fp32 tensor in the reference's layout (Conv2d: (Cout,Cin,kh,kw), Conv1d:
 * (Cout,Cin,1), BN vectors, bin_score: scalar).  The library copies.  Unknown keys are an error;
 * '*.num_batches_tracked' need not be passed. */
IMX_API int imx_load_weight(imx_handle_t h, int net, const char* name, const float* host,
                    int ndim, const int64_t* shape);
/* Folds eval-mode BatchNorm (eps 1e-5) into the preceding conv, re-lays weights out for the
 * kernels and uploads them.  Fails listing the first missing key. */
IMX_API int imx_finalize_weights(imx_handle_t h, int net);

/* SuperPoint.forward, detection half: encoder, heads, softmax+pixel-shuffle, simple_nms,
 * threshold, remove_borders, top-k (superpoint_test.py:113-149).  img_dev: (B,1,H,W).
 * counts_dev (may be NULL): B int32, number of keypoints kept per image. */
IMX_API int imx_superpoint_detect(imx_handle_t h, const float* img_dev, int B, int H, int W,
                          int32_t* counts_dev, void* stream);
/* SuperPoint.forward, description half (superpoint_test.py:151-155) for the images of the last
 * imx_superpoint_detect: writes, per image b, rows [0,count_b) of
 *   kpts_dev   (B,Kcap,2)  (x,y) float         scores_dev (B,Kcap)
 *   desc_dev   (B,Kcap,d)  one L2-normalised descriptor per row (the transpose of the
 *                          reference's (d,K) tensor; rows >= count_b are zero-filled)
 * Kcap must be >= every count (for max_keypoints >= 0, Kcap = max_keypoints suffices). */
IMX_API int imx_superpoint_describe(imx_handle_t h, int B, int Kcap, float* kpts_dev, float* scores_dev,
                            float* desc_dev, void* stream);

/* Dense SuperPoint forward used by training / pseudo-label export (superpoint/models/superpoint_train.py:31-57):
 * semi_dev (B,65,H/8,W/8) and desc_dev (B,d,H/8,W/8) in the reference's channel-major layout, descriptors divided
 * by their channel norm (:53-54). */
IMX_API int imx_superpoint_dense(imx_handle_t h, const float* img_dev, int B, int H, int W,
                         float* semi_dev, float* desc_dev, void* stream);

/* SuperGlue.forward (superglue_test.py:230-285) on B pairs.
 *   kpts{0,1}_dev (B,N{0,1},2) (x,y) px;  scores{0,1}_dev (B,N{0,1});
 *   desc{0,1}_dev: element (b, c, i) at  b*desc_stride_b + c*desc_stride_c + i*desc_stride_n
 *                  (reference layout (B,d,N): stride_c = N, stride_n = 1);
 *   n{0,1}_dev: B int32 valid counts per pair, or NULL = all N{0,1} valid;
 *   H,W per side: only the image *shape* is used (normalize_keypoints, :63-70).
 * Outputs (B,N0)/(B,N1): matches int64 (-1 = unmatched), matching_scores fp32; entries at
 * i >= count are -1 / 0.  A pair with a zero count yields all -1 / 0 (:235-242). */
IMX_API int imx_superglue_forward(imx_handle_t h, int B,
                          const float* kpts0_dev, const float* scores0_dev, const float* desc0_dev,
                          int64_t desc0_stride_b, int64_t desc0_stride_c, int64_t desc0_stride_n,
                          const int32_t* n0_dev, int N0, int H0, int W0,
                          const float* kpts1_dev, const float* scores1_dev, const float* desc1_dev,
                          int64_t desc1_stride_b, int64_t desc1_stride_c, int64_t desc1_stride_n,
                          const int32_t* n1_dev, int N1, int H1, int W1,
                          int64_t* matches0_dev, int64_t* matches1_dev,
                          float* mscores0_dev, float* mscores1_dev, void* stream);

/* Matching.forward (matching_test.py:54-82) fused for B pairs of equal-size images with
 * max_keypoints = K >= 0: SuperPoint on img0/img1 (each (B,1,H,W)), then SuperGlue, no host
 * synchronisation.  Outputs per side s: kpts (B,K,2), scores (B,K), counts (B) int32,
 * desc (B,K,d) or NULL, matches (B,K) int64, mscores (B,K). */
IMX_API int imx_match_pairs(imx_handle_t h, const float* img0_dev, const float* img1_dev, int B, int H, int W,
                    float* kpts0_dev, float* kpts1_dev, float* scores0_dev, float* scores1_dev,
                    int32_t* counts0_dev, int32_t* counts1_dev, float* desc0_dev, float* desc1_dev,
                    int64_t* matches0_dev, int64_t* matches1_dev,
                    float* mscores0_dev, float* mscores1_dev, void* stream);

/* Multi-GPU result collection: packs the outputs of imx_match_pairs for B pairs into fixed-size match records,
 * one row of 3 + 8K 32-bit words per pair:
 *   [pair_id | n0 | n1 | kpts0 2K f32 | kpts1 2K f32 | matches0 K i32 | matches1 K i32 | mscores0 K f32 | mscores1 K f32]
 * (floats as bit patterns; match indices narrowed to int32).  rec_dev (rows, 3+8K) int32 with rows >= B: rows past B are padding
 * (pair id -1, zeros) so that every rank contributes the same row count to the gather.  The rows are what the ranks exchange --
 * one gather to the rank that writes the results (the reference's loop over pairs has no cross-pair state:
 * superpoint_glue_test.py:66,72-78); image-matching_amd/shard.py does it with torch.distributed over RCCL, a C host with
 * ncclGroupStart / ncclSend / ncclRecv on the same buffers. */
IMX_API int imx_pack_records(imx_handle_t h, const int32_t* pair_ids_dev, int B, int K,
                     const float* kpts0_dev, const float* kpts1_dev, const int32_t* counts0_dev, const int32_t* counts1_dev,
                     const int64_t* matches0_dev, const int64_t* matches1_dev,
                     const float* mscores0_dev, const float* mscores1_dev, int32_t* rec_dev, int rows, void* stream);

/* The one collective of the path, for hosts without torch.distributed: every rank's (rows, width) int32 record buffer
 * (imx_pack_records; the same `rows` on every rank) is collected on rank `dst` -- out_dev there (world*rows, width), ordered by
 * rank; ignored on the other ranks -- as ONE group of ncclSend / ncclRecv over xGMI, enqueued on `stream`.  nccl_comm: an
 * ncclComm_t the host created with RCCL (ncclCommInitRank); rank and world size are read from it.  The RCCL entry points are
 * resolved at run time from the RCCL already loaded in the process (else librccl.so). */
IMX_API int imx_gather_records(imx_handle_t h, const int32_t* rec_dev, int rows, int width, int32_t* out_dev, int dst,
                       void* nccl_comm, void* stream);

/* Registration post-step inside the reference's timed region: RANSAC partial-affine (4-DoF similarity) fit
 * of kpts0[valid] -> kpts1[matches0[valid]], replacing cv2.estimateAffinePartial2D(..., cv2.RANSAC,
 * ransacReprojThreshold) at superpoint_glue_test.py:86-92.  Per pair: `hypotheses`
 * two-point models from a counter-based RNG (`seed`), best inlier count wins (ties: lowest hypothesis id),
 * closed-form least-squares refit on its inliers.  M_dev (B,2,3); inlier_dev (B,K) uint8 in keypoints0 index
 * space; n_inliers_dev (B) = 0 when the pair has <= 3 matches (no fit, M = 0).  counts0_dev may be NULL.  Any K (the matched
 * coordinates are staged in LDS up to K = 8192 and in HBM above). */
IMX_API int imx_estimate_affine_partial(imx_handle_t h, const float* kpts0_dev, const float* kpts1_dev,
                                const int64_t* matches0_dev, const int32_t* counts0_dev, int B, int K,
                                float ransac_threshold, int hypotheses, uint32_t seed,
                                float* M_dev, uint8_t* inlier_dev, int32_t* n_inliers_dev, void* stream);

/* SuperPoint + nearest-neighbour matcher: exact 2-NN over descriptor rows and the ratio test
 * `m.distance < ratio * n.distance`, replacing cv2.FlannBasedMatcher(...).knnMatch(Desc1, Desc2, k=2) and the loop at
 * superpoint_flann_test.py:66-74 (FLANN's KD-tree search is approximate; this is the exact search it approximates).
 * desc{0,1}_dev addressed like imx_superglue_forward's descriptors; n{0,1}_dev optional valid counts.
 * matches_dev (B,N0) int64 nearest index in side 1 or -1; dist{1,2}_dev (B,N0) L2 distances to the two neighbours. */
IMX_API int imx_knn_ratio_match(imx_handle_t h, int B,
                        const float* desc0_dev, int64_t desc0_stride_b, int64_t desc0_stride_c, int64_t desc0_stride_n,
                        const int32_t* n0_dev, int N0,
                        const float* desc1_dev, int64_t desc1_stride_b, int64_t desc1_stride_c, int64_t desc1_stride_n,
                        const int32_t* n1_dev, int N1, float ratio,
                        int64_t* matches_dev, float* dist1_dev, float* dist2_dev, void* stream);

/* Image ingest: cv2.resize(uint8 gray, (W,H)) [INTER_LINEAR] followed by `/255`
 * (datasets/SSHIDataset.py:19-27) on the GPU: src_dev (B,Hs,Ws) uint8 (batch stride src_stride_b bytes) ->
 * dst_dev (B,H,W) float32 in [0,1], ready for imx_superpoint_detect / imx_match_pairs.  H==Hs, W==Ws is the
 * `resize_scale is None` path (plain /255). */
IMX_API int imx_ingest_resize_u8(imx_handle_t h, const uint8_t* src_dev, int B, int Hs, int Ws, int64_t src_stride_b,
                         float* dst_dev, int H, int W, void* stream);

/* Warp post-step: cv2.warpAffine(source_original*255, Matrix, (W,H)) as written by cv2.imwrite
 * (superpoint_glue_test.py:101-113, superpoint_flann_test.py:88-92).  M_host: forward 2x3 matrix, 6 doubles on
 * the host (row-major).  src_dev (Hs,Ws) uint8 -> dst_dev (H,W) uint8. */
IMX_API int imx_warp_affine_u8(imx_handle_t h, const uint8_t* src_dev, int Hs, int Ws, const double* M_host,
                       uint8_t* dst_dev, int H, int W, void* stream);

/* Single-stage entry point: simple_nms (superpoint_test.py:7-22) on a caller-supplied score map
 * (B,H,W) -> out (B,H,W).  Compare-only arithmetic: bit-exact given identical input. */
IMX_API int imx_op_nms(imx_handle_t h, const float* scores_dev, float* out_dev, int B, int H, int W,
               int radius, void* stream);

/* Parity-test taps: when enabled, forwards keep copies of named intermediates
 * ("x4","semi","desc","score_map","nms","kenc","gnn<i>","mdesc","scores_in","u","v", ...).
 * imx_debug_fetch copies one to HOST (synchronises the device); shape_out gets up to 4 dims. */
IMX_API int imx_set_debug(imx_handle_t h, int enable);
IMX_API int imx_debug_fetch(imx_handle_t h, const char* name, float* host_out, int64_t capacity,
                    int64_t* shape_out, int* ndim_out);

/* Per-kernel timing for bench.py's roofline block: when enabled, every kernel launch is
 * bracketed by HIP events on the launch stream.  imx_timing_report(h, -1, ...) synchronises,
 * aggregates by kernel name and returns the number of rows; imx_timing_report(h, i>=0, ...)
 * returns row i as (name, launches, total_ms). */
IMX_API int imx_set_timing(imx_handle_t h, int enable);
IMX_API int imx_timing_report(imx_handle_t h, int index, const char** name_out, int64_t* launches_out,
                      double* total_ms_out);
IMX_API int imx_timing_reset(imx_handle_t h);
/* The kernel form row `index` of the last report ran as: "<kernel family>:<pipe>", pipe = f32 (fp32 MFMA), bf16x3 (fp32
 * products as six bf16 term products on the bf16 MFMA) or hbm (streaming kernel); "" for single-form kernels.  A name whose
 * launches took different forms has one row per form.  bench.py prices each row against the peak of the pipe named HERE. */
IMX_API const char* imx_timing_form(imx_handle_t h, int index);

/* Kernel-form options of a handle.  Defaults come from the environment ONCE, at imx_create (IMX_MFMA, IMX_LATENCY_FORMS,
 * IMX_CONV, IMX_GNN_TAIL, IMX_ATTENTION); afterwards only this call changes them -- nothing reads the environment on the launch path.
 *   "mfma"           "x3"   (default) fp32 products as six bf16 term products on the bf16 matrix pipe where a kernel has that
 *                           form (every linear layer, attention at head dims 32/64); "f32" keeps every product on the fp32 MFMA
 *                           (the A/B reference the parity tests hold the default against);
 *   "latency_forms"  "auto" (default) one or two pairs take the latency forms of the linear layers (M <= 4096 rows) and of
 *                           the attention (grids of <= 256 workgroups) and one launch per GNN layer tail; "off" never (results
 *                           then do not depend on the batch size bit for bit); "on" whenever the shape allows; "unfused" = "on" with the GNN layer tail as three
 *                           launches instead of one (same bytes: the A/B reference of the fused latency kernel);
 *   "conv"           "wino" (default) Winograd F(2x4,3x3) with its products on the fp16 matrix pipe: both transformed operands as two
 *                           fp16 planes scaled by a power of two (per tile in the fused first layer, per image -- from the producing
 *                           layer's maximum -- in the others), three plane products ("mfma" = "x3" only); a layer with at least one
 *                           (tile pair x 64 output channels) item per CU runs on tile pairs (conv3x3_wino24p: a transformed-weight
 *                           fragment serves two tiles), the others one tile per workgroup (conv3x3_wino24h) -- the same arithmetic,
 *                           bit for bit; "wino_h" never uses the pair form (the A/B reference of that choice); "wino32" the same with every
 *                           product on the fp32 MFMA (the A/B reference); "direct" the direct implicit-GEMM kernel for every 3x3 layer
 *                           (the fallback for shapes Winograd rejects);
 *   "gnn_tail"       "auto" (default): wherever the throughput forms run (more than 4096 feature rows, descriptor_dim 128) the
 *                           tail of a GNN layer (mlp.0 -> mlp.3 + residual -> the next layer's q|k|v or final_proj) is ONE launch: three fp16
 *                           plane products of two-plane operands beside the two-plane attention (the operands' powers of two come from
 *                           bounds: the (side, pair) maxima of x and v, the weights' column L1 norms) -- EXCEPT on layers whose bounds
 *                           the weights-derived guard finds too loose for fp16's range (read-only option "arith_guard" lists them),
 *                           which run the six-bf16-product launch; "fused" forces the fp16 launch on every layer (the guard's A/B);
 *                           "bf16x3" the six-product launch everywhere; "unfused" three launches (the A/B reference: another summation order);
 *   "attention"      "auto" (default): the throughput attention (head dims 32 / 64, "mfma" = "x3") cuts q, k, v and the softmax
 *                           weights into TWO fp16 planes (22 bits; every operand scaled by a power of two taken from the maximum of its
 *                           (side, pair) over the valid rows) and keeps three term products per k-step -- EXCEPT in layers whose q, k or v
 *                           projection has an output channel more than 2^12 above the median one (column L2 norms; "arith_guard" lists
 *                           them), which run three bf16 planes; "f16x2" forces the fp16 form everywhere (the guard's A/B); "bf16x3"
 *                           three bf16 planes and six term products everywhere (the A/B reference; no range limit; both are closer to
 *                           a float64 evaluation than the fp32 MFMA form).
 *   "attention_qblocks" "auto" (default) | "1" | "2": 32-query blocks per wave of the two-plane attention at head dim 32.  Two (a workgroup of
 *                           256 queries: every staged key / value tile and every fragment read serves two blocks; bit-identical results)
 *                           where the padded keypoint count is a multiple of 256 and one block per wave would still leave 1024 workgroups;
 *   "linear"         "auto" (default) = "f16x2": the plain linear layers of the GNN in the throughput path (every layer's q|k|v, mlp.0',
 *                           mlp.3 and final_proj where the layer tail is not fused -- descriptor_dim 256 --, layer 0's q|k|v otherwise) as
 *                           three fp16 plane products (gemm_h2): both operands as two fp16 planes, the weights scaled by one power of
 *                           two per matrix, the activations by the power of two of their ACTUAL maximum over the valid rows of each
 *                           (side, pair) -- written by the producing kernel's epilogue (the attention output by max |v|, of whose rows
 *                           it is a convex combination).  Needs "attention" = f16x2 (its tables carry the maxima), padded keypoint
 *                           counts that are multiples of 128 and weights inside the spread guard (largest |w| at most 2^14 above the
 *                           median column's largest); otherwise, and under "bf16x3", gemm_x3's six bf16 plane products;
 * A/B switches of the bit-identity tests and of tools/ (results agree bit for bit, the Sinkhorn group to 2e-6 in the potentials with
 * equal matches); none of them is read from the environment:
 *   "conv_swizzle"      "on" (default) the tensor between two pair-form 3x3 layers without a pool is tile-swizzled; "off": blocked;
 *   "qkv_amax"          "epilogue" (default) a plain q|k|v projection writes the (side, pair) maxima in its epilogue; "kernel": a separate pass;
 *   "sinkhorn_group"    "auto" (default: the most slabs per workgroup -- 4, 2 or 1 -- whose groups still fill the chip's resident
 *                       workgroup slots: 1024 up to 1024 columns, 512 above) | "1" | "2" | "4";
 *   "sinkhorn_prefetch" "auto" (default: off since round 6 -- two 16-wave workgroups per CU cover each other) | "off" | "on";
 *   "sinkhorn_merge"    "auto" (default: = "kernel") | "kernel" (sinkhorn_vmerge: a second launch per iteration) | "fused" (the last
 *                       slab workgroups of a pair merge its column partials: one launch per iteration, bit-identical, measured slower);
 *   "keypoints"         "auto" (default: "bits" where nms_radius is 1..4 and keypoint_threshold >= 0, else "dense") | "dense" (the
 *                       keypoint kernels read the NMS'd score map, three passes) | "bits" (they read the candidate bit rows the last
 *                       NMS stage writes; the "nms" debug tap is then computed when it is fetched).
 * Read-only (imx_get_option only): "arith_guard" -- what the weights-derived guards decided at imx_finalize_weights: the largest spread
 * of a layer's transformed convolution weights and the pipe the 3x3 chain runs on, the GNN layers whose tail runs bf16x3, the largest
 * bound looseness, the layers whose attention runs bf16x3 with the largest q|k|v channel spread, the largest spread of the plain
 * linear layers' weights and the form they run.  The guards cover the convolution weights' per-output-channel spread (all eight 3x3 layers, conv1a by its plain weights), the layer
 * tails' bounds, the q / k / v projections' channel spread (round 6) and the linear layers' weight spread; they are estimates from
 * the WEIGHTS -- an input-dependent outlier (one activation 2^16 above the rest of its (side, pair)) is not seen by them.
 * "conv" also accepts "wx3" (round 3's removed bf16-plane convolution: runs "wino32" and says so on stderr).
 * Unknown keys / values are an error.  imx_get_option returns the current value ("" for an unknown key); the pointer is valid
 * until the next call on the handle. */
IMX_API int imx_set_option(imx_handle_t h, const char* key, const char* value);
IMX_API const char* imx_get_option(imx_handle_t h, const char* key);

/* Library build string, e.g. "imx 0.4 gfx950 hip-7.2 fp32 build 3f2a91c07d1e" (the id is a digest of the library
 * sources: measurements taken on one build are only quoted for that build). */
IMX_API const char* imx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IMX_H */
