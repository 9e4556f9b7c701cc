/* Plain-C use of the imx C ABI (include/imx.h) with the HIP runtime only -- no Python, no torch.
 *
 *   gcc -std=c99 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/abi_example.c \
 *       -Limage-matching_amd -limx -L/opt/rocm/lib -lamdhip64 -o abi_example
 *   LD_LIBRARY_PATH=image-matching_amd:/opt/rocm/lib ./abi_example [weights.bin pair.bin H W K]
 *
 * Without arguments: creates a handle with the BatchNorm SuperPoint configuration of superpoint_test.py:57-63, hands
 * over ONE weight record and shows the error paths ("missing key ...", "weights not finalized").
 *
 * With arguments it is the whole Matching.forward (matching_test.py:54-82) from C -- the sequence the Python drop-in
 * classes issue through ctypes:
 *   weights.bin  state-dict records as a checkpoint reader would hand them over:
 *                repeated { int32 net; int32 keylen; char key[keylen]; int32 ndim; int64 shape[ndim]; float data[prod(shape)] }
 *   pair.bin     two float32 images (H x W each, values in [0,1]): image0 then image1
 *   H W K        image size and max_keypoints
 * It loads every record with imx_load_weight, finalizes both networks, uploads the pair into HBM, runs the fused
 * imx_match_pairs on a stream it owns, copies the results back and prints
 *   keypoints <n0> <n1>
 *   matches <count> checksum <sum over i of (i+1)*(matches0[i]+2) mod 2^31>
 * (tests/test_gpu_abi.py compares that line with the ctypes path on the same inputs). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "imx.h"

static int check(imx_handle_t h, int rc, const char* what) {
  if (rc != 0) fprintf(stderr, "%s failed: %s\n", what, imx_last_error(h));
  return rc;
}

/* reads the records of weights.bin into the handle; returns the number of records or -1 */
static int load_records(imx_handle_t h, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return -1; }
  int n = 0;
  for (;;) {
    int32_t net, keylen, ndim;
    if (fread(&net, 4, 1, f) != 1) break;                          /* clean end of file */
    char key[256];
    int64_t shape[8];
    if (fread(&keylen, 4, 1, f) != 1 || keylen <= 0 || keylen >= (int)sizeof key || fread(key, 1, (size_t)keylen, f) != (size_t)keylen ||
        fread(&ndim, 4, 1, f) != 1 || ndim < 0 || ndim > 8 || fread(shape, 8, (size_t)ndim, f) != (size_t)ndim) {
      fprintf(stderr, "%s: malformed record %d\n", path, n);
      fclose(f);
      return -1;
    }
    key[keylen] = 0;
    size_t count = 1;
    for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
    float* data = (float*)malloc(count * sizeof(float));
    if (!data || fread(data, sizeof(float), count, f) != count) {
      fprintf(stderr, "%s: short data for %s\n", path, key);
      free(data);
      fclose(f);
      return -1;
    }
    int rc = check(h, imx_load_weight(h, net, key, data, ndim, shape), key);
    free(data);
    if (rc != 0) { fclose(f); return -1; }
    ++n;
  }
  fclose(f);
  return n;
}

static int match_one_pair(imx_handle_t h, const char* pair_path, int H, int W, int K, int d) {
  const size_t npx = (size_t)H * W;
  float* host = (float*)malloc(2 * npx * sizeof(float));
  FILE* f = fopen(pair_path, "rb");
  if (!host || !f || fread(host, sizeof(float), 2 * npx, f) != 2 * npx) {
    fprintf(stderr, "cannot read two %dx%d float32 images from %s\n", H, W, pair_path);
    return 4;
  }
  fclose(f);
  /* every I/O buffer lives in HBM and belongs to the caller; the library borrows the pointers for the call */
  float *img = NULL, *kpts = NULL, *scores = NULL, *ms = NULL;
  int32_t* counts = NULL;
  int64_t* matches = NULL;
  hipStream_t stream = NULL;
  if (hipMalloc((void**)&img, 2 * npx * sizeof(float)) != hipSuccess || hipMalloc((void**)&kpts, sizeof(float) * 2 * K * 2) != hipSuccess ||
      hipMalloc((void**)&scores, sizeof(float) * 2 * K) != hipSuccess || hipMalloc((void**)&ms, sizeof(float) * 2 * K) != hipSuccess ||
      hipMalloc((void**)&counts, sizeof(int32_t) * 2) != hipSuccess || hipMalloc((void**)&matches, sizeof(int64_t) * 2 * K) != hipSuccess ||
      hipStreamCreate(&stream) != hipSuccess) {
    fprintf(stderr, "HIP allocation failed\n");
    return 3;
  }
  (void)d;
  hipMemcpyAsync(img, host, 2 * npx * sizeof(float), hipMemcpyHostToDevice, stream);
  /* Matching.forward for B = 1: SuperPoint on both images, SuperGlue, match extraction; asynchronous on `stream` */
  int rc = imx_match_pairs(h, img, img + npx, 1, H, W, kpts, kpts + 2 * K, scores, scores + K, counts, counts + 1,
                           NULL, NULL, matches, matches + K, ms, ms + K, stream);
  if (check(h, rc, "imx_match_pairs") != 0) return 5;
  int32_t n[2];
  int64_t* m0 = (int64_t*)malloc(sizeof(int64_t) * K);
  hipMemcpyAsync(n, counts, sizeof n, hipMemcpyDeviceToHost, stream);
  hipMemcpyAsync(m0, matches, sizeof(int64_t) * K, hipMemcpyDeviceToHost, stream);
  if (hipStreamSynchronize(stream) != hipSuccess) { fprintf(stderr, "stream failed\n"); return 6; }
  long count = 0;
  unsigned long sum = 0;
  for (int i = 0; i < K; ++i) {
    if (m0[i] >= 0) ++count;
    sum = (sum + (unsigned long)(i + 1) * (unsigned long)(m0[i] + 2)) & 0x7fffffffUL;
  }
  printf("keypoints %d %d\n", (int)n[0], (int)n[1]);
  printf("matches %ld checksum %lu\n", count, sum);
  free(m0);
  free(host);
  hipFree(img); hipFree(kpts); hipFree(scores); hipFree(ms); hipFree(counts); hipFree(matches);
  hipStreamDestroy(stream);
  return 0;
}

int main(int argc, char** argv) {
  const int full = argc >= 6;
  imx_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.descriptor_dim = 128;
  cfg.nms_radius = 4;
  cfg.keypoint_threshold = 0.005f;
  cfg.max_keypoints = full ? atoi(argv[5]) : 1024;
  cfg.remove_borders = 4;
  cfg.align_corners = 0;                                               /* superpoint_test.py:47 under torch >= 1.10 */
  cfg.sp_variant = IMX_SP_VARIANT_BN;
  cfg.num_gnn_layers = 18;
  for (int i = 0; i < 18; ++i) cfg.gnn_layer_is_cross[i] = i & 1;       /* ['self', 'cross'] * 9 */
  cfg.kenc_n = 3;
  cfg.kenc_channels[0] = 32; cfg.kenc_channels[1] = 64; cfg.kenc_channels[2] = 128;
  cfg.sinkhorn_iterations = 30;
  cfg.match_threshold = 0.1f;

  printf("%s\n", imx_version());
  imx_handle_t h = NULL;
  if (imx_create(0, &cfg, &h) != 0) {                 /* no GPU: a clean error, never a CPU fallback */
    fprintf(stderr, "imx_create: %s\n", imx_last_error(NULL));
    return 2;
  }

  if (full) {
    int n = load_records(h, argv[1]);
    if (n < 0) return 7;
    printf("loaded %d weight records\n", n);
    if (check(h, imx_finalize_weights(h, IMX_NET_SUPERPOINT), "imx_finalize_weights(SuperPoint)") != 0) return 8;
    if (check(h, imx_finalize_weights(h, IMX_NET_SUPERGLUE), "imx_finalize_weights(SuperGlue)") != 0) return 8;
    int rc = match_one_pair(h, argv[2], atoi(argv[3]), atoi(argv[4]), cfg.max_keypoints, cfg.descriptor_dim);
    imx_destroy(h);
    return rc;
  }

  /* one weight record, as a checkpoint reader would hand it over (reference key names and layouts) */
  const int64_t shape[4] = {64, 1, 3, 3};
  float* w = (float*)calloc(64 * 9, sizeof(float));
  check(h, imx_load_weight(h, IMX_NET_SUPERPOINT, "inc.conv.conv.0.weight", w, 4, shape), "imx_load_weight");
  free(w);
  if (imx_finalize_weights(h, IMX_NET_SUPERPOINT) != 0)      /* lists the first key that is still missing */
    fprintf(stderr, "finalize (expected with a partial state dict): %s\n", imx_last_error(h));

  /* images and outputs live in HBM and belong to the caller */
  const int B = 2, H = 480, W = 640;
  float* img_dev = NULL;
  int32_t* counts_dev = NULL;
  hipStream_t stream = NULL;
  if (hipMalloc((void**)&img_dev, sizeof(float) * B * H * W) != hipSuccess ||
      hipMalloc((void**)&counts_dev, sizeof(int32_t) * B) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) {
    fprintf(stderr, "HIP allocation failed\n");
    return 3;
  }
  hipMemsetAsync(img_dev, 0, sizeof(float) * B * H * W, stream);
  int rc = imx_superpoint_detect(h, img_dev, B, H, W, counts_dev, stream);     /* asynchronous on `stream` */
  if (rc != 0) fprintf(stderr, "imx_superpoint_detect (expected without weights): %s\n", imx_last_error(h));
  hipStreamSynchronize(stream);

  hipFree(img_dev);
  hipFree(counts_dev);
  hipStreamDestroy(stream);
  imx_destroy(h);
  return 0;
}
