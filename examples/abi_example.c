/* Plain-C use of the imx C ABI (include/imx.h) with the HIP runtime only -- no Python, no torch.
 *
 *   gcc -std=c99 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/abi_example.c \
 *       -Limage-matching_amd -limx -L/opt/rocm/lib -lamdhip64 -o abi_example
 *   LD_LIBRARY_PATH=image-matching_amd:/opt/rocm/lib ./abi_example
 *
 * It creates a handle with the BatchNorm SuperPoint configuration of superpoint_test.py:57-63, loads a state dict
 * given as (key, shape, data) records, runs the detector on a batch of images resident in HBM and reads the
 * keypoint counts back: the same sequence the Python drop-in classes issue through ctypes.  Weight records come from
 * the caller (a checkpoint reader); here they are zero-filled placeholders, so finalize only demonstrates the error
 * path ("missing key ...") unless every key of the network is supplied. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "imx.h"

static int check(imx_handle_t h, int rc, const char* what) {
  if (rc != 0) fprintf(stderr, "%s failed: %s\n", what, imx_last_error(h));
  return rc;
}

int main(void) {
  imx_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.descriptor_dim = 128;
  cfg.nms_radius = 4;
  cfg.keypoint_threshold = 0.005f;
  cfg.max_keypoints = 1024;
  cfg.remove_borders = 4;
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
