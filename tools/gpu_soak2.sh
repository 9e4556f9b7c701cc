#!/bin/bash
# soak of the callers either side of the path: RANSAC, kNN ratio matcher, resize / warp fuzz tests over a seed range
r=${1:-100-200}
mkdir -p gpurun_out
IMX_FUZZ_SEEDS=$r timeout ${SOAK_TIMEOUT:-1200} python -m pytest -q -m gpu \
  "tests/test_gpu_registration.py::test_ransac_random_sizes_counts_and_outliers_vs_host_restatement" \
  "tests/test_gpu_registration.py::test_knn_ratio_random_shapes_vs_brute_force" \
  "tests/test_gpu_ingest.py::test_resize_and_warp_random_sizes_bit_exact" > gpurun_out/soak2_$r.log 2>&1
echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/soak2_$r.log | tail -40
