#!/bin/bash
# soak: the shape / config fuzz tests of the parity suite over many more seeds than the suite runs (IMX_FUZZ_SEEDS=a-b)
# usage: tools/gpu_soak.sh 100-220   -> gpurun_out/soak_<range>.log
r=${1:-100-160}
mkdir -p gpurun_out
IMX_FUZZ_SEEDS=$r timeout ${SOAK_TIMEOUT:-1500} python -m pytest -q -m gpu \
  "tests/test_gpu_superpoint.py::test_superpoint_random_shapes_and_configs_vs_oracle" \
  "tests/test_gpu_parity_r2.py::test_superglue_random_shapes_batches_and_counts_vs_oracle" \
  "tests/test_gpu_parity_r2.py::test_superglue_random_shapes_on_the_throughput_forms" > gpurun_out/soak_$r.log 2>&1
echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/soak_$r.log | tail -40
