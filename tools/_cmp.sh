python -m pytest tests/test_gpu_superglue.py tests/test_gpu_matching.py tests/test_gpu_superpoint.py -x -q 2>&1 | tail -3
for m in ws tiled; do echo "IMX_GEMM=$m"; IMX_GEMM=$m bash tools/gpu_bench_only.sh; done
