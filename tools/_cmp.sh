python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_matching.py -x -q 2>&1 | tail -3
for m in f24 f22; do echo "IMX_CONVN=$m"; IMX_CONVN=$m bash tools/gpu_bench_only.sh; done
