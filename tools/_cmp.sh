python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_matching.py -x -q 2>&1 | tail -5
for m in f24 f22; do echo "IMX_CONV1=$m"; IMX_CONV1=$m bash tools/gpu_bench_only.sh; done
