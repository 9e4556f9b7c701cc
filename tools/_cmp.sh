python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_matching.py -x -q 2>&1 | tail -3
IMX_WINO_TRACE=1 python tools/run_pairs.py --pairs 32 --iters 1 2>&1 | grep "wino24 trace" | head -1
bash tools/gpu_bench_only.sh
