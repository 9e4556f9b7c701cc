python -m pytest tests/test_gpu_superpoint.py -x -q 2>&1 | tail -3
IMX_WINO_TRACE=1 python tools/run_pairs.py --pairs 32 --iters 2 2>&1 | grep "wino24 trace" | tail -2; bash tools/gpu_bench_only.sh
