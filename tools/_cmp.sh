python -m pytest tests/test_gpu_superpoint.py -x -q 2>&1 | tail -2
IMX_WINO_TRACE=1 python tools/run_pairs.py --pairs 32 --iters 1 2>&1 | grep "wino24n trace" | head -2
bash tools/gpu_bench_only.sh
