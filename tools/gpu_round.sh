#!/bin/bash
# one GPU round trip: the -m gpu suite (log kept), then the default bench line.   usage: tools/gpu_round.sh <tag> [pytest args]
tag=${1:-r03}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s "$@" > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_gpu_tests.log
grep -E "\[sweep|\[fp64-anchored\]|\[strict|\[sinkhorn-anchored\]" gpurun_out/${tag}_gpu_tests.log > gpurun_out/${tag}_gpu_tests_parity_log.txt
python bench.py > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.log
echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench_c3.log; cut -c1-600 gpurun_out/${tag}_bench_c3.json
