#!/bin/bash
tools/gpu_profile_all.sh r03_v5 > gpurun_out/r03_v5_profile_all.log 2>&1
cp gpurun_out/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
python bench.py > gpurun_out/r03_v5_bench_c3.json 2> gpurun_out/r03_v5_bench_c3.log      # again, now WITH the PMC file of this build
python -m pytest tests -m gpu -q -s > gpurun_out/r03_v5_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r03_v5_gpu_tests.log
grep -E "\[sweep|\[fp64-anchored\]" gpurun_out/r03_v5_gpu_tests.log > gpurun_out/r03_v5_gpu_tests_parity_log.txt
tail -1 gpurun_out/r03_v5_bench_c3.json | cut -c1-300
