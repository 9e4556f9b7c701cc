#!/bin/bash
# the round's measurement set in one gpurun call: tools/gpu_final.sh <tag>   (e.g. r06_v1)
#   profiles (bench lines c3 / c2 / c5, kernel stats, PMC traffic + limiter passes) -> gpurun_out/<tag>_*, the two JSON files bench.py
#   reads copied into profiles/ on the box, then the bench once more WITH them, then the GPU test suite with its parity log
tag=${1:-r06_v1}; round=${tag%%_*}
tools/gpu_profile_all.sh $tag > gpurun_out/${tag}_profile_all.log 2>&1
cp gpurun_out/${round}_pmc_traffic.json profiles/${round}_pmc_traffic.json
cp gpurun_out/${round}_pmc_limiter.json profiles/${round}_pmc_limiter.json
python bench.py > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.log      # again, now WITH the PMC files of this build
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -m gpu -q -s > gpurun_out/${tag}_gpu_tests.log 2>&1
  echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_gpu_tests.log
  grep -E "\[sweep|\[fp64-anchored\]|\[strict|\[heavy\]|\[scales\]" gpurun_out/${tag}_gpu_tests.log > gpurun_out/${tag}_gpu_tests_parity_log.txt
fi
tail -1 gpurun_out/${tag}_bench_c3.json | cut -c1-300
