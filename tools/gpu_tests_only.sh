mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r03_v2_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_v2_gpu_tests.log
grep -E "^\[sweep|\[fp64-anchored\]" gpurun_out/r03_v2_gpu_tests.log > gpurun_out/r03_v2_gpu_tests_parity_log.txt
