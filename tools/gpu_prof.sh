#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>_kernel_stats.txt   usage: tools/gpu_prof.sh tag [bench args]
tag=$1; shift
R=$(pwd); export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$tag
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -- python $R/bench.py --no-cpu-baseline --no-extras --no-roofline-pass --steps 3 --warmup 1 "$@" > $R/gpurun_out/prof_$tag/bench.log 2>&1
cd $R
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats.txt
rm -rf gpurun_out/prof_$tag
head -40 gpurun_out/${tag}_kernel_stats.txt
