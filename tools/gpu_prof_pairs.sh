#!/bin/bash
# rocprofv3 kernel trace of tools/run_pairs.py (no result validation) -> per-kernel averages.  usage: tools/gpu_prof_pairs.sh tag [grep-pattern]
tag=$1; pat=${2:-conv}
R=$(pwd); export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$tag
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -- python $R/tools/run_pairs.py --pairs 32 --iters 3 > $R/gpurun_out/prof_$tag/run.log 2>&1
cd $R
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_summary.py $db > gpurun_out/${tag}_kernel_stats.txt
rm -rf gpurun_out/prof_$tag
echo "== $tag"; grep -E "$pat|TOTAL" gpurun_out/${tag}_kernel_stats.txt
