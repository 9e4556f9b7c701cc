#!/bin/bash
# rocprofv3 kernel stats of single-pair calls (the latency path) -> gpurun_out/kstats_b1.txt
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out; rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 1 --iters ${ITERS:-20} > $out/prof_tmp/run.log 2>&1)
db=$(find $out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_summary.py $db > $out/kstats_b1.txt
rm -rf $out/prof_tmp
head -${1:-30} $out/kstats_b1.txt | cut -c1-130
