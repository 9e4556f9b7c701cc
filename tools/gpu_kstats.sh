#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/kstats.txt (top rows printed)
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out; rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof_tmp -- python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > $out/prof_tmp/run.log 2>&1)
db=$(find $out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_summary.py $db > $out/kstats.txt
rm -rf $out/prof_tmp
head -${1:-14} $out/kstats.txt | cut -c1-130
