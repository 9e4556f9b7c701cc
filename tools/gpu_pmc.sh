#!/bin/bash
# one rocprofv3 PMC pass over tools/run_pairs.py: tools/gpu_pmc.sh <tag> <counter> [counter...]   -> gpurun_out/<tag>_pmc.txt
tag=$1; shift
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out; rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
(cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 32 --iters 2 > $out/prof_tmp/run.log 2>&1)
db=$(find $out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_pmc.py $db > $out/${tag}_pmc.txt
rm -rf $out/prof_tmp
head -14 $out/${tag}_pmc.txt
