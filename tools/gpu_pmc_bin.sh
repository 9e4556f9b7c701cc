#!/bin/bash
# one rocprofv3 PMC pass over a standalone binary: tools/gpu_pmc_bin.sh <tag> <binary> <counter> [counter...]  -> gpurun_out/<tag>_pmc.txt
tag=$1; bin=$2; shift; shift
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out; rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
(cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_tmp -- $R/$bin > $out/prof_tmp/run.log 2>&1)
db=$(find $out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_pmc.py $db > $out/${tag}_pmc.txt
rm -rf $out/prof_tmp
head -8 $out/${tag}_pmc.txt
