#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r06_ab3.txt; : > $out
for v in old olde1 olde2 o0 o0e16 o1e16 o0e1; do
  echo "=== $v" >> $out
  timeout 200 tools/tmp_ab/conv_h_bench_$v 128 2>&1 | grep "blocked" | sed -e 's/| max.*//' -e 's/fp32 wino.*pair/pair/' >> $out
done
cat $out
