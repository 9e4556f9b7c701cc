#!/bin/bash
# round 6, call 2: where the pipelined pair kernel's time goes -- patch loads removed / cache-hot / late, U refills batched
mkdir -p gpurun_out
out=gpurun_out/r06_ab2.txt; : > $out
for v in old o0 o0e1 o0e2 o0e4 o0e8 o0e12 o1e12 o1e4 o1e1; do
  echo "=== $v" >> $out
  timeout 200 tools/tmp_ab/conv_h_bench_$v 128 2>&1 | grep "blocked" | sed -e 's/| max.*//' -e 's/fp32 wino.*pair/pair/' >> $out
done
cat $out
