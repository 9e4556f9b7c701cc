#!/bin/bash
# bench summaries only (no tests): tools/gpu_bench_only.sh [c3|c5|both]
for w in "--workload c3"; do
python bench.py $w --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); k=j['roofline']['kernels']
print(round(j['value'],1), 'ms/step', j['ms_per_step'], {n: k[n]['ms_per_step'] for n in ('conv1ab_pool','conv2a','conv2b_pool','conv3a','conv3b_pool','conv4a','convPaDa','attention','qkv_proj','gnn_mlp1','gnn_mlp2','sinkhorn','nms')})"
done
