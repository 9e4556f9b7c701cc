#!/bin/bash
# quick GPU check: SuperGlue/matching parity tests + C3 and C5 bench summaries
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in "--workload c3" "--workload c5 --pairs-per-gpu 8 --steps 5"; do
python bench.py $w --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); k=j['roofline']['kernels']
print(round(j['value'],1), 'ms/step', j['ms_per_step'], {n: k[n]['ms_per_step'] for n in ('conv1ab_pool','conv2a','conv2b_pool','conv3a','conv3b_pool','conv4a','convPaDa','attention','qkv_proj','gnn_mlp1','gnn_mlp2','sinkhorn','nms')})"
done
