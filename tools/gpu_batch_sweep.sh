#!/bin/bash
# conv-stack time per pair against the batch size: does a step whose activations fit the 256 MB Infinity Cache run the layers faster?
for b in 2 4 8 16 64; do
python bench.py --pairs-per-gpu $b --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); k=j['roofline']['kernels']; B=j['config']['pairs_per_gpu_per_step']
names=('conv1ab_pool','conv2a','conv2b_pool','conv3a','conv3b_pool','conv4a','conv4b','convPaDa')
print('B',B,'pairs/s',round(j['value'],1),'us/pair: step',round(1e3*j['ms_per_step']/B,1),' '.join(f'{n}={1e3*k[n][\"ms_per_step\"]/B:.1f}' for n in names))"
done
