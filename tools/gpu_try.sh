#!/bin/bash
# quick GPU iteration: SuperPoint parity tests (guarded by a timeout: a hung kernel must not hold the box), then the C3 step's kernel table
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_superpoint.py -x -q 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 2>gpurun_out/try_bench.log | tail -1 > gpurun_out/try_bench.json
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/try_bench.json').read())
    k = j['roofline']['kernels']
    print(j['value'], 'pairs/s', j['ms_per_step'], 'ms/step', j.get('parity_in_run'))
    for n, v in k.items():
        if v['ms_per_step'] > 0.15: print(' ', n, v['ms_per_step'], v.get('form'), v.get('executed_frac'))
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/try_bench.log').read()[-3000:])
PY
