#!/bin/bash
# per-phase cycle traces of the Winograd kernels on one 16-pair step (IMX_WINO_TRACE: developer instrumentation);
# IMX_WX3_EXP=1..3: timing experiments that drop one ingredient of conv3x3_wx3's phase
for e in ${EXPS:-0}; do
  echo "== IMX_WX3_EXP=$e"
  IMX_CONV=wx3 IMX_WX3_EXP=$e IMX_WINO_TRACE=1 timeout 300 python tools/run_pairs.py --pairs 16 --iters 1 2>&1 | grep -E "wx3 trace" | head -${LINES_:-3}
done
