#!/bin/bash
# The counters behind roofline.limiter: three rocprofv3 --pmc passes (kernel trace only, one block family per
# pass) over the bench's own 64-pair C3 step, summarised per kernel, and profiles/<tag>_pmc_limiter.json for bench.py.
# usage: tools/gpu_pmc_limiter.sh <tag>      -> gpurun_out/<tag>_pmc_{sqa,sqb,ta,ta2,tcp,tcp2}.txt, gpurun_out/<tag>_pmc_limiter.json
# (SKIP_SQ=1: only the TA / TCP passes; each pass is ~2-3 minutes of box time)
tag=${1:-r06}
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out
pass() {  # name, counters...
  name=$1; shift
  rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
  # (a pass that rocprofv3 cannot schedule aborts and then hangs: bounded)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 64 --iters 2 > $out/prof_tmp/run.log 2>&1)
  db=$(find $out/prof_tmp -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "pass $name: no database"; tail -5 $out/prof_tmp/run.log; return; fi
  python tools/rocpd_pmc.py $db > $out/${tag}_pmc_${name}.txt
  head -6 $out/${tag}_pmc_${name}.txt | cut -c1-260
}
[ -n "$SKIP_SQ" ] || pass sqa GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
[ -n "$SKIP_SQ" ] || pass sqb GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# (TA_TA_BUSY / TA_BUFFER_* make rocprofv3 abort on this box: not collected)
pass ta2 GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp2 GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
rm -rf $out/prof_tmp
python tools/pmc_limiter.py $out/${tag}_pmc_ 64 $out/${tag}_pmc_limiter.json
