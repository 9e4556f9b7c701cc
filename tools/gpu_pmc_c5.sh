#!/bin/bash
# Counters of the C5 step (1280x960, d = 256, 2048 keypoints, 100 Sinkhorn iterations; 8 pairs per step): kernel stats and the PMC
# passes of tools/gpu_pmc_limiter.sh / gpu_profile_all.sh on tools/run_pairs.py --workload c5 (kernel trace only, one block family per pass).
# usage: tools/gpu_pmc_c5.sh <tag>      -> gpurun_out/<tag>_c5_{kernel_stats,pmc_sqa,pmc_sqb,pmc_tcp,pmc_tcp2,pmc_fetch,pmc_write}.txt
tag=${1:-r05}
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out
run() {  # rocprof args...
  rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
  (cd /tmp && timeout 300 rocprofv3 "$@" -d $out/prof_tmp -- python $R/tools/run_pairs.py --workload c5 --pairs 8 --iters 2 > $out/prof_tmp/run.log 2>&1)
  find $out/prof_tmp -name "*.db" | head -1
}
db=$(run --kernel-trace --stats)
[ -n "$db" ] && python tools/rocpd_summary.py $db > $out/${tag}_c5_kernel_stats.txt
pass() {  # name, counters...
  name=$1; shift
  db=$(run --kernel-trace --pmc "$@")
  if [ -z "$db" ]; then echo "pass $name: no database"; tail -5 $out/prof_tmp/run.log; return; fi
  python tools/rocpd_pmc.py $db > $out/${tag}_c5_pmc_${name}.txt
  head -8 $out/${tag}_c5_pmc_${name}.txt | cut -c1-240
}
pass sqa GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
pass sqb GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass tcp GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp2 GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
rm -rf $out/prof_tmp
head -14 $out/${tag}_c5_kernel_stats.txt | cut -c1-130
