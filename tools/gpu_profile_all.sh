#!/bin/bash
# Full measurement set for profiles/: bench JSON lines (c3 default, c2, c5), rocprofv3 kernel stats of the bench command,
# and PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE -- separate runs, kernel trace only) of the 64-pair step (the bench's own step).
# usage: tools/gpu_profile_all.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r06}
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out; mkdir -p $out
python bench.py > $out/${tag}_bench_c3.json 2> $out/${tag}_bench_c3.log
python bench.py --workload c2 --no-cpu-baseline --no-extras > $out/${tag}_bench_c2.json 2>/dev/null
python bench.py --workload c5 --pairs-per-gpu 8 --steps 5 --no-cpu-baseline --no-extras > $out/${tag}_bench_c5.json 2>/dev/null
prof() {  # name, rocprof args..., -- cmd
  name=$1; shift
  rm -rf $out/prof_tmp; mkdir -p $out/prof_tmp
  (cd /tmp && timeout 420 rocprofv3 "$@" > $out/prof_tmp/run.log 2>&1)
  find $out/prof_tmp -name "*.db" | head -1
}
db=$(prof stats --kernel-trace --stats -d $out/prof_tmp -- python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1)
python tools/rocpd_summary.py $db > $out/${tag}_kernel_stats.txt
db=$(prof sq --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 64 --iters 2)
python tools/rocpd_pmc.py $db > $out/${tag}_pmc_sq.txt
db=$(prof fetch --kernel-trace --pmc FETCH_SIZE -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 64 --iters 2)
python tools/rocpd_pmc.py $db > $out/${tag}_pmc_fetch.txt
db=$(prof write --kernel-trace --pmc WRITE_SIZE -d $out/prof_tmp -- python $R/tools/run_pairs.py --pairs 64 --iters 2)
python tools/rocpd_pmc.py $db > $out/${tag}_pmc_write.txt
rm -rf $out/prof_tmp
round=${tag%%_*}
python tools/pmc_traffic.py $out/${tag}_pmc_fetch.txt $out/${tag}_pmc_write.txt 64 $out/${round}_pmc_traffic.json > /dev/null
# the counters behind roofline.limiter (SQ x2, TA stalls, TCP x2) -> ${round}_pmc_limiter.json
tools/gpu_pmc_limiter.sh ${tag} > $out/${tag}_pmc_limiter.log 2>&1
cp $out/${tag}_pmc_limiter.json $out/${round}_pmc_limiter.json
tail -1 $out/${tag}_bench_c3.json | cut -c1-400
head -12 $out/${tag}_kernel_stats.txt
head -8 $out/${tag}_pmc_sq.txt; head -4 $out/${tag}_pmc_fetch.txt; head -4 $out/${tag}_pmc_write.txt
