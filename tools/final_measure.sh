#!/bin/bash
# Final measurements of a round on the GPU box (run through gpurun; part 1 | 2 | 3 keep each call inside gpurun's limit).
#   tools/final_measure.sh <tag> <part>
# Writes gpurun_out/<tag>/: bench lines, the rocprofv3 kernel trace of the default bench command with the averages of the
# SETTLED crowd (last 500 launches per kernel), and the --pmc counter summaries (tools/counters.py) of C3 / C5 / C2.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PART=$2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
if [ "$PART" = 1 ]; then
  $B > $O/bench_C3_step.json 2> $O/err.txt || exit 1
  $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_C3_step_driver_like_20_steps.json 2>> $O/err.txt || exit 1
  $B --mode orca --no-cpu-baseline > $O/bench_C3_orca.json 2>> $O/err.txt || exit 1
  for W in C2 C5; do
    $B --workload $W > $O/bench_${W}.json 2>> $O/err.txt || exit 1
    $B --workload $W --mode orca > $O/bench_${W}_orca.json 2>> $O/err.txt || exit 1
  done
  for W in A16 A50 A100 C3; do
    $B --mode alan --workload $W --steps 1000 --warmup 200 --cpu-seconds 6 > $O/bench_alan_${W}.json 2>> $O/err.txt || exit 1
  done
  $B --variant free --no-cpu-baseline > $O/bench_C3_step_free.json 2>> $O/err.txt || exit 1
  $B --starts separated --no-cpu-baseline > $O/bench_C3_step_separated.json 2>> $O/err.txt || exit 1
  python3 $R/tools/diag/env_default_world.py > $O/reference_world_batch.txt 2>> $O/err.txt || exit 1
fi
if [ "$PART" = 2 ]; then
  rm -rf $O/trace_C3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_C3 -o trace -- python3 $R/bench.py --no-cpu-baseline --steps 2000 --warmup 200 > $O/trace_C3.log 2>&1 || exit 1
  python3 $R/tools/settled_kernel_stats.py $O/trace_C3 500 > $O/kernel_stats_rocprofv3_settled.csv || exit 1
  cp $(find $O/trace_C3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_rocprofv3_all_launches.csv
  rm -rf $O/trace_C3
  PART=pmc; WL="C3"
fi
if [ "$PART" = 3 ]; then PART=pmc; WL="C5 C2"; fi
if [ "$PART" = pmc ]; then
  for W in $WL; do
    case $W in C3) A=4096; N=64;; C5) A=256; N=512;; C2) A=1024; N=16;; esac
    CMD="python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline"
    rm -rf $O/pmc_$W; mkdir -p $O/pmc_$W
    rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_$W/fetch -- $CMD > $O/pmc_$W/fetch.log 2>&1 || exit 1
    rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_$W/write -- $CMD > $O/pmc_$W/write.log 2>&1 || exit 1
    rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/pmc_$W/sq1 -- $CMD > $O/pmc_$W/sq1.log 2>&1 || exit 1
    rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_WAIT_ANY -d $O/pmc_$W/sq2 -- $CMD > $O/pmc_$W/sq2.log 2>&1 || exit 1
    (cd $R && python3 tools/counters.py $O/pmc_$W/fetch $O/pmc_$W/write $O/pmc_$W/sq1,$O/pmc_$W/sq2 $O/counters_${W}_step.json $W step $A $N) > $O/counters_${W}.log 2>&1 || exit 1
    rm -rf $O/pmc_$W
  done
fi
echo "part $2 done"
