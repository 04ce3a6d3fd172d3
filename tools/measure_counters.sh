#!/bin/bash
# Counter passes (FETCH_SIZE | WRITE_SIZE | SQ_*) for the workloads other than the default one, so that bench.py can
# quote measured traffic for them too.  usage: tools/measure_counters.sh <tag>   (outputs under gpurun_out/<tag>/)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {  # workload mode A N
  local wl=$1 mode=$2 A=$3 N=$4 args="--workload $1 --mode $2 --steps 20 --warmup 5 --rollout-chunk 1 --no-cpu-baseline"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/sq --output-format csv -- python3 $R/bench.py $args > $O/sq.log 2>&1 || return 1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d $O/sq2 --output-format csv -- python3 $R/bench.py $args > $O/sq2.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $args > $O/fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $args > $O/write.log 2>&1 || return 1
  python3 $R/tools/counters.py $O/fetch $O/write $O/sq,$O/sq2 $O/${T}_counters_${wl}_${mode}.json $wl $mode $A $N > /dev/null || return 1
  rm -rf $O/sq $O/sq2 $O/fetch $O/write
  # the bench line of this workload again, with its counter summary in place (bench.py reads profiles/)
  local out=$O/${T}_bench_${wl}$([ $mode = orca ] && echo _orca).json
  cp $O/${T}_counters_${wl}_${mode}.json $R/profiles/ && timeout -k 10 300 python3 $R/bench.py --workload $wl --mode $mode > $out 2>>$O/bench.err || return 1
  echo "$wl $mode done"
}
run C2 step 1024 16 && run C5 step 256 512 && run C3 orca 4096 64 && run C2 orca 1024 16
