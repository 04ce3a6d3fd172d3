#!/bin/bash
# Measurement batch on the GPU box: tests, bench lines, rocprofv3 kernel stats, SQ and TCC counter passes.
# usage: tools/measure.sh <tag> [quick]     (outputs under gpurun_out/<tag>/, to be copied into profiles/)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1 || { tail -30 $O/${T}_gpu_tests.log; exit 1; }
tail -1 $O/${T}_gpu_tests.log
timeout -k 10 300 python3 $R/bench.py > $O/${T}_bench_C3_step.json 2>$O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/${T}_bench_C3_step.json'));print('C3 step', d['value'], {k:v for k,v in d['kernels_ms'].items() if isinstance(v,float)}, d['roofline']['kernel'], d['roofline']['frac'])"
[ "$2" = quick ] && exit 0
for wl in C2 C5; do timeout -k 10 300 python3 $R/bench.py --workload $wl > $O/${T}_bench_$wl.json 2>>$O/bench.err || exit 1; done
# ORCA-only policy rollouts (BASELINE config C2 is one): ca_rollout in chunks of 50 steps, and one ca_orca_step per step
for wl in C2 C3 C5; do
  timeout -k 10 300 python3 $R/bench.py --workload $wl --mode orca > $O/${T}_bench_${wl}_orca.json 2>>$O/bench.err || exit 1
  timeout -k 10 300 python3 $R/bench.py --workload $wl --mode orca --rollout-chunk 1 --no-cpu-baseline > $O/${T}_bench_${wl}_orca_per_step_calls.json 2>>$O/bench.err || exit 1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/${T}_bench_*.json")):
    d=json.load(open(f)); print("%-48s %8.1f M  %s" % (f.split("/")[-1], d["value"]/1e6, {k:v for k,v in d["kernels_ms"].items() if isinstance(v,float)}))
PY
timeout -k 10 300 python3 $R/bench.py --variant free --no-cpu-baseline > $O/${T}_bench_C3_step_free.json 2>>$O/bench.err || exit 1
timeout -k 10 300 python3 $R/bench.py --starts separated --no-cpu-baseline > $O/${T}_bench_C3_step_separated.json 2>>$O/bench.err || exit 1
timeout -k 10 600 python3 $R/tools/cpu_baseline_table.py $O/${T}_cpu_baseline_table.json 4 > $O/${T}_cpu_baseline_table.txt 2>>$O/bench.err || exit 1
echo bench done
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_rocprofv3.csv
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/sq --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d $O/sq2 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
python3 $R/tools/counters.py $O/fetch $O/write $O/sq,$O/sq2 $O/${T}_counters_C3_step.json C3 step 4096 64 > /dev/null || exit 1
# the headline line again, now that its counter summary exists (bench.py quotes profiles/*_counters_*.json of the same sources)
cp $O/${T}_counters_C3_step.json $R/profiles/ && timeout -k 10 300 python3 $R/bench.py > $O/${T}_bench_C3_step.json 2>>$O/bench.err || exit 1
rm -rf $O/kt $O/sq $O/sq2 $O/fetch $O/write
ls $O
