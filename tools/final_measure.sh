#!/bin/bash
# Round-end measurement batch (GPU box): tests, bench lines, rocprofv3 kernel stats, SQ and TCC counter passes.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; T=$1
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests -m gpu -q > $O/${T}_gpu_tests.log 2>&1 || { tail -20 $O/${T}_gpu_tests.log; exit 1; }
tail -1 $O/${T}_gpu_tests.log
timeout -k 10 300 python3 $R/bench.py > $O/${T}_bench_C3_step.json 2>$O/bench.err || exit 1
for wl in C2 C5; do timeout -k 10 300 python3 $R/bench.py --workload $wl > $O/${T}_bench_$wl.json 2>>$O/bench.err || exit 1; done
timeout -k 10 300 python3 $R/bench.py --mode orca > $O/${T}_bench_C3_orca.json 2>>$O/bench.err || exit 1
echo bench done
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_rocprofv3.csv
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/sq --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/sq.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O/sq | grep -A9 "nbr_kernel\|obs_kernel\|step_kernel" > $O/${T}_sq_pmc.txt
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/write.log 2>&1 || exit 1
python3 $R/tools/hbm_traffic.py $O/fetch $O/write $O/${T}_hbm_traffic_pmc.json 4096 64 > /dev/null || exit 1
cd $R && timeout -k 10 300 python3 tools/stamps.py C3 step > $O/${T}_kernel_phase_stamps.txt 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/stamps.py C3 step rt 2>/dev/null | grep "timeline\|duration\|end last" > $O/${T}_wave_timelines.txt
rm -rf $O/kt $O/sq $O/fetch $O/write $R/gpurun_out/libcaenv_stamps.so
ls $O
