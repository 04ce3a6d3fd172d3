#!/bin/bash
# Round 3, GPU batch 1: baseline of this round (tests, bench lines at HEAD), C5/C2/C3 A/B against the round-1 tree,
# instruction-mix / instruction-cache counter passes, clock trace with sclk.   usage: tools/r03_run1.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r03a}; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
B="timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline"
$B > $O/bench_C3_step.json 2>$O/bench.err || { tail $O/bench.err; exit 1; }
for wl in C2 C5; do for m in step orca; do $B --workload $wl --mode $m > $O/bench_${wl}_$m.json 2>>$O/bench.err || exit 1; done; done
$B --mode orca > $O/bench_C3_orca.json 2>>$O/bench.err || exit 1
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "%.1f M" % (d["value"]/1e6), d["kernels_ms"])
PY
echo "--- round-1 tree (a4ec6c8), >= 1 s of warm-up"
R1=$R/build/r01tree
( cd $R1 && timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload C5 --warmup 10000 --steps 2000 > $O/r01tree_C5.json 2>>$O/bench.err \
  && timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload C2 --warmup 30000 --steps 2000 > $O/r01tree_C2.json 2>>$O/bench.err \
  && timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload C3 --warmup 7000 --steps 2000 > $O/r01tree_C3.json 2>>$O/bench.err ) || { tail $O/bench.err; exit 1; }
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/r01tree_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "%.1f M" % (d["value"]/1e6), d.get("kernels_ms"))
PY
echo "--- counters"
P="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH -d $O/pmcA --output-format csv -- $P > $O/pmcA.log 2>&1 || { tail $O/pmcA.log; exit 1; }
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d $O/pmcB --output-format csv -- $P > $O/pmcB.log 2>&1 || { tail $O/pmcB.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY -d $O/pmcC --output-format csv -- $P > $O/pmcC.log 2>&1 || { tail $O/pmcC.log; exit 1; }
python3 - <<PY > $O/counters_C3_instruction_mix.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmcA","pmcB","pmcC"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ca::","")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "step_kernel" in k or "obs_kernel" in k:
        print(k)
        for c,v in sorted(acc[k].items()): print("   %-32s %16.0f  (%d launches)" % (c, sum(v)/len(v), len(v)))
PY
cat $O/counters_C3_instruction_mix.txt
rm -rf $O/pmcA $O/pmcB $O/pmcC
echo "--- clock trace"
timeout -k 10 120 python3 $R/tools/diag/clock_trace.py > $O/clock_trace.txt 2>&1; tail -6 $O/clock_trace.txt
ls $O
