#!/bin/bash
# round 5, first GPU call: the GPU suite with the new episode-end / settled-regime tests, then bench lines with --verify
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "$1" != bench ]; then
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
fi
B="python3 $R/bench.py"
timeout -k 10 300 $B --steps 20 --warmup 5 > $O/bench_C3_step_driver_like.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
timeout -k 10 300 $B > $O/bench_C3_step.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
for W in C2 C5; do timeout -k 10 300 $B --workload $W --no-cpu-baseline > $O/bench_$W.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }; done
timeout -k 10 300 $B --workload C2 --mode orca --no-cpu-baseline > $O/bench_C2_orca.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
timeout -k 10 300 $B --mode alan --workload A16 --steps 1000 --warmup 200 --no-cpu-baseline > $O/bench_alan_A16.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); print("%-40s %8.1f M  %s  verified=%s" % (f.split("/")[-1], d["value"]/1e6, d["kernels_ms"], d.get("verified")))
PY
