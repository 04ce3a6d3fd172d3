#!/bin/bash
# quick check on the GPU box: tools/quick_check.sh <tag> [bench] -- the GPU suite (unless "bench" is given), then bench lines of C3 / C2 / C5 (full step and ORCA-only)
# with their verified.bit_exact verdicts, under gpurun_out/<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-q}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "$2" != bench ]; then
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
fi
B="python3 $R/bench.py --no-cpu-baseline"
timeout -k 10 300 $B > $O/bench_C3_step.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
timeout -k 10 300 $B --mode orca > $O/bench_C3_orca.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
for W in C2 C5; do timeout -k 10 300 $B --workload $W > $O/bench_$W.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }; done
timeout -k 10 300 $B --workload C2 --mode orca > $O/bench_C2_orca.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
timeout -k 10 300 $B --workload C5 --mode orca > $O/bench_C5_orca.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); print("%-28s %8.1f M  %s  verified=%s" % (f.split("/")[-1], d["value"]/1e6, {k:v for k,v in d["kernels_ms"].items() if isinstance(v,float)}, d["verified"]["bit_exact"]))
PY
