#!/bin/bash
# the whole GPU suite WITHOUT -x (every failure listed), then optionally bench lines: tools/gpu_suite.sh <tag> [bench]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-s}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -q > $O/gpu_tests.log 2>&1; rc=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -40
if [ "$2" = bench ]; then
  B="python3 $R/bench.py --no-cpu-baseline"
  timeout -k 10 300 $B > $O/bench_C3_step.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
  timeout -k 10 300 $B --mode orca > $O/bench_C3_orca.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
  for W in C2 C5; do timeout -k 10 300 $B --workload $W > $O/bench_$W.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }; done
  python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); print("%-28s %8.1f M  %s  verified=%s" % (f.split("/")[-1], d["value"]/1e6, {k:v for k,v in d["kernels_ms"].items() if isinstance(v,float)}, d["verified"]["bit_exact"]))
PY
fi
exit $rc
