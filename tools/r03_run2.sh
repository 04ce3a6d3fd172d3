#!/bin/bash
# Round 3, GPU batch 2: the quad kernel against the lane kernel (CA_QUAD=0/1), per-step launches against ca_rollout chunks
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r03b}; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline"
run() { # name env... -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" $B "$@" > $O/$name.json 2>>$O/bench.err || { echo "$name FAILED"; tail -5 $O/bench.err; return 1; }
  python3 -c "import json;d=json.load(open('$O/$name.json'));print('%-28s %8.1f M  %s' % ('$name', d['value']/1e6, d['kernels_ms']))"
}
run C2_step_lane CA_QUAD=0 -- --workload C2 &&
run C2_step_quad CA_QUAD=1 -- --workload C2 &&
run C2_orca_lane_c1 CA_QUAD=0 -- --workload C2 --mode orca --rollout-chunk 1 &&
run C2_orca_quad_c1 CA_QUAD=1 -- --workload C2 --mode orca --rollout-chunk 1 &&
run C2_orca_lane_c50 CA_QUAD=0 -- --workload C2 --mode orca &&
run C2_orca_quad_c50 CA_QUAD=1 -- --workload C2 --mode orca &&
run C2_orca_quad_c200 CA_QUAD=1 -- --workload C2 --mode orca --rollout-chunk 200 &&
run C3_step_quad CA_QUAD=1 -- --workload C3 &&
run C3_orca_quad_c50 CA_QUAD=1 -- --workload C3 --mode orca &&
run C3_orca_lane_c50 CA_QUAD=0 -- --workload C3 --mode orca &&
echo done
