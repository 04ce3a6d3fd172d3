#!/bin/bash
# where does four-lanes-per-agent stop paying?  C2-shaped arenas (16 agents, K = 5) and C3-shaped (64 agents, K = 10), lane vs quad
R=$GRAFT_REPO_ROOT; cd /tmp
for spec in "C2 1024" "C2 2048" "C2 4096" "C2 8192" "C2 16384" "C3 128" "C3 256" "C3 512" "C3 1024"; do
  set -- $spec
  for q in 0 1; do
    for mode in step orca; do
      CA_QUAD=$q python3 $R/bench.py --workload $1 --arenas $2 --mode $mode --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 A=$2 quad=$q $mode %8.1f M  %.2f us/step  %s' % (d['value']/1e6, d['ms_per_step']*1e3, d['kernels_ms']))" || exit 1
    done
  done
done
