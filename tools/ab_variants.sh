#!/bin/bash
# A/B of library variants on one box: the product library against variants/libcaenv_<name>.so (built by hand with extra
# flags), alternating twice.  usage: tools/ab_variants.sh <tag> <name> [<name> ...]   (BENCH_ARGS="--workload C5" to change the workload)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r03d}; shift; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS=${BENCH_ARGS:---workload C3}
for rep in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then lib=$R/collision_avoidance_amd/libcaenv.so; else lib=$R/variants/libcaenv_$v.so; fi
    timeout -k 10 300 python3 $R/tools/run_variant.py $lib --no-cpu-baseline $ARGS > $O/${v}_$rep.json 2>>$O/bench.err || { echo "$v failed"; tail -5 $O/bench.err; exit 1; }
    python3 -c "import json;d=json.load(open('$O/${v}_$rep.json'));print('%-12s rep $rep %8.1f M  %s' % ('$v', d['value']/1e6, {k:v for k,v in d['kernels_ms'].items() if isinstance(v,float)}))"
  done
done
