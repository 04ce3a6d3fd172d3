#!/bin/bash
# What limits C5 (256 arenas x 512 agents)?  The same workload at 128 / 256 / 512 / 1024 / 2048 arenas, full step and
# ORCA-only: a rate that doubles from 256 to 512 arenas says "fill" (one workgroup per CU), a flat one "pair_kernel".
#   tools/c5_batch_sweep.sh <tag>   -> gpurun_out/<tag>/C5_batch_sweep.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-c5sweep}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
OUT=$O/C5_batch_sweep.txt
echo "# C5 batch sweep: bench.py --workload C5 --arenas A (512 agents per arena), settled crowd, verified lines" > $OUT
echo "# arenas mode    G agent-steps/s   ms/step   kernels_ms   verified" >> $OUT
for A in 128 256 512 1024 2048; do
  for M in step orca; do
    timeout -k 10 300 python3 $R/bench.py --workload C5 --arenas $A --mode $M --no-cpu-baseline --steps 1000 --warmup 100 > $O/c5_${A}_$M.json 2>> $O/err.txt || { tail $O/err.txt; exit 1; }
    python3 -c "
import json;d=json.load(open('$O/c5_${A}_$M.json'))
print('%6d %-5s %8.3f G  %8.4f ms  %s  verified=%s' % ($A,'$M',d['value']/1e9,d['ms_per_step'],{k:round(v,4) for k,v in d['kernels_ms'].items() if isinstance(v,float)},d['verified']['bit_exact']))" | tee -a $OUT
  done
done
