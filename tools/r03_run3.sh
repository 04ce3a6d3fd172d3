#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-r03c}; O=$R/gpurun_out/$T
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CA_QUAD=1 timeout -k 10 600 python3 $R/tools/stamps.py C2 orca > $O/stamps_quad_C2_orca.txt 2>&1 || { tail $O/stamps_quad_C2_orca.txt; exit 1; }
cat $O/stamps_quad_C2_orca.txt
CA_QUAD=1 timeout -k 10 600 python3 $R/tools/stamps.py C2 step > $O/stamps_quad_C2_step.txt 2>&1 || { tail $O/stamps_quad_C2_step.txt; exit 1; }
cat $O/stamps_quad_C2_step.txt
