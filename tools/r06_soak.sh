#!/bin/bash
# round-6 parity soaks on the final sources: tools/r06_soak.sh <tag> <seed offset>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-soak}; S=${2:-600}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'$R'); from collision_avoidance_amd import build as b; print('sources', b.loaded_sha())" > $O/soak_parity_$S.txt
timeout -k 10 1000 python3 $R/tools/soak_parity.py 2048 1200 $S >> $O/soak_parity_$S.txt 2>&1 || { tail -5 $O/soak_parity_$S.txt; exit 1; }
tail -2 $O/soak_parity_$S.txt
