#!/bin/bash
# round-6 parity soaks on the final sources: tools/r06_soak.sh <tag> <seed offset> [<seed offset> ...] [alan]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-soak}; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for S in "$@"; do
  if [ "$S" = alan ]; then
    timeout -k 10 600 python3 $R/tools/soak_alan.py 600 60 > $O/soak_alan.txt 2>&1 || { tail -5 $O/soak_alan.txt; exit 1; }
    tail -2 $O/soak_alan.txt
    continue
  fi
  python3 -c "import sys; sys.path.insert(0,'$R'); from collision_avoidance_amd import build as b; print('sources', b.loaded_sha())" > $O/soak_parity_$S.txt
  timeout -k 10 600 python3 $R/tools/soak_parity.py 2048 1200 $S >> $O/soak_parity_$S.txt 2>&1 || { tail -5 $O/soak_parity_$S.txt; exit 1; }
  tail -1 $O/soak_parity_$S.txt
done
