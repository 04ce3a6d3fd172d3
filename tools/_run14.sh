set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04final3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/tools/soak_parity.py 2048 1200 > $O/soak1.txt 2>&1 || { tail -3 $O/soak1.txt; exit 1; }
tail -1 $O/soak1.txt
timeout -k 10 500 python3 $R/tools/soak_parity.py 4096 600 2000 > $O/soak3.txt 2>&1 || { tail -3 $O/soak3.txt; exit 1; }
tail -1 $O/soak3.txt
