cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02o
show() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['value']/1e6,1), d['kernels_ms'])"; }
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02o/t.log 2>&1 || { tail -30 gpurun_out/r02o/t.log; exit 1; }
tail -1 gpurun_out/r02o/t.log
for wl in C3 C5 C2; do
  (cd variants/r01 && timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline --steps 1500 --warmup 6000 > ../../gpurun_out/r02o/old_$wl.json 2>../../gpurun_out/r02o/err.log) || exit 1
  show gpurun_out/r02o/old_$wl.json "r01 tree $wl"
  timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline --steps 1500 > gpurun_out/r02o/new_$wl.json 2>gpurun_out/r02o/err.log || { tail gpurun_out/r02o/err.log; exit 1; }
  show gpurun_out/r02o/new_$wl.json "current  $wl"
done
timeout -k 10 200 python3 bench.py --mode orca --no-cpu-baseline --steps 1500 > gpurun_out/r02o/new_orca.json 2>gpurun_out/r02o/err.log || exit 1
show gpurun_out/r02o/new_orca.json "current C3 orca"
