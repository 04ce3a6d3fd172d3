cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02p
show() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['value']/1e6,1), d['kernels_ms'])"; }
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02p/t.log 2>&1 || { tail -30 gpurun_out/r02p/t.log; exit 1; }
tail -1 gpurun_out/r02p/t.log
cp collision_avoidance_amd/libcaenv.so /tmp/cur.so; cp variants/libcaenv_bw.so collision_avoidance_amd/libcaenv.so
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or step_with" > gpurun_out/r02p/t_bw.log 2>&1 || { tail -30 gpurun_out/r02p/t_bw.log; exit 1; }
tail -1 gpurun_out/r02p/t_bw.log
cp /tmp/cur.so collision_avoidance_amd/libcaenv.so
for rep in 1 2; do
(cd variants/r01 && timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 6000 > ../../gpurun_out/r02p/old.json 2>../../gpurun_out/r02p/err.log) || exit 1
show gpurun_out/r02p/old.json "r01 tree        "
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 > gpurun_out/r02p/new.json 2>gpurun_out/r02p/err.log || { tail gpurun_out/r02p/err.log; exit 1; }
show gpurun_out/r02p/new.json "current (branch)"
timeout -k 10 200 python3 tools/run_variant.py variants/libcaenv_bw.so --no-cpu-baseline --steps 1500 > gpurun_out/r02p/bw.json 2>gpurun_out/r02p/err.log || { tail gpurun_out/r02p/err.log; exit 1; }
show gpurun_out/r02p/bw.json "bitwise clip    "
done
