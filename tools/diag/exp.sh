cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02l
show() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['value']/1e6,1), d['kernels_ms'])"; }
(cd variants/r01 && timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 300 > ../../gpurun_out/r02l/old.json 2>../../gpurun_out/r02l/err.log) || exit 1
show gpurun_out/r02l/old.json "r01 tree, 300-step warm-up       "
CA_BENCH_MIN_WARM=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 300 > gpurun_out/r02l/new_short.json 2>gpurun_out/r02l/err.log || exit 1
show gpurun_out/r02l/new_short.json "current, 300-step warm-up        "
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 > gpurun_out/r02l/new_long.json 2>gpurun_out/r02l/err.log || exit 1
show gpurun_out/r02l/new_long.json "current, 1 s warm-up             "
CA_BENCH_MIN_WARM=0 timeout -k 10 200 python3 tools/run_variant.py variants/libcaenv_nb16.so --no-cpu-baseline --steps 1500 --warmup 300 > gpurun_out/r02l/nb16_short.json 2>gpurun_out/r02l/err.log || { tail gpurun_out/r02l/err.log; exit 1; }
show gpurun_out/r02l/nb16_short.json "16-bit nb ids, 300-step warm-up  "
(cd variants/r01 && timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 6000 > ../../gpurun_out/r02l/old_long.json 2>../../gpurun_out/r02l/err.log) || exit 1
show gpurun_out/r02l/old_long.json "r01 tree, 6000-step warm-up      "
