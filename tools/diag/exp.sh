cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02m
show() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['value']/1e6,1), d['kernels_ms'])"; }
CA_OBS_GPB=2 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3like or c5like or c2like or step_with_actions or golden or full_size" > gpurun_out/r02m/t.log 2>&1 || { tail -20 gpurun_out/r02m/t.log; exit 1; }
tail -1 gpurun_out/r02m/t.log
(cd variants/r01 && timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 6000 > ../../gpurun_out/r02m/old.json 2>../../gpurun_out/r02m/err.log) || exit 1
show gpurun_out/r02m/old.json "r01 tree (6000 warm-up steps)"
for g in 1 2 4; do
  CA_OBS_GPB=$g timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1500 > gpurun_out/r02m/w7_g$g.json 2>gpurun_out/r02m/err.log || { tail gpurun_out/r02m/err.log; exit 1; }
  show gpurun_out/r02m/w7_g$g.json "7 waves/SIMD gpb $g"
  CA_OBS_GPB=$g timeout -k 10 200 python3 tools/run_variant.py variants/libcaenv_w6.so --no-cpu-baseline --steps 1500 > gpurun_out/r02m/w6_g$g.json 2>gpurun_out/r02m/err.log || { tail gpurun_out/r02m/err.log; exit 1; }
  show gpurun_out/r02m/w6_g$g.json "6 waves/SIMD gpb $g"
done
for g in 1 2 4 8; do
  CA_OBS_GPB=$g timeout -k 10 200 python3 bench.py --workload C5 --no-cpu-baseline --steps 1500 > gpurun_out/r02m/c5_g$g.json 2>gpurun_out/r02m/err.log || exit 1
  show gpurun_out/r02m/c5_g$g.json "C5 7 waves/SIMD gpb $g"
done
