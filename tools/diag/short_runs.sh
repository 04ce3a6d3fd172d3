for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 100 --warmup 5" ""; do python bench.py --no-cpu-baseline $a 2>>gpurun_out/r03w.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['steps'], round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2), d['kernels_ms'])"; done
