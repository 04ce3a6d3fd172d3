#!/usr/bin/env python3
"""Diagnostic: HBM bytes per launch of the three kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950).  Usage: hbm_traffic.py <fetch_dir> <write_dir> <out.json> A N
FETCH_SIZE is in KB and, on gfx950, tallies 128-B requests of wide coalesced reads at 64 B, so the true
value lies between the reported one and twice it (MI355X_MICROARCH.md, HBM section); both are written."""
import csv
import glob
import json
import sys
from collections import defaultdict


def means(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for name in ("nbr_kernel", "step_kernel", "obs_kernel"):
                if name in k:
                    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, \
        {k: max(len(v) for v in d.values()) for k, d in acc.items()}


fetch, n1 = means(sys.argv[1])
write, n2 = means(sys.argv[2])
A, N = int(sys.argv[4]), int(sys.argv[5])
alg = {"nbr_kernel": 0, "step_kernel": 60 * A * N, "obs_kernel": 256 * A * N}
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py --steps 20 --warmup 5 "
               "--no-cpu-baseline`; KB per launch, mean over the launches of the pass; hbm_bytes_low uses FETCH_SIZE as "
               "reported, hbm_bytes_high doubles it (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B)",
       "launches_per_pass": {"fetch": n1, "write": n2}, "kernels": {}}
for k in ("nbr_kernel", "step_kernel", "obs_kernel"):
    if k not in fetch or k not in write:   # the neighbour search is normally fused into step_kernel
        continue
    f, w = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
    out["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_low": (f + w) * 1024,
                         "hbm_bytes_high": (2 * f + w) * 1024, "algorithmic_bytes": alg[k]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
