#!/usr/bin/env python3
"""Per-kernel counter summary of one bench.py command from three rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
cannot share a pass on gfx950; the SQ counters take a third), tagged with the hash of the kernel sources so that
bench.py quotes it only for the build it was taken from.

  counters.py <fetch_dir> <write_dir> <sq_dir>[,<sq_dir2>...] <out.json> <workload> <mode> A N [note]

FETCH_SIZE is in KB and, on gfx950, tallies the 128-B requests of wide coalesced reads at 64 B, so the true value
lies between the reported one and twice it (MI355X_MICROARCH.md, HBM section); both bounds are written
(hbm_bytes_low / hbm_bytes_high).  WRITE_SIZE is exact for streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collision_avoidance_amd import build as b  # noqa: E402

KERNELS = ("nbr_kernel", "step_kernel", "obs_kernel", "lp3_kernel")
ALIAS = {"quad_kernel": "step_kernel", "pair_kernel": "step_kernel"}   # the four- / two-lanes-per-agent solve kernels are reported in the step_kernel slot (the full name is kept)


def means(d):
    acc = defaultdict(lambda: defaultdict(list))
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for name in KERNELS + tuple(ALIAS):
                if name in k:
                    key = ALIAS.get(name, name)
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    names[key] = k.split("(")[0].replace("void ca::", "")
    return ({k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()},
            {k: max(len(v) for v in d.values()) for k, d in acc.items()}, names)


def main():
    fetch, n1, _ = means(sys.argv[1])
    write, n2, _ = means(sys.argv[2])
    sq, n3, names = {}, {}, {}
    for d in sys.argv[3].split(","):   # several SQ passes (8 counters each): wave / wait counters, instruction classes
        sq_d, n_d, names_d = means(d)
        for k, v in sq_d.items():
            sq.setdefault(k, {}).update(v)
        n3.update(n_d); names.update(names_d)
    out_path, workload, mode, A, N = sys.argv[4], sys.argv[5], sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
    step_bytes = 60 if mode == "step" else 52
    alg = {"nbr_kernel": 0, "step_kernel": step_bytes * A * N, "obs_kernel": 256 * A * N, "lp3_kernel": 0}
    out = {"src_sha": b.loaded_sha(), "workload": workload, "mode": mode,
           "note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of `python3 bench.py --workload %s --mode %s "
                   "--steps 20 --warmup 5 --no-cpu-baseline`; means over the launches of a pass; KB per launch; "
                   "hbm_bytes_low uses FETCH_SIZE as reported, hbm_bytes_high doubles it (gfx950 tallies the 128-B "
                   "requests of wide coalesced reads at 64 B). %s" % (workload, mode, " ".join(sys.argv[9:])),
           "launches_per_pass": {"fetch": n1, "write": n2, "sq": n3}, "kernels": {}}
    for k in KERNELS:
        if k not in fetch or k not in write:
            continue
        f, w = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
        d = {"name": names.get(k, k), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_low": (f + w) * 1024,
             "hbm_bytes_high": (2 * f + w) * 1024, "algorithmic_bytes": alg[k]}
        d.update(sq.get(k, {}))
        out["kernels"][k] = d
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
