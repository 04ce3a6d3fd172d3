#!/usr/bin/env python3
"""Diagnostic: idle time between consecutive kernels of the step loop, from a rocprofv3 --kernel-trace CSV.
usage: launch_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ca::", "")))
rows.sort()
rows = rows[len(rows) // 2:]          # the settled second half
gaps, durs = {}, {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps.setdefault((n0[:24], n1[:24]), []).append(s1 - e0)
    durs.setdefault(n0[:24], []).append(e0 - s0)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
    v = np.array(v)
    print("gap %-26s -> %-26s n=%5d  p10/p50/p90 = %6.2f / %6.2f / %6.2f us" % (k[0], k[1], len(v), np.percentile(v, 10) / 1e3, np.median(v) / 1e3, np.percentile(v, 90) / 1e3))
for k, v in sorted(durs.items(), key=lambda kv: -len(kv[1]))[:4]:
    v = np.array(v)
    print("dur %-26s n=%5d  p10/p50/p90 = %6.2f / %6.2f / %6.2f us" % (k, len(v), np.percentile(v, 10) / 1e3, np.median(v) / 1e3, np.percentile(v, 90) / 1e3))
