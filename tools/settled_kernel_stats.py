#!/usr/bin/env python3
"""Average duration of each kernel over its LAST n launches in a rocprofv3 --kernel-trace output directory: the crowd of
the benchmark settles over its first seconds, so the whole-run averages of `--stats` mix a young crowd
in; the last launches of a 2000-step region are the settled kernel that bench.py's live event timing reports.
  settled_kernel_stats.py <trace_dir> [n=500]   -> CSV on stdout"""
import csv
import glob
import sys
from collections import defaultdict

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 500
rows = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("kernel,launches_total,launches_averaged,avg_ns_last_n,avg_ns_all,min_ns_last_n,max_ns_last_n")
for k, v in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    v.sort()
    last = v[-n:]
    dl = [e - s for s, e in last]
    da = [e - s for s, e in v]
    print('"%s",%d,%d,%.1f,%.1f,%d,%d' % (k.replace('"', "'"), len(v), len(dl), sum(dl) / len(dl), sum(da) / len(da), min(dl), max(dl)))
