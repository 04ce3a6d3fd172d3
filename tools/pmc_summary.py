#!/usr/bin/env python3
"""Diagnostic: per-kernel means of the counters in a rocprofv3 --pmc run (counter_collection.csv)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ca::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, "launches", max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f" % (c, sum(v) / len(v)))
