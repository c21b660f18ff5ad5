#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean over dispatches.
Usage: python tools/pmc_summary.py DIR [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "skr_"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if want not in row["Kernel_Name"]:
                continue
            per_dispatch[(row["Kernel_Name"], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, d, c), v in per_dispatch.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    print(k)
    for c, vs in sorted(cs.items()):
        print("  %-28s mean %.6g  (n=%d)" % (c, sum(vs) / len(vs), len(vs)))
