#!/usr/bin/env python3
"""tools/pmc_summarize.py <dir> -- per-kernel sums of the rocprofv3 --pmc CSVs written by tools/pmc_run.sh"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "tyr::" not in k:
            continue
        k = k.replace("void ", "").replace("tyr::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(f"== {k}")
    for c in sorted(agg[k]):
        n = calls[k][c]
        print(f"   {c:34s} {agg[k][c]:18.0f}   /dispatch {agg[k][c]/n:16.1f}  ({n} dispatches)")
