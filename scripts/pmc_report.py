#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel: pmc_report.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
import os
f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if pat in k:
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(k, "dispatches", len(disp[k]))
    for c, x in sorted(v.items()):
        print("   %-26s %.4g" % (c, x))
