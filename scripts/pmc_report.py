#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel: pmc_report.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
import os
files = glob.glob(d + "/*/*_counter_collection.csv")
if not files:
    sys.exit("pmc_report: no counter CSV under %s (did the profiled run fail? see gpurun_out/pmc_passes.log)" % d)
f = max(files, key=os.path.getmtime)
import time
if time.time() - os.path.getmtime(f) > 3600:
    sys.exit("pmc_report: %s is older than an hour -- a stale pass, not this run's" % f)
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
