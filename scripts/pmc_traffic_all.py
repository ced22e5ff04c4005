#!/usr/bin/env python3
"""HBM traffic per kernel and launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command:
pmc_traffic_all.py <fetch dir> <write dir> <samples per launch>   ->  (2 x FETCH_SIZE + WRITE_SIZE) bytes per sample."""
import collections, csv, glob, os, sys

def load(d, counter):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Grid_Size"])
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
n = float(sys.argv[3])
rows = []
for k in fetch:
    if k not in write or not k[0].startswith("k_"):
        continue
    f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2       # gfx950: FETCH_SIZE counts half (profiles/r03_fetch_calibration.txt)
    w = sum(write[k]) / len(write[k]) * 1024
    rows.append(((f + w) * len(fetch[k]), k, len(fetch[k]), f / n, w / n))
for tot, k, calls, f, w in sorted(rows, reverse=True)[:30]:
    print("%-44s grid=%-10s calls=%3d  read %6.2f B/sample  written %6.2f B/sample" % (k[0], k[1], calls, f, w))
