#!/usr/bin/env python3
"""Vector / scalar / LDS instructions per sample and launch of every kernel: pmc_valu_all.py <pmc dir> <samples per launch>
(one rocprofv3 --pmc pass with SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES)."""
import collections, csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"), key=os.path.getmtime)
n = float(sys.argv[2])
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Grid_Size"])
    per.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in per.items():
    if not k[0].startswith("k_") or "SQ_INSTS_VALU" not in c:
        continue
    m = lambda name: sum(c[name]) / max(len(c[name]), 1)
    rows.append((m("SQ_INSTS_VALU") * len(c["SQ_INSTS_VALU"]), k, len(c["SQ_INSTS_VALU"]), m("SQ_INSTS_VALU") * 64 / n, m("SQ_INSTS_SALU") * 64 / n, m("SQ_INSTS_LDS") * 64 / n))
for tot, k, calls, v, s_, l in sorted(rows, reverse=True)[:32]:
    print("%-44s grid=%-10s calls=%3d  VALU %6.1f  SALU %6.1f  LDS %5.1f  per sample (lane instructions)" % (k[0], k[1], calls, v, s_, l))
