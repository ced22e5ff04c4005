#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of `bench.py --roofline-only`)
into the per-kernel traffic files bench.py reads: profiles/r04_pmc_<name>.json.  The kernels of a roofline leg are
the symbols the library's kernel log reported for it (`device_kernels` in the bench line) -- nothing is reconstructed
from radii here -- and every file carries the hash of the kernel sources it was recorded with (`lib_sha16`), which
bench.py compares with the build it runs (mismatch => traffic = null).
usage: pmc_roofline.py <fetch dir> <write dir> <bench json line file> <out dir>"""
import collections, csv, glob, json, os, sys


def load(d, counter):
    files = glob.glob(d + "/*/*_counter_collection.csv")
    if not files:
        sys.exit("pmc_roofline: no counter CSV under %s" % d)
    f = max(files, key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[(r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), r["Grid_Size"])].append(float(r["Counter_Value"]))
    if not per:
        sys.exit("pmc_roofline: no %s rows in %s" % (counter, f))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
doc = json.loads(line)
roof, sha = doc["roofline_kernels"], doc.get("lib_sha16")
out = sys.argv[4]


def kb(per, name):
    """total KB and launches of the kernel with exactly this symbol (largest grid only)"""
    keys = [k for k in per if k[0] == name]
    if not keys:
        return None, 0
    g = max(int(k[1]) for k in keys)
    keys = [k for k in keys if int(k[1]) == g]
    return sum(sum(per[k]) for k in keys), sum(len(per[k]) for k in keys)


for e in roof:
    if "error" in e or "device_kernels" not in e:
        continue
    k = e["kernel"]
    if "SumThreshold" in k:
        name = "sumthreshold"
    elif k.startswith("rejection step"):
        name = "reject"
    else:
        rad = int(k.split("r = ", 1)[1].split(":")[0])
        name = "boxfilter_s%d_r%d" % (0 if "time-axis" in k else 1, rad)
    f = w = 0.0
    launches = 0
    missing = []
    for sym in e["device_kernels"]:
        a, n1 = kb(fetch, sym)
        b, _ = kb(write, sym)
        if a is None or b is None:
            missing.append(sym)
            continue
        f += a; w += b; launches += n1
    if missing:
        print("no counters for", missing, "-- skipped", name)
        continue
    # one MEASUREMENT of a leg = launches_per_measurement launches (a launch pair of the time stage, the six launches of a
    # rejection step -- several of one symbol): totals over everything counted / number of measurements
    nmeas = launches / max(e.get("launches_per_measurement", 1), 1)
    f /= nmeas; w /= nmeas
    hbm = int((2 * f + w) * 1024)       # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM)
    d = {"kernel": k, "device_kernels": e["device_kernels"], "launches_counted": launches, "lib_sha16": sha,
         "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
         "samples_per_launch": e["samples_per_launch"],
         "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
         "hbm_bytes_per_launch": hbm,
         "traffic_over_algorithmic": hbm / e["algorithmic_bytes_per_launch"],
         "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes, scripts/pmc_run.sh) -- python3 bench.py --roofline-only"}
    json.dump(d, open(os.path.join(out, "r04_pmc_%s.json" % name), "w"), indent=1)
    print(name, e["device_kernels"], "traffic/algorithmic = %.3f" % d["traffic_over_algorithmic"])
