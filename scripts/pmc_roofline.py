#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of `bench.py --roofline-only`)
into the per-kernel traffic files bench.py reads: profiles/r03_pmc_<name>.json.
usage: pmc_roofline.py <fetch dir> <write dir> <bench json line file> <out dir>"""
import collections, csv, glob, json, os, sys

def load(d, counter):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
roof = json.loads(line)["roofline_kernels"]
out = sys.argv[4]

def kb(per, pat, nth_largest_grid=0):
    """mean KB per launch of the kernels whose name contains pat (largest grid only)"""
    keys = [k for k in per if pat in k[0]]
    if not keys:
        return None, 0, []
    g = max(int(k[1]) for k in keys)
    keys = [k for k in keys if int(k[1]) == g]
    tot = sum(sum(per[k]) / len(per[k]) for k in keys)     # one launch of every matching kernel (both images)
    return tot, sum(len(per[k]) for k in keys), sorted(k[0] for k in keys)

def emit(name, entry, pats, note):
    f = w = 0.0
    names = []
    launches = 0
    for p in pats:
        a, n1, k1 = kb(fetch, p)
        b, n2, k2 = kb(write, p)
        if a is None or b is None:
            print("no counters for", p)
            return
        f += a; w += b; names += k1; launches += n1
    hbm = int((2 * f + w) * 1024)       # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM)
    d = {"kernel": entry["kernel"], "device_kernels": names, "launches_counted": launches,
         "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
         "samples_per_launch": entry["samples_per_launch"],
         "algorithmic_bytes_per_launch": entry["algorithmic_bytes_per_launch"],
         "hbm_bytes_per_launch": hbm,
         "traffic_over_algorithmic": hbm / entry["algorithmic_bytes_per_launch"],
         "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes, scripts/pmc_any.sh) -- python3 bench.py --roofline-only",
         "note": note}
    json.dump(d, open(os.path.join(out, "r03_pmc_%s.json" % name), "w"), indent=1)
    print(name, "traffic/algorithmic = %.3f" % d["traffic_over_algorithmic"])

for e in roof:
    k = e["kernel"]
    if "error" in e:
        continue
    if k.startswith("k_colst_mask"):
        emit("sumthreshold", e, ["k_colst_mask"], "dword-per-lane row loads (x2 on FETCH_SIZE, calibrated in round 1 on this kernel's known byte count), byte stores")
    elif "time-axis" in k:
        rad = int(k.rsplit("r = ", 1)[1])
        pats = (["k_boxq_deep<"] if (88 <= 2 * rad < 96 or 104 <= 2 * rad < 112) else ["k_boxq<"]) if 56 <= 2 * rad < 112 else (["k_boxt"] if rad >= 16 else ["k_colfilter_lds<2"])
        emit("boxfilter_s0_r%d" % rad, e, pats, "one launch per image (weight: packed flag words in, float32 out; data: + float32 in); dword-per-lane accesses")
    elif "frequency-axis" in k:
        rad = int(k.rsplit("r = ", 1)[1])
        if 34 <= 2 * rad < 112:
            pat = "k_boxqf<%d, 1," % (2 * rad // 16 * 16)
        else:
            ks = 80 if 2 * rad >= 80 else 64 if 2 * rad >= 64 else 32 if 2 * rad >= 32 else 16 if 2 * rad >= 16 else 8
            pat = "k_boxf<%d, %s, 1," % (ks, "true" if 2 * rad > ks else "false")
        emit("boxfilter_s1_r%d" % rad, e, [pat], "both images + amplitudes in, |data - background| out; 64-byte row segments in, dword-per-lane out")
