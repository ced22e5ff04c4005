#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the stream_mix kernels, whose bytes are known by construction (calibrates the counters per access shape)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_$c -- $GRAFT_REPO_ROOT/scripts/ubench/stream_mix.bin > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
n = 1008 * 1024 * 4096
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(root + "pmc_cal_%s/*/*_counter_collection.csv" % c), key=os.path.getmtime)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        kb = sum(v) / len(v)
        print("%-10s %-40s %.4g KB per launch = %.3f B per sample" % (c, k, kb, kb * 1024 / n))
PY
