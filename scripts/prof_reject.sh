# GPU box: per-launch durations of the rejection step's kernels (bench.py --roofline-only under rocprofv3 --kernel-trace)
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_rej
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rej -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $GRAFT_REPO_ROOT/gpurun_out/prof_rej.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/reject_launches.txt
import csv, glob
f = max(glob.glob("gpurun_out/prof_rej/*/*_kernel_trace.csv"))
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_mr_", "k_median_reject"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t0 is None else (s - t0) / 1e3
    print("%-18s %8.1f us   gap after previous %6.1f us   grid %s" % (r["Kernel_Name"].split("(")[0], (e - s) / 1e3, gap, "%s x %s x %s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])))
    t0 = e
PY
rm -rf gpurun_out/prof_rej
cat gpurun_out/reject_launches.txt
