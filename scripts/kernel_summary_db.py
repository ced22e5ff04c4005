"""Per-kernel summary (name, grid, calls, total / average time) of a rocprofv3 --kernel-trace results
database (the default sqlite output); scripts/kernel_summary.py does the same for the CSV output.
    python scripts/kernel_summary_db.py gpurun_out/prof_x/p_results.db [rows]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = ("select s.display_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.end - d.start "
     "from %s d join %s s on d.kernel_id = s.id" % (kd, ks))
agg = collections.OrderedDict()
for n, gx, gy, gz, dur in c.execute(q):
    if not (n.startswith('k_') or n.startswith('void k_')):
        continue
    a = agg.setdefault((n.split('(')[0][:44], gx, gy, gz), [0, 0.0])
    a[0] += 1
    a[1] += dur / 1e3
tot = sum(a[1] for a in agg.values())
print('total ms', tot / 1e3)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:rows]:
    print("%-44s grid=%10d,%5d,%4d calls=%3d total_ms=%8.2f avg_us=%9.1f %5.1f%%"
          % (k[0], k[1], k[2], k[3], a[0], a[1] / 1e3, a[1] / a[0], 100 * a[1] / tot))
