"""Steady-state per-kernel summary from a rocprofv3 kernel_trace.csv: only dispatches in the last `frac` of the
traced time range are counted (skips warm-up / autotune).  usage: prof_summary.py trace.csv [frac] [steps_in_window]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = list(csv.DictReader(open(path)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < cut:
        continue
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
    agg[key][0] += 1
    agg[key][1] += e - s
tot = sum(v[1] for v in agg.values())
busy_window = t1 - cut
print("window %.2f ms, kernel time %.2f ms (%.1f%% busy)" % (busy_window / 1e6, tot / 1e6, 100.0 * tot / busy_window))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-62s grid=%-9s wg=%-4s n=%-5d total=%8.3f ms avg=%8.1f us  %5.2f%%" % (k[0], k[1], k[2], v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100.0 * v[1] / tot))
