"""One training step's kernel timeline from a rocprofv3 kernel_trace.csv: the dispatches between the last two fused_adam launches,
in start order, with start offset, duration and the idle gap in front of each.  usage: prof_timeline.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "fused_adam" in r["Kernel_Name"]]
lo, hi = adam[-3] + 1, adam[-1] + 1      # two fused_adam launches per step (decay / no-decay group)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
gaps = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-44:]
    gap = s - prev_end
    gaps += max(gap, 0)
    print("%9.1f us  dur %7.1f  gap %6.1f  grid=%-8s wg=%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                                                 r.get("Workgroup_Size_X", ""), name))
    prev_end = max(prev_end, e)
print("step span %.1f us, idle gaps %.1f us, %d launches" % ((prev_end - t0) / 1e3, gaps / 1e3, hi - lo))
