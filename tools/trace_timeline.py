#!/usr/bin/env python3
"""The last step of a rocprofv3 --kernel-trace CSV as a timeline: every launch >= MIN_US in start order with its queue, grid, start offset,
duration and the number of launches of OTHER queues in flight at its start.  usage: trace_timeline.py KERNEL_TRACE_CSV [min_us]"""
import csv
import re
import sys

floor = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
rows.sort()
marks = [s for s, e, q, n, g, w in rows if "splice_input" in n]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if t0 <= r[0] < t1]
print("step of %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
for s, e, q, n, g, w in step:
    if (e - s) / 1e3 < floor:
        continue
    nm = re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n).split("(")[0]
    others = [re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n2).split("(")[0][:24] for s2, e2, q2, n2, g2, w2 in step if q2 != q and s2 <= s < e2]
    print("%9.3f ms  q%-2s %-58s %6d blk %9.1f us   %s" % ((s - t0) / 1e6, q, nm[:58], int(g) // max(1, int(w)), (e - s) / 1e3, ", ".join(others)))
