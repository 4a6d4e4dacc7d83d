#!/usr/bin/env python3
"""Per (kernel, grid) launch table of the last STEPS steps of a rocprofv3 --kernel-trace CSV: launches/step, average and total
time -- which shapes a kernel class spends its time on.  usage: trace_shapes.py KERNEL_TRACE_CSV [steps] [min_ms_per_step]"""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_util import step_window  # noqa: E402

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
rows.sort()
marks = [s for s, e, q, n, g, w in rows if "splice_input" in n]  # the first kernel of a step's forward pass: once per step (the denominator kernels are not: the multi-workgroup form launches its fallback behind it)
t0, t1, steps = step_window(marks, steps)
acc = {}
for s, e, q, n, g, w in rows:
    if s < t0 or s >= t1:
        continue
    nm = re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n).split("(")[0]
    k = (q, nm, int(g) // max(1, int(w)))
    a = acc.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
print("%-3s %-64s %8s %8s %10s %10s" % ("q", "kernel", "blocks", "n/step", "avg us", "ms/step"))
for (q, nm, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if t / 1e6 / steps < floor:
        continue
    print("%-3s %-64s %8d %8.1f %10.1f %10.3f" % (q, nm[:64], g, n / steps, t / 1e3 / n, t / 1e6 / steps))
