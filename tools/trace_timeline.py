#!/usr/bin/env python3
"""Timeline of the LAST step of a rocprofv3 --kernel-trace CSV: per stream, the kernels in launch order with start offset and
duration (one line each, coalescing runs of the same kernel name) -- what runs beside what, where the gaps are.
usage: trace_timeline.py KERNEL_TRACE_CSV"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r["Kernel_Name"]))
rows.sort()
marks = [s for s, e, q, n in rows if "splice_input" in n]
t0, t1 = marks[-2], marks[-1]
step = [(s, e, q, re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n).split("(")[0][:44]) for s, e, q, n in rows if t0 <= s < t1]
print("step of %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
qs = sorted({q for _, _, q, _ in step})
for q in qs:
    ks = [(s, e, n) for s, e, qq, n in step if qq == q]
    busy = sum(e - s for s, e, _ in ks)
    print("\n== queue %s: %d launches, busy %.3f ms, first at %.3f, last ends %.3f" % (q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6))
    i = 0
    while i < len(ks):
        j = i
        while j + 1 < len(ks) and ks[j + 1][2] == ks[i][2]:
            j += 1
        dur = sum(e - s for s, e, _ in ks[i:j + 1])
        gap = (ks[i][0] - ks[i - 1][1]) / 1e3 if i else 0.0
        print("  %9.3f  +%7.1f us gap  %3d x %-44s %8.1f us" % ((ks[i][0] - t0) / 1e6, gap, j - i + 1, ks[i][2], dur / 1e3))
        i = j + 1
