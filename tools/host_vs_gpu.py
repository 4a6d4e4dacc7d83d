#!/usr/bin/env python3
"""Is a gap on the caller's queue the host's or the GPU's?  Reads rocprofv3 --hip-trace --kernel-trace CSVs (same run), takes the last
step (between the last two splice_input kernels) and prints, for every kernel of the busiest queue that starts >= MIN_US after its
predecessor on that queue ended: when the host CALLED its launch relative to the predecessor's end.  Called long before the gap ended =
the GPU waited on an event; called at the gap's end = the host was late.
usage: host_vs_gpu.py HIP_API_TRACE.csv KERNEL_TRACE.csv [min_us]"""
import csv
import re
import sys
from collections import Counter
api, ker = sys.argv[1], sys.argv[2]
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
call = {}
with open(api) as f:
    for r in csv.DictReader(f):
        if "Launch" in r["Function"]:
            call[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = []
with open(ker) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r["Correlation_Id"]))
rows.sort()
marks = [s for s, e, q, n, c in rows if "splice_input" in n]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if t0 <= r[0] < t1]
qmain = Counter(r[2] for r in step).most_common(1)[0][0]
q1 = [r for r in step if r[2] == qmain]
short = lambda n: re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n).split("(")[0][:44]
first_call = min(call[r[4]][0] for r in step if r[4] in call)
print("step %.2f ms on the GPU; its first launch was called %.2f ms before the first kernel started" % ((t1 - t0) / 1e6, (t0 - first_call) / 1e6))
for a, b in zip(q1, q1[1:]):
    gap = (b[0] - a[1]) / 1e3
    if gap < floor or b[4] not in call:
        continue
    cs, ce = call[b[4]]
    print("%9.3f ms  gap %7.1f us before %-44s  launch called %8.1f us %s the predecessor ended (returned %6.1f us later)" %
          ((a[1] - t0) / 1e6, gap, short(b[3]), abs(cs - a[1]) / 1e3, "AFTER" if cs > a[1] else "before", (ce - cs) / 1e3))
# how far ahead is the host over the step: for every 50th kernel of the queue, call time vs start time
print("host lead (kernel start - launch call), every 40th launch of the queue:")
for r in q1[::40]:
    if r[4] in call:
        print("   %9.3f ms  %-44s lead %9.1f us" % ((r[0] - t0) / 1e6, short(r[3]), (r[0] - call[r[4]][0]) / 1e3))
