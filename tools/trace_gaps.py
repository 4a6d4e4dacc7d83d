#!/usr/bin/env python3
"""Where is no kernel in flight?  From a rocprofv3 --kernel-trace CSV: over the last STEPS steps, the idle intervals (no kernel running on
any queue) longer than MIN_US, with the kernel that ended before and the one that started after each, summed per (before, after) pair.
usage: trace_gaps.py KERNEL_TRACE_CSV [steps] [min_us]"""
import csv
import re
import sys
from collections import defaultdict

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r["Kernel_Name"]))
rows.sort()
marks = [s for s, e, q, n in rows if "den_forward" in n or "den_wide_init" in n or "den_mw_kernel<0>" in n]
t0, t1 = marks[-steps - 1], marks[-1]
win = [r for r in rows if r[1] > t0 and r[0] < t1]


def short(n):
    return re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", n).split("(")[0][:44]


acc = defaultdict(lambda: [0, 0.0])
cur_end, cur_name, total = t0, "(window start)", 0.0
for s, e, q, n in win:
    if s > cur_end:
        gap = (s - cur_end) / 1e3
        total += gap
        if gap >= min_us:
            a = acc[(cur_name, "q%s %s" % (q, short(n)))]
            a[0] += 1
            a[1] += gap
    if e > cur_end:
        cur_end, cur_name = e, "q%s %s" % (q, short(n))
print("idle (no kernel on any queue): %.2f ms per step over %d steps" % (total / 1e3 / steps, steps))
for (a, b), (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%7.1f us/step  %5.1f x/step  after %-48s before %s" % (us / steps, c / steps, a, b))
