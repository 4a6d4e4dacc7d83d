#!/usr/bin/env python3
"""Which stream is the step waiting for?  From a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): over the last STEPS
steps (delimited by the once-per-step splice_input kernel) the busy time of every queue, the time it is the ONLY queue with a
kernel in flight, and the time no kernel is in flight at all.
usage: stream_overlap.py KERNEL_TRACE_CSV [steps]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_util import step_window  # noqa: E402

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        q = r.get("Stream_Id") or r.get("Queue_Id")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "%s/%s" % (r.get("Queue_Id"), q), r["Kernel_Name"]))
rows.sort()
marks = [s for s, e, q, n in rows if "splice_input" in n]  # the first kernel of a step's forward pass: once per step (the denominator kernels are not: the multi-workgroup form launches its fallback behind it)
t0, t1, steps = step_window(marks, steps)
win = [(max(s, t0), min(e, t1), q, n) for s, e, q, n in rows if e > t0 and s < t1]
ev = []
for s, e, q, n in win:
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
active = {}
busy, only = {}, {}
idle = 0
last = t0
for t, d, q in ev:
    dt = t - last
    live = [k for k, v in active.items() if v > 0]
    if not live:
        idle += dt
    for k in live:
        busy[k] = busy.get(k, 0) + dt
    if len(live) == 1:
        only[live[0]] = only.get(live[0], 0) + dt
    active[q] = active.get(q, 0) + d
    last = t
span = (t1 - t0) / 1e6
print("window %.2f ms = %d steps of %.2f ms; no kernel in flight %.2f ms/step" % (span, steps, span / steps, idle / 1e6 / steps))
for q in sorted(busy, key=lambda k: -busy[k]):
    n = sum(1 for w in win if w[2] == q)
    print("queue/stream %-10s %6d launches/step  busy %7.2f ms/step  alone %7.2f ms/step" % (q, n // steps, busy[q] / 1e6 / steps, only.get(q, 0) / 1e6 / steps))
# what runs while a queue is alone: top kernels by alone time
for q in only:
    acc = {}
    # recompute: time when this queue is alone, attributed to its kernels
    others = sorted((s, e) for s, e, qq, n in win if qq != q)
    merged = []
    for s, e in others:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [m[0] for m in merged]
    for s, e, qq, n in win:
        if qq != q:
            continue
        cov = 0
        i = max(0, bisect.bisect_right(starts, s) - 1)
        while i < len(merged) and merged[i][0] < e:
            cov += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        name = n.split("(")[0][-60:]
        acc[name] = acc.get(name, 0) + (e - s - cov)
    print("alone time of", q)
    for name, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
        print("   %8.3f ms/step  %s" % (v / 1e6 / steps, name))
