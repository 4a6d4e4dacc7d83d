#!/usr/bin/env python3
"""Where does the host spend a step?  Reads a rocprofv3 --hip-trace CSV (…_hip_api_trace.csv), takes the thread that launches the kernels and, over the
last `steps` steps' worth of calls, prints the time inside HIP calls by function, the time between calls, and the longest gaps with the calls around them.
usage: python tools/host_gaps.py TRACE.csv [launches_per_step]"""
import csv
import sys
from collections import Counter, defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    r = csv.DictReader(f)
    for x in r:
        rows.append((int(x["Thread_Id"]), x["Function"], int(x["Start_Timestamp"]), int(x["End_Timestamp"])))
by_thread = defaultdict(list)
for t, fn, s, e in rows:
    by_thread[t].append((s, e, fn))
main = max(by_thread, key=lambda t: sum(1 for c in by_thread[t] if c[2] == "hipLaunchKernel"))
calls = sorted(by_thread[main])
# nested calls (hipLaunchKernel inside hipLaunchKernelGGL wrappers etc.): keep outermost
flat, end = [], 0
for s, e, fn in calls:
    if s >= end:
        flat.append((s, e, fn))
        end = e
# the tail: the timed region = the calls before the last hipDeviceSynchronize-bounded stretch; take the last 40 % of launches
launch_idx = [i for i, c in enumerate(flat) if c[2] == "hipLaunchKernel"]
lo = launch_idx[int(len(launch_idx) * 0.5)]
hi = launch_idx[int(len(launch_idx) * 0.95)]
seg = flat[lo:hi]
span = seg[-1][1] - seg[0][0]
inside = Counter()
cnt = Counter()
gaps = []
for i, (s, e, fn) in enumerate(seg):
    inside[fn] += e - s
    cnt[fn] += 1
    if i:
        gaps.append((s - seg[i - 1][1], seg[i - 1][2], fn, s))
tot_in = sum(inside.values())
tot_gap = sum(g[0] for g in gaps)
nl = cnt["hipLaunchKernel"]
print("thread %d: %d calls, %d launches over %.1f ms: inside HIP %.1f ms (%.1f us / launch-equivalent), between calls %.1f ms" % (main, len(seg), nl, span / 1e6, tot_in / 1e6, tot_in / 1e3 / nl, tot_gap / 1e6))
for fn, ns in inside.most_common(12):
    print("  %-28s %7d calls %9.2f ms  %7.2f us avg" % (fn, cnt[fn], ns / 1e6, ns / 1e3 / cnt[fn]))
hist = Counter()
for g in gaps:
    b = 0 if g[0] < 1e3 else 1 if g[0] < 1e4 else 2 if g[0] < 1e5 else 3 if g[0] < 1e6 else 4
    hist[b] += g[0]
print("gap time by size: <1us %.2f ms, 1-10us %.2f, 10-100us %.2f, 0.1-1ms %.2f, >1ms %.2f" % tuple(hist[b] / 1e6 for b in range(5)))
pair = Counter()
for g in gaps:
    if g[0] >= 1e5:
        pair[(g[1], g[2])] += g[0]
print("gaps >= 0.1 ms by (call before, call after):")
for k, ns in pair.most_common(12):
    print("  %-28s -> %-28s %9.2f ms" % (k[0], k[1], ns / 1e6))
big = [g for g in gaps if g[0] >= 1e6]
print("%d gaps >= 1 ms; in order (ms since the segment's start: gap ms, launches since the previous one):" % len(big))
t0 = seg[0][0]
starts = [c[0] for c in seg if c[2] == "hipLaunchKernel"]
import bisect
prev = 0
for g in big[:60]:
    k = bisect.bisect_left(starts, g[3])
    print("  %8.1f: %6.2f  (%d launches)  %s -> %s" % ((g[3] - t0) / 1e6, g[0] / 1e6, k - prev, g[1], g[2]))
    prev = k
