#!/usr/bin/env python3
"""Idle gaps of the caller's stream (queue 1) in a trace_timeline.py listing made with floor 0: every gap >= MIN_US with the launches on
either side.  usage: q1_gaps.py TIMELINE.txt [min_us]"""
import re
import sys
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = []
for ln in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) ms\s+q(\d+)\s+(\S.*?)\s+(\d+) blk\s+([\d.]+) us', ln)
    if m:
        t, q, n, b, d = m.groups()
        ev.append((float(t) * 1e3, int(q), n, float(d)))
q1 = [e for e in ev if e[1] == 1]
tot = 0.0
for a, b in zip(q1, q1[1:]):
    gap = b[0] - (a[0] + a[3])
    if gap >= floor:
        tot += gap
        others = sorted({e[2][:28] for e in ev if e[1] != 1 and e[0] < b[0] and e[0] + e[3] > a[0] + a[3]})
        print("%9.1f us  gap %7.1f us  after %-40s before %-40s | %s" % (a[0] + a[3], gap, a[2][:40], b[2][:40], ", ".join(others)[:150]))
print("total idle in gaps >= %.0f us: %.1f us; q1 busy %.1f us" % (floor, tot, sum(e[3] for e in q1)))
