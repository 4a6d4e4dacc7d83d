#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc counter_collection CSV.  usage: pmc_summary.py DIR_OR_CSV [name filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|tdnnf::|void ", "", r["Kernel_Name"]).split("(")[0]
        if flt and flt not in n:
            continue
        a = acc[n][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for n, cs in acc.items():
    print(n[:60])
    for c, (k, v) in sorted(cs.items()):
        print("   %-24s launches %6d  mean %16.1f" % (c, k, v / k))
