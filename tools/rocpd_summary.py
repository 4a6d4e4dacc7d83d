#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME).
usage: rocpd_summary.py DB [steps] [top]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(db.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel ms {tot:.2f}  ({tot / steps:.2f} per step over {steps} steps)")
for name, n, ms in rows[:top]:
    nm = re.sub(r"\(anonymous namespace\)::|tdnnf::", "", name)[:120]
    print(f"{ms:9.2f} ms {ms / steps:8.3f}/step {n:6d} avg {1e3 * ms / n:8.1f} us  {nm}")
