#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for o in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/r5_alt${o}_$c; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- python3 tools/gemm_bench.py 2 "tdnnf." gemm_alt_taps=$o > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$o" "$c" <<'P'
import csv,sys,collections,re
t=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r'(rows_gemm\w*<[^>]*>|wgrad_kernel<[^>]*>)', r["Kernel_Name"])
    if m: k=(m.group(1), r.get("Grid_Size") or r.get("Grid_Size_X")); t[k][0]+=1; t[k][1]+=float(r["Counter_Value"])
for (n,g),(k,v) in sorted(t.items()): print("alt=%s %s %-50s grid %-9s launches %d  MB/launch %.1f"%(sys.argv[2],sys.argv[3],n,g,k,v/k/1024))
P
    rm -rf $d
  done
done
