#!/bin/bash
# GPU box: alternating tap order -- parity, stand-alone GEMMs, PMC traffic of the 160-wide forward GEMM, the step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for o in 0 1; do python3 tools/gemm_bench.py 10 "tdnnf.linear" gemm_alt_taps=$o 2>&1 | grep -v wgrad | sed "s/^/alt=$o /"; done
for o in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/r5_alt${o}_$c; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- python3 tools/gemm_bench.py 2 "tdnnf.linear full" gemm_alt_taps=$o > $d.log 2>&1
    f=$(ls $d/*counter_collection.csv $d/*/*counter_collection.csv 2>/dev/null | head -1)
    python3 - "$f" "$o" "$c" <<'P'
import csv,sys,collections
t=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].split("(")[0][-60:]
    if "rows_gemm" in n or "wgrad" in n: t[n][0]+=1; t[n][1]+=float(r["Counter_Value"])
for n,(k,v) in t.items(): print("alt=%s %s %-62s launches %d  MB/launch %.1f"%(sys.argv[2],sys.argv[3],n,k,v/k/1024))
P
    rm -rf $d
  done
done
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4"
for rep in 1 2; do for o in 0 1; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option gemm_alt_taps=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('alt=$o full', j['ms_per_step'], [(k['kernel'][-7:],k['tflops']) for k in r['all_kernels']])"
done; done
