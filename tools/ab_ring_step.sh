#!/bin/bash
# same-box A/B at the bench step: old kernels / ring for forward GEMMs only (TDNNF_WT=0) / ring everywhere
set -o pipefail
OUT=gpurun_out/ab_ring
mkdir -p $OUT
i=0
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do
  set -- $cfg
  i=$((i+1))
  TDNNF_GEMM_RING=$1 TDNNF_WT=$2 TDNNF_RING_PRIO=0 timeout -k 10 400 python bench.py --no-also --no-alt --no-cpu-baseline --no-parity --steps 8 --warmup 3 > $OUT/step$i.json 2> $OUT/step$i.err || { echo "bench failed"; tail -5 $OUT/step$i.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/step$i.json").read().strip().splitlines()[-1])
print("ring=$1 wt=$2", d["ms_per_step"], [ (k["kernel"], round(k["tflops"],1), round(k["ms"],1)) for k in d["roofline"]["all_kernels"]])
PY
done
