#!/bin/bash
# same-box A/B of the persistent LDS-DMA-ring GEMM (TDNNF_GEMM_RING=0|1): kernel shapes alone, then the bench step
set -o pipefail
OUT=gpurun_out/ab_ring
mkdir -p $OUT
for r in 0 1; do
  TDNNF_GEMM_RING=$r timeout -k 10 240 python tools/gemm_bench.py 5 rate > $OUT/gemm_ring$r.log 2>&1 || { echo "gemm_bench ring=$r failed"; tail -5 $OUT/gemm_ring$r.log; exit 1; }
  TDNNF_GEMM_RING=$r timeout -k 10 240 python tools/gemm_bench.py 5 prefinal >> $OUT/gemm_ring$r.log 2>&1
  TDNNF_GEMM_RING=$r timeout -k 10 240 python tools/gemm_bench.py 5 probe1 >> $OUT/gemm_ring$r.log 2>&1
done
paste -d'\n' $OUT/gemm_ring0.log $OUT/gemm_ring1.log | grep " fwd " 
for r in 0 1 0 1; do
  TDNNF_GEMM_RING=$r timeout -k 10 400 python bench.py --no-also --no-alt --no-cpu-baseline --steps 8 --warmup 3 > $OUT/bench_ring$r.json 2> $OUT/bench_ring$r.err || { echo "bench ring=$r failed"; tail -5 $OUT/bench_ring$r.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/bench_ring$r.json").read().strip().splitlines()[-1])
print("ring=$r", d["ms_per_step"], [ (k["kernel"], round(k["tflops"],1)) for k in d["roofline"]["all_kernels"]], d.get("parity"))
PY
done
