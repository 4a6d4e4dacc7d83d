#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 2 --steps 12 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]], [round(c["ms"],1) for c in d["roofline"]["all_kernels"]])'
for rep in 1 2; do
for o in "--option ng_pform=0" "--option ng_pform=1"; do
  for g in f32 f16x3; do
  echo -n "[$o] $g: "; timeout -k 10 200 python3 bench.py $Q --gemm $g $o 2>/dev/null | python3 -c "$P"
  done
done
done
} 2>&1 | tee gpurun_out/r5b_valu3.txt
