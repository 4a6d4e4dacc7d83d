#!/bin/bash
# GPU box: upper bound of what moving the input-side statistics passes off the matrix cores can give (timing only)
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 2 --steps 12 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]], [round(c["ms"],1) for c in d["roofline"]["all_kernels"]])'
for rep in 1 2; do
for o in "" "--option ng_diag_skip=1" "--option ng_diag_skip=3" "--natural-gradient 0"; do
  for g in f32 f16x3; do
  echo -n "[$o] $g: "; python3 bench.py $Q --gemm $g $o 2>/dev/null | python3 -c "$P"
  done
done
done 2>&1 | tee gpurun_out/r5b_skip.txt
