#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 600 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "early" 2>&1 | tail -3
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4 --steps 16 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]])'
for rep in 1 2 3; do
for o in "--option ng_early_fork=0" ""; do
  for g in f32 f16x3; do
  echo -n "[$o] $g: "; timeout -k 10 200 python3 bench.py $Q --gemm $g $o 2>/dev/null | python3 -c "$P"
  done
done
done
} 2>&1 | tee gpurun_out/r5b_fork.txt
