#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4 --steps 16 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]])'
for rep in 1 2 3; do
for o in "--option xent_behind_den=0" ""; do
  echo -n "[$o] f32: "; timeout -k 10 200 python3 bench.py $Q $o 2>/dev/null | python3 -c "$P"
done
done 2>&1 | tee gpurun_out/r5b_xbd.txt
for o in "--option xent_behind_den=0" "--option xent_behind_den=1"; do
  echo -n "[$o] f16x3: "; timeout -k 10 200 python3 bench.py $Q --gemm f16x3 $o 2>/dev/null | python3 -c "$P"
done 2>&1 | tee -a gpurun_out/r5b_xbd.txt
