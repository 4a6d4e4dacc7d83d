#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 300 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "grouped_weight" 2>&1 | tail -3
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "f16" 2>&1 | tail -3
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 2 --steps 10 --warmup 3 --ng-burn-in 6 --gemm f16x3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for rep in 1 2 3; do
for w in darts-offset bn-supernet; do
for o in "--option planes_group=0" ""; do
  echo -n "$w [$o] f16x3: "; timeout -k 10 300 python3 bench.py $Q --workload $w $o 2>/dev/null | python3 -c "$P"
done
done
done
} 2>&1 | tee gpurun_out/r5b_pg2.txt
