#!/bin/bash
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["alt"]["ms_per_step"])'
for rep in 1 2; do
for q in 4 8 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q f32, alt: "; GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --no-parity --no-cpu-baseline --no-also --steps 20 --warmup 5 2>/dev/null | python3 -c "$P"
done
done 2>&1 | tee gpurun_out/r5b_alt2.txt
