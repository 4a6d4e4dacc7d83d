#!/bin/bash
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["alt"]["ms_per_step"])'
for rep in 1 2; do
for o in "--option planes_group=0" ""; do
  echo -n "[$o] f32, alt: "; timeout -k 10 300 python3 bench.py --no-parity --no-cpu-baseline --no-also --steps 20 --warmup 5 $o 2>/dev/null | python3 -c "$P"
done
done 2>&1 | tee gpurun_out/r5b_alt.txt
