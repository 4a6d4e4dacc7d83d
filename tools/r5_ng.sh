#!/bin/bash
# GPU box: what natural gradient costs per phase (on / off), full size and the small shapes
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1 --phases"
for sh in "--chunk 1500 --minibatch 128 --steps 8" "--chunk 1500 --minibatch 16 --steps 16" "--chunk 150 --minibatch 64 --steps 40"; do
 for ng in 1 0; do
  echo "== $sh ng=$ng"; timeout -k 10 300 python3 bench.py $Q $sh --natural-gradient $ng 2>gpurun_out/err.tmp | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', j['ms_per_step'])"; grep "phases" gpurun_out/err.tmp
 done
done
