#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1"
for rep in 1 2 3; do for q in 4 8; do
  for sh in "--chunk 150 --minibatch 64 --steps 40" "--chunk 1500 --minibatch 16 --steps 16" "--chunk 1500 --minibatch 128 --steps 6"; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py $Q $sh 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HWQ=$q $sh', j['ms_per_step'])"
  done
done; done
