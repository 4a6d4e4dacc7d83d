#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1 --phases"
for rep in 1 2; do
for sh in "--chunk 150 --minibatch 64 --steps 40" "--chunk 1500 --minibatch 16 --steps 16" "--chunk 1500 --minibatch 128 --steps 6" "$@"; do
  echo "== $sh"; timeout -k 10 300 python3 bench.py $Q $sh 2>gpurun_out/err.tmp | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', j['ms_per_step'])"; grep "phases" gpurun_out/err.tmp
done; done
