#!/bin/bash
# GPU box: unprofiled host lead at the small shapes, and a few same-box A/B runs
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1"
run() { echo "== $*"; timeout -k 10 200 python3 bench.py $Q "$@" 2>gpurun_out/err.tmp | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', j['ms_per_step'], 'frames/s', j['value'])"; grep "host lead" gpurun_out/err.tmp | cut -c1-400; }
run --chunk 150 --minibatch 64 --steps 40 --host-lead
run --chunk 1500 --minibatch 16 --steps 16 --host-lead
run --chunk 150 --minibatch 64 --steps 40 --option ng_early_in=0
run --chunk 1500 --minibatch 16 --steps 16 --option ng_early_in=0
export GPU_MAX_HW_QUEUES=8
run --chunk 150 --minibatch 64 --steps 40
run --chunk 1500 --minibatch 16 --steps 16
