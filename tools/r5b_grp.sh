#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 600 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "early or natural or ng" 2>&1 | tail -4
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --no-alt --chunk 150 --minibatch 64 --steps 8 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parity', d.get('parity'), d['ms_per_step'])"
bash tools/r5_ab.sh 3 "--option ng_early_in=0" ""
} 2>&1 | tee gpurun_out/r5b_grp.txt
