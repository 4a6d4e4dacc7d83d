#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 1; do python3 tools/gemm_bench_small.py 20 splitk_partial_round=$o 2>&1 | grep "1500x16 .* full-rate" | grep -v wgrad | sed "s/^/spr=$o /"; done
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/r5_ab.sh 3 "" "--option splitk_partial_round=0" 2>&1 | tail -4
