#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 2 1; do python3 tools/gemm_bench_small.py 20 splitk_per_cu=$o 2>&1 | grep "1/3-rate\|150x64 .*full" | grep -v wgrad | sed "s/^/spc=$o /"; done
bash tools/r5_ab.sh 3 "" "--option splitk_per_cu=1" 2>&1 | tail -4
