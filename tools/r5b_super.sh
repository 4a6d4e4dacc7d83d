#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 2 --steps 8 --warmup 3 --ng-burn-in 6"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"])'
for rep in 1 2; do
for w in darts-offset bn-supernet; do
for o in "--option xent_behind_den=0 --option ng_early_fork=0 --option ng_pform=0" ""; do
  echo -n "$w [$o]: "; timeout -k 10 300 python3 bench.py $Q --workload $w $o 2>/dev/null | python3 -c "$P"
done
done
done 2>&1 | tee gpurun_out/r5b_super.txt
