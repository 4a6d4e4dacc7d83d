#!/bin/bash
# same-box A/B of TDNNF_NG_EARLY_IN (input-side natural-gradient statistics ahead of the backward pass) at three shapes
set -o pipefail
for shape in "1500 128" "1500 16" "150 64"; do
  set -- $shape
  for v in 0 1 0 1; do
    TDNNF_NG_EARLY_IN=$v timeout -k 10 400 python bench.py --no-also --no-alt --no-cpu-baseline --no-parity --chunk $1 --minibatch $2 --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('early=$v chunk $1 x $2:', d['ms_per_step'])"
  done
done
