#!/bin/bash
# same-box A/B of TDNNF_DEN_MW (denominator recursions with several workgroups per sequence) at the small-minibatch shapes
for shape in "1500 16" "1500 32" "150 64" "150 16"; do
  set -- $shape
  for v in 0 1 0 1; do
    TDNNF_DEN_MW=$v timeout -k 10 400 python bench.py --no-also --no-alt --no-cpu-baseline --no-parity --chunk $1 --minibatch $2 --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('den_mw=$v chunk $1 x $2:', d['ms_per_step'])"
  done
done
