#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4"
for rep in 1 2; do for o in 0 2; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option ng_bk=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j.get('roofline_secondary',{}); print('ng_bk=$o full', j['ms_per_step'], 'ng class GB/s', s.get('achieved'), 'ms/step', s.get('ms_per_step'))"
done; done
