#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 1; do python3 tools/gemm_bench.py 10 "ngshape in" ng_bk=$o 2>&1 | grep fwd | sed "s/^/ng_bk=$o /"; done
timeout -k 10 200 python3 -m pytest tests/test_gpu_ng_group.py tests/test_gpu_update_ng.py -x -q -m gpu 2>&1 | tail -2
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4"
for rep in 1 2; do for o in 0 1; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option ng_bk=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j.get('roofline_secondary',{}); print('ng_bk=$o full', j['ms_per_step'], 'ng class GB/s', s.get('achieved'), 'ms/step', s.get('ms_per_step'))"
done; done
