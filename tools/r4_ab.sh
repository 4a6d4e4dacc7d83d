#!/bin/bash
# same-box A/B of bench configurations (GPU box): tools/r4_ab.sh "ARGS A" "ARGS B" ... ; each twice, interleaved
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --steps 8 --warmup 4"
for rep in 1 2; do
  for spec in "$@"; do
    python3 bench.py $Q $spec 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %8.3f ms  %10.0f frames/s' % ('$spec', d['ms_per_step'], d['value']))"
  done
done
