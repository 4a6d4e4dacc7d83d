#!/bin/bash
# GPU box: the denominator alone (128, 64 and 16 sequences, 500 frames), its tests, and the step in both arithmetics (twice)
python3 tools/den_bench.py 4000 12 128 500 2>&1 | tail -1
python3 tools/den_bench.py 4000 12 64 500 2>&1 | tail -1
python3 tools/den_bench.py 4000 12 16 500 2>&1 | tail -1
python3 -m pytest tests -m gpu -x -q -k "chain or denominator or objective" 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --no-parity --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', j['ms_per_step'], 'f16x3', j['alt']['ms_per_step'], [(h['kernel'], h['ms_per_step']) for h in j['roofline_hbm']])"
done
