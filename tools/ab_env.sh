#!/bin/bash
# several environment settings on the same box: tools/ab_env.sh "A=1 B=2" "A=2" ... ; prints ms/step and the class table of each (twice)
for rep in 1 2; do
  for cfg in "$@"; do
    printf "%-44s " "$cfg"
    env $cfg python bench.py --no-parity --no-alt --no-also $BENCH_ARGS | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [(k['kernel'][-7:], round(k['ms']/d['steps'],1), k['tflops']) for k in d['roofline']['all_kernels']])"
  done
done
