#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_bench.sh VAR A B [bench args]; prints ms/step of both, twice
var=$1; a=$2; b=$3; shift 3
for rep in 1 2; do
  for v in $a $b; do
    printf "%s=%s " $var $v
    env $var=$v python bench.py --no-cpu-baseline --no-alt "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [(k['kernel'][-12:], round(k['ms'],1)) for k in d['roofline']['all_kernels']])"
  done
done
