#!/bin/bash
# headline shape: the trainer's one-kernel backward denominator (default above 96 sequences) against the split form with two workgroups per
# sequence, the recursions one after the other (TDNNF_DEN_TRAINER_SPLIT=1), and against the split form with one workgroup per sequence
for cfg in "0 1" "1 1" "1 0" "0 1" "1 1"; do
  set -- $cfg
  TDNNF_DEN_TRAINER_SPLIT=$1 TDNNF_DEN_MW_SEQ=$2 timeout -k 10 400 python bench.py --no-also --no-alt --no-cpu-baseline --no-parity --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('trainer_split=$1 mw_seq=$2:', d['ms_per_step'])"
done
