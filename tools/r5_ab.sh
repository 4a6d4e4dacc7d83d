#!/bin/bash
# GPU box: interleaved same-box A/B of option sets at the two small shapes.  usage: r5_ab.sh REPS "optset1" "optset2" ...   (an optset = bench.py flags, may be empty)
cd "$GRAFT_REPO_ROOT"
REPS=$1; shift
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1"
for rep in $(seq $REPS); do
  i=0
  for o in "$@"; do
    for sh in "--chunk 150 --minibatch 64 --steps 40" "--chunk 1500 --minibatch 16 --steps 16"; do
      timeout -k 10 200 python3 bench.py $Q $sh $o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB set$i [$o] $sh', j['ms_per_step'])"
    done
    i=$((i+1))
  done
done | tee gpurun_out/r5_ab_raw.txt
python3 - <<'P'
import re,collections
d=collections.defaultdict(list)
for ln in open('gpurun_out/r5_ab_raw.txt'):
    m=re.match(r'AB (set\d+ \[.*?\]) --chunk (\d+) --minibatch (\d+) --steps \d+ ([\d.]+)',ln)
    if m: d[(m.group(1),m.group(2)+'x'+m.group(3))].append(float(m.group(4)))
for k,v in sorted(d.items()): print("%-60s %-8s mean %.3f min %.3f  %s"%(k[0],k[1],sum(v)/len(v),min(v),v))
P
