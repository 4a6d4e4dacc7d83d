#!/bin/bash
# GPU box: synchronised BatchNorm through the library's own one-rank RCCL communicators against none, interleaved (>= 5 runs each)
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1 --chunk 1500 --minibatch 16 --steps 16"
for rep in 1 2 3 4 5 6; do for o in off on; do
  timeout -k 10 200 python3 bench.py $Q --sync-batchnorm $o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('syncbn $rep $o', j['ms_per_step'], j['config']['sync_batchnorm'], (j['config'].get('rccl_library') or '')[-60:])"
done; done | tee gpurun_out/r5_syncbn_raw.txt
python3 - <<'P'
import statistics as st
on=[float(l.split()[3]) for l in open('gpurun_out/r5_syncbn_raw.txt') if l.split()[2]=='on']
off=[float(l.split()[3]) for l in open('gpurun_out/r5_syncbn_raw.txt') if l.split()[2]=='off']
print("sync-BN on : n=%d mean %.3f median %.3f min %.3f max %.3f"%(len(on),st.mean(on),st.median(on),min(on),max(on)))
print("sync-BN off: n=%d mean %.3f median %.3f min %.3f max %.3f"%(len(off),st.mean(off),st.median(off),min(off),max(off)))
print("difference of medians %+.3f ms, of means %+.3f ms (interleaved pairs, one box)"%(st.median(on)-st.median(off), st.mean(on)-st.mean(off)))
P
