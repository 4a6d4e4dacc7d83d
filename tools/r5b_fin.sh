#!/bin/bash
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 1"
for rep in 1 2 3; do
for v in old new; do
  cp ab_libs/lib_$v.so tdnn-f_nas_amd/libtdnnf_hip.so
  for sh in "--chunk 150 --minibatch 64 --steps 40" "--chunk 1500 --minibatch 16 --steps 16"; do
    echo -n "$v $sh: "; timeout -k 10 200 python3 bench.py $Q $sh 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"
  done
done
done 2>&1 | tee gpurun_out/r5b_fin.txt
cp ab_libs/lib_new.so tdnn-f_nas_amd/libtdnnf_hip.so
