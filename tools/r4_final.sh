#!/bin/bash
# GPU box: the round's final measurements -- the default bench line, and the one-GPU rehearsal of the library-issued sync-BN exchange
cd "$GRAFT_REPO_ROOT"
Q="--no-parity --no-cpu-baseline --no-also --no-alt"
for rep in 1 2; do for sb in off on; do
  python3 bench.py $Q --chunk 1500 --minibatch 16 --steps 16 --warmup 4 --sync-batchnorm $sb 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1500x16 sync-batchnorm $sb', d['ms_per_step'], 'ms', d['config'].get('sync_batchnorm'))"
done; done > gpurun_out/r4_syncbn.txt 2>&1
cat gpurun_out/r4_syncbn.txt
python3 bench.py > gpurun_out/r4_final.json 2> gpurun_out/r4_final.err
tail -c 600 gpurun_out/r4_final.json
