#!/bin/bash
# GPU box: the small shapes (150 x 64, 1500 x 16, the supernet shards) with one and with two weight-gradient streams, same box
Q="--no-parity --no-cpu-baseline --no-also --no-alt"
for rep in 1 2; do for o in 1 -1; do
  for sh in "--chunk 150 --minibatch 64 --steps 40" "--chunk 1500 --minibatch 16 --steps 16"; do
    python3 bench.py $Q $sh --option wgrad_stream=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream=$o', '$sh', j['ms_per_step'], j['value'])"
  done
done; done
for o in 1 -1; do for w in darts-offset bn-supernet; do
  python3 bench.py $Q --workload $w --chunk 1500 --minibatch 16 --steps 12 --option wgrad_stream=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream=$o', '$w 1500x16', j['ms_per_step'], j['value'])"
done; done
