#!/bin/bash
# GPU box: the round's final default bench line (what the driver runs) and the sync-BN interleaved A/B
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err || { tail -5 gpurun_out/r5_final_bench.err; exit 1; }
python3 - <<'P'
import json
j=json.loads(open('gpurun_out/r5_final_bench.json').read().strip().splitlines()[-1])
print('value',j['value'],'ms',j['ms_per_step'],'frac',j['roofline']['frac'], 'alt', j['alt']['value'], j['alt']['ms_per_step'])
for a in j['also']: print('  ', a['what'][:90], a['ms_per_step'], a.get('value'))
print('parity', j['parity']); print('cpu', j['cpu_baseline'])
P
bash tools/r5_syncbn.sh > gpurun_out/r5_syncbn.txt 2>&1; tail -4 gpurun_out/r5_syncbn.txt
