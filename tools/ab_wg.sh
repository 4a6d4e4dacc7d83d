set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_net.py -m gpu -q -x -k "chain or net" > gpurun_out/ap_t.log 2>&1; tail -2 gpurun_out/ap_t.log
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])'
for i in 1 2; do $B --steps 8 --warmup 4 2>/dev/null | python -c "$P"; done
$B --chunk 150 --minibatch 64 --steps 40 --warmup 8 2>/dev/null | python -c "$P"
