set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "GPU_MAX_HW_QUEUES=4 TDNNF_WGRAD_STREAM=1 TDNNF_NG_STREAMS=1" "GPU_MAX_HW_QUEUES=8 TDNNF_WGRAD_STREAM=1 TDNNF_NG_STREAMS=1" "GPU_MAX_HW_QUEUES=8 TDNNF_WGRAD_STREAM=1 TDNNF_NG_STREAMS=4" "GPU_MAX_HW_QUEUES=12 TDNNF_WGRAD_STREAM=1 TDNNF_NG_STREAMS=4" "GPU_MAX_HW_QUEUES=8 TDNNF_WGRAD_STREAM=0 TDNNF_NG_STREAMS=4"; do
  run "$cfg 150x64" $cfg $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
  run "$cfg 150x128" $cfg $B --chunk 150 --minibatch 128 --steps 30 --warmup 8
  run "$cfg 1500x128" $cfg $B --steps 8 --warmup 4
done
