set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "TDNNF_NG_SHARE=0" "TDNNF_NG_SHARE=1" "TDNNF_NG_SHARE=0" "TDNNF_NG_SHARE=1"; do
  run "$cfg 150x64" $cfg $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
  run "$cfg 150x128" $cfg $B --chunk 150 --minibatch 128 --steps 32 --warmup 8
  echo "$cfg host_launch"; env $cfg python tools/host_launch_time.py 150 64 2>/dev/null | tail -1
done
