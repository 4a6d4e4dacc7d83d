set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "TDNNF_DEN_SPLIT=0" "TDNNF_DEN_SPLIT=1"; do
  run "$cfg 1500x128" $cfg $B --steps 8 --warmup 4
  run "$cfg 150x64" $cfg $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
  run "$cfg 1500x128 10k states" $cfg $B --steps 4 --warmup 2 --den-states 10000
  echo "$cfg den_bench 4000"; env $cfg DEN_MODE=1 python tools/den_bench.py 4000 12 128 500 2>/dev/null | tail -1
  echo "$cfg den_bench 10000"; env $cfg DEN_MODE=1 python tools/den_bench.py 10000 12 128 500 2>/dev/null | tail -1
done
