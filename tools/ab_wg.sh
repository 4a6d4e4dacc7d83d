set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "TDNNF_DEN_MODE=1" "TDNNF_DEN_MODE=2"; do
  run "$cfg 1500x128 10k states" $cfg $B --steps 6 --warmup 3 --den-states 10000
  run "$cfg 1500x128 7k states" $cfg $B --steps 6 --warmup 3 --den-states 7000
  run "$cfg 1500x128 4k states" $cfg $B --steps 6 --warmup 3 --den-states 4000
done
