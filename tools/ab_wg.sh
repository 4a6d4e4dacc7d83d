set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
run "150x64" X=1 $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
run "150x128" X=1 $B --chunk 150 --minibatch 128 --steps 32 --warmup 8
run "1500x128" X=1 $B --steps 8 --warmup 4
python tools/host_launch_time.py 150 64
