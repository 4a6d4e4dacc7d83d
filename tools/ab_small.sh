# A/B runs of environment switches at the small shapes (GPU box): tools/ab_small.sh "VAR=VAL ..." ...
R=$GRAFT_REPO_ROOT; cd $R
Q="--no-parity --no-cpu-baseline --no-also --no-alt"
run() {  # label, env..., shape args
  local label="$1"; shift
  for shape in "--chunk 150 --minibatch 64 --steps 40 --warmup 8" "--chunk 1500 --minibatch 16 --steps 16 --warmup 4"; do
    env "$@" python3 bench.py $Q $shape 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['config']['frames_per_chunk'], d['config']['sequences_per_gpu'], d['ms_per_step'])"
  done
}
for spec in "$@"; do
  run "$spec" $spec
done
