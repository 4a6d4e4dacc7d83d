# same-box A/B of an environment switch at the recipes' egs shapes (GPU box): tools/ab_env.sh VAR [chunk minibatch]...
V=$1; shift
[ $# -eq 0 ] && set -- 150 64 150 128
while [ $# -ge 2 ]; do
  for rep in 1 2; do for v in 0 1; do
    echo -n "$V=$v chunk $1 x $2: "; env $V=$v python3 tools/host_launch_time.py $1 $2 | tail -1 || exit 1
  done; done
  shift 2
done
