# kernel traces of the bench step with and without natural gradient (GPU box): where the preconditioning's cost sits, per kernel and stream
cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-parity --no-cpu-baseline --no-also --no-alt --steps 8 --warmup 4"
for ng in 1 0; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ngcost$ng -o r -- $B --natural-gradient $ng > $GRAFT_REPO_ROOT/gpurun_out/ngcost$ng.log 2>&1
  F=$(ls $GRAFT_REPO_ROOT/gpurun_out/ngcost$ng/r_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/ngcost$ng/*/r_kernel_trace.csv 2>/dev/null | head -1)
  python3 $GRAFT_REPO_ROOT/tools/trace_shapes.py "$F" 8 0.0 > $GRAFT_REPO_ROOT/gpurun_out/ngcost${ng}_shapes.txt
  python3 $GRAFT_REPO_ROOT/tools/stream_overlap.py "$F" 8 > $GRAFT_REPO_ROOT/gpurun_out/ngcost${ng}_overlap.txt
  rm -f "$F"
done
