# kernel trace of the step at one shape (GPU box): tools/shape_trace.sh CHUNK MINIBATCH TAG [bench args...]
# -> gpurun_out/trace_TAG_{shapes,overlap}.txt (+ the bench line in trace_TAG.log)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CH=$1; MB=$2; TAG=$3; shift 3
rm -rf $R/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o r -- python3 $R/bench.py --no-parity --no-cpu-baseline --no-also --no-alt --chunk $CH --minibatch $MB --steps 8 --warmup 4 "$@" > $R/gpurun_out/trace_$TAG.log 2>&1 || exit 1
F=$(ls $R/gpurun_out/trace_$TAG/r_kernel_trace.csv $R/gpurun_out/trace_$TAG/*/r_kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/trace_shapes.py "$F" 8 0.0 > $R/gpurun_out/trace_${TAG}_shapes.txt
python3 $R/tools/stream_overlap.py "$F" 8 > $R/gpurun_out/trace_${TAG}_overlap.txt
python3 $R/tools/trace_timeline.py "$F" > $R/gpurun_out/trace_${TAG}_timeline.txt
rm -rf $R/gpurun_out/trace_$TAG
grep '^{' $R/gpurun_out/trace_$TAG.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['ms_per_step'], d['value'])"
