#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of one bench configuration: tools/r4_prof.sh TAG [bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o r -- python3 bench.py --no-parity --no-alt --no-also --no-cpu-baseline "$@" > gpurun_out/$TAG.log 2>&1
grep '^{' gpurun_out/$TAG.log | tail -1 > gpurun_out/${TAG}_bench_line.json
F=$(ls gpurun_out/$TAG/r_kernel_trace.csv gpurun_out/$TAG/*/r_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/stream_overlap.py "$F" 8 > gpurun_out/${TAG}_overlap.txt || true
python3 tools/trace_shapes.py "$F" 8 0.1 > gpurun_out/${TAG}_shapes.txt || true
python3 tools/trace_timeline.py "$F" ${TIMELINE_MIN_US:-100} > gpurun_out/${TAG}_timeline.txt || true
S=$(ls gpurun_out/$TAG/r_kernel_stats.csv gpurun_out/$TAG/*/r_kernel_stats.csv 2>/dev/null | head -1)
cp "$S" gpurun_out/${TAG}_kernel_stats.csv
rm -f "$F"
head -40 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
