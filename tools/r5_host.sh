#!/bin/bash
# GPU box: hip-trace + kernel-trace of a small shape -> is the caller's queue waiting for the host or for events?  tools/r5_host.sh TAG [bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/$TAG
rocprofv3 --hip-trace --kernel-trace --output-format csv -d gpurun_out/$TAG -o r -- python3 bench.py --no-parity --no-alt --no-also --no-cpu-baseline "$@" > gpurun_out/$TAG.log 2>&1
A=$(ls gpurun_out/$TAG/r_hip_api_trace.csv gpurun_out/$TAG/*/r_hip_api_trace.csv 2>/dev/null | head -1)
K=$(ls gpurun_out/$TAG/r_kernel_trace.csv gpurun_out/$TAG/*/r_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/host_vs_gpu.py "$A" "$K" 40 > gpurun_out/${TAG}_host_vs_gpu.txt
python3 tools/host_gaps.py "$A" > gpurun_out/${TAG}_host_gaps.txt || true
grep '^{' gpurun_out/$TAG.log | tail -1 | cut -c1-200
rm -f "$A" "$K"
cat gpurun_out/${TAG}_host_vs_gpu.txt
