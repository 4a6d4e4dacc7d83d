#!/bin/bash
# GPU box, round 5 first call: the new data-parallel tests, then full timelines (every launch) of the two small shapes
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python3 -m pytest tests/test_gpu_data_parallel.py -x -q -m gpu > gpurun_out/r5_dp_tests.log 2>&1 || { tail -30 gpurun_out/r5_dp_tests.log; exit 1; }
tail -3 gpurun_out/r5_dp_tests.log
TIMELINE_MIN_US=0 timeout -k 10 300 bash tools/r4_prof.sh r5_150x64 --chunk 150 --minibatch 64 --steps 12 > /dev/null
TIMELINE_MIN_US=0 timeout -k 10 300 bash tools/r4_prof.sh r5_1500x16 --chunk 1500 --minibatch 16 --steps 8 > /dev/null
cat gpurun_out/r5_150x64_bench_line.json | cut -c1-300
