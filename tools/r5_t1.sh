#!/bin/bash
# GPU box: the trainer's parity tests after the grouped optimizer step / wgrad lag 3, then an interleaved A/B of the new options
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py tests/test_gpu_parity.py tests/test_gpu_update_ng.py tests/test_gpu_data_parallel.py -x -q -m gpu > gpurun_out/r5_t1_tests.log 2>&1 || { tail -40 gpurun_out/r5_t1_tests.log; exit 1; }
tail -3 gpurun_out/r5_t1_tests.log
bash tools/r5_ab.sh 3 "" "--option wgrad_lag=1" "--option wgrad_on_caller=1" "--option ng_early_in=0" "--option wgrad_on_caller=1 --option wgrad_lag=1" > gpurun_out/r5_ab1.log 2>&1
tail -12 gpurun_out/r5_ab1.log
