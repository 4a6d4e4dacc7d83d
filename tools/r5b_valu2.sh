#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 300 python3 -m pytest tests/test_gpu_ng_valu.py -x -q -m gpu 2>&1 | tail -15
timeout -k 10 300 python3 tools/ng_pass_bench.py 10
} 2>&1 | tee gpurun_out/r5b_valu2.txt
