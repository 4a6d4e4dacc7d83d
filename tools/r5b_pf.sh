#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/r5b_valu3.sh
timeout -k 10 800 python3 -m pytest tests/test_gpu_ng_valu.py tests/test_gpu_net.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ng or NG or natural or early or one_pass or bench_shape_properties or fused" 2>&1 | tail -4
