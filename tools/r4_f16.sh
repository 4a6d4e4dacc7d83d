#!/bin/bash
# GPU box: the plane GEMM tests, the stand-alone shapes, and the f16x3 step (twice)
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_planes_gemm.py tests/test_gpu_net.py -x -q -k "planes or plane or f16 or bf16x6" 2>&1 | tail -3
python3 tools/planes_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/bf16x6.*| f16x3/| f16x3/' | cut -c1-200
for i in 1 2; do
python3 bench.py --gemm f16x3 --no-parity --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3', j['ms_per_step'], [(c['kernel'][-12:], c['tflops']) for c in j['roofline']['all_kernels']])"
done
