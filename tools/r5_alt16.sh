#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_planes_gemm.py tests/test_gpu_net.py -x -q -m gpu -k "planes" 2>&1 | tail -2
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4 --gemm f16x3"
for rep in 1 2 3; do for o in 0 1; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option gemm_alt_taps=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('f16x3 alt=$o', j['ms_per_step'], [(k['kernel'][-7:],k['tflops']) for k in r['all_kernels']])"
done; done
