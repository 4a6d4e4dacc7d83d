#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_planes_gemm.py tests/test_gpu_net.py -x -q -m gpu -k "planes" 2>&1 | tail -2; python3 - <<P
import sys; sys.path.insert(0,".")
import __graft_entry__ as ge
pkg=ge.load_package(); lib=pkg.hipabi.load(); pkg.hipabi.check(lib.tdnnf_set_option(b"gemm_alt_taps", 2))
import pytest
sys.exit(pytest.main(["tests/test_gpu_net.py","-x","-q","-m","gpu","-k","f16x3"]))
P
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4 --gemm f16x3"
for rep in 1 2 3; do for o in 1 2; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option gemm_alt_taps=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('f16x3 alt=$o', j['ms_per_step'], [(k['kernel'][-7:],k['tflops']) for k in r['all_kernels']])"
done; done
