#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - <<P
import sys; sys.path.insert(0,".")
import __graft_entry__ as ge
pkg=ge.load_package(); lib=pkg.hipabi.load(); pkg.hipabi.check(lib.tdnnf_set_option(b"reverse_passes", 3))
import pytest
sys.exit(pytest.main(["tests/test_gpu_net.py","tests/test_gpu_parity.py","-x","-q","-m","gpu","-k","NG or batchnorm or 7q-shape-small"]))
P
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 4"
for rep in 1 2; do for o in 0 1 2 3; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --option reverse_passes=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h={e['kernel']:e['ms_per_step'] for e in j['roofline_hbm']}; print('rev=$o', j['ms_per_step'], h)"
done; done
for rep in 1 2; do for o in 0 3; do
  timeout -k 10 300 python3 bench.py $Q --steps 8 --gemm f16x3 --option reverse_passes=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3 rev=$o', j['ms_per_step'])"
done; done
