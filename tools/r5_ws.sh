#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 30000; do python3 tools/gemm_bench_small.py 20 wgrad_small=$o 2>&1 | grep "wgrad" | sed "s/^/ws=$o /"; done
timeout -k 10 300 python3 - <<'P'
import sys; sys.path.insert(0,'.')
import __graft_entry__ as ge
pkg=ge.load_package(); lib=pkg.hipabi.load()
pkg.hipabi.check(lib.tdnnf_set_option(b"wgrad_small", 30000))
import pytest
sys.exit(pytest.main(["tests/test_gpu_parity.py","-x","-q","-m","gpu","-k","update"]))
P
bash tools/r5_ab.sh 3 "" "--option wgrad_small=12000" "--option wgrad_small=30000" 2>&1 | tail -6
