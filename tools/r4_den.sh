#!/bin/bash
# GPU box: the denominator alone (128, 64 and 16 sequences, 500 frames) for the two builds of tools/ab_lib.sh, its tests on the new one, and the step
for v in old new old new; do
  cp ab_libs/lib_$v.so tdnn-f_nas_amd/libtdnnf_hip.so
  for b in 128 16; do echo -n "$v: "; python3 tools/den_bench.py 4000 12 $b 500 2>&1 | tail -1; done
done
python3 -m pytest tests -m gpu -x -q -k "chain or denominator or objective" 2>&1 | tail -2
bash tools/ab_lib.sh run
bash tools/ab_lib.sh run --gemm f16x3
bash tools/ab_lib.sh run --chunk 1500 --minibatch 16 --steps 16
