#!/bin/bash
# GPU box, round 5 second session: (1) old / new library at the headline shape, exact f32 and f16x3 (finalize kernels as four-wave blocks);
# (2) early input statistics on their own stream at the small shapes.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== f32 headline old/new"; bash tools/ab_lib.sh run --steps 12 --warmup 4
echo "== f16x3 headline old/new"; bash tools/ab_lib.sh run --steps 12 --warmup 4 --gemm f16x3
cp ab_libs/lib_new.so tdnn-f_nas_amd/libtdnnf_hip.so
echo "== small shapes, ng_early_in 3"; bash tools/r5_ab.sh 3 "" "--option ng_early_in=3"
} 2>&1 | tee gpurun_out/r5b_first.txt
