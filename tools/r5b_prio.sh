#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
echo "== f32"; bash tools/ab_lib.sh run --steps 16 --warmup 4
echo "== f16x3"; bash tools/ab_lib.sh run --steps 16 --warmup 4 --gemm f16x3
bash tools/r5b_fin.sh
} 2>&1 | tee gpurun_out/r5b_prio.txt
