#!/bin/bash
# PMC passes over the wide denominator (GPU box): L2 hit rate, fabric fetch bytes, SQ wait breakdown.  usage: tools/pmc_den.sh OUTDIR [states] [T]
set -e
OUT=$(realpath -m ${1:-gpurun_out/pmc_den}); ST=${2:-30000}; T=${3:-60}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export DEN_MODE=2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 $ROOT/tools/den_bench.py $ST 12 128 $T > $OUT/l2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/den_bench.py $ST 12 128 $T > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ROOT/tools/den_bench.py $ST 12 128 $T > $OUT/sq.log 2>&1
