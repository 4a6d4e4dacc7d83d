#!/bin/bash
# SQ counters of the rows GEMM kernels alone (GPU box): old kernel (TDNNF_GEMM_RING=0) against the persistent ring.  usage: tools/pmc_gemm.sh OUTDIR [shape filter]
set -e
OUT=$(realpath -m ${1:-gpurun_out/pmc_gemm}); F=${2:-"affine 1/3"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for r in 0 1; do
  export TDNNF_GEMM_RING=$r
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq$r -- python3 $ROOT/tools/gemm_bench.py 2 "$F" > $OUT/sq$r.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/lds$r -- python3 $ROOT/tools/gemm_bench.py 2 "$F" > $OUT/lds$r.log 2>&1 || true
  python3 $ROOT/tools/pmc_summary.py $OUT/sq$r rows_gemm > $OUT/summary$r.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/lds$r rows_gemm >> $OUT/summary$r.txt || true
done
cat $OUT/summary0.txt $OUT/summary1.txt
