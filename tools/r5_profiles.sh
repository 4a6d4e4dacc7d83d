#!/bin/bash
# On the GPU box: the rocprofv3 runs behind profiles/r05_* (DESIGN.md 5).  tools/r5_profiles.sh STAGE...
#   f32      kernel trace + stats of the bench's own command line, then FETCH_SIZE / WRITE_SIZE in passes of their own (counters and traces
#            are never combined), fewer steps -- the per-launch averages do not depend on the count
#   f16x3    the same for --gemm f16x3
#   small    the recipes' egs shape and the 8-GPU shard (150 x 64, 1500 x 16): kernel classes, stream overlap, dispatches per step
#   super    the supernets at 1500 x 128 (kernel classes) and their 8-GPU shards at 1500 x 16 (also stream overlap)
#   sq       SQ counters of the GEMM kernels alone (plain tile kernel, persistent ring, plane kernels)
# Then here: python tools/make_profiles.py r05 r5_stats
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-parity --no-alt --no-also --no-cpu-baseline"
trace_reports() {  # dir tag [marker kernel]
  local F=$(ls gpurun_out/$1/r_kernel_trace.csv gpurun_out/$1/*/r_kernel_trace.csv 2>/dev/null | head -1)
  python3 tools/stream_overlap.py "$F" 8 > gpurun_out/${1}_overlap.txt || true
  python3 tools/trace_shapes.py "$F" 8 0.1 > gpurun_out/${1}_shapes.txt || true
  python3 - "$F" <<'PY' > gpurun_out/${1}_dispatches.txt || true
import csv, sys
sys.path.insert(0, "tools")
from trace_util import step_window
rows = [(int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [s for s, n in rows if "splice_input" in n]
t0, t1, k = step_window(marks, 8)
print("kernel dispatches per step over the last %d steps: %.1f" % (k, sum(1 for s, n in rows if t0 <= s < t1) / float(k)))
PY
  rm -f "$F"
}
stats_run() {  # tag, bench args...
  local t=$1; shift
  rm -rf gpurun_out/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$t -o r -- $B "$@" > gpurun_out/$t.log 2>&1
  grep '^{' gpurun_out/$t.log | tail -1 > gpurun_out/${t}_bench_line.json
  trace_reports $t
  echo "$t: $(python3 -c "import json; d=json.load(open('gpurun_out/${t}_bench_line.json')); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')")"
}
pmc_run() {  # tag, bench args...
  local t=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    local d=gpurun_out/${t}_$(echo $c | tr A-Z a-z | cut -d_ -f1)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- $B "$@" --steps 2 --warmup 1 > $d.log 2>&1
    rm -f $d/r_kernel_trace.csv $d/*/r_kernel_trace.csv
  done
}
for stage in "$@"; do
  case $stage in
    f32) stats_run r5_stats; pmc_run r5_pmc ;;
    f16x3) stats_run r5_stats_f16x3 --gemm f16x3; pmc_run r5_pmc_f16x3 --gemm f16x3 ;;
    small)
      stats_run r5_stats_150x64 --chunk 150 --minibatch 64 --steps 8 --warmup 4
      stats_run r5_stats_1500x16 --chunk 1500 --minibatch 16 --steps 8 --warmup 4 ;;
    super)
      for w in darts-offset darts-offset-cvupdate bn-supernet; do stats_run r5_stats_$w --workload $w --steps 4 --warmup 2 --ng-burn-in 6; done
      for w in darts-offset bn-supernet; do stats_run r5_stats_${w}_1500x16 --workload $w --chunk 1500 --minibatch 16 --steps 8 --warmup 4; done ;;
    sq)
      OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_sq; rm -rf $OUT; mkdir -p $OUT; cd /tmp
      C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
      C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES"
      for r in 0 1; do
        rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/f32_ring${r}_a -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py 2 "affine 1/3" gemm_ring=$r > $OUT/f32_ring${r}_a.log 2>&1
        rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/f32_ring${r}_b -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py 2 "affine 1/3" gemm_ring=$r > $OUT/f32_ring${r}_b.log 2>&1 || true
      done
      rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/planes_a -- python3 $GRAFT_REPO_ROOT/tools/planes_bench.py 2 > $OUT/planes_a.log 2>&1
      rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/planes_b -- python3 $GRAFT_REPO_ROOT/tools/planes_bench.py 2 > $OUT/planes_b.log 2>&1 || true
      cd $GRAFT_REPO_ROOT
      for d in f32_ring0 f32_ring1 planes; do
        python3 tools/pmc_summary.py $OUT/${d}_a gemm > $OUT/summary_$d.txt; python3 tools/pmc_summary.py $OUT/${d}_b gemm >> $OUT/summary_$d.txt || true
      done
      find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
      cat $OUT/summary_planes.txt | head -60 ;;
  esac
done
