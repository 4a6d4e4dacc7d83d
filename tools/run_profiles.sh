#!/bin/bash
# On the GPU box: the rocprofv3 runs behind profiles/ (DESIGN.md 5).  1: kernel trace + stats of the bench's own command line
# (timed region = the default 8 steps after 4 warm-ups and 10 setup minibatches); 2, 3: FETCH_SIZE / WRITE_SIZE in passes of
# their own (counters and traces are never combined), fewer steps -- the per-launch averages do not depend on the count.
# Then: python tools/make_profiles.py r02 r3_stats
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-parity --no-alt --no-also"
rm -rf gpurun_out/r3_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats -o r -- $B > gpurun_out/r3_stats.log 2>&1
grep '^{' gpurun_out/r3_stats.log | tail -1 > gpurun_out/r3_stats_bench_line.json
F=$(ls gpurun_out/r3_stats/r_kernel_trace.csv gpurun_out/r3_stats/*/r_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/stream_overlap.py "$F" 8 > gpurun_out/r3_stats_overlap.txt
python3 tools/trace_shapes.py "$F" 8 0.1 > gpurun_out/r3_stats_shapes.txt
python3 - "$F" <<'PY' > gpurun_out/r3_stats_dispatches.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [s for s, n in rows if "den_forward" in n or "den_wide_init" in n or "den_mw_kernel<0>" in n]
t0, t1 = marks[-9], marks[-1]
print("kernel dispatches per step over the last 8 steps: %.1f" % (sum(1 for s, n in rows if t0 <= s < t1) / 8.0))
PY
rm -f "$F"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o r -- $B --steps 2 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o r -- $B --steps 2 --warmup 1 > gpurun_out/pmc_write.log 2>&1
rm -f gpurun_out/pmc_fetch/r_kernel_trace.csv gpurun_out/pmc_write/r_kernel_trace.csv gpurun_out/pmc_*/*/r_kernel_trace.csv
# the supernets and the small shapes: kernel-class tables (rocprofv3 --stats) and the dispatch count per step
for w in darts-offset darts-offset-cvupdate bn-supernet; do
  rm -rf gpurun_out/r3_stats_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats_$w -o r -- $B --workload $w --steps 4 --warmup 2 --ng-burn-in 6 > gpurun_out/r3_stats_$w.log 2>&1
  grep '^{' gpurun_out/r3_stats_$w.log | tail -1 > gpurun_out/r3_stats_${w}_bench_line.json
  rm -f gpurun_out/r3_stats_$w/r_kernel_trace.csv gpurun_out/r3_stats_$w/*/r_kernel_trace.csv
done
for shape in "150 64" "1500 16"; do
  set -- $shape
  t=r3_stats_${1}x${2}
  rm -rf gpurun_out/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$t -o r -- $B --chunk $1 --minibatch $2 --steps 8 --warmup 4 > gpurun_out/$t.log 2>&1
  grep '^{' gpurun_out/$t.log | tail -1 > gpurun_out/${t}_bench_line.json
  F=$(ls gpurun_out/$t/r_kernel_trace.csv gpurun_out/$t/*/r_kernel_trace.csv 2>/dev/null | head -1)
  python3 tools/stream_overlap.py "$F" 8 > gpurun_out/${t}_overlap.txt
  python3 - "$F" <<'PY' > gpurun_out/${t}_dispatches.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [s for s, n in rows if "splice_input" in n]
t0, t1 = marks[-9], marks[-1]
print("kernel dispatches per step over the last 8 steps: %.1f" % (sum(1 for s, n in rows if t0 <= s < t1) / 8.0))
PY
  rm -f "$F"
done
ls -la gpurun_out/r3_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
