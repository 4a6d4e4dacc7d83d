# kernel traces of the step at the recipes' egs shape (chunk 150 x 64 sequences), natural gradient on and off (GPU box); the traces of
# the last 4 steps come back under gpurun_out/ for tools/trace_gaps.py / stream_overlap.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-parity --no-cpu-baseline --no-also --no-alt --chunk 150 --minibatch 64 --steps 8 --warmup 4"
for ng in 1 0; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/recipe$ng -o r -- $B --natural-gradient $ng > $R/gpurun_out/recipe$ng.log 2>&1 || exit 1
  F=$(ls $R/gpurun_out/recipe$ng/r_kernel_trace.csv $R/gpurun_out/recipe$ng/*/r_kernel_trace.csv 2>/dev/null | head -1)
  python3 $R/tools/trace_shapes.py "$F" 8 0.0 > $R/gpurun_out/recipe${ng}_shapes.txt
  python3 $R/tools/stream_overlap.py "$F" 8 > $R/gpurun_out/recipe${ng}_overlap.txt
  python3 - "$F" $R/gpurun_out/recipe${ng}_last4.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "den_forward" in r["Kernel_Name"] or "den_wide_init" in r["Kernel_Name"] or "den_mw_kernel<0>" in r["Kernel_Name"]]
t0 = marks[-5]
keep = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=["Kernel_Name", "Stream_Id", "Queue_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size", "Workgroup_Size"], extrasaction="ignore")
w.writeheader()
w.writerows(keep)
PY
  rm -f "$F"
  grep '^{' $R/gpurun_out/recipe$ng.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ng', $ng, d['ms_per_step'], d['value'])"
done
