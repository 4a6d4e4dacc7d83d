set -e
python -m pytest tests -m gpu -q -x -k "chain or denominator" > gpurun_out/r2_wide_t.log 2>&1
export DEN_MODE=2
python tools/den_bench.py 30000 12 128 500 > gpurun_out/r2_wide_b.log 2>&1
for sl in 1 4 8; do echo "slices $sl" >> gpurun_out/r2_wide_b.log; TDNNF_WIDE_SLICES=$sl python tools/den_bench.py 30000 12 128 500 >> gpurun_out/r2_wide_b.log 2>&1; done
echo serial >> gpurun_out/r2_wide_b.log; TDNNF_WIDE_SERIAL=1 python tools/den_bench.py 30000 12 128 500 >> gpurun_out/r2_wide_b.log 2>&1
echo sg16 >> gpurun_out/r2_wide_b.log; TDNNF_WIDE_SG=16 python tools/den_bench.py 30000 12 128 500 >> gpurun_out/r2_wide_b.log 2>&1
python tools/den_bench.py 10000 12 128 500 >> gpurun_out/r2_wide_b.log 2>&1
python tools/den_bench.py 4000 12 128 500 >> gpurun_out/r2_wide_b.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wide5 -- python3 $GRAFT_REPO_ROOT/tools/den_bench.py 30000 12 128 500 > $GRAFT_REPO_ROOT/gpurun_out/r2_wide_p.log 2>&1
TDNNF_WIDE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wide5s -- python3 $GRAFT_REPO_ROOT/tools/den_bench.py 30000 12 128 500 > $GRAFT_REPO_ROOT/gpurun_out/r2_wide_p.log 2>&1
