cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-parity --no-cpu-baseline --no-also --no-alt --steps 6 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ap1 -o r -- $B > $GRAFT_REPO_ROOT/gpurun_out/ap1.log 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/ap1/r_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/ap1/*/r_kernel_trace.csv
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py tests/test_gpu_parity.py -m gpu -q -x -k "batchnorm or dropout or net_matches" > gpurun_out/ap_t.log 2>&1; tail -2 gpurun_out/ap_t.log
