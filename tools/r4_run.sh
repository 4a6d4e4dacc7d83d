cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_net.py -x -q -k "planes" > gpurun_out/r4_net_planes2.log 2>&1; tail -5 gpurun_out/r4_net_planes2.log
python -m pytest tests/test_gpu_data_parallel.py -x -q > gpurun_out/r4_dp.log 2>&1; tail -8 gpurun_out/r4_dp.log
Q="--no-parity --no-cpu-baseline --no-also --no-alt --steps 8 --warmup 4"
for g in f32 f16x3 bf16x6; do python bench.py $Q --gemm $g > gpurun_out/r4_b_$g.json 2> gpurun_out/r4_b_$g.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/r4_b_$g.json').read().strip().splitlines()[-1]); print('$g', d['ms_per_step'], d['value'], [(k['kernel'], k.get('ms'), k.get('tflops')) for k in d['roofline'].get('all_kernels', [])])"; done
