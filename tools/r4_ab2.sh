for i in 1 2 3; do
python3 bench.py --gemm f16x3 --no-parity --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3 $*', j['ms_per_step'], [(c['kernel'][-8:], c['ms'], c['tflops']) for c in j['roofline']['all_kernels']])"
done
