set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "TDNNF_GEMM_SERIAL_EPILOGUE=1" "X=1" "TDNNF_GEMM_SERIAL_EPILOGUE=1" "X=1"; do
  run "$cfg 1500x128" $cfg $B --steps 8 --warmup 4
done




