set -e
B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]])'
run() { echo "$1"; shift; env "$@" 2>/dev/null | python -c "$P"; }
for cfg in "TDNNF_GEMM_SERIAL_EPILOGUE=1" "X=1" "TDNNF_GEMM_SERIAL_EPILOGUE=1" "X=1"; do
  run "$cfg 1500x128" $cfg $B --steps 8 --warmup 4
done
run "bf16x3 serial" TDNNF_GEMM_SERIAL_EPILOGUE=1 $B --steps 8 --warmup 4 --gemm bf16x3
run "bf16x3 new" X=1 $B --steps 8 --warmup 4 --gemm bf16x3
run "150x64 serial" TDNNF_GEMM_SERIAL_EPILOGUE=1 $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
run "150x64 new" X=1 $B --chunk 150 --minibatch 64 --steps 40 --warmup 8
for cfg in "TDNNF_GEMM_SERIAL_EPILOGUE=1" "X=1"; do echo $cfg; env $cfg python tools/gemm_bench.py 8 prefinal 2>/dev/null | grep -E "fwd|bwd" | cut -c1-100; env $cfg python tools/gemm_bench.py 8 "tdnnf.affine" 2>/dev/null | grep -E "fwd|bwd" | cut -c1-100; done
