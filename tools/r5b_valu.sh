#!/bin/bash
# GPU box: the vector-ALU statistics passes (option ng_valu): natural-gradient tests, then interleaved A/B at the headline shape and the small ones
cd "$GRAFT_REPO_ROOT"
{
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_update_ng.py tests/test_gpu_ng_group.py tests/test_gpu_net.py -x -q -m gpu -k "natural or ng or NG or grouped or early or fused_output" 2>&1 | tail -5 || exit 1
Q="--no-parity --no-cpu-baseline --no-also --no-alt --roofline-steps 2 --steps 12 --warmup 4"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]], [round(c["ms"],1) for c in d["roofline"]["all_kernels"]])'
for rep in 1 2; do
for o in "--option ng_valu=0" "--option ng_valu=1"; do
  for g in f32 f16x3; do
  echo -n "[$o] $g: "; timeout -k 10 200 python3 bench.py $Q --gemm $g $o 2>/dev/null | python3 -c "$P"
  done
done
done
bash tools/r5_ab.sh 2 "--option ng_valu=0" "--option ng_valu=1"
} 2>&1 | tee gpurun_out/r5b_valu.txt
