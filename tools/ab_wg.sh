B="python bench.py --no-parity --no-cpu-baseline --no-also --no-alt"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launches"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]], [round(c["flops_per_step"]/1e12,3) for c in d["roofline"]["all_kernels"]])'
for rs in 0 4 4 0; do echo "roofline-steps $rs"; $B --steps 8 --warmup 4 --roofline-steps $rs 2>/dev/null | python -c "$P"; done
