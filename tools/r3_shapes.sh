# round-3 tracking runs (GPU box): the shapes VERDICT r2 asks for, one bench line each under gpurun_out/r3_<tag>_*.json
# usage: tools/r3_shapes.sh TAG
TAG=${1:-base}
R=$GRAFT_REPO_ROOT
cd $R
Q="--no-parity --no-cpu-baseline --no-also --no-alt"
python3 bench.py $Q --chunk 150 --minibatch 64 --steps 40 --warmup 8 > gpurun_out/r3_${TAG}_150x64.json 2> gpurun_out/r3_${TAG}_150x64.err || exit 1
python3 bench.py $Q --chunk 1500 --minibatch 16 --steps 16 --warmup 4 > gpurun_out/r3_${TAG}_1500x16.json 2> gpurun_out/r3_${TAG}_1500x16.err || exit 1
python3 bench.py $Q --steps 8 --warmup 4 > gpurun_out/r3_${TAG}_1500x128.json 2> gpurun_out/r3_${TAG}_1500x128.err || exit 1
python3 - <<'PY' $TAG
import json, sys, glob
for f in sorted(glob.glob("gpurun_out/r3_%s_*.json" % sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"], [ (c["kernel"], c["ms"], c["tflops"]) for c in d["roofline"]["all_kernels"]])
    except Exception as e:
        print(f, "ERR", e)
PY
