#!/bin/bash
# GPU box: bench.py's N > 1 code path with two ranks on ONE GPU (gloo collectives, host-staged) -- not a measurement, a rehearsal of the launch
# contract, the weak + strong blocks and the JSON line.  (RCCL itself cannot run two ranks on one device.)
export TDNNF_BENCH_REHEARSE_ON_ONE_GPU=1
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --ng-burn-in 2 \
  --chunk 150 --minibatch 8 > gpurun_out/r4_rehearse.json 2> gpurun_out/r4_rehearse.err
echo "exit $?"
tail -c 1500 gpurun_out/r4_rehearse.json
tail -5 gpurun_out/r4_rehearse.err
