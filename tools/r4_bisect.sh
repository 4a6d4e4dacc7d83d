#!/bin/bash
# GPU box: one process per variant of tools/shard_bisect.py (does a job's step time depend on what ran before it in the process?)
# usage: bash tools/r4_bisect.sh [variant ...]    (alone as-bench small-big-small big-64 big-32 big-no-ng ...)
mkdir -p gpurun_out
for v in ${@:-alone as-bench small-big-small}; do
  python tools/shard_bisect.py $v 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r4_bisect.txt
done
