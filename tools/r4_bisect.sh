#!/bin/bash
# one process per variant of tools/shard_bisect.py
mkdir -p gpurun_out
for v in ${@:-alone as-bench}; do
  python tools/shard_bisect.py $v 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r4_bisect.txt
done
