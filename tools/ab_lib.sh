#!/bin/bash
# Same-box A/B of two BUILDS of the library (boxes differ by up to +-8 % on single kernels, so two gpurun calls cannot be compared).
# Here (build container):   tools/ab_lib.sh stash        -> builds HEAD's library as ab_libs/lib_old.so and the working tree's as lib_new.so
# On the GPU box (gpurun):  bash tools/ab_lib.sh run [bench.py arguments]   -> old / new / old / new, one line each
# Remove ab_libs/ afterwards (it travels with the snapshot, it is not tracked).
set -e
cd "$(dirname "$0")/.."
if [ "$1" = stash ]; then
  mkdir -p ab_libs
  make -s -C tdnn-f_nas_amd/csrc && cp tdnn-f_nas_amd/libtdnnf_hip.so ab_libs/lib_new.so
  git stash -q && make -s -C tdnn-f_nas_amd/csrc && cp tdnn-f_nas_amd/libtdnnf_hip.so ab_libs/lib_old.so
  git stash pop -q && make -s -C tdnn-f_nas_amd/csrc
  ls -la ab_libs
  exit 0
fi
shift || true
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [round(c["tflops"],1) for c in d["roofline"]["all_kernels"]])'
for v in old new old new; do
  cp ab_libs/lib_$v.so tdnn-f_nas_amd/libtdnnf_hip.so
  echo -n "$v: "; python bench.py --no-parity --no-cpu-baseline --no-also --no-alt "$@" 2>/dev/null | python -c "$P"
done
