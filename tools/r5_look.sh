#!/bin/bash
# GPU box: phases (unprofiled) and full timelines of the two small shapes
cd "$GRAFT_REPO_ROOT"
bash tools/r5_phases.sh 2>&1 | grep -v "^$" > gpurun_out/r5_phases.txt
TIMELINE_MIN_US=0 timeout -k 10 300 bash tools/r4_prof.sh r5b_150x64 --chunk 150 --minibatch 64 --steps 12 > /dev/null
TIMELINE_MIN_US=0 timeout -k 10 300 bash tools/r4_prof.sh r5b_1500x16 --chunk 1500 --minibatch 16 --steps 8 > /dev/null
cat gpurun_out/r5_phases.txt
