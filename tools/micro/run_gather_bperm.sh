#!/bin/bash
cd $(dirname $0)
for sg in 16 32 64; do for m in 0 1 2 3; do ./gather_bperm $m $sg 4 1024; done; done
./gather_bperm 1 32 4 4096; ./gather_bperm 1 32 4 8192; ./gather_bperm 1 16 4 4096; ./gather_bperm 2 32 2 1024; ./gather_bperm 2 32 6 1024; ./gather_bperm 2 16 6 1024
