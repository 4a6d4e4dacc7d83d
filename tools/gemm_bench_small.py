#!/usr/bin/env python3
"""The f32 GEMM kernels stand-alone on the shapes of the SMALL minibatches (chunk 150 x 64 sequences, chunk 1500 x 16): forward, backward-data and
weight gradient of the TDNN-F layer's two components, at the full frame rate and on the 1/3 grid.  usage: gemm_bench_small.py [reps [OPTION=VALUE ...]]"""
import os
import runpy
import sys
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "20", "NOTHING"] + sys.argv[2:]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py"))
run = g["run"]
g["only"] = ""
import builtins
for B, tag, nt_full, nt_third in ((64, "150x64", 156, 52), (16, "1500x16", 1506, 502)):
    g["run"].__globals__["only"] = ""
    run("%s linear full-rate" % tag, [-1, 0], nt_full, B, 1536, 160)
    run("%s affine full-rate" % tag, [0, 1], nt_full, B, 160, 1536)
    run("%s linear 1/3-rate" % tag, [-3, 0], nt_third, B, 1536, 160, step=3)
    run("%s affine 1/3-rate" % tag, [0, 3], nt_third, B, 160, 1536, step=3)
