#!/usr/bin/env python3
"""Which action of bench.py's main job slows the 1500 x 16 line item that follows it?  One variant per process.
usage (GPU box): python tools/shard_bisect.py VARIANT"""
import argparse
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

variant = sys.argv[1]
pkg = ge.load_package()
sys.argv = ["bench.py"]
# bench.main()'s parser defaults
args = argparse.Namespace(gpus=1, steps=8, warmup=4, ng_burn_in=10, roofline_steps=4, chunk=1500, minibatch=128, scaling="weak", sync_batchnorm="auto",
                          no_strong=False, no_overlap=False, den_states=4000, den_degree=12.0, cpu_sequences=16, workload="7q", darts_offsets=7,
                          bn_choices="reference", gemm="f32", no_alt=True, dropout=0.0, natural_gradient=1, also_only=None, option=[])
sync = torch.cuda.synchronize


def shard(tag, ng=None):
    import time
    j = bench.Job(pkg, args, 1500, 16, 4000, 0, 1, natural_gradient=ng)
    d = j.run(10, 4, 16, sync)
    t0 = time.perf_counter()
    for _ in range(16):
        j.step()
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    j.close()
    print("%-30s %-44s %7.2f ms   (again: host issue %.2f, with the sync %.2f)" % (variant, tag, 1e3 * d / 16, 1e3 * (t1 - t0) / 16, 1e3 * (t2 - t0) / 16), flush=True)


if variant == "alone":
    shard("no job before")
    sys.exit(0)
if variant == "small-big-small":
    shard("first")
big = dict(chunk=1500, seqs=128, ng=None)
if variant == "big-short-chunks":
    big["chunk"] = 150
if variant == "big-64":
    big["seqs"] = 64
if variant == "big-32":
    big["seqs"] = 32
if variant in ("big-no-ng", "both-no-ng"):
    big["ng"] = 0
job = bench.Job(pkg, args, big["chunk"], big["seqs"], 4000, 0, 1, natural_gradient=big["ng"])
if variant == "no-generator":
    job.net.set_random_draws = lambda **kw: None
d = job.run(10, 4, 8, sync, profile=variant != "no-profile")
if variant == "results":
    job.net.results.cpu()
job.close()
if variant == "free":
    del job
    gc.collect()
    torch.cuda.empty_cache()
shard("after the big job (%.1f ms)" % (1e3 * d / 8), ng=0 if variant in ("small-no-ng", "both-no-ng") else None)
