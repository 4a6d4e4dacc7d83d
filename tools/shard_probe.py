#!/usr/bin/env python3
"""Does a job's step time depend on what ran in the process before it?  Times the 1500 x 16 step alone, then after a 1500 x 128 job in
exact f32, then after one in f16x3 (bench.py runs its line items in one process).  usage (GPU box): python tools/shard_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
LR = pkg.trainer.learning_rate(0, 1, 100, 0, 100)
print('lr', LR)


def job(chunk, mb, prec, steps, tag, profile=False):
    cfg = pkg.trainer.make_config(frames_per_chunk=chunk, num_sequences=mb, use_natural_gradient=1, gemm_precision=prec)
    net = pkg.trainer.ChainNet(cfg)
    net.set_params(net.init_params_numpy(seed=0, output_stddev=0.05))
    feats, iv = pkg.trainer.synthetic_egs(net, seed=1)
    graph = pkg.synth.make_den_graph(4000, cfg.num_pdfs, mean_out_degree=12.0, seed=2)
    den = pkg.hipabi.DenGraph(graph)
    sup = pkg.hipabi.Supervision(pkg.synth.make_supervision_from_den(graph, mb, chunk // 3, num_paths=2, seed=3))
    fd, ivd = torch.from_numpy(feats).cuda(), torch.from_numpy(iv).cuda()
    for i in range(14):
        net.forward_backward(fd, ivd, den, sup, step=i)
        net.update(LR, step=i)
    torch.cuda.synchronize()
    if profile:
        pkg.hipabi.check(pkg.hipabi.load().tdnnf_profile_enable(1))
    t0 = time.perf_counter()
    for i in range(14, 14 + steps):
        net.forward_backward(fd, ivd, den, sup, step=i)
        net.update(LR, step=i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if profile:
        import ctypes as C
        lib = pkg.hipabi.load()
        pkg.hipabi.check(lib.tdnnf_profile_enable(0))
        for k in range(8):
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            pkg.hipabi.check(lib.tdnnf_profile_read(k, C.byref(a), C.byref(b), C.byref(c)))
    fb, off = __import__("ctypes").c_int(), __import__("ctypes").c_int()
    pkg.hipabi.load().tdnnf_chain_den_mw_status(__import__("ctypes").byref(fb), __import__("ctypes").byref(off), 0)
    print("%-40s %8.2f ms / step   (den_mw fallbacks %d, disabled %d; torch reserved %.1f GB)" % (tag, 1e3 * dt, fb.value, off.value, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    net.close()
    del fd, ivd
    torch.cuda.empty_cache()


job(1500, 16, 0, 16, "1500 x 16 first")
job(1500, 128, 0, 4, "1500 x 128 f32, events on", profile=True)
job(1500, 16, 0, 16, "1500 x 16 after events were on")
job(1500, 128, 0, 4, "1500 x 128 f32")
job(1500, 16, 0, 16, "1500 x 16 after the f32 job")
job(1500, 128, 3, 4, "1500 x 128 f16x3")
job(1500, 16, 0, 16, "1500 x 16 after the f16x3 job")
job(150, 64, 0, 16, "150 x 64")
job(1500, 16, 0, 16, "1500 x 16 after 150 x 64")
