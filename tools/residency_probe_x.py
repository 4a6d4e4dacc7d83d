#!/usr/bin/env python3
"""Residency of the split-bf16 rows kernels: TDNNF_GEMM_NOBAL=1 TDNNF_GEMM_PREC=1|3 python tools/residency_probe_x.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.hipabi
lib = abi.load()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def probe(name, Di, Do):
    out = []
    for tiles in (200, 256, 257, 500, 512, 513, 760, 768, 769, 1024, 1025):
        M = 128 * tiles
        x = torch.randn(M, Di, device="cuda")
        W = torch.randn(Do, Di, device="cuda")
        y = torch.zeros(M, Do, device="cuda")
        s = abi.stream()
        t = timed(lambda: abi.check(lib.tdnnf_affine_propagate(abi.pmat(x), abi.ptr(W), Di, None, Do, abi.pmat(y), s)))
        out.append(f"{tiles}:{t:.0f}")
    print(f"{name:34s}", "  ".join(out))


probe("N=160 K=3072", 3072, 160)
probe("N=128 K=1536", 1536, 128)
