#!/usr/bin/env python3
"""How many blocks of each rows_gemm tile variant are resident per CU?  Times plain launches (TDNNF_GEMM_NOBAL=1) with
tile counts around 2x and 3x the CU count: the time steps up when one more tile needs one more round.
usage (GPU box): TDNNF_GEMM_NOBAL=1 python tools/residency_probe.py"""
import ctypes as C
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


def probe(name, Di, Do, K, bwd):
    out = []
    for tiles in (500, 512, 513, 760, 768, 769, 1024, 1025):
        ntn = (Do + 127) // 128 if not bwd else 1
        M = 128 * tiles // max(ntn, 1)
        x = torch.randn(M, Di, device="cuda")
        W = torch.randn(Do, Di, device="cuda")
        y = torch.zeros(M, Do, device="cuda")
        s = abi.stream()
        if not bwd:
            t = timed(lambda: abi.check(lib.tdnnf_affine_propagate(abi.pmat(x), abi.ptr(W), Di, None, Do, abi.pmat(y), s)))
        else:
            dx = torch.zeros(M, Di, device="cuda")
            t = timed(lambda: abi.check(lib.tdnnf_affine_backprop(abi.pmat(y), abi.ptr(W), Di, Di, abi.pmat(dx), s)))
        out.append(f"{tiles}:{t:.0f}")
    print(f"{name:34s}", "  ".join(out))


probe("128x160 BK16 b_kc (fwd N=160 K=3072)", 3072, 160, 1, False)
probe("128x160 BK16 !b_kc (bwd N=160 K=1536)", 160, 1536, 1, True)
probe("128x128 BK32 b_kc (fwd N=128 K=1536)", 1536, 128, 1, False)
probe("128x128 BK16 b_kc (fwd N=128 K=256)", 256, 128, 1, False)
probe("128x128 BK32 !b_kc (bwd N=128 K=1536)", 128, 1536, 1, True)
probe("128x128 BK16 !b_kc (bwd N=128 K=256)", 128, 256, 1, True)
