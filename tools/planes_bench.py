#!/usr/bin/env python3
"""Micro-benchmark of the pre-split bf16-plane GEMM (csrc/planes_gemm.hip) on the K = 3072 shapes of the 7q step (the .linear
forward / .affine backward-data GEMMs: M rows x 160 columns, two taps of 1536) and on the 1536-wide ones, beside the exact-f32
kernel.  f32-equivalent TFLOP/s = 2 M N K / time.  Usage (GPU box): python tools/planes_bench.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.hipabi
lib = abi.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, M, N, Di, offs):
    K = len(offs)
    rows_in = M + max(offs) - min(offs)
    X = torch.randn(rows_in, Di, device="cuda")
    W = torch.randn(N, K * Di, device="cuda") / (K * Di) ** 0.5
    Cm = torch.zeros(M, N, device="cuda")
    BN = 160 if ((N + 159) // 160) * 160 - N < ((N + 127) // 128) * 128 - N else (256 if N % 256 == 0 else 128)
    Nb = ((N + BN - 1) // BN) * BN
    if Nb % 256 == 0:  # a chunk stride (rows x 32 bytes) that is a multiple of 8 KB puts every plane and K block on the same memory channels: 2.7x slower
        Nb += 8
    lead, tail = 0, 256
    if (rows_in + tail) % 256 == 0:
        tail += 8
    ap = torch.zeros(lib.tdnnf_planes_bytes(rows_in, Di, lead, tail) // 2, dtype=torch.bfloat16, device="cuda")
    bp = torch.zeros(lib.tdnnf_planes_bytes(N, K * Di, 0, Nb - N) // 2, dtype=torch.bfloat16, device="cuda")
    s = abi.stream()
    t_split = timed(lambda: abi.check(lib.tdnnf_planes_split(abi.pmat(X), lead, tail, abi.ptr(ap), s)))
    abi.check(lib.tdnnf_planes_split(abi.pmat(W), 0, Nb - N, abi.ptr(bp), s))
    a_row = (C.c_longlong * K)(*[lead + o - min(offs) for o in offs])
    zero = (C.c_int * K)(*([0] * K))
    b_col = (C.c_int * K)(*[i * Di for i in range(K)])
    cols = (C.c_int * K)(*([Di] * K))
    t = timed(lambda: abi.check(lib.tdnnf_planes_gemm(abi.ptr(ap), lead + rows_in + tail, abi.ptr(bp), Nb, K, a_row, zero, b_col, cols, None, 2, 0, abi.pmat(Cm), s)))
    flops = 2.0 * M * N * K * Di
    ref = sum(X[o - min(offs):o - min(offs) + M].double() @ W[:, i * Di:(i + 1) * Di].double().T for i, o in enumerate(offs))
    err = float((Cm.double() - ref).norm() / ref.norm())
    print(f"{name:34s} M={M:7d} N={N:5d} K={K}x{Di:5d}  planes {t * 1e3:8.1f} us {flops / t / 1e9:7.1f} TF-eq   split of A {t_split * 1e3:7.1f} us "
          f"({X.numel() * 10 / t_split / 1e9:6.2f} TB/s)   err {err:.1e}", flush=True)


B = 128
run("tdnnf.linear fwd full-rate", 1564 * B, 160, 1536, [-B, 0])
run("tdnnf.linear fwd 1/3-rate", 520 * B, 160, 1536, [-3 * B, 0])
run("tdnnf.affine fwd 1/3-rate", 519 * B, 1536, 160, [0, 3 * B])
run("tdnnf.affine fwd full-rate", 1563 * B, 1536, 160, [0, B])
run("prefinal.linear 1536->256", 500 * B, 256, 1536, [0])
run("output.affine 256->6034", 500 * B, 6034, 256, [0])
run("long K 1536x1536", 512 * B, 1536, 1536, [-B, 0])
