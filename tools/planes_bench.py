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
DATA = sys.argv[2] if len(sys.argv) > 2 else "randn"  # "zeros" / "ones": the same kernels on operands that toggle no multiplier bits (clock / power check)


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


ELEM = {3: torch.bfloat16, 2: torch.float16}


def planes(np_, x, lead, R, t_rows=0):
    P = torch.zeros(lib.tdnnf_planes_bytes(np_, R, (x.shape[1] + 15) // 16) // 2, dtype=ELEM[np_], device="cuda")
    PT = torch.zeros(lib.tdnnf_planes_bytes(np_, t_rows, ((x.shape[0] + 63) // 64) * 4) // 2, dtype=ELEM[np_], device="cuda") if t_rows else None
    scale = torch.zeros(4, device="cuda")
    ws = torch.zeros(lib.tdnnf_planes_split_workspace_bytes() // 4 + 4, device="cuda")
    s = abi.stream()
    fn = lambda: abi.check(lib.tdnnf_planes_split(np_, abi.pmat(x), lead, R, abi.ptr(P), t_rows, abi.ptr(PT) if PT is not None else None, abi.ptr(scale), abi.ptr(ws), s))
    return P, PT, scale, fn


def run(name, M, N, Di, offs):
    K = len(offs)
    rows_in = M + max(offs) - min(offs)
    X = torch.randn(rows_in, Di, device="cuda")
    W = torch.randn(N, K * Di, device="cuda") / (K * Di) ** 0.5
    if DATA == "zeros":
        X, W = X * 0, W * 0
    elif DATA == "ones":
        X, W = X * 0 + 1, W * 0 + 1
    Cm = torch.zeros(M, N, device="cuda")
    BN = 160 if ((N + 159) // 160) * 160 - N < ((N + 127) // 128) * 128 - N else (256 if N % 256 == 0 else 128)
    Nb = ((N + BN - 1) // BN) * BN
    if Nb % 256 == 0:  # a chunk stride (rows x 32 bytes) that is a multiple of 8 KB puts every plane and K block on the same memory channels: 2.7x slower
        Nb += 8
    lead, tail = 0, 256
    if (rows_in + tail) % 256 == 0:
        tail += 8
    ref = sum(X[o - min(offs):o - min(offs) + M].double() @ W[:, i * Di:(i + 1) * Di].double().T for i, o in enumerate(offs))
    flops = 2.0 * M * N * K * Di
    out = f"{name:30s} M={M:7d} N={N:5d} K={K}x{Di:5d}"
    for np_, tag in ((3, "bf16x6"), (2, "f16x3")):
        ap, _, sa, split_a = planes(np_, X, lead, lead + rows_in + tail)
        apt_rows = Di + 8
        _, _, _, split_both = planes(np_, X, lead, lead + rows_in + tail, apt_rows)
        bp, _, sb, split_b = planes(np_, W, 0, Nb)
        t_split = timed(split_a)
        t_both = timed(split_both)
        split_a()
        split_b()
        a_row = (C.c_longlong * K)(*[lead + o - min(offs) for o in offs])
        zero = (C.c_int * K)(*([0] * K))
        b_col = (C.c_int * K)(*[i * Di for i in range(K)])
        cols = (C.c_int * K)(*([Di] * K))
        s = abi.stream()
        t = timed(lambda: abi.check(lib.tdnnf_planes_gemm(np_, abi.ptr(ap), lead + rows_in + tail, abi.ptr(sa), abi.ptr(bp), Nb, abi.ptr(sb), K, a_row, None, zero, b_col, cols,
                                                          None, 2, 0, abi.pmat(Cm), s)))
        err = float((Cm.double() - ref).norm() / ref.norm())
        out += f" | {tag} {t * 1e3:7.1f} us {flops / t / 1e9:6.1f} TF-eq  split {t_split * 1e3:6.1f} us (+T {t_both * 1e3:6.1f})  err {err:.1e}"
    t32 = timed(lambda: torch.matmul(X[:M], W[:, :Di].T))
    print(out, flush=True)


B = 128
run("tdnnf.linear fwd full-rate", 1564 * B, 160, 1536, [-B, 0])
run("tdnnf.linear fwd 1/3-rate", 520 * B, 160, 1536, [-3 * B, 0])
run("tdnnf.affine fwd 1/3-rate", 519 * B, 1536, 160, [0, 3 * B])
run("tdnnf.affine fwd full-rate", 1563 * B, 1536, 160, [0, B])
run("prefinal.linear 1536->256", 500 * B, 256, 1536, [0])
run("output.affine 256->6034", 500 * B, 6034, 256, [0])
run("long K 1536x1536", 512 * B, 1536, 1536, [-B, 0])
