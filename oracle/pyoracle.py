"""ctypes binding of the CPU oracle (oracle/*.c) for numpy arrays.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  PARITY UNPINNED
(see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OMat(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("rows", C.c_int), ("cols", C.c_int),
                ("stride", C.c_int)]


def build(force=False):
    libs = [os.path.join(_HERE, n) for n in ("liboracle.so", "liboracle_fast.so")]
    if force or not all(os.path.exists(p) for p in libs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return libs


_libs = {}


def lib(fast=False):
    key = "fast" if fast else "ref"
    if key not in _libs:
        path = build()[1 if fast else 0]
        _libs[key] = C.CDLL(path)
        _declare(_libs[key])
    return _libs[key]


def omat(a):
    """View a 2-D float32 numpy array (row stride = a.strides[0]/4) as an omat."""
    assert a.dtype == np.float32 and a.ndim == 2 and (a.shape[1] <= 1 or a.strides[1] == 4), (a.dtype, a.shape, a.strides)
    assert a.strides[0] % 4 == 0
    return OMat(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1], a.strides[0] // 4)


def fptr(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dptr(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int))


class DenGraph(C.Structure):
    _fields_ = [("num_states", C.c_int), ("num_arcs", C.c_int), ("num_pdfs", C.c_int),
                ("arc_src", C.POINTER(C.c_int)), ("arc_dst", C.POINTER(C.c_int)),
                ("arc_pdf", C.POINTER(C.c_int)), ("arc_prob", C.POINTER(C.c_float)),
                ("initial_probs", C.POINTER(C.c_float))]


class Supervision(C.Structure):
    _fields_ = [("num_sequences", C.c_int), ("frames_per_sequence", C.c_int),
                ("seq_state_begin", C.POINTER(C.c_int)), ("seq_arc_begin", C.POINTER(C.c_int)),
                ("state_time", C.POINTER(C.c_int)), ("final_logprob", C.POINTER(C.c_float)),
                ("arc_src", C.POINTER(C.c_int)), ("arc_dst", C.POINTER(C.c_int)),
                ("arc_pdf", C.POINTER(C.c_int)), ("arc_logprob", C.POINTER(C.c_float)),
                ("weight", C.c_float)]


def _declare(L):
    P = C.POINTER
    M = P(OMat)
    f, i, d = C.c_float, C.c_int, C.c_double
    pf, pi, pd = P(f), P(i), P(d)
    sig = {
        "oracle_tdnn_share_index": (i, [pi, i]),
        "oracle_tdnn_darts_coef": (None, [pf, i, i, f, pf, f, pf]),
        "oracle_tdnn_darts_effective_coef": (None, [pf, i, i, i, pf]),
        "oracle_tdnn_propagate": (None, [M, pf, i, i, i, i, i, pi, pf, pf, i, M]),
        "oracle_tdnn_backprop_data": (None, [M, pf, i, i, i, i, i, pi, pf, M]),
        "oracle_tdnn_update_simple": (None, [M, M, i, i, i, i, pi, pf, f, pf, i, pf]),
        "oracle_tdnn_darts_tap_dots": (None, [M, M, pf, i, i, i, i, i, pi, pd]),
        "oracle_tdnn_darts_alpha_update": (None, [pd, pf, i, i, i, f, f, pf]),
        "oracle_tdnn_splice": (None, [M, i, i, i, i, pi, pf, i, M]),
        "oracle_batchnorm_propagate": (None, [M, f, f, M, pf]),
        "oracle_batchnorm_backprop": (None, [M, M, f, pf, M]),
        "oracle_batchnorm_store_stats": (None, [pf, i, i, pd, pd, pd]),
        "oracle_batchnorm_compute_derived": (None, [d, pd, pd, i, f, f, pf, pf]),
        "oracle_batchnorm_test_propagate": (None, [M, pf, pf, M]),
        "oracle_batchnorm_test_backprop": (None, [M, pf, M]),
        "oracle_gumbel_noise": (None, [pf, i, pf]),
        "oracle_softmax_flops_propagate": (None, [M, pf, f, M]),
        "oracle_softmax_flops_backprop": (None, [M, M, f, pf, i, f, M]),
        "oracle_onehot_index": (i, [f, i]),
        "oracle_onehot_propagate": (None, [f, M]),
        "oracle_copyn_propagate": (None, [M, f, M]),
        "oracle_copyn_backprop": (None, [M, f, M]),
        "oracle_constant_function_propagate": (None, [pf, M]),
        "oracle_constant_function_backprop": (None, [M, f, pf]),
        "oracle_flops_constraint_backprop": (None, [pf, f, i, i, M]),
        "oracle_elementwise_product_propagate": (None, [M, i, M]),
        "oracle_elementwise_product_backprop": (None, [M, M, i, M]),
        "oracle_relu_propagate": (None, [M, M]),
        "oracle_relu_backprop": (None, [M, M, M]),
        "oracle_relu_repair": (None, [pd, d, i, f, f, f, M]),
        "oracle_relu_store_stats": (None, [M, pd, pd, pd]),
        "oracle_affine_propagate": (None, [M, pf, i, pf, i, M]),
        "oracle_affine_backprop": (None, [M, pf, i, i, M]),
        "oracle_affine_update_simple": (None, [M, M, f, pf, i, pf]),
        "oracle_log_softmax_propagate": (None, [M, M]),
        "oracle_log_softmax_backprop": (None, [M, M, M]),
        "oracle_sum_scaled": (None, [M, f, M, f, M]),
        "oracle_general_dropout_propagate": (None, [M, pf, i, M]),
        "oracle_den_initial_probs": (None, [i, i, pi, pi, pf, i, i, pf]),
        "oracle_chain_denominator": (i, [P(DenGraph), M, i, f, f, pd, M]),
        "oracle_chain_numerator": (d, [P(Supervision), M, M]),
        "oracle_chain_objf_and_deriv": (i, [P(DenGraph), P(Supervision), M, f, f, f, pd, pd, pd, M, M]),
        "oracle_ng_create": (C.c_void_p, [i, i, f, f]),
        "oracle_ng_destroy": (None, [C.c_void_p]),
        "oracle_ng_precondition": (None, [C.c_void_p, M, pf]),
        "oracle_ng_state": (i, [C.c_void_p, pf, pf, pf, pi]),
        "oracle_constrain_orthonormal": (None, [f, pf, i, i, i]),
        "oracle_apply_l2": (None, [pf, pf, C.c_long, f]),
        "oracle_max_change_scales": (None, [pd, pf, i, f, f, f, pf, pi]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


SYMBOLS = None  # filled lazily by tests that check exports


# --------------------------------------------------------------------------- helpers
def den_graph_struct(g):
    """g: dict with int32 src,dst,pdf; float32 prob, init; ints H,P."""
    return DenGraph(int(g["H"]), len(g["src"]), int(g["P"]), iptr(g["src"]), iptr(g["dst"]),
                    iptr(g["pdf"]), fptr(g["prob"]), fptr(g["init"]))


def supervision_struct(s):
    return Supervision(int(s["B"]), int(s["T"]), iptr(s["seq_state_begin"]), iptr(s["seq_arc_begin"]),
                       iptr(s["state_time"]), fptr(s["final_logprob"]), iptr(s["arc_src"]),
                       iptr(s["arc_dst"]), iptr(s["arc_pdf"]), fptr(s["arc_logprob"]),
                       float(s.get("weight", 1.0)))
