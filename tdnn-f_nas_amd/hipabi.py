"""ctypes binding of libtdnnf_hip.so (the C-ABI in include/tdnnf_hip.h) for torch tensors.

Plumbing only: torch provides device memory and streams; every computation is a
call into the HIP library.  Fails loudly if the library is missing -- there is no
CPU or PyTorch fallback for any op.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtdnnf_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tdnnf_hip.h")
MAX_OFFSETS = 16

_lib = None


class Mat(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("stride", C.c_int)]


class TdnnIndexes(C.Structure):
    _fields_ = [("row_stride", C.c_int), ("num_offsets", C.c_int), ("row_offsets", C.c_int * MAX_OFFSETS)]


class HipAbiError(RuntimeError):
    pass


def _prototypes():
    """{name: (restype, [argtypes])} parsed from include/tdnnf_hip.h, so that floats, size_t and
    pointers are marshalled with their C types (ctypes' defaults would pass float as double)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(tdnnf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()

        def ctype(decl):
            decl = decl.strip()
            if "*" in decl or decl.startswith("tdnnf_stream"):
                return C.c_void_p
            if decl.startswith("size_t"):
                return C.c_size_t
            if decl.startswith("float"):
                return C.c_float
            if decl.startswith("double"):
                return C.c_double
            if decl.startswith("long long"):
                return C.c_longlong
            if decl.startswith("int") or decl.startswith("unsigned"):
                return C.c_int
            raise HipAbiError(f"cannot marshal parameter '{decl}' of {name}")

        args = [] if params in ("", "void") else [ctype(p) for p in params.split(",")]
        if "*" in ret:
            res = C.c_char_p if "char" in ret else C.c_void_p
        elif ret.startswith("void"):
            res = None
        elif ret.startswith("size_t"):
            res = C.c_size_t
        elif ret.startswith("long long"):
            res = C.c_longlong
        else:
            res = C.c_int
        protos[name] = (res, args)
    return protos


def declared_symbols():
    """Every function name declared in include/tdnnf_hip.h."""
    return sorted(_prototypes())


def load():
    """dlopen the library and check that it exports every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipAbiError(f"{LIB_PATH} is not built: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950)")
    # torch first: its wheel carries a libamdhip64 of its own (soname libamdhip64.so.7), which the library's NEEDED entry
    # then resolves to.  The other order maps /opt/rocm's runtime as well, and the second HIP runtime of a process finds no
    # device (hipErrorNoDevice in the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    missing = []
    for s in declared_symbols():
        try:
            getattr(lib, s)
        except AttributeError:
            missing.append(s)
    if missing:
        raise HipAbiError(f"libtdnnf_hip.so does not export: {missing}")
    for name, (res, args) in _prototypes().items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HipAbiError(f"tdnnf error {rc}: {load().tdnnf_last_error().decode()}")


def mat(t):
    """torch 2-D float32 CUDA tensor (unit column stride) -> tdnnf_mat by value."""
    import torch
    assert t.dtype == torch.float32 and t.dim() == 2 and t.is_cuda, (t.dtype, t.shape, t.device)
    assert t.shape[1] <= 1 or t.stride(1) == 1
    stride = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return Mat(t.data_ptr(), t.shape[0], t.shape[1], stride)


def pmat(t):
    return C.byref(mat(t)) if t is not None else None


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def indexes(row_stride, row_offsets):
    ix = TdnnIndexes()
    ix.row_stride = int(row_stride)
    ix.num_offsets = len(row_offsets)
    for i, o in enumerate(row_offsets):
        ix.row_offsets[i] = int(o)
    return ix


def iarr(a):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def farr(a):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def workspace(nbytes, device="cuda"):
    import torch
    return torch.empty((int(nbytes) + 3) // 4 + 4, dtype=torch.float32, device=device)


class option:
    """`with option("ng_fuse", 0): ...` -- a tuning option of the library (tdnnf_set_option) for the duration of a block."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        lib = load()
        old = C.c_int()
        check(lib.tdnnf_get_option(self.name, C.byref(old)))
        self.old = old.value
        check(lib.tdnnf_set_option(self.name, self.value))
        return self

    def __exit__(self, *exc):
        check(load().tdnnf_set_option(self.name, self.old))
        return False


class DenGraph:
    def __init__(self, g):
        lib = load()
        self.h = C.c_void_p()
        self._keep = [iarr(g["src"]), iarr(g["dst"]), iarr(g["pdf"]), farr(g["prob"]), farr(g["init"])]
        check(lib.tdnnf_den_graph_create(int(g["H"]), len(g["src"]), int(g["P"]), self._keep[0][1], self._keep[1][1],
                                         self._keep[2][1], self._keep[3][1], self._keep[4][1], 0, C.byref(self.h)))
        self.H, self.P = int(g["H"]), int(g["P"])

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.tdnnf_den_graph_destroy(self.h)
            self.h = None


class Supervision:
    def __init__(self, s):
        lib = load()
        self.h = C.c_void_p()
        k = [iarr(s["seq_state_begin"]), iarr(s["seq_arc_begin"]), iarr(s["state_time"]), farr(s["final_logprob"]),
             iarr(s["arc_src"]), iarr(s["arc_dst"]), iarr(s["arc_pdf"]), farr(s["arc_logprob"])]
        check(lib.tdnnf_supervision_create(int(s["B"]), int(s["T"]), k[0][1], k[1][1], k[2][1], k[3][1], k[4][1], k[5][1],
                                           k[6][1], k[7][1], float(s.get("weight", 1.0)), C.byref(self.h)))
        self.B, self.T = int(s["B"]), int(s["T"])

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.tdnnf_supervision_destroy(self.h)
            self.h = None
