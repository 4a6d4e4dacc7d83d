"""Helpers for the -m gpu parity tests: call the C-ABI (through tdnn-f_nas_amd/hipabi.py)
on torch CUDA tensors and compare with the CPU oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import torch

F = np.float32


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def padded(a, pad=4):
    """Device copy of a 2-D array inside a wider buffer (row stride > cols), like a Kaldi sub-matrix."""
    a = np.ascontiguousarray(a)
    stride = ((a.shape[1] + 3) // 4) * 4 + pad
    buf = torch.full((a.shape[0], stride), 7.0, dtype=torch.float32, device="cuda")
    buf[:, :a.shape[1]] = torch.from_numpy(a).cuda()
    return buf[:, :a.shape[1]], buf


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


class Hip:
    """Thin call layer: methods mirror the C-ABI names without the tdnnf_ prefix."""

    def __init__(self, pkg):
        self.abi = pkg.hipabi
        self.lib = pkg.hipabi.load()

    def __getattr__(self, name):
        fn = getattr(self.lib, "tdnnf_" + name)
        abi = self.abi

        def call(*args):
            conv = []
            for a in args:
                if isinstance(a, torch.Tensor):
                    conv.append(abi.pmat(a) if a.dim() == 2 and a.dtype == torch.float32 and getattr(a, "_as_mat", True)
                                else abi.ptr(a))
                else:
                    conv.append(a)
            rc = fn(*conv)
            if fn.restype is C.c_int:
                abi.check(rc)
            return rc

        return call

    def vec(self, t):
        """Mark a tensor to be passed as a raw device pointer."""
        return self.abi.ptr(t)

    def ws(self, nbytes):
        return self.abi.workspace(nbytes)

    def stream(self):
        return self.abi.stream()
