"""CPU: the host-side linear algebra of the natural-gradient refresh (tdnn-f_nas_amd/csrc/host_linalg.h: Householder +
implicit-QL symmetric eigen-solver, Cholesky inverse, deterministic Gram-Schmidt) built with g++ and checked against
numpy.  The oracle solves the same eigen-problems with cyclic Jacobi (oracle/oracle_ng.c), so the two are independent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "host_linalg.h"
extern "C" void hl_eig(const double *A, int n, double *c, double *U) {
  std::vector<double> a(A, A + (size_t)n * n), cc, uu;
  tdnnf::hostla::sym_eig(a, n, cc, uu);
  for (int i = 0; i < n; i++) c[i] = cc[i];
  for (int i = 0; i < n * n; i++) U[i] = uu[i];
}
extern "C" int hl_cholinv(const double *O, int n, double *Cm, double *Ci) {
  std::vector<double> o(O, O + (size_t)n * n), c, ci;
  const bool ok = tdnnf::hostla::cholesky_inverse(o, n, c, ci);
  for (int i = 0; i < n * n; i++) { Cm[i] = c[i]; Ci[i] = ci[i]; }
  return ok ? 1 : 0;
}
extern "C" void hl_gs(float *W, int R, int D, int ld) {
  std::vector<float> w(W, W + (size_t)R * ld);
  tdnnf::hostla::orthogonalize_rows(w, R, D, ld);
  for (size_t i = 0; i < w.size(); i++) W[i] = w[i];
}
'''


@pytest.fixture(scope="module")
def hl(tmp_path_factory):
    d = tmp_path_factory.mktemp("hl")
    (d / "hl.cc").write_text(SRC)
    so = str(d / "libhl.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tdnn-f_nas_amd", "csrc"),
                           str(d / "hl.cc"), "-o", so])
    return C.CDLL(so)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n", [1, 2, 3, 20, 80])
def test_sym_eig_matches_numpy(hl, n):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T * 0.37
    if n == 20:  # a zero row/column and a repeated eigenvalue
        A[:, 5] = 0
        A[5, :] = 0
        A[7, 7] = A[8, 8] = 2.0
        A[7, 8] = A[8, 7] = 0.0
        A[7, :7] = A[:7, 7] = A[8, :7] = A[:7, 8] = 0
        A[7, 9:] = A[9:, 7] = A[8, 9:] = A[9:, 8] = 0
    c, U = np.zeros(n), np.zeros((n, n))
    hl.hl_eig(p(np.ascontiguousarray(A)), n, p(c), p(U))
    w = np.linalg.eigvalsh(A)[::-1]
    assert np.all(np.diff(c) <= 1e-12)  # sorted descending
    assert np.abs(c - w).max() <= 1e-12 * max(1.0, np.abs(w).max())
    assert np.abs(U @ np.diag(c) @ U.T - A).max() <= 1e-11 * max(1.0, np.abs(A).max())
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12


def test_cholesky_inverse(hl):
    rng = np.random.default_rng(1)
    n = 24
    B = rng.standard_normal((n, n + 5))
    O = B @ B.T / n + np.eye(n)
    Cm, Ci = np.zeros((n, n)), np.zeros((n, n))
    assert hl.hl_cholinv(p(np.ascontiguousarray(O)), n, p(Cm), p(Ci)) == 1
    assert np.allclose(Cm @ Cm.T, O, atol=1e-12) and np.allclose(np.triu(Cm, 1), 0)
    assert np.allclose(Ci @ Cm, np.eye(n), atol=1e-12)
    O[3, 3] = -1.0
    assert hl.hl_cholinv(p(np.ascontiguousarray(O)), n, p(Cm), p(Ci)) == 0  # not positive definite


def test_gram_schmidt_replaces_dependent_rows(hl):
    rng = np.random.default_rng(2)
    R, D, ld = 6, 17, 20
    W = np.zeros((R, ld), np.float32)
    W[:, :D] = rng.standard_normal((R, D))
    W[3, :D] = 2.0 * W[1, :D] - W[0, :D]  # linearly dependent
    W[5, :] = 0                            # zero row
    hl.hl_gs(p(W), R, D, ld)
    G = W[:, :D].astype(np.float64) @ W[:, :D].T.astype(np.float64)
    assert np.abs(G - np.eye(R)).max() < 1e-5
    assert not W[:, D:].any()
