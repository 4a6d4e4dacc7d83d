// host_linalg.h -- the small dense host-side linear algebra of the OnlineNaturalGradient state update
// (UPSTREAM Kaldi does these R x R problems, R <= 80, on the CPU as well: SymPosSemiDefEig, Cholesky, Invert).
// Plain C++ (no HIP) so that tests/test_host_linalg.py can build it with g++ and check it against numpy.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

namespace tdnnf {
namespace hostla {

// Symmetric eigen-decomposition A = U diag(c) U^T by Householder tridiagonalisation followed by the implicit
// QL iteration.  A: n x n row-major (only read); c: eigenvalues sorted descending; U: eigenvectors in columns.
inline void sym_eig(const std::vector<double> &A, int n, std::vector<double> &c, std::vector<double> &U) {
  std::vector<double> V(A), d(n, 0.0), e(n, 0.0);
  auto v = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
  if (n == 0) {
    c.clear();
    U.clear();
    return;
  }
  // ---- reduction to tridiagonal form
  for (int j = 0; j < n; j++) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
        v(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1], g = sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += v(k, j) * d[k];
          e[k] += v(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; k++) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  // ---- accumulate the Householder transformations
  for (int i = 0; i < n - 1; i++) {
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; k++) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; j++) {
    d[j] = v(n - 1, j);
    v(n - 1, j) = 0.0;
  }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- implicit QL on the tridiagonal matrix (d diagonal, e sub-diagonal)
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
    int m = l;
    while (m < n - 1 && fabs(e[m]) > eps * tst1) m++;
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double cc = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; i--) {
          c3 = c2;
          c2 = cc;
          s2 = s;
          g = cc * e[i];
          h = cc * p;
          r = hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          cc = p / r;
          p = cc * d[i] - s * g;
          d[i + 1] = h + s * (cc * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + cc * h;
            v(k, i) = cc * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = cc * p;
      } while (fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] += f;
    e[l] = 0.0;
  }
  // ---- sort descending
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] > d[b]; });
  c.resize(n);
  U.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) {
    c[j] = d[order[j]];
    for (int k = 0; k < n; k++) U[(size_t)k * n + j] = v(k, order[j]);
  }
}

// Lower Cholesky factor C (O = C C^T) and its inverse Ci; returns false when O is not positive definite.
inline bool cholesky_inverse(const std::vector<double> &O, int n, std::vector<double> &C, std::vector<double> &Ci) {
  C.assign((size_t)n * n, 0.0);
  Ci.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double sum = O[(size_t)i * n + j];
      for (int k = 0; k < j; k++) sum -= C[(size_t)i * n + k] * C[(size_t)j * n + k];
      if (i == j) {
        if (!(sum > 0.0)) return false;
        C[(size_t)i * n + i] = sqrt(sum);
      } else {
        C[(size_t)i * n + j] = sum / C[(size_t)j * n + j];
      }
    }
  for (int i = 0; i < n; i++) {
    Ci[(size_t)i * n + i] = 1.0 / C[(size_t)i * n + i];
    for (int j = 0; j < i; j++) {
      double sum = 0;
      for (int k = j; k < i; k++) sum += C[(size_t)i * n + k] * Ci[(size_t)k * n + j];
      Ci[(size_t)i * n + j] = -sum / C[(size_t)i * n + i];
    }
  }
  return true;
}

// Gram-Schmidt over the rows of W (R x D, leading dimension ld) with a deterministic replacement for (numerically)
// dependent rows: stand-in for Kaldi's OrthogonalizeRows(), which re-randomises such rows.
inline void orthogonalize_rows(std::vector<float> &W, int R, int D, int ld) {
  std::vector<double> row(D);
  for (int i = 0; i < R; i++) {
    int cand = i;
    for (int attempt = 0;; attempt++) {
      double n0 = 0;
      for (int k = 0; k < D; k++) {
        row[k] = attempt == 0 ? W[(size_t)i * ld + k] : (k == cand % D ? 1.0 : 0.0);
        n0 += row[k] * row[k];
      }
      for (int pass = 0; pass < 2; pass++)
        for (int j = 0; j < i; j++) {
          double dot = 0;
          for (int k = 0; k < D; k++) dot += row[k] * W[(size_t)j * ld + k];
          for (int k = 0; k < D; k++) row[k] -= dot * W[(size_t)j * ld + k];
        }
      double n1 = 0;
      for (int k = 0; k < D; k++) n1 += row[k] * row[k];
      if (n0 > 0 && n1 > 1e-8 * n0 && n1 > 1e-30) {
        const double inv = 1.0 / sqrt(n1);
        for (int k = 0; k < D; k++) W[(size_t)i * ld + k] = (float)(row[k] * inv);
        break;
      }
      cand = attempt == 0 ? i : cand + 1;
    }
  }
}

}  // namespace hostla
}  // namespace tdnnf
