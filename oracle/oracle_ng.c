/*
 * oracle_ng.c -- CPU restatement of OnlineNaturalGradient::PreconditionDirections.
 * Test infrastructure only (see oracle.h).  PARITY UNPINNED.
 *
 * UPSTREAM (nnet3/natural-gradient-online.{h,cc}, not shipped).  Restated from
 * Povey, Zhang, Khudanpur 2015, "Parallel training of DNNs with natural
 * gradient and parameter averaging" (appendix: the online low-rank Fisher
 * update) and SURVEY.md 8(a) row A8.  Reference call sites:
 * src/nnet3/nnet-tdnn-component.cc:598-599 (rank 20 / 80, alpha 4, history
 * 2000, update period 4: :183-210), nnet-simple-component.cc:3001-3002.
 *
 * State: W_t (R x D) = E_t^{1/2} R_t, d_t (R), rho_t, step counter t.
 * X_hat = X - (X W^T) W ; scale = sqrt(tr(X X^T) / tr(X_hat X_hat^T)).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct oracle_ng {
  int rank, update_period, t, D, frozen;
  float num_samples_history, alpha, epsilon, delta;
  float *W;  /* R x D */
  float *d;  /* R */
  float rho;
};

oracle_ng *oracle_ng_create(int rank, int update_period,
                            float num_samples_history, float alpha) {
  oracle_ng *ng = (oracle_ng *)calloc(1, sizeof(oracle_ng));
  ng->rank = rank;
  ng->update_period = update_period;
  ng->num_samples_history = num_samples_history;
  ng->alpha = alpha;
  ng->epsilon = 1.0e-10f;
  ng->delta = 5.0e-04f;
  return ng;
}

void oracle_ng_destroy(oracle_ng *ng) {
  if (!ng) return;
  free(ng->W);
  free(ng->d);
  free(ng);
}

int oracle_ng_state(const oracle_ng *ng, float *W, float *d, float *rho, int *t) {
  if (!ng->W) return 0;
  if (W) memcpy(W, ng->W, sizeof(float) * ng->rank * ng->D);
  if (d) memcpy(d, ng->d, sizeof(float) * ng->rank);
  if (rho) *rho = ng->rho;
  if (t) *t = ng->t;
  return ng->rank;
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (double):
   A = U diag(c) U^T, columns of U are eigenvectors; sorted descending. */
static void sym_eig(double *A, int n, double *c, double *U) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) U[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; k++) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = cs * akp - sn * akq;
          A[k * n + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = cs * apk - sn * aqk;
          A[q * n + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; k++) {
          double ukp = U[k * n + p], ukq = U[k * n + q];
          U[k * n + p] = cs * ukp - sn * ukq;
          U[k * n + q] = sn * ukp + cs * ukq;
        }
      }
  }
  for (int i = 0; i < n; i++) c[i] = A[i * n + i];
  for (int i = 0; i < n; i++) { /* SortSvd: descending */
    int m = i;
    for (int j = i + 1; j < n; j++)
      if (c[j] > c[m]) m = j;
    if (m != i) {
      double tc = c[i];
      c[i] = c[m];
      c[m] = tc;
      for (int k = 0; k < n; k++) {
        double tu = U[k * n + i];
        U[k * n + i] = U[k * n + m];
        U[k * n + m] = tu;
      }
    }
  }
}

/* InitOrthonormalSpecial + InitDefault */
static void ng_init_default(oracle_ng *ng, int D) {
  if (ng->rank >= D) ng->rank = D - 1;
  int R = ng->rank;
  ng->D = D;
  free(ng->W);
  free(ng->d);
  ng->W = (float *)calloc((size_t)R * D, sizeof(float));
  ng->d = (float *)calloc(R, sizeof(float));
  if (R == 0) return;
  ng->rho = ng->epsilon;
  for (int i = 0; i < R; i++) ng->d[i] = ng->epsilon;
  const float first_elem = 1.1f;
  for (int r = 0; r < R; r++) {
    int ncols = 0;
    for (int c = r; c < D; c += R) ncols++;
    float normalizer = 1.0f / sqrtf(first_elem * first_elem + ncols - 1);
    int i = 0;
    for (int c = r; c < D; c += R, i++)
      ng->W[(size_t)r * D + c] = normalizer * (i == 0 ? first_elem : 1.0f);
  }
  float E_tii = 1.0f / (2.0f + (D + R) * ng->alpha / D);
  for (size_t i = 0; i < (size_t)R * D; i++) ng->W[i] *= sqrtf(E_tii);
  ng->t = 0;
}

/* Gram-Schmidt with a deterministic replacement for (numerically) dependent rows -- stands in for Kaldi's
   OrthogonalizeRows(), which re-randomises such rows; both the oracle and the product use this exact routine. */
static void orthogonalize_rows(float *W, int R, int D) {
  double *row = (double *)malloc(sizeof(double) * D);
  for (int i = 0; i < R; i++) {
    int cand = i;
    for (int attempt = 0;; attempt++) {
      double n0 = 0;
      for (int k = 0; k < D; k++) {
        row[k] = attempt == 0 ? W[(size_t)i * D + k] : (k == cand % D ? 1.0 : 0.0);
        n0 += row[k] * row[k];
      }
      for (int pass = 0; pass < 2; pass++)
        for (int j = 0; j < i; j++) {
          double dot = 0;
          for (int k = 0; k < D; k++) dot += row[k] * W[(size_t)j * D + k];
          for (int k = 0; k < D; k++) row[k] -= dot * W[(size_t)j * D + k];
        }
      double n1 = 0;
      for (int k = 0; k < D; k++) n1 += row[k] * row[k];
      if (n0 > 0 && n1 > 1e-8 * n0 && n1 > 1e-30) {
        const double inv = 1.0 / sqrt(n1);
        for (int k = 0; k < D; k++) W[(size_t)i * D + k] = (float)(row[k] * inv);
        break;
      }
      cand = attempt == 0 ? i : cand + 1; /* try unit vectors e_i, e_{i+1}, ... */
    }
  }
  free(row);
}

/* ReorthogonalizeRt1 (UPSTREAM): make R_{t+1} = E_{t+1}^{-1/2} W_{t+1} orthonormal again. */
static void reorthogonalize(const oracle_ng *ng, const float *d_t1, float rho_t1, float *W1) {
  const int R = ng->rank, D = ng->D;
  const float threshold = 1.0e-03f;
  double d1_sum = 0;
  for (int i = 0; i < R; i++) d1_sum += d_t1[i];
  const double beta = rho_t1 * (1.0 + ng->alpha) + ng->alpha * d1_sum / D;
  double *sqrt_e = malloc(sizeof(double) * R), *inv_sqrt_e = malloc(sizeof(double) * R);
  for (int i = 0; i < R; i++) {
    const double e = 1.0 / (beta / d_t1[i] + 1.0);
    sqrt_e[i] = sqrt(e);
    inv_sqrt_e[i] = 1.0 / sqrt_e[i];
  }
  double *O = malloc(sizeof(double) * R * R), *Cm = calloc((size_t)R * R, sizeof(double));
  int is_unit = 1;
  for (int i = 0; i < R; i++)
    for (int j = 0; j <= i; j++) {
      double a = 0;
      for (int k = 0; k < D; k++) a += (double)W1[(size_t)i * D + k] * W1[(size_t)j * D + k];
      a = (float)a * inv_sqrt_e[i] * inv_sqrt_e[j];
      O[i * R + j] = O[j * R + i] = a;
      if (fabs(a - (i == j ? 1.0 : 0.0)) > threshold) is_unit = 0;
    }
  if (!is_unit) {
    /* Cholesky O = C C^T, then C^{-1} */
    int ok = 1;
    for (int i = 0; i < R && ok; i++)
      for (int j = 0; j <= i; j++) {
        double sum = O[i * R + j];
        for (int k = 0; k < j; k++) sum -= Cm[i * R + k] * Cm[j * R + k];
        if (i == j) {
          if (!(sum > 0.0)) { ok = 0; break; }
          Cm[i * R + i] = sqrt(sum);
        } else {
          Cm[i * R + j] = sum / Cm[j * R + j];
        }
      }
    double *Ci = calloc((size_t)R * R, sizeof(double));
    double cmax = 0;
    if (ok) { /* invert the lower-triangular factor */
      for (int i = 0; i < R; i++) {
        Ci[i * R + i] = 1.0 / Cm[i * R + i];
        for (int j = 0; j < i; j++) {
          double sum = 0;
          for (int k = j; k < i; k++) sum += Cm[i * R + k] * Ci[k * R + j];
          Ci[i * R + j] = -sum / Cm[i * R + i];
        }
      }
      for (int i = 0; i < R * R; i++)
        if (Ci[i] > cmax) cmax = Ci[i];
      if (!(cmax < 100.0)) ok = 0;
    }
    if (!ok) { /* Gram-Schmidt on R_{t+1}' rows, then W = E^{1/2} R */
      orthogonalize_rows(W1, R, D);
      for (int i = 0; i < R; i++)
        for (int k = 0; k < D; k++) W1[(size_t)i * D + k] *= (float)sqrt_e[i];
    } else { /* W <- E^{1/2} C^{-1} E^{-1/2} W */
      float *tmp = malloc(sizeof(float) * (size_t)R * D);
      memcpy(tmp, W1, sizeof(float) * (size_t)R * D);
      for (int i = 0; i < R; i++)
        for (int k = 0; k < D; k++) {
          double a = 0;
          for (int j = 0; j <= i; j++) a += (float)(Ci[i * R + j] * sqrt_e[i] * inv_sqrt_e[j]) * (double)tmp[(size_t)j * D + k];
          W1[(size_t)i * D + k] = (float)a;
        }
      free(tmp);
    }
    free(Ci);
  }
  free(sqrt_e); free(inv_sqrt_e); free(O); free(Cm);
}

static float ng_eta(const oracle_ng *ng, int N) {
  float ans = 1.0f - expf(-(float)N / ng->num_samples_history);
  if (ans > 0.9f) ans = 0.9f;
  return ans;
}

static int ng_updating(const oracle_ng *ng) {
  const int num_initial_updates = 10;
  return !ng->frozen && (ng->t <= num_initial_updates ||
                         (ng->t - num_initial_updates) % ng->update_period == 0);
}

static void compute_et(const float *d, int R, double beta, double *e, double *sqrt_e,
                       double *inv_sqrt_e) {
  for (int i = 0; i < R; i++) {
    e[i] = 1.0 / (beta / d[i] + 1.0);
    sqrt_e[i] = sqrt(e[i]);
    inv_sqrt_e[i] = 1.0 / sqrt_e[i];
  }
}

static void ng_precondition_internal(oracle_ng *ng, omat *X, int updating,
                                     double tr_X_Xt) {
  int N = X->rows, D = X->cols, R = ng->rank;
  float *W = ng->W;
  /* H = X W^T */
  float *H = (float *)malloc(sizeof(float) * (size_t)N * R);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++)
    for (int r = 0; r < R; r++) {
      double a = 0;
#pragma omp simd reduction(+ : a)
      for (int k = 0; k < D; k++) a += (double)X->data[(long)X->stride * n + k] * W[(size_t)r * D + k];
      H[(size_t)n * R + r] = (float)a;
    }
  float *J = NULL;
  double *Kt = NULL, *Lt = NULL;
  if (updating) {
    J = (float *)malloc(sizeof(float) * (size_t)R * D);
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < R; r++) { /* J = H^T X, one row of J per thread, X walked row by row */
      double *acc = (double *)calloc(D, sizeof(double));
      for (int n = 0; n < N; n++) {
        const double h = H[(size_t)n * R + r];
        const float *x = X->data + (long)X->stride * n;
#pragma omp simd
        for (int k = 0; k < D; k++) acc[k] += h * x[k];
      }
      for (int k = 0; k < D; k++) J[(size_t)r * D + k] = (float)acc[k];
      free(acc);
    }
    Kt = (double *)malloc(sizeof(double) * R * R);
    Lt = (double *)malloc(sizeof(double) * R * R);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < R; i++)
      for (int j = 0; j < R; j++) {
        double k = 0, l = 0;
        for (int c = 0; c < D; c++) k += (double)J[(size_t)i * D + c] * J[(size_t)j * D + c];
        for (int n = 0; n < N; n++) l += (double)H[(size_t)n * R + i] * H[(size_t)n * R + j];
        Kt[i * R + j] = (float)k; /* K = J J^T */
        Lt[i * R + j] = (float)l; /* L = H^T H */
      }
  }
  /* X_hat = X - H W  (uses W_t, before it is replaced) */
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    double *acc = (double *)calloc(D, sizeof(double));
    for (int r = 0; r < R; r++) {
      const double h = H[(size_t)n * R + r];
      const float *w = W + (size_t)r * D;
#pragma omp simd
      for (int k = 0; k < D; k++) acc[k] += h * w[k];
    }
    float *x = X->data + (long)X->stride * n;
    for (int k = 0; k < D; k++) x[k] = (float)((double)x[k] - acc[k]);
    free(acc);
  }
  free(H);
  if (!updating) return;

  float eta = ng_eta(ng, N), rho_t = ng->rho, alpha = ng->alpha;
  double d_sum = 0;
  for (int i = 0; i < R; i++) d_sum += ng->d[i];
  double beta_t = rho_t * (1.0 + alpha) + alpha * d_sum / D;
  double *e = malloc(sizeof(double) * R), *sqrt_e = malloc(sizeof(double) * R),
         *inv_sqrt_e = malloc(sizeof(double) * R);
  compute_et(ng->d, R, beta_t, e, sqrt_e, inv_sqrt_e);
  /* Z_t = (eta/N)^2 E^-.5 K E^-.5 + (eta/N)(1-eta) E^-.5 L E^-.5 (D+rho I)
         + (eta/N)(1-eta) (D+rho I) E^-.5 L E^-.5 + (1-eta)^2 (D+rho I)^2 */
  double *Z = malloc(sizeof(double) * R * R);
  double eN = (double)eta / N, eN1 = eN * (1.0 - eta);
  for (int i = 0; i < R; i++)
    for (int j = 0; j < R; j++) {
      double di = ng->d[i] + rho_t, dj = ng->d[j] + rho_t;
      double z = eN * eN * inv_sqrt_e[i] * Kt[i * R + j] * inv_sqrt_e[j] +
                 eN1 * inv_sqrt_e[i] * Lt[i * R + j] * inv_sqrt_e[j] * (di + dj);
      if (i == j) z += (1.0 - eta) * (1.0 - eta) * di * di;
      Z[i * R + j] = z;
    }
  for (int i = 0; i < R; i++) /* symmetrise against roundoff */
    for (int j = 0; j < i; j++) Z[i * R + j] = Z[j * R + i] = 0.5 * (Z[i * R + j] + Z[j * R + i]);
  double *c = malloc(sizeof(double) * R), *U = malloc(sizeof(double) * R * R);
  sym_eig(Z, R, c, U);
  double c_floor = pow(rho_t * (1.0 - eta), 2);
  const double condition_threshold = 1.0e+06;
  int must_reorthogonalize = c[0] > condition_threshold * c[R - 1];
  for (int i = 0; i < R; i++)
    if (c[i] < c_floor) {
      c[i] = c_floor;
      must_reorthogonalize = 1;
    }
  double *sqrt_c = malloc(sizeof(double) * R);
  double sqrt_c_sum = 0, sqrt_c_max = 0;
  for (int i = 0; i < R; i++) {
    sqrt_c[i] = sqrt(c[i]);
    sqrt_c_sum += sqrt_c[i];
    if (sqrt_c[i] > sqrt_c_max) sqrt_c_max = sqrt_c[i];
  }
  /* rho_{t+1} = 1/(D-R) ( eta/N tr(X X^T) + (1-eta)(D rho_t + tr(D_t)) - tr(C^.5) ) */
  float rho_t1 = (float)(1.0 / (D - R) *
                         (eta / N * tr_X_Xt + (1 - eta) * (D * rho_t + d_sum) - sqrt_c_sum));
  float *d_t1 = malloc(sizeof(float) * R);
  float floor_val = fmaxf(ng->epsilon, ng->delta * (float)sqrt_c_max);
  for (int i = 0; i < R; i++) {
    d_t1[i] = (float)sqrt_c[i] - rho_t1;
    if (d_t1[i] < floor_val) d_t1[i] = floor_val;
  }
  if (rho_t1 < floor_val) rho_t1 = floor_val;
  /* W_{t+1} = A_t B_t,  B_t = J + (1-eta)/(eta/N) (D_t + rho_t I) W_t,
     A_t = (eta/N) E_{t+1}^.5 C^-.5 U^T E_t^-.5 */
  double d1_sum = 0;
  for (int i = 0; i < R; i++) d1_sum += d_t1[i];
  double beta_t1 = rho_t1 * (1.0 + alpha) + alpha * d1_sum / D;
  double *e1 = malloc(sizeof(double) * R), *sqrt_e1 = malloc(sizeof(double) * R),
         *inv_sqrt_e1 = malloc(sizeof(double) * R);
  compute_et(d_t1, R, beta_t1, e1, sqrt_e1, inv_sqrt_e1);
  for (int r = 0; r < R; r++) {
    float coeff = (float)((1.0 - eta) / (eta / N) * (ng->d[r] + rho_t));
    for (int k = 0; k < D; k++) J[(size_t)r * D + k] += coeff * W[(size_t)r * D + k];
  }
  float *At = malloc(sizeof(float) * R * R);
  for (int i = 0; i < R; i++)
    for (int j = 0; j < R; j++)
      At[i * R + j] = (float)(U[j * R + i] * (eta / N) * sqrt_e1[i] / sqrt_c[i] * inv_sqrt_e[j]);
  float *W1 = malloc(sizeof(float) * (size_t)R * D);
  for (int i = 0; i < R; i++)
    for (int k = 0; k < D; k++) {
      double a = 0;
      for (int j = 0; j < R; j++) a += (double)At[i * R + j] * J[(size_t)j * D + k];
      W1[(size_t)i * D + k] = (float)a;
    }
  if (must_reorthogonalize) reorthogonalize(ng, d_t1, rho_t1, W1);
  memcpy(ng->W, W1, sizeof(float) * (size_t)R * D);
  memcpy(ng->d, d_t1, sizeof(float) * R);
  ng->rho = rho_t1;
  free(J); free(Kt); free(Lt); free(e); free(sqrt_e); free(inv_sqrt_e); free(Z);
  free(c); free(U); free(sqrt_c); free(d_t1); free(e1); free(sqrt_e1);
  free(inv_sqrt_e1); free(At); free(W1);
}

static double trace_xxt(const omat *X) {
  double t = 0;
  for (int n = 0; n < X->rows; n++)
    for (int k = 0; k < X->cols; k++) {
      double v = X->data[(long)X->stride * n + k];
      t += v * v;
    }
  return t;
}

static void ng_precondition_noinit(oracle_ng *ng, omat *X, float *scale) {
  int updating = ng_updating(ng);
  double initial_product = trace_xxt(X);
  ng_precondition_internal(ng, X, updating, initial_product);
  if (scale) {
    if (initial_product <= 0.0)
      *scale = 1.0f;
    else
      *scale = (float)sqrt(initial_product / trace_xxt(X));
  }
  ng->t += 1;
}

void oracle_ng_precondition(oracle_ng *ng, omat *X, float *scale) {
  if (X->cols == 1) { /* preconditioning a single column is pointless */
    if (scale) *scale = 1.0f;
    return;
  }
  if (ng->t == 0 && ng->W == NULL) {
    /* Init(): default init, then self-train on this very minibatch 3 times
       (1 if N <= rank), discarding the outputs. */
    ng_init_default(ng, X->cols);
    if (ng->rank > 0) {
      int iters = X->rows <= ng->rank ? 1 : 3;
      int was_frozen = ng->frozen;
      ng->frozen = 0;
      ng->t = 1;
      omat copy = {malloc(sizeof(float) * (size_t)X->rows * X->cols), X->rows, X->cols, X->cols};
      for (int i = 0; i < iters; i++) {
        for (int n = 0; n < X->rows; n++)
          memcpy(copy.data + (size_t)n * X->cols, X->data + (long)X->stride * n, sizeof(float) * X->cols);
        float s;
        ng_precondition_noinit(ng, &copy, &s);
      }
      free(copy.data);
      ng->frozen = was_frozen;
      ng->t = 0;
    }
  }
  if (ng->rank == 0) {
    if (scale) *scale = 1.0f;
    return;
  }
  ng_precondition_noinit(ng, X, scale);
}
