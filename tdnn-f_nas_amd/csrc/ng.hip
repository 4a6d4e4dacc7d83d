// ng.hip -- OnlineNaturalGradient::PreconditionDirections on gfx950 (UPSTREAM Kaldi
// nnet3/natural-gradient-online.{h,cc}; call sites /root/reference/src/nnet3/nnet-tdnn-component.cc:598-599,
// nnet-simple-component.cc:3001-3002; configuration :183-210).  SURVEY.md 8(a) row A8.
//
// The N x D work (H = X W^T, X_hat = X - H W, J = H^T X, L = H^T H, K = J J^T) runs on the f32 MFMA
// GEMM kernels; the R x R symmetric eigen-problem (R <= 80) is solved on the host in double, exactly
// where the reference does it, on the steps where the low-rank state is refreshed.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "gemm_f32.h"

struct tdnnf_ng {
  int rank, update_period, t, D, frozen;
  float num_samples_history, alpha, epsilon, delta, rho;
  float *W;  // device R x D
  std::vector<float> d;
  float *scratch;
  size_t scratch_floats;
  float *neg_one;  // device constant {-1}
};

namespace tdnnf {
namespace {

constexpr int kSumBlocks = 256;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(MatView x, double *partial) {
  __shared__ double red[4];
  double s = 0;
  const long long total = (long long)x.rows * x.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += kSumBlocks * 256LL) {
    const double v = x.data[(size_t)(e / x.cols) * x.stride + e % x.cols];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void add_diag_rows_kernel(float *J, const float *W, const float *coeff, int R, int D) {
  const long long total = (long long)R * D;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) J[e] += coeff[e / D] * W[e];
}

// cyclic Jacobi for a symmetric n x n matrix (row-major), eigenvalues sorted descending, eigenvectors in columns of U
void jacobi_eig(std::vector<double> &A, int n, std::vector<double> &c, std::vector<double> &U) {
  U.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) U[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
        for (int k = 0; k < n; k++) {
          const double x = A[k * n + p], y = A[k * n + q];
          A[k * n + p] = cs * x - sn * y;
          A[k * n + q] = sn * x + cs * y;
        }
        for (int k = 0; k < n; k++) {
          const double x = A[p * n + k], y = A[q * n + k];
          A[p * n + k] = cs * x - sn * y;
          A[q * n + k] = sn * x + cs * y;
        }
        for (int k = 0; k < n; k++) {
          const double x = U[k * n + p], y = U[k * n + q];
          U[k * n + p] = cs * x - sn * y;
          U[k * n + q] = sn * x + cs * y;
        }
      }
  }
  c.resize(n);
  for (int i = 0; i < n; i++) c[i] = A[i * n + i];
  for (int i = 0; i < n; i++) {
    int m = i;
    for (int j = i + 1; j < n; j++)
      if (c[j] > c[m]) m = j;
    if (m != i) {
      std::swap(c[i], c[m]);
      for (int k = 0; k < n; k++) std::swap(U[k * n + i], U[k * n + m]);
    }
  }
}

void compute_et(const std::vector<float> &d, double beta, std::vector<double> &sqrt_e, std::vector<double> &inv_sqrt_e) {
  const int R = (int)d.size();
  sqrt_e.resize(R);
  inv_sqrt_e.resize(R);
  for (int i = 0; i < R; i++) {
    const double e = 1.0 / (beta / d[i] + 1.0);
    sqrt_e[i] = sqrt(e);
    inv_sqrt_e[i] = 1.0 / sqrt_e[i];
  }
}

// Gram-Schmidt with a deterministic replacement for (numerically) dependent rows (stand-in for Kaldi's
// OrthogonalizeRows(), which re-randomises such rows)
void orthogonalize_rows(std::vector<float> &W, int R, int D) {
  std::vector<double> row(D);
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
      cand = attempt == 0 ? i : cand + 1;
    }
  }
}

int init_default(tdnnf_ng *ng, int D) {
  if (ng->rank >= D) ng->rank = D - 1;
  const int R = ng->rank;
  ng->D = D;
  if (ng->W) hipFree(ng->W);
  ng->W = nullptr;
  ng->d.assign(R, ng->epsilon);
  ng->rho = ng->epsilon;
  ng->t = 0;
  if (R == 0) return TDNNF_OK;
  std::vector<float> W((size_t)R * D, 0.f);
  const float first_elem = 1.1f;
  for (int r = 0; r < R; r++) {  // InitOrthonormalSpecial
    int ncols = 0;
    for (int c = r; c < D; c += R) ncols++;
    const float normalizer = 1.0f / sqrtf(first_elem * first_elem + ncols - 1);
    int i = 0;
    for (int c = r; c < D; c += R, i++) W[(size_t)r * D + c] = normalizer * (i == 0 ? first_elem : 1.0f);
  }
  const float E_tii = 1.0f / (2.0f + (D + R) * ng->alpha / D);
  for (auto &w : W) w *= sqrtf(E_tii);
  TDNNF_HIP(hipMalloc((void **)&ng->W, sizeof(float) * (size_t)R * D));
  TDNNF_HIP(hipMemcpy(ng->W, W.data(), sizeof(float) * W.size(), hipMemcpyHostToDevice));
  return TDNNF_OK;
}

bool updating(const tdnnf_ng *ng) {
  return !ng->frozen && (ng->t <= 10 || (ng->t - 10) % ng->update_period == 0);
}

int ensure_scratch(tdnnf_ng *ng, size_t floats) {
  if (ng->scratch_floats >= floats) return TDNNF_OK;
  if (ng->scratch) hipFree(ng->scratch);
  ng->scratch = nullptr;
  ng->scratch_floats = 0;
  TDNNF_HIP(hipMalloc((void **)&ng->scratch, sizeof(float) * floats));
  ng->scratch_floats = floats;
  return TDNNF_OK;
}

int sumsq_host(MatView x, double *partial_dev, hipStream_t s, double *out) {
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kSumBlocks), dim3(256), 0, s, x, partial_dev);
  double h[kSumBlocks];
  TDNNF_HIP(hipMemcpyAsync(h, partial_dev, sizeof(h), hipMemcpyDeviceToHost, s));
  TDNNF_HIP(hipStreamSynchronize(s));
  double t = 0;
  for (int i = 0; i < kSumBlocks; i++) t += h[i];
  *out = t;
  return TDNNF_OK;
}

// one PreconditionDirections step on X (state must be initialised); increments t
int precondition_step(tdnnf_ng *ng, MatView X, float *scale, hipStream_t s) {
  const int N = X.rows, D = X.cols, R = ng->rank;
  const bool upd = updating(ng);
  const size_t wg_bytes = std::max(wgrad_workspace_bytes(R, D, 1, N), wgrad_workspace_bytes(R, R, 1, N));
  const size_t f_H = (size_t)N * R, f_J = (size_t)R * D, f_RR = (size_t)R * R;
  const size_t need = f_H + 2 * f_J + 3 * f_RR + R + 2 * kSumBlocks * 2 + wg_bytes / 4 + 64;
  int rc = ensure_scratch(ng, need);
  if (rc) return rc;
  float *H = ng->scratch, *J = H + ((f_H + 3) & ~(size_t)3), *W1 = J + ((f_J + 3) & ~(size_t)3);
  float *Kd = W1 + ((f_J + 3) & ~(size_t)3), *Ld = Kd + ((f_RR + 3) & ~(size_t)3), *Ad = Ld + ((f_RR + 3) & ~(size_t)3);
  float *coeff = Ad + ((f_RR + 3) & ~(size_t)3);
  double *partial = (double *)(coeff + ((R + 3) & ~3) + 2);
  partial = (double *)(((uintptr_t)partial + 15) & ~(uintptr_t)15);
  void *wg_ws = (void *)(partial + 2 * kSumBlocks);
  double tr0;
  if ((rc = sumsq_host(X, partial, s, &tr0))) return rc;

  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = X.data; a.lda = X.stride; a.B = ng->W; a.ldb = D; a.C = H; a.ldc = R; a.M = N; a.N = R; a.init_mode = 2; a.nseg = 1;
  a.seg[0].klen = D; a.seg[0].m_lo = 0; a.seg[0].m_hi = N;
  TDNNF_HIP(rows_gemm(a, true, s));  // H = X W^T
  std::vector<float> Kh, Lh;
  if (upd) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.dY = H; w.lddy = R; w.X = X.data; w.ldx = X.stride; w.Do = R; w.Di = D; w.K = 1; w.N = N; w.row_stride = 1;
    w.scale = 1.f; w.G = J; w.ldg = D; w.accumulate = 0;
    TDNNF_HIP(wgrad(w, wg_ws, wg_bytes, s));  // J = H^T X
    w.X = H; w.ldx = R; w.Di = R; w.G = Ld; w.ldg = R;
    TDNNF_HIP(wgrad(w, wg_ws, wg_bytes, s));  // L = H^T H
    RowsGemmArgs k;
    memset(&k, 0, sizeof(k));
    k.A = J; k.lda = D; k.B = J; k.ldb = D; k.C = Kd; k.ldc = R; k.M = R; k.N = R; k.init_mode = 2; k.nseg = 1;
    k.seg[0].klen = D; k.seg[0].m_lo = 0; k.seg[0].m_hi = R;
    TDNNF_HIP(rows_gemm(k, true, s));  // K = J J^T
    Kh.resize(f_RR);
    Lh.resize(f_RR);
    TDNNF_HIP(hipMemcpyAsync(Kh.data(), Kd, sizeof(float) * f_RR, hipMemcpyDeviceToHost, s));
    TDNNF_HIP(hipMemcpyAsync(Lh.data(), Ld, sizeof(float) * f_RR, hipMemcpyDeviceToHost, s));
  }
  RowsGemmArgs b;
  memset(&b, 0, sizeof(b));
  b.A = H; b.lda = R; b.B = ng->W; b.ldb = D; b.C = X.data; b.ldc = X.stride; b.M = N; b.N = D; b.init_mode = 0; b.nseg = 1;
  b.coef = ng->neg_one;
  b.seg[0].klen = R; b.seg[0].m_lo = 0; b.seg[0].m_hi = N;
  TDNNF_HIP(rows_gemm(b, false, s));  // X_hat = X - H W
  double tr1;
  if ((rc = sumsq_host(X, partial, s, &tr1))) return rc;  // also completes the K/L copies
  if (scale) *scale = tr0 <= 0.0 ? 1.0f : (float)sqrt(tr0 / tr1);

  if (upd) {
    float eta = 1.0f - expf(-(float)N / ng->num_samples_history);
    if (eta > 0.9f) eta = 0.9f;
    const float rho_t = ng->rho, alpha = ng->alpha;
    double d_sum = 0;
    for (int i = 0; i < R; i++) d_sum += ng->d[i];
    const double beta_t = rho_t * (1.0 + alpha) + alpha * d_sum / D;
    std::vector<double> sqrt_e, inv_sqrt_e;
    compute_et(ng->d, beta_t, sqrt_e, inv_sqrt_e);
    std::vector<double> Z(f_RR), c, U;
    const double eN = (double)eta / N, eN1 = eN * (1.0 - eta);
    for (int i = 0; i < R; i++)
      for (int j = 0; j < R; j++) {
        const double di = ng->d[i] + rho_t, dj = ng->d[j] + rho_t;
        double z = eN * eN * inv_sqrt_e[i] * Kh[i * R + j] * inv_sqrt_e[j] + eN1 * inv_sqrt_e[i] * Lh[i * R + j] * inv_sqrt_e[j] * (di + dj);
        if (i == j) z += (1.0 - eta) * (1.0 - eta) * di * di;
        Z[i * R + j] = z;
      }
    for (int i = 0; i < R; i++)
      for (int j = 0; j < i; j++) Z[i * R + j] = Z[j * R + i] = 0.5 * (Z[i * R + j] + Z[j * R + i]);
    jacobi_eig(Z, R, c, U);
    const double c_floor = pow(rho_t * (1.0 - eta), 2);
    bool must_reorthogonalize = c[0] > 1.0e+06 * c[R - 1];  // condition_threshold
    std::vector<double> sqrt_c(R);
    double sqrt_c_sum = 0, sqrt_c_max = 0;
    for (int i = 0; i < R; i++) {
      if (c[i] < c_floor) {
        c[i] = c_floor;
        must_reorthogonalize = true;
      }
      sqrt_c[i] = sqrt(c[i]);
      sqrt_c_sum += sqrt_c[i];
      sqrt_c_max = std::max(sqrt_c_max, sqrt_c[i]);
    }
    float rho_t1 = (float)(1.0 / (D - R) * (eta / N * tr0 + (1 - eta) * (D * rho_t + d_sum) - sqrt_c_sum));
    const float floor_val = std::max(ng->epsilon, ng->delta * (float)sqrt_c_max);
    std::vector<float> d_t1(R);
    for (int i = 0; i < R; i++) d_t1[i] = std::max((float)sqrt_c[i] - rho_t1, floor_val);
    if (rho_t1 < floor_val) rho_t1 = floor_val;
    double d1_sum = 0;
    for (int i = 0; i < R; i++) d1_sum += d_t1[i];
    const double beta_t1 = rho_t1 * (1.0 + alpha) + alpha * d1_sum / D;
    std::vector<double> sqrt_e1, inv_sqrt_e1;
    compute_et(d_t1, beta_t1, sqrt_e1, inv_sqrt_e1);
    std::vector<float> coeff_h(R), At(f_RR);
    for (int r = 0; r < R; r++) coeff_h[r] = (float)((1.0 - eta) / (eta / N) * (ng->d[r] + rho_t));
    for (int i = 0; i < R; i++)
      for (int j = 0; j < R; j++) At[i * R + j] = (float)(U[j * R + i] * (eta / N) * sqrt_e1[i] / sqrt_c[i] * inv_sqrt_e[j]);
    TDNNF_HIP(hipMemcpyAsync(coeff, coeff_h.data(), sizeof(float) * R, hipMemcpyHostToDevice, s));
    TDNNF_HIP(hipMemcpyAsync(Ad, At.data(), sizeof(float) * f_RR, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(add_diag_rows_kernel, dim3(grid_for((long long)R * D, 256)), dim3(256), 0, s, J, ng->W, coeff, R, D);
    RowsGemmArgs w1;
    memset(&w1, 0, sizeof(w1));
    w1.A = Ad; w1.lda = R; w1.B = J; w1.ldb = D; w1.C = W1; w1.ldc = D; w1.M = R; w1.N = D; w1.init_mode = 2; w1.nseg = 1;
    w1.seg[0].klen = R; w1.seg[0].m_lo = 0; w1.seg[0].m_hi = R;
    TDNNF_HIP(rows_gemm(w1, false, s));  // W_{t+1} = A_t B_t
    if (must_reorthogonalize) {  // ReorthogonalizeRt1 (UPSTREAM): R_{t+1} = E_{t+1}^{-1/2} W_{t+1} back to orthonormal rows
      RowsGemmArgs o;
      memset(&o, 0, sizeof(o));
      o.A = W1; o.lda = D; o.B = W1; o.ldb = D; o.C = Kd; o.ldc = R; o.M = R; o.N = R; o.init_mode = 2; o.nseg = 1;
      o.seg[0].klen = D; o.seg[0].m_lo = 0; o.seg[0].m_hi = R;
      TDNNF_HIP(rows_gemm(o, true, s));  // O = W W^T
      std::vector<float> Oh(f_RR);
      TDNNF_HIP(hipMemcpyAsync(Oh.data(), Kd, sizeof(float) * f_RR, hipMemcpyDeviceToHost, s));
      TDNNF_HIP(hipStreamSynchronize(s));
      std::vector<double> O(f_RR), Cm(f_RR, 0.0), Ci(f_RR, 0.0);
      bool is_unit = true;
      for (int i = 0; i < R; i++)
        for (int j = 0; j <= i; j++) {
          const double a = (double)Oh[i * R + j] * inv_sqrt_e1[i] * inv_sqrt_e1[j];
          O[i * R + j] = O[j * R + i] = a;
          if (fabs(a - (i == j ? 1.0 : 0.0)) > 1.0e-03) is_unit = false;
        }
      if (!is_unit) {
        bool ok = true;
        for (int i = 0; i < R && ok; i++)
          for (int j = 0; j <= i; j++) {
            double sum = O[i * R + j];
            for (int k = 0; k < j; k++) sum -= Cm[i * R + k] * Cm[j * R + k];
            if (i == j) {
              if (!(sum > 0.0)) { ok = false; break; }
              Cm[i * R + i] = sqrt(sum);
            } else {
              Cm[i * R + j] = sum / Cm[j * R + j];
            }
          }
        double cmax = 0;
        if (ok) {
          for (int i = 0; i < R; i++) {
            Ci[i * R + i] = 1.0 / Cm[i * R + i];
            for (int j = 0; j < i; j++) {
              double sum = 0;
              for (int k = j; k < i; k++) sum += Cm[i * R + k] * Ci[k * R + j];
              Ci[i * R + j] = -sum / Cm[i * R + i];
            }
          }
          for (auto v : Ci) cmax = std::max(cmax, v);
          if (!(cmax < 100.0)) ok = false;
        }
        if (!ok) {  // Gram-Schmidt on the host, then W = E^{1/2} R
          std::vector<float> Wh(f_J);
          TDNNF_HIP(hipMemcpyAsync(Wh.data(), W1, sizeof(float) * f_J, hipMemcpyDeviceToHost, s));
          TDNNF_HIP(hipStreamSynchronize(s));
          orthogonalize_rows(Wh, R, D);
          for (int i = 0; i < R; i++)
            for (int k = 0; k < D; k++) Wh[(size_t)i * D + k] *= (float)sqrt_e1[i];
          TDNNF_HIP(hipMemcpyAsync(W1, Wh.data(), sizeof(float) * f_J, hipMemcpyHostToDevice, s));
          TDNNF_HIP(hipStreamSynchronize(s));
        } else {  // W <- (E^{1/2} C^{-1} E^{-1/2}) W
          std::vector<float> Th(f_RR, 0.f);
          for (int i = 0; i < R; i++)
            for (int j = 0; j <= i; j++) Th[i * R + j] = (float)(Ci[i * R + j] * sqrt_e1[i] * inv_sqrt_e1[j]);
          TDNNF_HIP(hipMemcpyAsync(Ad, Th.data(), sizeof(float) * f_RR, hipMemcpyHostToDevice, s));
          TDNNF_HIP(hipStreamSynchronize(s));
          RowsGemmArgs t2;
          memset(&t2, 0, sizeof(t2));
          t2.A = Ad; t2.lda = R; t2.B = W1; t2.ldb = D; t2.C = J; t2.ldc = D; t2.M = R; t2.N = D; t2.init_mode = 2; t2.nseg = 1;
          t2.seg[0].klen = R; t2.seg[0].m_lo = 0; t2.seg[0].m_hi = R;
          TDNNF_HIP(rows_gemm(t2, false, s));
          TDNNF_HIP(hipMemcpyAsync(W1, J, sizeof(float) * f_J, hipMemcpyDeviceToDevice, s));
        }
      }
    }
    TDNNF_HIP(hipMemcpyAsync(ng->W, W1, sizeof(float) * f_J, hipMemcpyDeviceToDevice, s));
    TDNNF_HIP(hipStreamSynchronize(s));  // host vectors coeff_h / At go out of scope
    ng->d = d_t1;
    ng->rho = rho_t1;
  }
  ng->t += 1;
  return TDNNF_OK;
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

int tdnnf_ng_create(int rank, int update_period, float num_samples_history, float alpha, tdnnf_ng **out) {
  TDNNF_REQUIRE(out && rank >= 0 && update_period >= 1 && num_samples_history > 0 && alpha >= 0, "ng_create: bad configuration");
  tdnnf_ng *ng = new tdnnf_ng();
  ng->rank = rank;
  ng->update_period = update_period;
  ng->t = 0;
  ng->D = 0;
  ng->frozen = 0;
  ng->num_samples_history = num_samples_history;
  ng->alpha = alpha;
  ng->epsilon = 1.0e-10f;
  ng->delta = 5.0e-04f;
  ng->rho = 0;
  ng->W = nullptr;
  ng->scratch = nullptr;
  ng->scratch_floats = 0;
  ng->neg_one = nullptr;
  *out = ng;
  return TDNNF_OK;
}

void tdnnf_ng_destroy(tdnnf_ng *ng) {
  if (!ng) return;
  hipFree(ng->W);
  hipFree(ng->scratch);
  hipFree(ng->neg_one);
  delete ng;
}

int tdnnf_ng_precondition(tdnnf_ng *ng, tdnnf_mat *X, float *scale_host, tdnnf_stream stream) {
  TDNNF_REQUIRE(ng && mat_ok(X) && X->rows > 0 && X->cols > 0, "ng_precondition: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (X->cols == 1) {  // preconditioning one column is pointless (UPSTREAM)
    if (scale_host) *scale_host = 1.0f;
    return TDNNF_OK;
  }
  if (!ng->neg_one) {
    const float m1 = -1.0f;
    TDNNF_HIP(hipMalloc((void **)&ng->neg_one, 16));
    TDNNF_HIP(hipMemcpy(ng->neg_one, &m1, sizeof(float), hipMemcpyHostToDevice));
  }
  MatView xv = view(X);
  if (ng->t == 0 && ng->W == nullptr && ng->D == 0) {  // Init(): default state + self-training on this minibatch
    int rc = init_default(ng, X->cols);
    if (rc) return rc;
    if (ng->rank > 0) {
      const int iters = X->rows <= ng->rank ? 1 : 3;
      float *copy = nullptr;
      TDNNF_HIP(hipMalloc((void **)&copy, sizeof(float) * (size_t)X->rows * X->cols));
      const int was_frozen = ng->frozen;
      ng->frozen = 0;
      ng->t = 1;
      for (int i = 0; i < iters && rc == 0; i++) {
        hipMemcpy2DAsync(copy, sizeof(float) * X->cols, X->data, sizeof(float) * X->stride, sizeof(float) * X->cols, X->rows,
                         hipMemcpyDeviceToDevice, s);
        float sc;
        rc = precondition_step(ng, MatView{copy, X->rows, X->cols, X->cols}, &sc, s);
      }
      hipStreamSynchronize(s);
      hipFree(copy);
      ng->frozen = was_frozen;
      ng->t = 0;
      if (rc) return rc;
    }
  }
  TDNNF_REQUIRE(ng->D == X->cols, "ng_precondition: dimension changed from %d to %d", ng->D, X->cols);
  if (ng->rank == 0) {
    if (scale_host) *scale_host = 1.0f;
    return TDNNF_OK;
  }
  return precondition_step(ng, xv, scale_host, s);
}

}  // extern "C"
