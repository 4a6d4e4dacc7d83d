// ng.hip -- OnlineNaturalGradient::PreconditionDirections on gfx950 (UPSTREAM Kaldi
// nnet3/natural-gradient-online.{h,cc}; call sites /root/reference/src/nnet3/nnet-tdnn-component.cc:598-599,
// nnet-simple-component.cc:3001-3002; configuration :183-210).  SURVEY.md 8(a) row A8.
//
// Fisher model F ~ R^T D R + rho I of rank R; the preconditioned directions are X^ = X - (X W^T) W with
// W = E^{1/2} R, returned with scale = sqrt(tr(X X^T) / tr(X^ X^^T)).  What is N-sized (N = frames x sequences):
//   every call   H = X W^T                         one pass over X on the MFMA rows-GEMM (128x32 tile for R <= 32);
//                tr(X X^T)                         by-product of staging X in that GEMM
//                L = H^T H,  tr(X^ X^^T) = tr(XX^T) - 2 tr(L) + <L, W W^T>      (R x R, no second pass over X)
//   refresh call J = H^T X, K = J J^T              second pass (first 10 calls, then every update_period-th)
// The R x R symmetric eigen-problem of the refresh is solved on the host in double, where the reference solves it,
// but off the critical path: K, L and tr(XX^T) are copied to pinned memory, an event behind the copies hands them to a
// worker thread, and W_{t+1} = A_t (J + diag(c) W_t) is formed on the device the next time this object is used
// (W_{t+1} is not needed earlier).  X itself is only rewritten by tdnnf_ng_precondition (the component-level entry
// point); the trainer never materialises X^ (ng.h).
#include "ng.h"

#include <math.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "gemm_f32.h"
#include "host_linalg.h"


namespace tdnnf {
namespace {

inline int pad4(int x) { return (x + 3) & ~3; }
inline size_t pad4z(size_t x) { return (x + 3) & ~(size_t)3; }

// ------------------------------------------------------------------ small device kernels
__global__ void add_diag_rows_kernel(float *J, const float *W, const float *coeff, int R, int D) {
  const long long total = (long long)R * D;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) J[e] += coeff[e / D] * W[e];
}
// WT[d][r] = W[r][d];  wlast[r] = W[r][D-1]
__global__ void derive_kernel(const float *W, int Rp, int D, int Dp, float *WT, float *wlast) {
  const long long total = (long long)D * Rp;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int dd = (int)(e / Rp), r = (int)(e % Rp);
    const float v = W[(size_t)r * Dp + dd];
    WT[e] = v;
    if (dd == D - 1) wlast[r] = v;
  }
}
// WP[(i * Rp + r) * Di + k] = W[r * Dp + i * Di + k]: the taps' blocks of W_t one below the other (K Rp x Di, k contiguous)
__global__ void stack_taps_kernel(const float *W, int Rp, int Dp, int Di, int K, float *WP) {
  const long long total = (long long)K * Rp * Di;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int k = (int)(e % Di), ir = (int)(e / Di), i = ir / Rp, r = ir % Rp;
    WP[e] = W[(size_t)r * Dp + (size_t)i * Di + k];
  }
}
// H[m][r] = sum_i P[m + o_i][i * Rp + r] (+ bias[r]);  part[b] = sum_i psum[b + o_i / 128] for the 128-row block b (o_i % 128 == 0)
struct PformTaps {
  int K, o[kMaxSeg];
};
__global__ __launch_bounds__(256) void pform_combine_kernel(const float *P, int ldp, PformTaps tp, int Rp, const float *bias, float *H, int N,
                                                            const double *psum, double *part, int part_cap) {
  const int m0 = blockIdx.x * 128, per = Rp / 4;  // float4 per row of H
  for (int e = threadIdx.x; e < 128 * per; e += 256) {
    const int m = m0 + e / per, q = e % per;
    if (m >= N) break;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) v = *reinterpret_cast<const float4 *>(bias + 4 * q);
    for (int i = 0; i < tp.K; i++) {
      const float4 t = *reinterpret_cast<const float4 *>(P + (size_t)(m + tp.o[i]) * ldp + i * Rp + 4 * q);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    *reinterpret_cast<float4 *>(H + (size_t)m * Rp + 4 * q) = v;
  }
  if (threadIdx.x == 0) {
    double sacc = 0;
    for (int i = 0; i < tp.K; i++) sacc += psum[blockIdx.x + tp.o[i] / 128];
    part[blockIdx.x] = sacc;
  }
  for (int i = gridDim.x + blockIdx.x * 256 + threadIdx.x; i < part_cap; i += gridDim.x * 256) part[i] = 0.0;
}
__global__ void scatter_col_kernel(const float *v, int R, float *J, int ld, int col) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) J[(size_t)r * ld + col] = v[r];
}
// tr0 = sum partial + ones_term;  tr1 = tr0 - 2 tr(L) + <L, WWT>;  scale = sqrt(tr0 / tr1)
__global__ __launch_bounds__(256) void ng_scalars_kernel(const double *partial, int nb, double ones_term, const float *L, const float *WWT,
                                                         int Rp, double *scal, float *scale_f) {
  __shared__ double red[3][4];
  double a = 0, b = 0, c = 0;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
  for (int i = threadIdx.x; i < Rp * Rp; i += 256) {
    const double l = L[i];
    c += l * (double)WWT[i];
    if (i / Rp == i % Rp) b += l;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tr0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]) + ones_term;
    const double trL = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double trLW = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const double tr1 = tr0 - 2.0 * trL + trLW;
    scal[0] = tr0;
    scal[1] = tr1;
    *scale_f = (tr0 <= 0.0 || !(tr1 > 0.0)) ? 1.0f : (float)sqrt(tr0 / tr1);
  }
}

// ------------------------------------------------------------------ worker pool for the host part of a refresh
void host_update(tdnnf_ng *ng);

struct NgPool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<tdnnf_ng *> q;
  bool started = false;
  void start() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
    for (int i = 0; i < n; i++) std::thread([this]() { run(); }).detach();
    started = true;
  }
  void push(tdnnf_ng *ng) {
    std::lock_guard<std::mutex> lk(mu);
    if (!started) start();
    q.push_back(ng);
    cv_job.notify_one();
  }
  void run() {
    for (;;) {
      tdnnf_ng *ng;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [this]() { return !q.empty(); });
        ng = q.front();
        q.pop_front();
      }
      // the job was queued when its copies were enqueued, not when they finished (a stream callback for that stalls the
      // stream for ~0.1 ms per refresh, 72 of them in a refresh step): wait for them here
      (void)hipSetDevice(ng->device);
      (void)hipEventSynchronize(ng->ev_wait ? ng->ev_wait : ng->ev_job);
      host_update(ng);
      {
        std::lock_guard<std::mutex> lk(mu);
        ng->job_done = 1;
      }
      cv_done.notify_all();
    }
  }
  void wait(tdnnf_ng *ng) {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [ng]() { return ng->job_done != 0; });
  }
  bool done(tdnnf_ng *ng) {
    std::lock_guard<std::mutex> lk(mu);
    return ng->job_done != 0;
  }
};
NgPool &pool() {
  static NgPool *p = new NgPool();  // never destroyed: worker threads outlive static destruction
  return *p;
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

// Z_t, its eigen-decomposition, d_{t+1}, rho_{t+1} and the R x R factor A_t of W_{t+1} = A_t B_t (UPSTREAM
// PreconditionDirectionsInternal, updating branch).  Reads only host memory; runs on a pool thread.
void host_update(tdnnf_ng *ng) {
  const int R = ng->rank, Rp = ng->Rp, D = ng->D, N = ng->job_N;
  const float *Kh = ng->h_K, *Lh = ng->h_L;
  const double tr0 = *ng->h_tr0;
  float eta = 1.0f - expf(-(float)N / ng->num_samples_history);
  if (eta > 0.9f) eta = 0.9f;
  const float rho_t = ng->rho, alpha = ng->alpha;
  double d_sum = 0;
  for (int i = 0; i < R; i++) d_sum += ng->d[i];
  const double beta_t = rho_t * (1.0 + alpha) + alpha * d_sum / D;
  std::vector<double> sqrt_e, inv_sqrt_e;
  compute_et(ng->d, beta_t, sqrt_e, inv_sqrt_e);
  std::vector<double> Z((size_t)R * R), c, U;
  const double eN = (double)eta / N, eN1 = eN * (1.0 - eta);
  for (int i = 0; i < R; i++)
    for (int j = 0; j < R; j++) {
      const double di = ng->d[i] + rho_t, dj = ng->d[j] + rho_t;
      double z = eN * eN * inv_sqrt_e[i] * Kh[i * Rp + j] * inv_sqrt_e[j] + eN1 * inv_sqrt_e[i] * Lh[i * Rp + j] * inv_sqrt_e[j] * (di + dj);
      if (i == j) z += (1.0 - eta) * (1.0 - eta) * di * di;
      Z[(size_t)i * R + j] = z;
    }
  for (int i = 0; i < R; i++)
    for (int j = 0; j < i; j++) Z[(size_t)i * R + j] = Z[(size_t)j * R + i] = 0.5 * (Z[(size_t)i * R + j] + Z[(size_t)j * R + i]);
  hostla::sym_eig(Z, R, c, U);
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
  ng->d_next.resize(R);
  for (int i = 0; i < R; i++) ng->d_next[i] = std::max((float)sqrt_c[i] - rho_t1, floor_val);
  if (rho_t1 < floor_val) rho_t1 = floor_val;
  ng->rho_next = rho_t1;
  double d1_sum = 0;
  for (int i = 0; i < R; i++) d1_sum += ng->d_next[i];
  const double beta_t1 = rho_t1 * (1.0 + alpha) + alpha * d1_sum / D;
  compute_et(ng->d_next, beta_t1, ng->sqrt_e1, ng->inv_sqrt_e1);
  memset(ng->h_coeff, 0, sizeof(float) * Rp);
  memset(ng->h_At, 0, sizeof(float) * (size_t)Rp * Rp);
  for (int r = 0; r < R; r++) ng->h_coeff[r] = (float)((1.0 - eta) / (eta / N) * (ng->d[r] + rho_t));
  for (int i = 0; i < R; i++)
    for (int j = 0; j < R; j++)
      ng->h_At[(size_t)i * Rp + j] = (float)(U[(size_t)j * R + i] * (eta / N) * ng->sqrt_e1[i] / sqrt_c[i] * inv_sqrt_e[j]);
  ng->must_reorth = must_reorthogonalize;
}

// ------------------------------------------------------------------ device state
int alloc_state(tdnnf_ng *ng, int D) {
  if (ng->rank >= D) ng->rank = D - 1;
  const int R = ng->rank, Rp = pad4(std::max(R, 1)), Dp = pad4(D);
  ng->D = D;
  ng->Dp = Dp;
  ng->Rp = Rp;
  const size_t fRD = (size_t)Rp * Dp, fRR = (size_t)Rp * Rp;
  // (W^T: kWtPadRows zero rows behind its D -- ng_valu.hip reads whole K steps of up to 64 rows, the tile's columns there are zeros)
  const size_t fWT = pad4z((size_t)(D + kWtPadRows) * Rp);
  const size_t floats = 3 * fRD + fWT + 4 * fRR + 3 * Rp + 8 + 8;
  TDNNF_HIP(hipMalloc((void **)&ng->dev, sizeof(float) * floats));
  TDNNF_HIP(hipMemset(ng->dev, 0, sizeof(float) * floats));
  float *p = ng->dev;
  ng->W = p; p += fRD;
  ng->J = p; p += fRD;
  ng->W1 = p; p += fRD;
  ng->WT = p; p += fWT;
  ng->WWT = p; p += fRR;
  ng->Kd = p; p += fRR;
  ng->Ld = p; p += fRR;
  ng->Ad = p; p += fRR;
  ng->wlast = p; p += Rp;
  ng->coeff = p; p += Rp;
  ng->tmpR = p; p += Rp;
  ng->neg_one = p; p += 4;
  ng->scale_f = p; p += 4;
  ng->scal = (double *)p;
  const size_t pin_floats = 3 * fRR + Rp + 4 + 4;
  TDNNF_HIP(hipHostMalloc((void **)&ng->pin, sizeof(float) * pin_floats, hipHostMallocDefault));
  if (!ng->ev_job) {
    TDNNF_HIP(hipGetDevice(&ng->device));
    TDNNF_HIP(hipEventCreateWithFlags(&ng->ev_job, hipEventDisableTiming | hipEventBlockingSync));
  }
  memset(ng->pin, 0, sizeof(float) * pin_floats);
  float *h = ng->pin;
  ng->h_K = h; h += fRR;
  ng->h_L = h; h += fRR;
  ng->h_At = h; h += fRR;
  ng->h_coeff = h; h += Rp;
  ng->h_scale = h; h += 4;
  ng->h_tr0 = (double *)h;
  const float consts[8] = {-1.0f, 0, 0, 0, 1.0f, 0, 0, 0};  // neg_one, scale_f
  TDNNF_HIP(hipMemcpy(ng->neg_one, consts, sizeof(consts), hipMemcpyHostToDevice));
  return TDNNF_OK;
}

int derive(tdnnf_ng *ng, hipStream_t s) {  // W^T, W W^T and the last column of W after W changed
  const int Rp = ng->Rp, D = ng->D, Dp = ng->Dp;
  hipLaunchKernelGGL(derive_kernel, dim3(grid_for((long long)D * Rp, 256)), dim3(256), 0, s, ng->W, Rp, D, Dp, ng->WT, ng->wlast);
  RowsGemmArgs k;
  memset(&k, 0, sizeof(k));
  k.A = ng->W; k.lda = Dp; k.B = ng->W; k.ldb = Dp; k.C = ng->WWT; k.ldc = Rp; k.M = Rp; k.N = Rp; k.init_mode = 2; k.nseg = 1;
  k.seg[0].klen = Dp; k.seg[0].m_lo = 0; k.seg[0].m_hi = Rp;
  TDNNF_HIP(rows_gemm(k, true, s));
  return TDNNF_OK;
}

int init_default(tdnnf_ng *ng, int D, hipStream_t s) {  // InitDefault (UPSTREAM)
  int rc = alloc_state(ng, D);
  if (rc) return rc;
  const int R = ng->rank, Dp = ng->Dp;
  ng->d.assign(R, ng->epsilon);
  ng->rho = ng->epsilon;
  ng->t = 0;
  if (R == 0) return TDNNF_OK;
  std::vector<float> W((size_t)ng->Rp * Dp, 0.f);
  const float first_elem = 1.1f;
  const float E_tii = 1.0f / (2.0f + (D + R) * ng->alpha / D);
  for (int r = 0; r < R; r++) {  // InitOrthonormalSpecial
    int ncols = 0;
    for (int c = r; c < D; c += R) ncols++;
    const float normalizer = 1.0f / sqrtf(first_elem * first_elem + ncols - 1);
    int i = 0;
    for (int c = r; c < D; c += R, i++) W[(size_t)r * Dp + c] = normalizer * (i == 0 ? first_elem : 1.0f) * sqrtf(E_tii);
  }
  TDNNF_HIP(hipMemcpyAsync(ng->W, W.data(), sizeof(float) * W.size(), hipMemcpyHostToDevice, s));
  TDNNF_HIP(hipStreamSynchronize(s));
  return derive(ng, s);
}

bool updating(const tdnnf_ng *ng) {
  return !ng->frozen && (ng->t <= 10 || (ng->t - 10) % ng->update_period == 0);
}

// ReorthogonalizeRt1 (UPSTREAM): bring R_{t+1} = E_{t+1}^{-1/2} W_{t+1} back to orthonormal rows.  Rare; synchronous.
int reorthogonalize(tdnnf_ng *ng, hipStream_t s) {
  const int R = ng->rank, Rp = ng->Rp, D = ng->D, Dp = ng->Dp;
  const size_t fRR = (size_t)Rp * Rp, fRD = (size_t)Rp * Dp;
  RowsGemmArgs o;
  memset(&o, 0, sizeof(o));
  o.A = ng->W1; o.lda = Dp; o.B = ng->W1; o.ldb = Dp; o.C = ng->Kd; o.ldc = Rp; o.M = Rp; o.N = Rp; o.init_mode = 2; o.nseg = 1;
  o.seg[0].klen = Dp; o.seg[0].m_lo = 0; o.seg[0].m_hi = Rp;
  TDNNF_HIP(rows_gemm(o, true, s));  // O = W W^T
  std::vector<float> Oh(fRR);
  TDNNF_HIP(hipMemcpyAsync(Oh.data(), ng->Kd, sizeof(float) * fRR, hipMemcpyDeviceToHost, s));
  TDNNF_HIP(hipStreamSynchronize(s));
  std::vector<double> O((size_t)R * R), Cm, Ci;
  bool is_unit = true;
  for (int i = 0; i < R; i++)
    for (int j = 0; j <= i; j++) {
      const double a = (double)Oh[(size_t)i * Rp + j] * ng->inv_sqrt_e1[i] * ng->inv_sqrt_e1[j];
      O[(size_t)i * R + j] = O[(size_t)j * R + i] = a;
      if (fabs(a - (i == j ? 1.0 : 0.0)) > 1.0e-03) is_unit = false;
    }
  if (is_unit) return TDNNF_OK;
  bool ok = hostla::cholesky_inverse(O, R, Cm, Ci);
  if (ok) {
    double cmax = 0;
    for (auto v : Ci) cmax = std::max(cmax, v);
    if (!(cmax < 100.0)) ok = false;
  }
  if (!ok) {  // Gram-Schmidt on the host, then W = E^{1/2} R
    std::vector<float> Wh(fRD);
    TDNNF_HIP(hipMemcpyAsync(Wh.data(), ng->W1, sizeof(float) * fRD, hipMemcpyDeviceToHost, s));
    TDNNF_HIP(hipStreamSynchronize(s));
    hostla::orthogonalize_rows(Wh, R, D, Dp);
    for (int i = 0; i < R; i++)
      for (int k = 0; k < D; k++) Wh[(size_t)i * Dp + k] *= (float)ng->sqrt_e1[i];
    TDNNF_HIP(hipMemcpyAsync(ng->W1, Wh.data(), sizeof(float) * fRD, hipMemcpyHostToDevice, s));
    TDNNF_HIP(hipStreamSynchronize(s));
    return TDNNF_OK;
  }
  std::vector<float> Th(fRR, 0.f);  // W <- (E^{1/2} C^{-1} E^{-1/2}) W
  for (int i = 0; i < R; i++)
    for (int j = 0; j <= i; j++) Th[(size_t)i * Rp + j] = (float)(Ci[(size_t)i * R + j] * ng->sqrt_e1[i] * ng->inv_sqrt_e1[j]);
  TDNNF_HIP(hipMemcpyAsync(ng->Ad, Th.data(), sizeof(float) * fRR, hipMemcpyHostToDevice, s));
  TDNNF_HIP(hipStreamSynchronize(s));
  RowsGemmArgs t2;
  memset(&t2, 0, sizeof(t2));
  t2.A = ng->Ad; t2.lda = Rp; t2.B = ng->W1; t2.ldb = Dp; t2.C = ng->J; t2.ldc = Dp; t2.M = Rp; t2.N = Dp; t2.init_mode = 2; t2.nseg = 1;
  t2.seg[0].klen = Rp; t2.seg[0].m_lo = 0; t2.seg[0].m_hi = Rp;
  TDNNF_HIP(rows_gemm(t2, false, s));
  TDNNF_HIP(hipMemcpyAsync(ng->W1, ng->J, sizeof(float) * fRD, hipMemcpyDeviceToDevice, s));
  return TDNNF_OK;
}

// Second half of a refresh: wait for the host part, then W_{t+1} = A_t (J + diag(coeff) W_t) on the device.
int finalize(tdnnf_ng *ng, hipStream_t s) {
  if (!ng->pending) return TDNNF_OK;
  pool().wait(ng);
  ng->pending = 0;
  const int Rp = ng->Rp, Dp = ng->Dp;
  const size_t fRR = (size_t)Rp * Rp, fRD = (size_t)Rp * Dp;
  TDNNF_HIP(hipMemcpyAsync(ng->coeff, ng->h_coeff, sizeof(float) * Rp, hipMemcpyHostToDevice, s));
  TDNNF_HIP(hipMemcpyAsync(ng->Ad, ng->h_At, sizeof(float) * fRR, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(add_diag_rows_kernel, dim3(grid_for((long long)fRD, 256)), dim3(256), 0, s, ng->J, ng->W, ng->coeff, Rp, Dp);
  RowsGemmArgs w1;
  memset(&w1, 0, sizeof(w1));
  w1.A = ng->Ad; w1.lda = Rp; w1.B = ng->J; w1.ldb = Dp; w1.C = ng->W1; w1.ldc = Dp; w1.M = Rp; w1.N = Dp; w1.init_mode = 2; w1.nseg = 1;
  w1.seg[0].klen = Rp; w1.seg[0].m_lo = 0; w1.seg[0].m_hi = Rp;
  TDNNF_HIP(rows_gemm(w1, false, s));
  if (ng->must_reorth) {
    int rc = reorthogonalize(ng, s);
    if (rc) return rc;
  }
  TDNNF_HIP(hipMemcpyAsync(ng->W, ng->W1, sizeof(float) * fRD, hipMemcpyDeviceToDevice, s));
  ng->d = ng->d_next;
  ng->rho = ng->rho_next;
  return derive(ng, s);
}

// P form of H = X~ W^T for K taps that are row shifts of ONE matrix: P = X [W_0^T | W_1^T | ...] in one pass over X (each row read
// once instead of K times), then H[m] = sum_i P[m + o_i][block i].  Workspace: W's blocks stacked, P, the pass's per-tile ||x||^2.
constexpr int kPformSpanCap = 4096;  // rows between the first and the last tap (<= 2 x 3 frames x 512 sequences)
size_t pform_ws_bytes(int Rp, int Di, int K, int N) {
  if (K < 2 || K * Rp > 64) return 0;
  const size_t M = (size_t)N + kPformSpanCap;
  return sizeof(float) * ((size_t)K * Rp * Di + 64) + sizeof(float) * (M * K * Rp + 64) + sizeof(double) * ((size_t)rows_gemm_sumsq_blocks((int)M) + 8) + 256;
}
size_t stats_ws_bytes(int Rp, int Di, int K, int N) {
  const size_t part = ((size_t)rows_gemm_sumsq_blocks(N) * sizeof(double) + 63) & ~(size_t)63;
  return part + std::max(std::max(wgrad_workspace_bytes(Rp, Rp, 1, N), wgrad_workspace_bytes(Rp, Di, K, N)), pform_ws_bytes(Rp, Di, K, N));
}

// What follows H in the first half: the bookkeeping of the call in flight and, on a refresh, J = H^T X.
int stats_after_h(tdnnf_ng *ng, const NgInput &in, const float *H, void *wg_ws, size_t wg_bytes, bool upd, hipStream_t s) {
  const int N = in.N, K = in.ix.num_offsets, Di = in.Di, Rp = ng->Rp, Dp = ng->Dp, D = ng->D;
  ng->cur_upd = upd;
  ng->cur_N = N;
  ng->cur_ones = in.ones;
  if (!upd) return TDNNF_OK;
  TDNNF_REQUIRE(wg_ws && wg_bytes >= wgrad_workspace_bytes(Rp, Di, K, N), "ng: workspace too small");
  TDNNF_HIP(hipMemsetAsync(ng->J, 0, sizeof(float) * (size_t)Rp * Dp, s));
  if (in.ones) TDNNF_HIP(hipMemsetAsync(ng->tmpR, 0, sizeof(float) * Rp, s));
  WgradArgs j;
  memset(&j, 0, sizeof(j));
  j.dY = H; j.lddy = Rp; j.X = in.x.data; j.ldx = in.x.stride; j.Do = Rp; j.Di = Di; j.K = K; j.N = N; j.row_stride = in.ix.row_stride;
  for (int i = 0; i < K; i++) j.row_offsets[i] = in.ix.row_offsets[i];
  j.coef = in.eff; j.scale = 1.f; j.G = ng->J; j.ldg = Dp; j.accumulate = 1; j.bias_acc = in.ones ? ng->tmpR : nullptr;
  j.active = in.active; j.max_active = in.max_active;
  TDNNF_HIP(wgrad(j, wg_ws, wg_bytes, s));  // J = H^T X  (last column: column sums of H)
  if (in.ones) hipLaunchKernelGGL(scatter_col_kernel, dim3((Rp + 63) / 64), dim3(64), 0, s, ng->tmpR, Rp, ng->J, Dp, D - 1);
  return TDNNF_OK;
}

// (see pform_ws_bytes)  Applies to: taps of one matrix at rows 128 apart in whole tiles (o_i % 128 == 0, N % 128 == 0: the pass's
// per-tile ||x||^2 then add up to each tap's window exactly), no tap coefficients, every row a row of X (row_stride 1), minibatches
// whose passes are bound by the read of X (>= 32 768 rows; below, the K-tap kernel's rows are latency-bound launches either way).
bool pform_ok(int Rp, const NgInput &in, size_t ws_bytes) {
  const int K = in.ix.num_offsets, N = in.N;
  if (!options().ng_pform || K < 2 || K * Rp > 64 || in.eff || in.active || in.ix.row_stride != 1 || N % 128 != 0 || N < 32768) return false;
  int lo = in.ix.row_offsets[0], hi = lo;
  for (int i = 1; i < K; i++) {
    lo = std::min(lo, in.ix.row_offsets[i]);
    hi = std::max(hi, in.ix.row_offsets[i]);
  }
  for (int i = 0; i < K; i++)
    if ((in.ix.row_offsets[i] - lo) % 128 != 0) return false;
  if (hi - lo > kPformSpanCap || in.Di % 4 != 0 || (in.x.stride % 4) != 0 || in.Di < 1024) return false;  // (narrow matrices -- 160 columns -- measured slower: 78 against 61 us)
  return ws_bytes >= pform_ws_bytes(Rp, in.Di, K, N);
}
// W: rank_padded x ldw (row r = [tap 0's Di columns | tap 1's | ... | the ones' column]); bias: that last column, or null
int pform_pass(const float *W, int Rp, int Dp, const float *bias, const NgInput &in, float *H, double *part, void *ws, hipStream_t s) {
  const int K = in.ix.num_offsets, N = in.N, Di = in.Di;
  PformTaps tp;
  memset(&tp, 0, sizeof(tp));
  tp.K = K;
  int lo = in.ix.row_offsets[0], hi = lo;
  for (int i = 1; i < K; i++) {
    lo = std::min(lo, in.ix.row_offsets[i]);
    hi = std::max(hi, in.ix.row_offsets[i]);
  }
  for (int i = 0; i < K; i++) tp.o[i] = in.ix.row_offsets[i] - lo;
  const int M = N + (hi - lo);
  float *WP = (float *)ws;
  float *P = WP + (((size_t)K * Rp * Di + 63) & ~(size_t)63);
  double *psum = (double *)(P + (((size_t)M * K * Rp + 63) & ~(size_t)63));
  hipLaunchKernelGGL(stack_taps_kernel, dim3(grid_for((long long)K * Rp * Di, 256)), dim3(256), 0, s, W, Rp, Dp, Di, K, WP);
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = in.x.data + (size_t)lo * in.x.stride; a.lda = in.x.stride; a.B = WP; a.ldb = Di; a.C = P; a.ldc = K * Rp; a.M = M; a.N = K * Rp;
  a.init_mode = 2; a.sumsq = psum; a.nseg = 1;
  a.seg[0].klen = Di; a.seg[0].m_lo = 0; a.seg[0].m_hi = M;
  TDNNF_HIP(rows_gemm(a, true, s));  // P = X [W_0^T | W_1^T ...]  (+ ||x||^2 per 128-row tile)
  hipLaunchKernelGGL(pform_combine_kernel, dim3(N / 128), dim3(256), 0, s, P, K * Rp, tp, Rp, bias, H, N, psum, part,
                     rows_gemm_sumsq_blocks(N));
  TDNNF_HIP(hipGetLastError());
  return TDNNF_OK;
}

// First half of one PreconditionDirections call, everything N x D sized: H = X W_t^T (with ||X||^2 per block into
// `part`) and, on a refresh, J = H^T X.  W_t is left untouched.
RowsGemmArgs stats_h_args(const tdnnf_ng *ng, const NgInput &in, float *H, double *part) {
  const int N = in.N, K = in.ix.num_offsets, Di = in.Di, Rp = ng->Rp, Dp = ng->Dp;
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = in.x.data; a.lda = (long long)in.x.stride * in.ix.row_stride; a.B = ng->W; a.ldb = Dp; a.C = H; a.ldc = Rp; a.M = N; a.N = Rp;
  a.bias = in.ones ? ng->wlast : nullptr;
  a.init_mode = in.ones ? 1 : 2;
  a.coef = in.eff;
  a.sumsq = part;
  a.nseg = K;
  for (int i = 0; i < K; i++) {
    a.seg[i].a_off = (long long)in.ix.row_offsets[i] * in.x.stride;
    a.seg[i].b_off = (long long)i * Di;
    a.seg[i].klen = Di;
    a.seg[i].m_lo = 0;
    a.seg[i].m_hi = N;
  }
  return a;
}
int stats_main(tdnnf_ng *ng, const NgInput &in, float *H, double *part, void *wg_ws, size_t wg_bytes, bool upd, hipStream_t s) {
  const int N = in.N, K = in.ix.num_offsets, Di = in.Di, Rp = ng->Rp, Dp = ng->Dp;
  const RowsGemmArgs a = stats_h_args(ng, in, H, part);
  // the same product on the vector ALUs, beside the matrix-core GEMMs of the other streams (ng_valu.hip), where its shape allows
  NgRowdotArgs v;
  memset(&v, 0, sizeof(v));
  v.X = in.x.data; v.ldx = in.x.stride; v.row_stride = in.ix.row_stride; v.nseg = K; v.Di = Di; v.eff = in.eff; v.WT = ng->WT; v.Rp = Rp;
  v.bias = in.ones ? ng->wlast : nullptr; v.H = H; v.ldh = Rp; v.N = N; v.part = part; v.part_cap = rows_gemm_sumsq_blocks(N);
  for (int i = 0; i < K; i++) v.seg_off[i] = (long long)in.ix.row_offsets[i] * in.x.stride;
  const int skip = options().ng_diag_skip;
  if (skip && ng->D != 0 && (((skip & 1) && K == 2 && Di >= 1024) || ((skip & 2) && !(K == 2 && Di >= 1024)))) {
    TDNNF_HIP(hipMemsetAsync(H, 0, sizeof(float) * (size_t)N * Rp, s));
    TDNNF_HIP(hipMemsetAsync(part, 0, sizeof(double) * rows_gemm_sumsq_blocks(N), s));
  } else if (pform_ok(Rp, in, wg_bytes)) {
    int rc = pform_pass(ng->W, Rp, Dp, in.ones ? ng->wlast : nullptr, in, H, part, wg_ws, s);
    if (rc) return rc;
  } else if (options().ng_valu && !in.active && ng_rowdot_ok(v)) {
    TDNNF_HIP(ng_rowdot(v, s));
  } else {
    TDNNF_HIP(rows_gemm(a, true, s));  // H = X W^T (+ ||X||_F^2 per block)
  }
  return stats_after_h(ng, in, H, wg_ws, wg_bytes, upd, s);
}

// Second half, R x R sized and latency bound: L = H^T H, traces and scale; on a refresh K = J J^T and the hand-off to the
// host worker.  May run on another stream than stats_main as long as it is ordered after it.
int stats_side(tdnnf_ng *ng, const float *H, const double *part, void *wg_ws, size_t wg_bytes, hipStream_t s) {
  const int N = ng->cur_N, Rp = ng->Rp, Dp = ng->Dp;
  TDNNF_REQUIRE(wg_ws && wg_bytes >= wgrad_workspace_bytes(Rp, Rp, 1, N), "ng: workspace too small");
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.dY = H; w.lddy = Rp; w.X = H; w.ldx = Rp; w.Do = Rp; w.Di = Rp; w.K = 1; w.N = N; w.row_stride = 1; w.scale = 1.f;
  w.G = ng->Ld; w.ldg = Rp; w.accumulate = 0;
  TDNNF_HIP(wgrad(w, wg_ws, wg_bytes, s));  // L = H^T H
  hipLaunchKernelGGL(ng_scalars_kernel, dim3(1), dim3(256), 0, s, part, rows_gemm_sumsq_blocks(N), ng->cur_ones ? (double)N : 0.0, ng->Ld, ng->WWT,
                     Rp, ng->scal, ng->scale_f);
  if (!ng->cur_upd) return TDNNF_OK;
  RowsGemmArgs k;
  memset(&k, 0, sizeof(k));
  k.A = ng->J; k.lda = Dp; k.B = ng->J; k.ldb = Dp; k.C = ng->Kd; k.ldc = Rp; k.M = Rp; k.N = Rp; k.init_mode = 2; k.nseg = 1;
  k.seg[0].klen = Dp; k.seg[0].m_lo = 0; k.seg[0].m_hi = Rp;
  TDNNF_HIP(rows_gemm(k, true, s));  // K = J J^T
  const size_t fRR = (size_t)Rp * Rp;
  TDNNF_HIP(hipMemcpyAsync(ng->h_K, ng->Kd, sizeof(float) * fRR, hipMemcpyDeviceToHost, s));
  TDNNF_HIP(hipMemcpyAsync(ng->h_L, ng->Ld, sizeof(float) * fRR, hipMemcpyDeviceToHost, s));
  TDNNF_HIP(hipMemcpyAsync(ng->h_tr0, ng->scal, sizeof(double), hipMemcpyDeviceToHost, s));
  ng->job_N = N;
  ng->job_done = 0;
  ng->pending = 1;
  TDNNF_HIP(hipEventRecord(ng->ev_job, s));
  ng->ev_wait = ng->ev_job;
  pool().push(ng);
  return TDNNF_OK;
}

// both halves on one stream; ws = [sumsq partials | wgrad workspace]
int stats_core(tdnnf_ng *ng, const NgInput &in, float *H, void *ws, size_t ws_bytes, bool upd, hipStream_t s) {
  TDNNF_REQUIRE(ws && ws_bytes >= stats_ws_bytes(ng->Rp, in.Di, in.ix.num_offsets, in.N), "ng: workspace too small");
  double *part = (double *)ws;
  const size_t part_bytes = ((size_t)rows_gemm_sumsq_blocks(in.N) * sizeof(double) + 63) & ~(size_t)63;
  void *wg_ws = (char *)ws + part_bytes;
  const size_t wg_bytes = ws_bytes - part_bytes;
  int rc = stats_main(ng, in, H, part, wg_ws, wg_bytes, upd, s);
  if (rc) return rc;
  return stats_side(ng, H, part, wg_ws, wg_bytes, s);
}

// Init(): default state, then self-training on this minibatch (3 refreshes from the same data), all on stream s
int init_from(tdnnf_ng *ng, const NgInput &in, int D, float *H, void *ws, size_t ws_bytes, hipStream_t s) {
  int rc = init_default(ng, D, s);
  if (rc) return rc;
  if (ng->rank > 0) {
    const int iters = in.N <= ng->rank ? 1 : 3;
    for (int i = 0; i < iters; i++) {
      if ((rc = stats_core(ng, in, H, ws, ws_bytes, true, s))) return rc;
      if ((rc = finalize(ng, s))) return rc;
    }
  }
  ng->t = 0;
  return TDNNF_OK;
}

}  // namespace

size_t ng_stats_workspace_bytes(int rank, int D, int K, int N) {
  const int Rp = pad4(std::max(1, std::min(rank, D - 1)));
  const int Di = K > 0 ? D / K : D;
  return stats_ws_bytes(Rp, Di, std::max(K, 1), N) + 64;
}
bool ng_updating(const tdnnf_ng *ng) { return updating(ng); }
void ng_pool_push(tdnnf_ng *ng) { pool().push(ng); }
void ng_pool_wait(tdnnf_ng *ng) { pool().wait(ng); }
bool ng_pool_done(tdnnf_ng *ng) { return pool().done(ng); }
int ng_finalize_one(tdnnf_ng *ng, hipStream_t s) {
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  return finalize(ng, s);
}
int ng_h_ld(const tdnnf_ng *ng) { return ng->Rp; }
int ng_dim(const tdnnf_ng *ng) { return ng->D; }
const float *ng_scale_dev(const tdnnf_ng *ng) { return ng->scale_f; }
const float *ng_w_dev(const tdnnf_ng *ng) { return ng->W; }
int ng_w_ld(const tdnnf_ng *ng) { return ng->Dp; }

int ng_stats_step(tdnnf_ng *ng, const NgInput &in, float *H, void *ws, size_t ws_bytes, hipStream_t s) {
  const int K = in.ix.num_offsets, D = K * in.Di + (in.ones ? 1 : 0);
  TDNNF_REQUIRE(ng && H && in.N > 0 && K >= 1 && K <= kMaxSeg && in.Di > 0, "ng_stats_step: bad arguments");
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  if (ng->D == 0) {
    int rc0 = init_from(ng, in, D, H, ws, ws_bytes, s);
    if (rc0) return rc0;
  }
  TDNNF_REQUIRE(ng->D == D, "ng: dimension changed from %d to %d", ng->D, D);
  if (ng->rank == 0) return TDNNF_OK;
  int rc = finalize(ng, s);  // a refresh started by the previous call on this object
  if (rc) return rc;
  rc = stats_core(ng, in, H, ws, ws_bytes, updating(ng), s);
  ng->t += 1;
  return rc;
}

int ng_stats_main(tdnnf_ng *ng, const NgInput &in, float *H, double *part, void *ws, size_t ws_bytes, hipStream_t s) {
  const int K = in.ix.num_offsets, D = K * in.Di + (in.ones ? 1 : 0);
  TDNNF_REQUIRE(ng && H && part && in.N > 0 && K >= 1 && K <= kMaxSeg && in.Di > 0, "ng_stats_main: bad arguments");
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  if (ng->D == 0) {
    int rc0 = init_from(ng, in, D, H, ws, ws_bytes, s);
    if (rc0) return rc0;
  }
  TDNNF_REQUIRE(ng->D == D, "ng: dimension changed from %d to %d", ng->D, D);
  ng->cur_N = 0;
  if (ng->rank == 0) return TDNNF_OK;
  int rc = finalize(ng, s);  // a refresh started by the previous call on this object
  if (rc) return rc;
  return stats_main(ng, in, H, part, ws, ws_bytes, updating(ng), s);
}

int ng_stats_main_prepare(tdnnf_ng *ng, const NgInput &in, float *H, double *part, hipStream_t s, RowsGemmArgs *out) {
  const int K = in.ix.num_offsets, D = K * in.Di + (in.ones ? 1 : 0);
  TDNNF_REQUIRE(ng && H && part && out && in.N > 0 && K >= 1 && K <= kMaxSeg && ng->D == D, "ng_stats_main_prepare: bad arguments");
  if (ng->rank == 0) {  // nothing to precondition: no pass (out->M = 0 fails rows_gemm_group_ok, the caller leaves the object to its own call)
    memset(out, 0, sizeof(*out));
    ng->cur_N = 0;
    return TDNNF_OK;
  }
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  ng->cur_N = 0;
  int rc = finalize(ng, s);  // a refresh started by the previous call on this object
  if (rc) return rc;
  *out = stats_h_args(ng, in, H, part);
  return TDNNF_OK;
}
int ng_stats_main_finish(tdnnf_ng *ng, const NgInput &in, const float *H, void *ws, size_t ws_bytes, hipStream_t s) {
  TDNNF_REQUIRE(ng && H && ng->D != 0 && ng->rank > 0 && in.N > 0, "ng_stats_main_finish: bad arguments");
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  return stats_after_h(ng, in, H, ws, ws_bytes, updating(ng), s);
}

int ng_finalize_if_ready(tdnnf_ng *ng, hipStream_t s, int *did) {
  *did = 0;
  if (!ng || ng->rank == 0 || !ng->pending || !pool().done(ng)) return TDNNF_OK;
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  *did = 1;
  return finalize(ng, s);
}

int ng_external_begin(tdnnf_ng *ng, int D, const float **W, int *Rp, int *ldw, hipStream_t s) {
  TDNNF_REQUIRE(ng && W && Rp && ldw, "ng_external_begin: bad arguments");
  *W = nullptr;
  if (ng->D == 0 || ng->rank == 0) return TDNNF_OK;  // first minibatch (W_0 comes from the data) / nothing to precondition
  TDNNF_REQUIRE(ng->D == D, "ng: dimension changed from %d to %d", ng->D, D);
  ng->cur_N = 0;
  int rc = finalize(ng, s);  // a refresh started by the previous call on this object
  if (rc) return rc;
  *W = ng->W;
  *Rp = ng->Rp;
  *ldw = ng->Dp;
  return TDNNF_OK;
}

int ng_external_end(tdnnf_ng *ng, const NgInput &in, const float *H, void *ws, size_t ws_bytes, hipStream_t s) {
  TDNNF_REQUIRE(ng && H && ng->D != 0 && ng->rank > 0 && in.N > 0, "ng_external_end: bad arguments");
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  return stats_after_h(ng, in, H, ws, ws_bytes, updating(ng), s);
}

int ng_stats_side(tdnnf_ng *ng, const float *H, const double *part, void *ws, size_t ws_bytes, hipStream_t s) {
  TDNNF_REQUIRE(ng && H && part, "ng_stats_side: bad arguments");
  if (ng->rank == 0 || ng->cur_N == 0) return TDNNF_OK;
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  int rc = stats_side(ng, H, part, ws, ws_bytes, s);
  ng->t += 1;
  return rc;
}

size_t ng_project_tmp_floats(const tdnnf_ng *in, const tdnnf_ng *out, int Do, int ldT) {
  const size_t q = in ? (size_t)Do * in->Rp : 0, p = out ? (size_t)out->Rp * ldT : 0;
  return std::max(q, p) + 16;
}

int ng_project(tdnnf_ng *in, tdnnf_ng *out, float *T, int Do, int Dx, int ldT, float *tmp, hipStream_t s) {
  TDNNF_REQUIRE(T && tmp && ldT % 4 == 0 && ldT >= Dx, "ng_project: bad arguments");
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  if (in && in->rank > 0) {
    TDNNF_REQUIRE(in->D == Dx && in->Dp == ldT, "ng_project: input-side dimension mismatch");
    const int Rp = in->Rp;
    RowsGemmArgs q;
    memset(&q, 0, sizeof(q));
    q.A = T; q.lda = ldT; q.B = in->W; q.ldb = in->Dp; q.C = tmp; q.ldc = Rp; q.M = Do; q.N = Rp; q.init_mode = 2; q.nseg = 1;
    q.seg[0].klen = ldT; q.seg[0].m_lo = 0; q.seg[0].m_hi = Do;
    TDNNF_HIP(rows_gemm(q, true, s));  // Q = T Wx^T
    RowsGemmArgs c;
    memset(&c, 0, sizeof(c));
    c.A = tmp; c.lda = Rp; c.B = in->W; c.ldb = in->Dp; c.C = T; c.ldc = ldT; c.M = Do; c.N = ldT; c.init_mode = 0; c.nseg = 1;
    c.coef = in->neg_one;
    c.seg[0].klen = Rp; c.seg[0].m_lo = 0; c.seg[0].m_hi = Do;
    TDNNF_HIP(rows_gemm(c, false, s));  // T -= Q Wx
  }
  if (out && out->rank > 0) {
    TDNNF_REQUIRE(out->D == Do, "ng_project: output-side dimension mismatch");
    const int Rp = out->Rp;
    RowsGemmArgs p;
    memset(&p, 0, sizeof(p));
    p.A = out->W; p.lda = out->Dp; p.B = T; p.ldb = ldT; p.C = tmp; p.ldc = ldT; p.M = Rp; p.N = ldT; p.init_mode = 2; p.nseg = 1;
    p.seg[0].klen = Do; p.seg[0].m_lo = 0; p.seg[0].m_hi = Rp;
    TDNNF_HIP(rows_gemm(p, false, s));  // P = Wy T
    RowsGemmArgs c;
    memset(&c, 0, sizeof(c));
    c.A = out->WT; c.lda = Rp; c.B = tmp; c.ldb = ldT; c.C = T; c.ldc = ldT; c.M = Do; c.N = ldT; c.init_mode = 0; c.nseg = 1;
    c.coef = out->neg_one;
    c.seg[0].klen = Rp; c.seg[0].m_lo = 0; c.seg[0].m_hi = Do;
    TDNNF_HIP(rows_gemm(c, false, s));  // T -= Wy^T P
  }
  return TDNNF_OK;
}

}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

const float *tdnnf_ng_scale_dev(const tdnnf_ng *ng) { return ng && ng->dev ? ng->scale_f : nullptr; }

size_t tdnnf_ng_stats_pass_workspace_bytes(int rank, int Di, int num_taps, int N) { return pform_ws_bytes((rank + 3) & ~3, Di, num_taps, N) + 64; }

int tdnnf_ng_stats_pass(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *X, int Di, const float *eff, const float *WT, const float *W, int ldw,
                        const float *bias, tdnnf_mat *H, double *sumsq, int sumsq_cap, int use_valu, void *workspace, size_t workspace_bytes,
                        tdnnf_stream stream) {
  TDNNF_REQUIRE(ix && X && X->data && H && H->data && Di > 0 && ix->num_offsets >= 1 && ix->num_offsets <= kMaxSeg && ix->row_stride >= 1,
                "ng_stats_pass: bad arguments");
  const int K = ix->num_offsets, N = H->rows, Rp = H->cols;
  for (int i = 0; i < K; i++)
    TDNNF_REQUIRE(ix->row_offsets[i] >= 0 && (long long)(N - 1) * ix->row_stride + ix->row_offsets[i] < X->rows && Di <= X->cols,
                  "ng_stats_pass: tap %d reads outside X (%d x %d)", i, X->rows, X->cols);
  TDNNF_REQUIRE(!sumsq || sumsq_cap >= rows_gemm_sumsq_blocks(N), "ng_stats_pass: sumsq needs %d entries", rows_gemm_sumsq_blocks(N));
  hipStream_t s = (hipStream_t)stream;
  if (use_valu == 2) {  // the P form: one pass over X for all taps
    NgInput in;
    memset(&in, 0, sizeof(in));
    in.x = view(X); in.ix = *ix; in.Di = Di; in.ones = bias ? 1 : 0; in.N = N; in.eff = eff;
    TDNNF_REQUIRE(W && ldw >= K * Di && sumsq && H->stride == Rp && workspace && pform_ok(Rp, in, workspace_bytes),
                  "ng_stats_pass: the one-pass form takes >= 2 taps of one matrix whole 128-row tiles apart, N %% 128 == 0, N >= 32768, taps x rank <= 64, no coefficients");
    return pform_pass(W, Rp, ldw, bias, in, H->data, sumsq, workspace, s);
  }
  if (use_valu) {
    NgRowdotArgs v;
    memset(&v, 0, sizeof(v));
    v.X = X->data; v.ldx = X->stride; v.row_stride = ix->row_stride; v.nseg = K; v.Di = Di; v.eff = eff; v.WT = WT; v.Rp = Rp;
    v.bias = bias; v.H = H->data; v.ldh = H->stride; v.N = N; v.part = sumsq; v.part_cap = sumsq_cap;
    for (int i = 0; i < K; i++) v.seg_off[i] = (long long)ix->row_offsets[i] * X->stride;
    TDNNF_REQUIRE(WT && ng_rowdot_ok(v), "ng_stats_pass: the vector-ALU kernel takes rank 20 / 40 / 80 and 16-byte aligned rows");
    TDNNF_HIP(ng_rowdot(v, s));
    return TDNNF_OK;
  }
  TDNNF_REQUIRE(W && ldw >= K * Di, "ng_stats_pass: the MFMA form needs W (rank x ldw)");
  GemmPrecisionScope exact_f32(2);
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = X->data; a.lda = (long long)X->stride * ix->row_stride; a.B = W; a.ldb = ldw; a.C = H->data; a.ldc = H->stride; a.M = N; a.N = Rp;
  a.bias = bias; a.init_mode = bias ? 1 : 2; a.coef = eff; a.sumsq = sumsq; a.nseg = K;
  for (int i = 0; i < K; i++) {
    a.seg[i].a_off = (long long)ix->row_offsets[i] * X->stride;
    a.seg[i].b_off = (long long)i * Di;
    a.seg[i].klen = Di;
    a.seg[i].m_lo = 0;
    a.seg[i].m_hi = N;
  }
  TDNNF_HIP(rows_gemm(a, true, s));
  return TDNNF_OK;
}

int tdnnf_ng_create(int rank, int update_period, float num_samples_history, float alpha, tdnnf_ng **out) {
  TDNNF_REQUIRE(out && rank >= 0 && update_period >= 1 && num_samples_history > 0 && alpha >= 0, "ng_create: bad configuration");
  tdnnf_ng *ng = new tdnnf_ng();
  ng->rank = rank;
  ng->Rp = 0;
  ng->update_period = update_period;
  ng->t = 0;
  ng->D = ng->Dp = 0;
  ng->frozen = 0;
  ng->num_samples_history = num_samples_history;
  ng->alpha = alpha;
  ng->epsilon = 1.0e-10f;
  ng->delta = 5.0e-04f;
  ng->rho = 0;
  ng->dev = nullptr;
  ng->pin = nullptr;
  ng->scratch = nullptr;
  ng->scratch_floats = 0;
  ng->pending = 0;
  ng->job_done = 0;
  ng->ev_job = nullptr;
  ng->ev_wait = nullptr;
  ng->must_reorth = false;
  *out = ng;
  return TDNNF_OK;
}

// OnlineNaturalGradient::Freeze (UPSTREAM; called by FreezeNaturalGradient, /root/reference/src/nnet3/nnet-tdnn-component.cc:979-982):
// a frozen object keeps preconditioning with its current state and never refreshes it
int tdnnf_ng_freeze(tdnnf_ng *ng, int freeze) {
  TDNNF_REQUIRE(ng, "ng_freeze: null object");
  ng->frozen = freeze ? 1 : 0;
  return TDNNF_OK;
}

void tdnnf_ng_destroy(tdnnf_ng *ng) {
  if (!ng) return;
  if (ng->pending) pool().wait(ng);  // the worker still reads this object's pinned buffers
  hipFree(ng->dev);
  hipHostFree(ng->pin);
  hipFree(ng->scratch);
  if (ng->ev_job) hipEventDestroy(ng->ev_job);
  delete ng;
}

// Component-level entry point: X (N x D, device) is replaced by X^; *scale_host (optional) receives the scale,
// which costs a stream synchronisation.
int tdnnf_ng_precondition(tdnnf_ng *ng, tdnnf_mat *X, float *scale_host, tdnnf_stream stream) {
  TDNNF_REQUIRE(ng && mat_ok(X) && X->rows > 0 && X->cols > 0, "ng_precondition: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfClassOverride prof_as_ng(3);
  GemmPrecisionScope exact_f32(2);
  if (X->cols == 1) {  // preconditioning one column is pointless (UPSTREAM)
    if (scale_host) *scale_host = 1.0f;
    return TDNNF_OK;
  }
  const int N = X->rows, D = X->cols;
  const int Rp = pad4(std::max(1, std::min(ng->rank, D - 1)));
  const size_t ws_bytes = stats_ws_bytes(Rp, D, 1, N) + 64;
  const size_t need = (size_t)N * Rp + ws_bytes / sizeof(float) + 64;
  if (ng->scratch_floats < need) {
    if (ng->scratch) hipFree(ng->scratch);
    ng->scratch = nullptr;
    ng->scratch_floats = 0;
    TDNNF_HIP(hipMalloc((void **)&ng->scratch, sizeof(float) * need));
    ng->scratch_floats = need;
  }
  float *H = ng->scratch;
  void *ws = (void *)(((uintptr_t)(H + (size_t)N * Rp) + 63) & ~(uintptr_t)63);
  NgInput in;
  memset(&in, 0, sizeof(in));
  in.x = view(X);
  in.ix.row_stride = 1;
  in.ix.num_offsets = 1;
  in.Di = D;
  in.N = N;
  int rc = ng_stats_step(ng, in, H, ws, ws_bytes - 64, s);
  if (rc) return rc;
  if (ng->rank == 0) {
    if (scale_host) *scale_host = 1.0f;
    return TDNNF_OK;
  }
  RowsGemmArgs b;
  memset(&b, 0, sizeof(b));
  b.A = H; b.lda = ng->Rp; b.B = ng->W; b.ldb = ng->Dp; b.C = X->data; b.ldc = X->stride; b.M = N; b.N = D; b.init_mode = 0; b.nseg = 1;
  b.coef = ng->neg_one;
  b.seg[0].klen = ng->Rp; b.seg[0].m_lo = 0; b.seg[0].m_hi = N;
  TDNNF_HIP(rows_gemm(b, false, s));  // X^ = X - H W
  if (scale_host) {
    TDNNF_HIP(hipMemcpyAsync(ng->h_scale, ng->scale_f, sizeof(float), hipMemcpyDeviceToHost, s));
    TDNNF_HIP(hipStreamSynchronize(s));
    *scale_host = *ng->h_scale;
  }
  return TDNNF_OK;
}

}  // extern "C"
