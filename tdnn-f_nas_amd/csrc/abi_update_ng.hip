// abi_update_ng.hip -- component-level natural-gradient updates behind the C-ABI (include/tdnnf_hip.h):
// the branch Backprop takes in every recipe of the reference ("use_natural_gradient_" defaults to true).
//
//   tdnnf_tdnn_update_natural_gradient    TdnnDARTSV3Component::UpdateNaturalGradient
//                                         /root/reference/src/nnet3/nnet-tdnn-component.cc:457-626 (and the plain
//                                         TdnnComponent's, UPSTREAM: the same without coefficients and logits)
//   tdnnf_affine_update_natural_gradient  NaturalGradientAffineComponent::Update nnet-simple-component.cc:2980-3024,
//                                         LinearComponent's natural-gradient branch :3240-3243 (bias_acc == NULL)
//
// These follow the reference's LITERAL order -- splice [c_i X_i ..., 1] into a temporary, copy out_deriv, precondition
// both copies in place, multiply by the product of the two scales -- because a caller that owns its own preconditioner
// objects and accumulators (a Kaldi component) expects exactly that state evolution.  The chain trainer (net.hip) uses
// the algebraically equal projection form of ng.h, which never materialises the N x D temporaries.
#include <string.h>

#include <algorithm>

#include "common.h"
#include "gemm_f32.h"
#include "ng.h"

namespace tdnnf {
namespace {

inline int pad4i(int x) { return (x + 3) & ~3; }

// in_value_temp of :482-532: X~[r][i*Di + d] = eff[i] * in[row_off[i] + r*rho][d] (a tap whose effective coefficient is
// zero is left zero, :512), X~[r][K*Di] = 1 when the component has a bias (:474-477)
__global__ __launch_bounds__(256) void splice_taps_kernel(MatView in, tdnnf_tdnn_indexes ix, const float *eff, int Di, int ones, MatView out) {
  const int KDi = ix.num_offsets * Di, C = KDi + ones;
  const long long total = (long long)out.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    float v = 1.0f;
    if (c < KDi) {
      const int i = c / Di, d = c % Di;
      const float cf = eff ? eff[i] : 1.0f;
      v = cf == 0.0f ? 0.0f : cf * in.data[((size_t)ix.row_offsets[i] + (size_t)r * ix.row_stride) * in.stride + d];
    }
    out.data[(size_t)r * out.stride + c] = v;
  }
}
__global__ __launch_bounds__(256) void copy_mat_kernel(MatView in, MatView out) {
  const long long total = (long long)in.rows * in.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / in.cols), c = (int)(e % in.cols);
    out.data[(size_t)r * out.stride + c] = in.data[(size_t)r * in.stride + c];
  }
}
// "local_lrate = scale * learning_rate_" (:604-605) with both scales still on the device:
// W_acc[o][c] += lr a b T[o][c] (c < KDi), bias_acc[o] += lr a b T[o][KDi]   (:606-624)
__global__ __launch_bounds__(256) void commit_scaled_kernel(const float *T, int ldT, int Do, int KDi, const float *sa, const float *sb, float lr,
                                                            float *W_acc, int ldw, float *bias_acc) {
  const float sc = lr * (sa ? sa[0] : 1.0f) * (sb ? sb[0] : 1.0f);
  const int C = KDi + (bias_acc ? 1 : 0);
  const long long total = (long long)Do * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int o = (int)(e / C), c = (int)(e % C);
    const float v = sc * T[(size_t)o * ldT + c];
    if (c < KDi) W_acc[(size_t)o * ldw + c] += v;
    else bias_acc[o] += v;
  }
}

struct Layout {
  size_t x, dy, t, tap, dots, upd, total;  // byte offsets
  int ldx, ldy, ldt;
  size_t upd_bytes;
};
Layout layout(int Do, int Di, int K, int N, int ones, bool alpha) {
  Layout L;
  const int Dx = K * Di + ones;
  L.ldx = pad4i(Dx);
  L.ldy = pad4i(Do);
  L.ldt = pad4i(Dx);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  L.x = take(sizeof(float) * (size_t)N * L.ldx);
  L.dy = take(sizeof(float) * (size_t)N * L.ldy);
  L.t = take(sizeof(float) * (size_t)Do * L.ldt);
  L.tap = take(alpha ? sizeof(float) * (size_t)Do * K * Di : 0);
  L.dots = take(sizeof(double) * TDNNF_TAP_DOTS_DOUBLES(TDNNF_MAX_OFFSETS));
  L.upd_bytes = std::max(wgrad_workspace_bytes(Do, Dx, 1, N), wgrad_workspace_bytes(Do, Di, K, N));
  L.upd = take(L.upd_bytes);
  L.total = off + 256;
  return L;
}

int update_ng_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do, int Di,
                   const float *linear_params, int ldw_params, const float *coef_memo, const float *eff_coef, int flags, int share_index,
                   float temp_proportion, tdnnf_ng *ng_in, tdnnf_ng *ng_out, float lr, float *W_acc, int ldw, float *bias_acc,
                   float *alpha_acc, void *ws, size_t ws_bytes, hipStream_t s, const char *who) {
  TDNNF_REQUIRE(ix && mat_ok(in_value) && mat_ok(out_deriv) && W_acc && ng_in && ng_out, "%s: null argument", who);
  const int K = ix->num_offsets, N = out_deriv->rows;
  TDNNF_REQUIRE(K >= 1 && K <= TDNNF_MAX_OFFSETS && ix->row_stride >= 1 && Do > 0 && Di > 0 && in_value->cols == Di && out_deriv->cols == Do && ldw >= K * Di,
                "%s: dims: in.cols=%d Di=%d out_deriv.cols=%d Do=%d ldw=%d K=%d", who, in_value->cols, Di, out_deriv->cols, Do, ldw, K);
  for (int i = 0; i < K; i++)
    TDNNF_REQUIRE(ix->row_offsets[i] >= 0 && (N == 0 || (long long)ix->row_offsets[i] + (long long)ix->row_stride * (N - 1) < in_value->rows),
                  "%s: in_value has too few rows for the time offsets", who);
  const bool darts = coef_memo != nullptr;
  TDNNF_REQUIRE(!darts || (eff_coef && linear_params && ldw_params >= K * Di && alpha_acc && share_index >= 0 && share_index < K),
                "%s: a TdnnDARTSV3Component needs coef_memo, eff_coef, its linear_params and the logit accumulator", who);
  TDNNF_REQUIRE(!darts || !(flags & TDNNF_DARTS_USE_GUMBEL) || temp_proportion > 0.f, "%s: gumbel mode needs temp-proportion > 0", who);
  if (N == 0 || lr == 0.0f) return TDNNF_OK;  // "if (to_update->learning_rate_ == 0.0) return" (:423-424)
  const int ones = bias_acc ? 1 : 0, KDi = K * Di, Dx = KDi + ones;
  TDNNF_REQUIRE(Dx >= 2 && Do >= 2, "%s: one-column operands are not preconditioned; use the simple update", who);
  const bool want_alpha = darts && !(flags & TDNNF_DARTS_UNIFORM_SAMPLE);
  const Layout L = layout(Do, Di, K, N, ones, darts);
  TDNNF_REQUIRE(ws && ws_bytes >= L.total, "%s: workspace too small (%zu < %zu bytes)", who, ws_bytes, L.total);
  char *base = (char *)ws;
  float *X = (float *)(base + L.x), *dY = (float *)(base + L.dy), *T = (float *)(base + L.t), *tap = (float *)(base + L.tap);
  double *dots = (double *)(base + L.dots);
  void *upd = base + L.upd;
  // ---- architecture logits (:490-590).  s_i = sum((X_i W_i^T) .* dY) = <dY^T X_i, W_i>: the unscaled tap gradients replace the
  // reference's extra forward GEMM per tap.  Uniform-sample mode adds no gradient (its GEMM result is discarded, :502-507) but the
  // accumulator is still scaled (quirk q5: the scalings multiply whatever it holds).
  if (darts) {
    if (want_alpha) {
      TDNNF_HIP(hipMemsetAsync(tap, 0, sizeof(float) * (size_t)Do * KDi, s));
      int rc = tdnn_update_simple_impl(ix, in_value, out_deriv, Do, Di, nullptr, 1.0f, tap, KDi, nullptr, upd, L.upd_bytes, nullptr, 0, s);
      if (rc) return rc;
    }
    int rc = tdnnf_tdnn_darts_alpha_update(want_alpha ? tap : nullptr, KDi, linear_params, ldw_params, Do, Di, K, coef_memo, flags, share_index,
                                           temp_proportion, lr, alpha_acc, dots, s);
    if (rc) return rc;
  }
  // ---- in_value_temp = [c_i X_i ..., 1], out_deriv_temp = out_deriv (:466-532, :592)
  tdnnf_mat Xm{X, N, Dx, L.ldx}, Ym{dY, N, Do, L.ldy};
  hipLaunchKernelGGL(splice_taps_kernel, dim3(grid_for((long long)N * Dx, 256)), dim3(256), 0, s, view(in_value), *ix, eff_coef, Di, ones, view(&Xm));
  hipLaunchKernelGGL(copy_mat_kernel, dim3(grid_for((long long)N * Do, 256)), dim3(256), 0, s, view(out_deriv), view(&Ym));
  TDNNF_LAUNCH_CHECK();
  // ---- PreconditionDirections on both copies, in the reference's order (:598-599); the scales stay on the device
  int rc = tdnnf_ng_precondition(ng_in, &Xm, nullptr, s);
  if (rc) return rc;
  rc = tdnnf_ng_precondition(ng_out, &Ym, nullptr, s);
  if (rc) return rc;
  // ---- T = dY'^T X~' ; linear_params_ += local_lrate T[:, :K Di], bias_params_[K:] += local_lrate T[:, K Di] (:606-624)
  TDNNF_HIP(hipMemsetAsync(T, 0, sizeof(float) * (size_t)Do * L.ldt, s));
  {
    GemmPrecisionScope exact_f32(2);
    rc = tdnnf_affine_update_simple(&Xm, &Ym, 1.0f, T, L.ldt, nullptr, upd, L.upd_bytes, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(commit_scaled_kernel, dim3(grid_for((long long)Do * Dx, 256)), dim3(256), 0, s, T, L.ldt, Do, KDi, tdnnf_ng_scale_dev(ng_in),
                     tdnnf_ng_scale_dev(ng_out), lr, W_acc, ldw, bias_acc);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

size_t tdnnf_tdnn_update_natural_gradient_workspace_bytes(int Do, int Di, int K, int num_rows, int has_bias) {
  if (Do <= 0 || Di <= 0 || K < 1 || K > TDNNF_MAX_OFFSETS || num_rows < 0) return 0;
  return layout(Do, Di, K, num_rows, has_bias ? 1 : 0, true).total;
}

int tdnnf_tdnn_update_natural_gradient(const tdnnf_tdnn_indexes *indexes, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do, int Di,
                                       const float *linear_params_dev, int ldw_params, const float *coef_memo_dev, const float *eff_coef_dev,
                                       int flags, int share_index, float temp_proportion, tdnnf_ng *preconditioner_in, tdnnf_ng *preconditioner_out,
                                       float learning_rate, float *W_acc_dev, int ldw, float *bias_acc_dev, float *alpha_acc_dev,
                                       void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream) {
  return update_ng_impl(indexes, in_value, out_deriv, Do, Di, linear_params_dev, ldw_params, coef_memo_dev, eff_coef_dev, flags, share_index,
                        temp_proportion, preconditioner_in, preconditioner_out, learning_rate, W_acc_dev, ldw, bias_acc_dev, alpha_acc_dev,
                        workspace_dev, workspace_bytes, (hipStream_t)stream, "tdnn_update_natural_gradient");
}

size_t tdnnf_affine_update_natural_gradient_workspace_bytes(int Do, int Di, int num_rows, int has_bias) {
  return tdnnf_tdnn_update_natural_gradient_workspace_bytes(Do, Di, 1, num_rows, has_bias);
}

int tdnnf_affine_update_natural_gradient(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, tdnnf_ng *preconditioner_in,
                                         tdnnf_ng *preconditioner_out, float learning_rate, float *W_acc_dev, int ldw, float *bias_acc_dev,
                                         void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in_value) && mat_ok(out_deriv) && in_value->rows == out_deriv->rows, "affine_update_natural_gradient: bad matrices");
  tdnnf_tdnn_indexes ix;
  memset(&ix, 0, sizeof(ix));
  ix.row_stride = 1;
  ix.num_offsets = 1;
  return update_ng_impl(&ix, in_value, out_deriv, out_deriv->cols, in_value->cols, nullptr, 0, nullptr, nullptr, 0, 0, 1.0f, preconditioner_in,
                        preconditioner_out, learning_rate, W_acc_dev, ldw, bias_acc_dev, nullptr, workspace_dev, workspace_bytes, (hipStream_t)stream,
                        "affine_update_natural_gradient");
}

}  // extern "C"
