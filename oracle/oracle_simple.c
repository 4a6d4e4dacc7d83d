/*
 * oracle_simple.c -- CPU restatement of the DARTS mixing ops and the thin stock
 * layers of nnet-simple-component.cc.  Test infrastructure only (see oracle.h).
 * PARITY UNPINNED.
 *
 * Follows /root/reference/src/nnet3/nnet-simple-component.cc:
 *   GumbelSoftmaxFlopsComponent :10088-10158, SoftmaxFlopsComponent :9968-10020,
 *   GumbelSoftmaxComponent :9774-9831, OnehotFunctionComponent :9504-9552,
 *   CopyNComponent :4843-4867, FlopsConstraintComponent :9454-9478,
 *   ConstantFunctionComponent :2602-2642, ElementwiseProductComponent :256-299,
 *   RectifiedLinearComponent :958-1091, AffineComponent :1235-1279,
 *   LogSoftmaxComponent :3607-3632.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

#ifdef ORACLE_F64ACC
typedef double acc_t;
#else
typedef float acc_t;
#endif

#define AT(m, r, c) ((m)->data[(long)(m)->stride * (r) + (c)])

/* G = -log(-log(U)) : :10095-10100 */
void oracle_gumbel_noise(const float *u, int n, float *g) {
  for (int i = 0; i < n; i++) g[i] = -logf(-logf(u[i]));
}

/* Softmax / Gumbel-softmax forward.  gumbel_u == NULL and temp == 1 gives
   SoftmaxFlopsComponent::Propagate (:9974-9978); otherwise the Gumbel version
   (:10095-10110) where ONE noise vector is shared by all rows. */
void oracle_softmax_flops_propagate(const omat *in, const float *gumbel_u,
                                    float temp_proportion, omat *out) {
  int C = in->cols;
  float *g = (float *)calloc(C, sizeof(float));
  if (gumbel_u) oracle_gumbel_noise(gumbel_u, C, g);
  for (int r = 0; r < in->rows; r++) {
    float mx = -INFINITY;
    for (int c = 0; c < C; c++) {
      float v = gumbel_u ? (AT(in, r, c) + g[c]) * (1.0f / temp_proportion) : AT(in, r, c);
      AT(out, r, c) = v;
      if (v > mx) mx = v;
    }
    double sum = 0;
    for (int c = 0; c < C; c++) sum += exp((double)AT(out, r, c) - mx);
    for (int c = 0; c < C; c++) {
      float p = (float)(exp((double)AT(out, r, c) - mx) / sum);
      AT(out, r, c) = p < 1.0e-20f ? 1.0e-20f : p; /* ApplyFloor(1e-20) */
    }
  }
  free(g);
}

/* :10006-10017 / :10144-10157.  NOTE the reference mutates out_deriv in place
   (it is declared const there); we reproduce that.  `flops` is the hard-coded
   -(25,50,80,100,120,160,200,240) in the reference, a parameter here
   (flops == NULL or scale == 0 skips the penalty: GumbelSoftmaxComponent :9829-9830). */
void oracle_softmax_flops_backprop(const omat *out_value, omat *out_deriv,
                                   float scale, const float *flops, int dim,
                                   float temp_proportion, omat *in_deriv) {
  int R = out_deriv->rows, C = out_deriv->cols;
  if (flops) {
    float a = scale / R / C;
    for (int r = 0; r < R; r++)
      for (int c = 0; c < dim; c++) AT(out_deriv, r, c) += a * flops[c];
  }
  for (int r = 0; r < R; r++) { /* DiffSoftmaxPerRow */
    double pe = 0;
    for (int c = 0; c < C; c++) pe += (double)AT(out_value, r, c) * AT(out_deriv, r, c);
    for (int c = 0; c < C; c++)
      AT(in_deriv, r, c) =
          (float)(AT(out_value, r, c) * ((double)AT(out_deriv, r, c) - pe)) *
          (1.0f / temp_proportion);
  }
}

/* :9512-9516 -- index i with i/C <= u < (i+1)/C in float arithmetic; -1 if none
   (u == 1.0 cannot happen with SetRandUniform but the loop would select nothing). */
int oracle_onehot_index(float u, int C) {
  for (int i = 0; i < C; i++)
    if (u >= (float)i / C && u < (float)(i + 1) / C) return i;
  return -1;
}

void oracle_onehot_propagate(float u, omat *out) {
  int idx = oracle_onehot_index(u, out->cols);
  for (int r = 0; r < out->rows; r++)
    for (int c = 0; c < out->cols; c++) AT(out, r, c) = (c == idx) ? 1.0f : 0.0f; /* :9517 */
}

/* AddMatBlocks(scale, in): out (N x k*d) += scale * [in in ... in]  (:4850) */
void oracle_copyn_propagate(const omat *in, float scale, omat *out) {
  int d = in->cols;
  for (int r = 0; r < out->rows; r++)
    for (int c = 0; c < out->cols; c++) AT(out, r, c) += scale * AT(in, r, c % d);
}

/* in_deriv (N x d) += scale * sum over blocks of out_deriv (:4865) */
void oracle_copyn_backprop(const omat *out_deriv, float scale, omat *in_deriv) {
  int d = in_deriv->cols;
  for (int r = 0; r < out_deriv->rows; r++)
    for (int c = 0; c < d; c++) {
      acc_t s = 0;
      for (int b = 0; b < out_deriv->cols / d; b++) s += AT(out_deriv, r, b * d + c);
      AT(in_deriv, r, c) += scale * (float)s;
    }
}

void oracle_constant_function_propagate(const float *output, omat *out) {
  for (int r = 0; r < out->rows; r++)
    for (int c = 0; c < out->cols; c++) AT(out, r, c) = output[c]; /* :2606 */
}

/* non-NG branch :2636: output_ += 5*lr*colsum(out_deriv)  (the x5 is a NAS edit) */
void oracle_constant_function_backprop(const omat *out_deriv, float lr,
                                       float *output_acc) {
  for (int c = 0; c < out_deriv->cols; c++) {
    acc_t s = 0;
    for (int r = 0; r < out_deriv->rows; r++) s += AT(out_deriv, r, c);
    output_acc[c] += 5.0f * lr * (float)s;
  }
}

/* :9474-9477 */
void oracle_flops_constraint_backprop(const float *flops, float scale, int rows_in,
                                      int cols_in, omat *in_deriv) {
  for (int r = 0; r < in_deriv->rows; r++)
    for (int c = 0; c < in_deriv->cols; c++)
      AT(in_deriv, r, c) = flops[c] * (scale / rows_in / cols_in);
}

/* :256-274 */
void oracle_elementwise_product_propagate(const omat *in, int output_dim,
                                          omat *out) {
  int n = in->cols / output_dim;
  for (int r = 0; r < in->rows; r++)
    for (int c = 0; c < output_dim; c++) {
      float p = AT(in, r, c);
      for (int i = 1; i < n; i++) p *= AT(in, r, i * output_dim + c);
      AT(out, r, c) = p;
    }
}

/* :276-299 */
void oracle_elementwise_product_backprop(const omat *in_value,
                                         const omat *out_deriv, int output_dim,
                                         omat *in_deriv) {
  int n = in_value->cols / output_dim;
  for (int r = 0; r < in_value->rows; r++)
    for (int i = 0; i < n; i++)
      for (int c = 0; c < output_dim; c++) {
        float p = AT(out_deriv, r, c);
        for (int j = 0; j < n; j++)
          if (j != i) p *= AT(in_value, r, j * output_dim + c);
        AT(in_deriv, r, i * output_dim + c) = p;
      }
}

void oracle_relu_propagate(const omat *in, omat *out) { /* :963-964 */
  for (int r = 0; r < in->rows; r++)
    for (int c = 0; c < in->cols; c++) {
      float x = AT(in, r, c);
      AT(out, r, c) = x < 0.0f ? 0.0f : x;
    }
}

void oracle_relu_backprop(const omat *out_value, const omat *out_deriv,
                          omat *in_deriv) { /* :978-979 */
  for (int r = 0; r < out_value->rows; r++)
    for (int c = 0; c < out_value->cols; c++)
      AT(in_deriv, r, c) = (AT(out_value, r, c) > 0.0f ? 1.0f : 0.0f) * AT(out_deriv, r, c);
}

/* RepairGradients :990-1074 once the coin flip (:1017) came up "repair".
   lower/upper are the thresholds as proportions (defaults .05/.95). */
void oracle_relu_repair(const double *deriv_sum, double count, int dim,
                        float self_repair_scale, float lower, float upper,
                        omat *in_deriv) {
  const float repair_probability = 0.5f;
  if (self_repair_scale == 0.0f || count == 0.0) return;
  float lo = lower * (float)count, hi = upper * (float)count;
  for (int c = 0; c < dim; c++) {
    float st = (float)deriv_sum[c];
    float v = (st - lo > 0.0f ? 1.0f : 0.0f) + (st - hi > 0.0f ? 1.0f : 0.0f) - 1.0f;
    v *= -self_repair_scale / repair_probability;
    if (v != 0.0f)
      for (int r = 0; r < in_deriv->rows; r++) AT(in_deriv, r, c) += v;
  }
}

/* StoreStatsInternal (nnet-component-itf.cc:433-459): value_sum += colsum(out),
   deriv_sum += colsum(out>0), count += rows. */
void oracle_relu_store_stats(const omat *out_value, double *value_sum,
                             double *deriv_sum, double *count) {
  for (int c = 0; c < out_value->cols; c++) {
    double vs = 0, ds = 0;
    for (int r = 0; r < out_value->rows; r++) {
      float v = AT(out_value, r, c);
      vs += v;
      ds += v > 0.0f ? 1.0 : 0.0;
    }
    value_sum[c] += vs;
    deriv_sum[c] += ds;
  }
  *count += out_value->rows;
}

/* AffineComponent::Propagate :1235-1244 (bias NULL = LinearComponent :3211-3216) */
void oracle_affine_propagate(const omat *in, const float *W, int ldw,
                             const float *bias, int Do, omat *out) {
  int Di = in->cols;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < in->rows; r++)
    for (int o = 0; o < Do; o++) {
      acc_t a = 0;
#pragma omp simd reduction(+ : a)
      for (int d = 0; d < Di; d++) a += (acc_t)AT(in, r, d) * (acc_t)W[(long)ldw * o + d];
      AT(out, r, o) = (float)((acc_t)(bias ? bias[o] : 0.0f) + a);
    }
}

/* :1262-1264: in_deriv = out_deriv * W (overwrites) */
void oracle_affine_backprop(const omat *out_deriv, const float *W, int ldw,
                            int Di, omat *in_deriv) {
  int Do = out_deriv->cols;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < out_deriv->rows; r++) {
    acc_t a[Di]; /* a[d] = sum_o dY[r][o] W[o][d], o ascending (W walked row by row) */
    for (int d = 0; d < Di; d++) a[d] = 0;
    for (int o = 0; o < Do; o++) {
      const acc_t g = (acc_t)AT(out_deriv, r, o);
      const float *w = W + (long)ldw * o;
#pragma omp simd
      for (int d = 0; d < Di; d++) a[d] += g * (acc_t)w[d];
    }
    for (int d = 0; d < Di; d++) AT(in_deriv, r, d) = (float)a[d];
  }
}

/* UpdateSimple :1246-1251 */
void oracle_affine_update_simple(const omat *in_value, const omat *out_deriv,
                                 float lr, float *W_acc, int ldw,
                                 float *bias_acc) {
  int N = out_deriv->rows, Do = out_deriv->cols, Di = in_value->cols;
#pragma omp parallel for schedule(static)
  for (int o = 0; o < Do; o++) {
    if (bias_acc) {
      acc_t s = 0;
      for (int r = 0; r < N; r++) s += AT(out_deriv, r, o);
      bias_acc[o] = (float)((acc_t)bias_acc[o] + (acc_t)lr * s);
    }
    acc_t *s = (acc_t *)calloc(Di, sizeof(acc_t)); /* s[d] = sum_r dY[r][o] X[r][d], r ascending */
    for (int r = 0; r < N; r++) {
      const acc_t g = (acc_t)AT(out_deriv, r, o);
      const float *x = in_value->data + (long)in_value->stride * r;
#pragma omp simd
      for (int d = 0; d < Di; d++) s[d] += g * (acc_t)x[d];
    }
    for (int d = 0; d < Di; d++)
      W_acc[(long)ldw * o + d] = (float)((acc_t)W_acc[(long)ldw * o + d] + (acc_t)lr * s[d]);
    free(s);
  }
}

void oracle_log_softmax_propagate(const omat *in, omat *out) { /* :3607-3614 */
  for (int r = 0; r < in->rows; r++) {
    float mx = -INFINITY;
    for (int c = 0; c < in->cols; c++)
      if (AT(in, r, c) > mx) mx = AT(in, r, c);
    double s = 0;
    for (int c = 0; c < in->cols; c++) s += exp((double)AT(in, r, c) - mx);
    float lse = mx + (float)log(s);
    for (int c = 0; c < in->cols; c++) AT(out, r, c) = AT(in, r, c) - lse;
  }
}

/* DiffLogSoftmaxPerRow :3616-3632: d_i = e_i - exp(y_i) * sum_j e_j */
void oracle_log_softmax_backprop(const omat *out_value, const omat *out_deriv,
                                 omat *in_deriv) {
  for (int r = 0; r < out_value->rows; r++) {
    double s = 0;
    for (int c = 0; c < out_value->cols; c++) s += AT(out_deriv, r, c);
    for (int c = 0; c < out_value->cols; c++)
      AT(in_deriv, r, c) = (float)(AT(out_deriv, r, c) - exp((double)AT(out_value, r, c)) * s);
  }
}

/* Descriptor Sum(Scale(sa, a), Scale(sb, b)) feeding NoOpComponent
   (composite_layers.py:205-213; NoOp Propagate :437-446 is a copy). */
void oracle_sum_scaled(const omat *a, float sa, const omat *b, float sb,
                       omat *out) {
  for (int r = 0; r < out->rows; r++)
    for (int c = 0; c < out->cols; c++) AT(out, r, c) = sa * AT(a, r, c) + sb * AT(b, r, c);
}

/* GeneralDropoutComponent (UPSTREAM), continuous, mask shared over time:
   row r of a t-major matrix belongs to sequence r % num_seq. */
void oracle_general_dropout_propagate(const omat *in, const float *mask,
                                      int num_seq, omat *out) {
  for (int r = 0; r < in->rows; r++)
    for (int c = 0; c < in->cols; c++)
      AT(out, r, c) = AT(in, r, c) * mask[(long)(r % num_seq) * in->cols + c];
}
