/*
 * oracle_tdnn.c -- CPU restatement of TdnnDARTSV3Component / TdnnComponent.
 * Test infrastructure only (see oracle.h).  PARITY UNPINNED.
 *
 * Follows /root/reference/src/nnet3/nnet-tdnn-component.cc:
 *   Propagate :214-333, Backprop :335-431, UpdateSimple :433-455,
 *   UpdateNaturalGradient :457-626, GetInputPart :806-820.
 * The plain TdnnComponent (UPSTREAM, not shipped) is the same with all
 * coefficients == 1 and the bias added unconditionally.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64ACC
typedef double acc_t;
#else
typedef float acc_t;
#endif

/* row pointer of tap i's strided view: GetInputPart, nnet-tdnn-component.cc:806-820 */
static inline const float *in_row(const omat *in, int row_stride, int row_offset,
                                  int r) {
  return in->data + (long)in->stride * ((long)row_offset + (long)r * row_stride);
}

/* nnet-tdnn-component.cc:232-240 and :359-364: the tap that is always kept at
   weight one is index 0 when offsets[1] > 0, else the last one. */
int oracle_tdnn_share_index(const int *time_offsets, int K) {
  if (K < 2) return 0; /* reference assumes K>=2 (quirk q3); a 1-tap layer shares tap 0 */
  return time_offsets[1] > 0 ? 0 : K - 1;
}

static void softmax_floor(float *v, int n, float floor_val) {
  /* CuVector::ApplySoftMax then ApplyFloor(1e-20): :267-268, :276-277 */
  float mx = v[0];
  for (int i = 1; i < n; i++)
    if (v[i] > mx) mx = v[i];
  double sum = 0.0;
  for (int i = 0; i < n; i++) sum += exp((double)v[i] - mx);
  for (int i = 0; i < n; i++) {
    float p = (float)(exp((double)v[i] - mx) / sum);
    v[i] = p < floor_val ? floor_val : p;
  }
}

/* nnet-tdnn-component.cc:250-289.  gumbel_u are the K uniform draws the
   reference takes from SetRandUniform(); sample_u the single uniform of :282. */
void oracle_tdnn_darts_coef(const float *log_alpha, int K, int flags,
                            float temp_proportion, const float *gumbel_u,
                            float sample_u, float *coef) {
  for (int i = 0; i < K; i++) coef[i] = log_alpha[i];
  if (flags & ORACLE_DARTS_USE_GUMBEL) {
    for (int i = 0; i < K; i++) {
      float g = -logf(-logf(gumbel_u[i])); /* :259-263 */
      coef[i] = (coef[i] + g) * (1.0f / temp_proportion);
    }
    softmax_floor(coef, K, 1.0e-20f);
  } else if (flags & ORACLE_DARTS_FREE_SELECT) {
    for (int i = 0; i < K; i++) coef[i] = 1.0f / (1.0f + expf(-coef[i])); /* :270-273 */
  } else {
    softmax_floor(coef, K, 1.0e-20f);
  }
  if (flags & ORACLE_DARTS_UNIFORM_SAMPLE) { /* :280-289 */
    for (int i = 0; i < K; i++)
      coef[i] = (sample_u >= (float)i / K && sample_u < (float)(i + 1) / K) ? 1.0f
                                                                           : 0.0f;
  }
}

/* The weight each tap's GEMM is actually given in Propagate/Backprop
   (:292-328, :366-416): uniform mode computes only the share tap and the
   sampled tap, both at 1.0; otherwise free_select uses c_i everywhere and the
   softmax/gumbel modes force the share tap to 1.0. */
void oracle_tdnn_darts_effective_coef(const float *coef, int K, int flags,
                                      int share_index, float *eff) {
  for (int i = 0; i < K; i++) {
    if (flags & ORACLE_DARTS_UNIFORM_SAMPLE)
      eff[i] = (i == share_index || coef[i] == 1.0f) ? 1.0f : 0.0f;
    else if (flags & ORACLE_DARTS_FREE_SELECT)
      eff[i] = coef[i];
    else
      eff[i] = (i == share_index) ? 1.0f : coef[i];
  }
}

/* init_mode: 0 = add into out (kPropagateAdds, no bias); 1 = out <- bias rows
   (:233-235); 2 = out <- 0 (:238, quirk q1).  eff_coef NULL means all ones. */
void oracle_tdnn_propagate(const omat *in, const float *W, int ldw, int Do,
                           int Di, int K, int row_stride,
                           const int *row_offsets, const float *bias,
                           const float *eff_coef, int init_mode, omat *out) {
  int N = out->rows;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < N; r++) {
    float *y = out->data + (long)out->stride * r;
    for (int o = 0; o < Do; o++) {
      acc_t a = 0;
      for (int i = 0; i < K; i++) {
        float c = eff_coef ? eff_coef[i] : 1.0f;
        if (c == 0.0f) continue;
        const float *x = in_row(in, row_stride, row_offsets[i], r);
        const float *w = W + (long)ldw * o + (long)i * Di;
        acc_t t = 0;
#pragma omp simd reduction(+ : t)
        for (int d = 0; d < Di; d++) t += (acc_t)x[d] * (acc_t)w[d];
        a += (acc_t)c * t; /* AddMatMat(c_i, in_part, kNoTrans, W_i, kTrans, 1.0) :302-324 */
      }
      float base = init_mode == 0 ? y[o] : (init_mode == 1 ? bias[o] : 0.0f);
      y[o] = (float)((acc_t)base + a);
    }
  }
}

/* :366-416.  in_deriv is ADDED to (kBackpropAdds).  Taps overlap, so this is
   done serially over taps and in parallel over rows within a tap. */
void oracle_tdnn_backprop_data(const omat *out_deriv, const float *W, int ldw,
                               int Do, int Di, int K, int row_stride,
                               const int *row_offsets, const float *eff_coef,
                               omat *in_deriv) {
  int N = out_deriv->rows;
  for (int i = 0; i < K; i++) {
    float c = eff_coef ? eff_coef[i] : 1.0f;
    if (c == 0.0f) continue;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < N; r++) {
      const float *dy = out_deriv->data + (long)out_deriv->stride * r;
      float *dx = in_deriv->data +
                  (long)in_deriv->stride * ((long)row_offsets[i] + (long)r * row_stride);
      /* t[d] = sum_o dy[o] W_i[o][d], o ascending for every d (W walked row by row) */
      acc_t t[Di];
      for (int d = 0; d < Di; d++) t[d] = 0;
      for (int o = 0; o < Do; o++) {
        const acc_t g = (acc_t)dy[o];
        const float *w = W + (long)ldw * o + (long)i * Di;
#pragma omp simd
        for (int d = 0; d < Di; d++) t[d] += g * (acc_t)w[d];
      }
      for (int d = 0; d < Di; d++) dx[d] = (float)((acc_t)dx[d] + (acc_t)c * t[d]);
    }
  }
}

/* UpdateSimple :433-455: bias += lr * colsum(dY);  W_i += lr * dY^T X_i.
   eff_coef (NULL = ones) gives the true gradient w.r.t. W_i when the forward
   pass weighted tap i by eff_coef[i]; the reference's UpdateSimple has no
   coefficients (it is only dimension-correct for the plain class, quirk q8). */
void oracle_tdnn_update_simple(const omat *in_value, const omat *out_deriv,
                               int Do, int Di, int K, int row_stride,
                               const int *row_offsets, const float *eff_coef,
                               float lr, float *W_acc, int ldw, float *bias_acc) {
  int N = out_deriv->rows;
  if (bias_acc) {
    for (int o = 0; o < Do; o++) {
      acc_t s = 0;
      for (int r = 0; r < N; r++) s += out_deriv->data[(long)out_deriv->stride * r + o];
      bias_acc[o] = (float)((acc_t)bias_acc[o] + (acc_t)lr * s);
    }
  }
#pragma omp parallel for schedule(static)
  for (int o = 0; o < Do; o++) {
    acc_t *tmp = (acc_t *)malloc(sizeof(acc_t) * Di);
    for (int i = 0; i < K; i++) {
      float c = eff_coef ? eff_coef[i] : 1.0f;
      if (c == 0.0f) continue;
      for (int d = 0; d < Di; d++) tmp[d] = 0;
      for (int r = 0; r < N; r++) {
        acc_t dy = out_deriv->data[(long)out_deriv->stride * r + o];
        const float *x = in_row(in_value, row_stride, row_offsets[i], r);
#pragma omp simd
        for (int d = 0; d < Di; d++) tmp[d] += dy * (acc_t)x[d];
      }
      float *w = W_acc + (long)ldw * o + (long)i * Di;
      for (int d = 0; d < Di; d++)
        w[d] = (float)((acc_t)w[d] + (acc_t)lr * (acc_t)c * tmp[d]);
    }
    free(tmp);
  }
}

/* s_i = sum((X_i W_i^T) .* dY)  -- the "out_temp.Sum()" of :534-557 (and the
   discarded one of :502-507). */
void oracle_tdnn_darts_tap_dots(const omat *in_value, const omat *out_deriv,
                                const float *W, int ldw, int Do, int Di, int K,
                                int row_stride, const int *row_offsets,
                                double *s) {
  int N = out_deriv->rows;
  for (int i = 0; i < K; i++) {
    double tot = 0.0;
#pragma omp parallel for reduction(+ : tot) schedule(static)
    for (int r = 0; r < N; r++) {
      const float *x = in_row(in_value, row_stride, row_offsets[i], r);
      const float *dy = out_deriv->data + (long)out_deriv->stride * r;
      for (int o = 0; o < Do; o++) {
        const float *w = W + (long)ldw * o + (long)i * Di;
        double t = 0;
        for (int d = 0; d < Di; d++) t += (double)x[d] * w[d];
        tot += t * dy[o];
      }
    }
    s[i] = tot;
  }
}

/* Architecture-logit update of UpdateNaturalGradient, :516-590, applied to the
   accumulator bias_params_[0:K] of `to_update`.  Reproduces quirk q5: the
   trailing scalings multiply the WHOLE accumulator, not just this minibatch's
   contribution.  In uniform-sample mode no alpha gradient is added (:491-514)
   but the scalings still run. */
void oracle_tdnn_darts_alpha_update(const double *s, const float *coef, int K,
                                    int flags, int share_index,
                                    float temp_proportion, float lr,
                                    float *alpha_acc) {
  if (!(flags & ORACLE_DARTS_UNIFORM_SAMPLE)) {
    for (int i = 0; i < K; i++) {
      float si = (float)s[i];
      if (flags & ORACLE_DARTS_FREE_SELECT) { /* :541-544 */
        alpha_acc[i] += si * coef[i];
        alpha_acc[i] += -1.0f * si * coef[i] * coef[i];
      } else if (i != share_index) { /* :546-559 */
        float tau = (flags & ORACLE_DARTS_USE_GUMBEL) ? temp_proportion : 1.0f;
        for (int j = 0; j < K; j++) alpha_acc[j] += (-1.0f * si / tau) * coef[i] * coef[j];
        alpha_acc[i] += (si / tau) * coef[i];
      }
    }
  }
  float mul = 1.0f;
  if (flags & ORACLE_DARTS_USE_ENTROPY) mul *= 5.0f; /* :565-569 */
  if (flags & ORACLE_DARTS_FREE_SELECT) mul *= 5.0f * lr; /* :574-577 */
  else if (flags & ORACLE_DARTS_USE_GUMBEL) mul *= lr;    /* :578-581 */
  else mul *= 5.0f * lr;                                  /* :582-586 */
  if (flags & ORACLE_DARTS_UPDATE_ALPHA) mul *= 10000.0f; /* :588-590 */
  for (int i = 0; i < K; i++) alpha_acc[i] *= mul;
}

/* in_value_temp of :482-532: [c_0 X_0 ... c_{K-1} X_{K-1}, 1]; a tap with
   eff_coef == 0 is left zero (:512). */
void oracle_tdnn_splice(const omat *in_value, int N, int Di, int K,
                        int row_stride, const int *row_offsets,
                        const float *eff_coef, int append_ones, omat *spliced) {
  for (int r = 0; r < N; r++) {
    float *dst = spliced->data + (long)spliced->stride * r;
    for (int i = 0; i < K; i++) {
      float c = eff_coef ? eff_coef[i] : 1.0f;
      const float *x = in_row(in_value, row_stride, row_offsets[i], r);
      for (int d = 0; d < Di; d++) dst[i * Di + d] = c == 0.0f ? 0.0f : c * x[d];
    }
    if (append_ones) dst[K * Di] = 1.0f;
  }
}
