/*
 * oracle_norm.c -- CPU restatement of BatchNormComponent / BatchNormTestComponent.
 * Test infrastructure only (see oracle.h).  PARITY UNPINNED.
 *
 * Follows /root/reference/src/nnet3/nnet-normalize-component.cc:
 *   BatchNormComponent::Propagate :401-465, Backprop :467-549, StoreStats :551-589,
 *   ComputeDerived :209-247; BatchNormTestComponent::ComputeDerived :682-715,
 *   Propagate :843-877, Backprop :879-922.  (block-dim < dim is a reshape of
 *   the same matrix, :406-416; callers pass the reshaped view.)
 */
#include "oracle.h"
#include <math.h>

/* memo layout = Memo::mean_uvar_scale rows: 0 mean, 1 uvar, 2 scale,
   3 var_deriv_mod (bwd temp), 4 temp (bwd temp)  (:429-432, :510-513). */
void oracle_batchnorm_propagate(const omat *in, float epsilon, float target_rms,
                                omat *out, float *memo) {
  int N = in->rows, D = in->cols;
  float *mean = memo, *uvar = memo + D, *scale = memo + 2 * D;
  float var_scale = 1.0f / (target_rms * target_rms); /* :438 */
#pragma omp parallel for schedule(static)
  for (int d = 0; d < D; d++) {
    double s = 0.0, s2 = 0.0;
    for (int r = 0; r < N; r++) {
      double x = in->data[(long)in->stride * r + d];
      s += x;
      s2 += x * x;
    }
    mean[d] = (float)(s / N);  /* AddRowSumMat(1/N) :433 */
    uvar[d] = (float)(s2 / N); /* AddDiagMat2(1/N)  :434 */
    float v = var_scale * uvar[d] - var_scale * mean[d] * mean[d]; /* :439 */
    if (v < 0.0f) v = 0.0f;                                        /* :441 */
    v += var_scale * epsilon;                                      /* :442 */
    scale[d] = powf(v, -0.5f);                                     /* :445 */
  }
#pragma omp parallel for schedule(static)
  for (int r = 0; r < N; r++)
    for (int d = 0; d < D; d++)
      out->data[(long)out->stride * r + d] =
          (in->data[(long)in->stride * r + d] - mean[d]) * scale[d]; /* :449-451 */
}

void oracle_batchnorm_backprop(const omat *out_value, const omat *out_deriv,
                               float target_rms, float *memo, omat *in_deriv) {
  int N = out_value->rows, D = out_value->cols;
  float *scale = memo + 2 * D, *vdm = memo + 3 * D, *temp = memo + 4 * D;
  float coeff = -1.0f / (target_rms * target_rms * N); /* :520 */
#pragma omp parallel for schedule(static)
  for (int d = 0; d < D; d++) {
    double zz = 0.0, sd = 0.0;
    for (int r = 0; r < N; r++) {
      double dz = out_deriv->data[(long)out_deriv->stride * r + d];
      zz += (double)out_value->data[(long)out_value->stride * r + d] * dz;
      sd += dz;
    }
    vdm[d] = (float)(coeff * zz) * scale[d]; /* :522-524 */
    temp[d] = (float)(-sd / N);              /* :526 */
  }
#pragma omp parallel for schedule(static)
  for (int r = 0; r < N; r++)
    for (int d = 0; d < D; d++) {
      float dz = out_deriv->data[(long)out_deriv->stride * r + d];
      float z = out_value->data[(long)out_value->stride * r + d];
      in_deriv->data[(long)in_deriv->stride * r + d] =
          (dz + temp[d]) * scale[d] + z * vdm[d]; /* :529-538 */
    }
}

/* :586-588 -- stats are doubles in the reference (nnet-normalize-component.h:458-462) */
void oracle_batchnorm_store_stats(const float *memo, int D, int num_frames,
                                  double *count, double *stats_sum,
                                  double *stats_sumsq) {
  *count += num_frames;
  for (int d = 0; d < D; d++) {
    stats_sum[d] += (double)num_frames * memo[d];
    stats_sumsq[d] += (double)num_frames * memo[D + d];
  }
}

/* :695-714 (identical text at :227-246) */
void oracle_batchnorm_compute_derived(double count, const double *stats_sum,
                                      const double *stats_sumsq, int D,
                                      float epsilon, float target_rms,
                                      float *scale, float *offset) {
  for (int d = 0; d < D; d++) {
    float off = (float)(stats_sum[d] * (-1.0 / count)); /* -mean */
    float sc = (float)(stats_sumsq[d] * (1.0 / count));
    sc += -1.0f * off * off; /* variance */
    if (sc < 0.0f) sc = 0.0f;
    sc += epsilon;
    sc = powf(sc, -0.5f);
    sc *= target_rms;
    scale[d] = sc;
    offset[d] = off * sc;
  }
}

void oracle_batchnorm_test_propagate(const omat *in, const float *scale,
                                     const float *offset, omat *out) {
  for (int r = 0; r < in->rows; r++)
    for (int d = 0; d < in->cols; d++) /* :872-874 */
      out->data[(long)out->stride * r + d] =
          in->data[(long)in->stride * r + d] * scale[d] + offset[d];
}

void oracle_batchnorm_test_backprop(const omat *out_deriv, const float *scale,
                                    omat *in_deriv) {
  for (int r = 0; r < out_deriv->rows; r++)
    for (int d = 0; d < out_deriv->cols; d++) /* :919-920 */
      in_deriv->data[(long)in_deriv->stride * r + d] =
          out_deriv->data[(long)out_deriv->stride * r + d] * scale[d];
}
