/*
 * oracle_optim.c -- CPU restatement of the per-minibatch optimizer step.
 * Test infrastructure only (see oracle.h).  PARITY UNPINNED.
 *
 * Follows /root/reference/src/nnet3/nnet-utils.cc:
 *   ConstrainOrthonormalInternal :914-1032, UpdateNnetWithMaxChange :2085-2175,
 *   ApplyL2Regularization :2223-2245.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

/* M (rows x cols, rows <= cols) <- M - 4*alpha*(P - scale^2 I) M, P = M M^T.
   scale < 0 selects the floating scale of :938-986. */
void oracle_constrain_orthonormal(float scale, float *M, int rows, int cols,
                                  int ld) {
  double *P = (double *)malloc(sizeof(double) * rows * rows);
  for (int i = 0; i < rows; i++)
    for (int j = 0; j <= i; j++) {
      double a = 0;
      for (int k = 0; k < cols; k++) a += (double)M[(long)ld * i + k] * M[(long)ld * j + k];
      P[i * rows + j] = P[j * rows + i] = (float)a; /* SymAddMat2 + CopyLowerToUpper, float storage */
    }
  float update_speed = 0.125f;
  if (scale < 0.0f) {
    double trace_P = 0, trace_P_P = 0;
    for (int i = 0; i < rows; i++) trace_P += P[i * rows + i];
    for (int i = 0; i < rows * rows; i++) trace_P_P += P[i] * P[i];
    scale = sqrtf((float)(trace_P_P / trace_P)); /* :964 */
    float ratio = (float)(trace_P_P * rows / (trace_P * trace_P));
    if (ratio > 1.02f) { /* :979-985 */
      update_speed *= 0.5f;
      if (ratio > 1.1f) update_speed *= 0.5f;
    }
  }
  for (int i = 0; i < rows; i++) P[i * rows + i] -= (double)scale * scale; /* :987 */
  float alpha = update_speed / (scale * scale);                              /* :1019 */
  float *upd = (float *)malloc(sizeof(float) * rows * cols);
  for (int i = 0; i < rows; i++)
    for (int k = 0; k < cols; k++) {
      double a = 0;
      for (int j = 0; j < rows; j++) a += P[i * rows + j] * M[(long)ld * j + k];
      upd[i * cols + k] = (float)(-4.0 * alpha * a); /* :1030 */
    }
  for (int i = 0; i < rows; i++)
    for (int k = 0; k < cols; k++) M[(long)ld * i + k] += upd[i * cols + k]; /* :1031 */
  free(P);
  free(upd);
}

/* delta += scale * params with scale = -2 * l2_scale * lr * l2  (:2237-2242) */
void oracle_apply_l2(const float *params, float *delta, long n, float scale) {
  if (scale == 0.0f) return;
  for (long i = 0; i < n; i++) delta[i] += scale * params[i];
}

/* Per-component then global max-change (:2095-2170).  dot_prods[i] = ||delta_i||^2.
   On return scale_factors[i] is the total factor to apply to component i's
   delta when adding it to the model; *ok = 0 for an infinite change (:2144-2147). */
void oracle_max_change_scales(const double *dot_prods, const float *max_change,
                              int n, float max_param_change,
                              float max_change_scale, float scale,
                              float *scale_factors, int *ok) {
  float param_delta_squared = 0.0f;
  for (int i = 0; i < n; i++) {
    float dot = (float)dot_prods[i], mc = max_change[i];
    if (mc != 0.0f && sqrtf(dot) * fabsf(scale) > mc * max_change_scale)
      scale_factors[i] = mc * max_change_scale / (sqrtf(dot) * fabsf(scale));
    else
      scale_factors[i] = 1.0f;
    param_delta_squared += powf(scale_factors[i], 2.0f) * dot;
  }
  float param_delta = sqrtf(param_delta_squared) * fabsf(scale);
  *ok = 1;
  if (max_param_change != 0.0f && param_delta > max_param_change * max_change_scale) {
    if (param_delta - param_delta != 0.0f) {
      *ok = 0;
      return;
    }
    scale *= max_param_change * max_change_scale / param_delta;
  }
  for (int i = 0; i < n; i++) scale_factors[i] *= scale; /* :2172 */
}
