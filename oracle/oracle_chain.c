/*
 * oracle_chain.c -- CPU restatement of the LF-MMI ("chain") objective:
 * chain::ComputeChainObjfAndDeriv, DenominatorComputation, NumeratorComputation.
 * Test infrastructure only (see oracle.h).  PARITY UNPINNED.
 *
 * UPSTREAM: none of this is shipped in /root/reference (it is reached only via
 * steps/nnet3/chain/train.py:515 -> nnet3-chain-train).  Restated from Povey et
 * al. 2016, "Purely sequence-trained neural networks for ASR based on
 * lattice-free MMI" and SURVEY.md 8(a) row A7; options pinned by
 * local/chain_NAS/run_TDNN_DARTSV3_fbk_stride_pretrain.sh:185-195
 * (leaky-hmm 0.1, l2 0.0, xent 0.1).
 *
 * nnet_output rows are t-major: row = t * num_sequences + s, cols = pdf-id.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(m, r, c) ((m)->data[(long)(m)->stride * (r) + (c)])

/* DenominatorGraph::SetInitialProbs (UPSTREAM): average state occupancy over
   num_iters (=100) steps of the row-normalised graph from the start state. */
void oracle_den_initial_probs(int H, int A, const int *src, const int *dst,
                              const float *prob, int start_state, int num_iters,
                              float *init) {
  double *norm = (double *)calloc(H, sizeof(double));
  double *cur = (double *)calloc(H, sizeof(double));
  double *nxt = (double *)calloc(H, sizeof(double));
  double *avg = (double *)calloc(H, sizeof(double));
  for (int a = 0; a < A; a++) norm[src[a]] += prob[a];
  cur[start_state] = 1.0;
  for (int it = 0; it < num_iters; it++) {
    for (int h = 0; h < H; h++) avg[h] += cur[h] / num_iters;
    memset(nxt, 0, sizeof(double) * H);
    for (int a = 0; a < A; a++)
      if (norm[src[a]] > 0) nxt[dst[a]] += cur[src[a]] * prob[a] / norm[src[a]];
    double *t = cur;
    cur = nxt;
    nxt = t;
  }
  for (int h = 0; h < H; h++) init[h] = (float)avg[h];
  free(norm);
  free(cur);
  free(nxt);
  free(avg);
}

/* DenominatorComputation::Forward + Backward (UPSTREAM chain-denominator.cc),
   per sequence.  x = exp(clamp(y,-30,30)); alpha renormalised every frame by
   A(t) = sum_i alpha(t,i) ("arbitrary_scale"); leaky-HMM transitions through
   the initial distribution.  deriv += deriv_weight * gamma_den.  Returns 1 if
   the result is finite. */
int oracle_chain_denominator(const oracle_den_graph *g, const omat *nnet_output,
                             int num_sequences, float leaky_hmm,
                             float deriv_weight, double *tot_logprob,
                             omat *deriv) {
  int H = g->num_states, A = g->num_arcs, B = num_sequences;
  int T = nnet_output->rows / B, P = nnet_output->cols;
  double total = 0.0;
  int ok = 1;
#pragma omp parallel for schedule(dynamic) reduction(+ : total) reduction(& : ok)
  for (int s = 0; s < B; s++) {
    float *x = (float *)malloc(sizeof(float) * (size_t)T * P);
    float *alpha = (float *)malloc(sizeof(float) * (size_t)(T + 1) * H); /* alpha-dash */
    double *asum = (double *)malloc(sizeof(double) * (T + 1));
    float *beta = (float *)malloc(sizeof(float) * 2 * (size_t)H);
    for (int t = 0; t < T; t++)
      for (int p = 0; p < P; p++) {
        float y = AT(nnet_output, t * B + s, p);
        y = y < -30.0f ? -30.0f : (y > 30.0f ? 30.0f : y); /* ApplyExpLimited(-30,30) */
        x[(size_t)t * P + p] = expf(y);
      }
    /* AlphaFirstFrame + AlphaDash(0) */
    double sum0 = 0;
    for (int h = 0; h < H; h++) sum0 += g->initial_probs[h];
    asum[0] = sum0;
    for (int h = 0; h < H; h++)
      alpha[h] = g->initial_probs[h] + leaky_hmm * (float)sum0 * g->initial_probs[h];
    double logcorr = 0.0;
    double *acc = (double *)malloc(sizeof(double) * H);
    for (int t = 1; t <= T; t++) {
      const float *prev = alpha + (size_t)(t - 1) * H;
      float *cur = alpha + (size_t)t * H;
      const float *xt = x + (size_t)(t - 1) * P;
      memset(acc, 0, sizeof(double) * H);
      for (int a = 0; a < A; a++)
        acc[g->arc_dst[a]] += (double)prev[g->arc_src[a]] * g->arc_prob[a] * xt[g->arc_pdf[a]];
      double inv = 1.0 / asum[t - 1]; /* arbitrary_scale */
      logcorr += log(asum[t - 1]);
      double sm = 0;
      for (int h = 0; h < H; h++) {
        cur[h] = (float)(acc[h] * inv);
        sm += cur[h];
      }
      asum[t] = sm;
      for (int h = 0; h < H; h++) /* AlphaDash(t) */
        cur[h] = cur[h] + leaky_hmm * (float)sm * g->initial_probs[h];
    }
    free(acc);
    double tot = 0;
    for (int h = 0; h < H; h++) tot += alpha[(size_t)T * H + h];
    double lp = log(tot) + logcorr;
    total += lp;
    if (!(lp - lp == 0.0)) ok = 0;
    if (deriv) {
      /* BetaDashLastFrame + Beta(T) */
      float *bnext = beta, *bcur = beta + H;
      double bd = 1.0 / tot, lsum = 0;
      for (int h = 0; h < H; h++) lsum += (double)g->initial_probs[h] * bd;
      for (int h = 0; h < H; h++) bnext[h] = (float)(bd + leaky_hmm * lsum);
      double *bacc = (double *)malloc(sizeof(double) * H);
      double *gam = (double *)malloc(sizeof(double) * P);
      for (int t = T - 1; t >= 0; t--) {
        const float *ad = alpha + (size_t)t * H;
        const float *xt = x + (size_t)t * P;
        double inv = 1.0 / asum[t];
        memset(bacc, 0, sizeof(double) * H);
        memset(gam, 0, sizeof(double) * P);
        for (int a = 0; a < A; a++) {
          double v = (double)g->arc_prob[a] * xt[g->arc_pdf[a]] * bnext[g->arc_dst[a]] * inv;
          bacc[g->arc_src[a]] += v;
          gam[g->arc_pdf[a]] += v * ad[g->arc_src[a]];
        }
        double ls = 0;
        for (int h = 0; h < H; h++) ls += (double)g->initial_probs[h] * bacc[h];
        for (int h = 0; h < H; h++) bcur[h] = (float)(bacc[h] + leaky_hmm * ls); /* Beta(t) */
        for (int p = 0; p < P; p++)
          AT(deriv, t * B + s, p) += deriv_weight * (float)gam[p];
        float *tmp = bnext;
        bnext = bcur;
        bcur = tmp;
      }
      free(bacc);
      free(gam);
    }
    free(x);
    free(alpha);
    free(asum);
    free(beta);
  }
  *tot_logprob = total;
  return ok;
}

static inline double log_add(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}

/* NumeratorComputation::Forward/Backward (UPSTREAM chain-numerator.cc): log-domain
   forward-backward over each sequence's time-synchronous supervision graph.
   Returns weight * sum_seq log p_num; post += weight * gamma_num. */
double oracle_chain_numerator(const oracle_supervision *sup,
                              const omat *nnet_output, omat *post) {
  int B = sup->num_sequences;
  double total = 0.0;
  for (int s = 0; s < B; s++) {
    int s0 = sup->seq_state_begin[s], s1 = sup->seq_state_begin[s + 1];
    int a0 = sup->seq_arc_begin[s], a1 = sup->seq_arc_begin[s + 1];
    int ns = s1 - s0;
    double *la = (double *)malloc(sizeof(double) * ns);
    double *lb = (double *)malloc(sizeof(double) * ns);
    for (int i = 0; i < ns; i++) la[i] = lb[i] = -INFINITY;
    la[0] = 0.0; /* first state of the sequence is its start state */
    for (int a = a0; a < a1; a++) { /* arcs sorted by source-state time */
      int t = sup->state_time[sup->arc_src[a]];
      double v = la[sup->arc_src[a] - s0] + sup->arc_logprob[a] +
                 AT(nnet_output, t * B + s, sup->arc_pdf[a]);
      la[sup->arc_dst[a] - s0] = log_add(la[sup->arc_dst[a] - s0], v);
    }
    double tot = -INFINITY;
    for (int i = 0; i < ns; i++)
      if (sup->final_logprob[s0 + i] != -INFINITY) {
        tot = log_add(tot, la[i] + sup->final_logprob[s0 + i]);
        lb[i] = sup->final_logprob[s0 + i];
      }
    total += tot;
    for (int a = a1 - 1; a >= a0; a--) {
      int t = sup->state_time[sup->arc_src[a]];
      double ll = sup->arc_logprob[a] + AT(nnet_output, t * B + s, sup->arc_pdf[a]);
      double v = ll + lb[sup->arc_dst[a] - s0];
      lb[sup->arc_src[a] - s0] = log_add(lb[sup->arc_src[a] - s0], v);
      if (post) {
        double g = exp(la[sup->arc_src[a] - s0] + v - tot);
        AT(post, t * B + s, sup->arc_pdf[a]) += sup->weight * (float)g;
      }
    }
    free(la);
    free(lb);
  }
  return sup->weight * total;
}

/* chain::ComputeChainObjfAndDeriv (UPSTREAM chain-training.cc) plus the xent
   handling of NnetChainTrainer::ProcessOutputs: xent_deriv receives the
   numerator posteriors (caller scales by xent_regularize). */
int oracle_chain_objf_and_deriv(const oracle_den_graph *g,
                                const oracle_supervision *sup,
                                const omat *nnet_output, float leaky_hmm,
                                float l2_regularize, float xent_regularize,
                                double *objf, double *l2_term, double *weight,
                                omat *nnet_output_deriv, omat *xent_deriv) {
  (void)xent_regularize;
  int B = sup->num_sequences;
  omat *d = nnet_output_deriv;
  if (d)
    for (int r = 0; r < d->rows; r++) memset(d->data + (long)d->stride * r, 0, sizeof(float) * d->cols);
  double den = 0.0;
  int ok = oracle_chain_denominator(g, nnet_output, B, leaky_hmm, -sup->weight, &den, d);
  den *= sup->weight;
  double num;
  if (xent_deriv) {
    for (int r = 0; r < xent_deriv->rows; r++)
      memset(xent_deriv->data + (long)xent_deriv->stride * r, 0, sizeof(float) * xent_deriv->cols);
    num = oracle_chain_numerator(sup, nnet_output, xent_deriv);
    if (d)
      for (int r = 0; r < d->rows; r++)
        for (int c = 0; c < d->cols; c++) AT(d, r, c) += AT(xent_deriv, r, c);
  } else {
    num = oracle_chain_numerator(sup, nnet_output, d);
  }
  *objf = num - den;
  *weight = (double)sup->weight * B * sup->frames_per_sequence;
  if (!((*objf) - (*objf) == 0.0) || !ok) { /* failure: objf = -10 * weight, zero derivs */
    if (d)
      for (int r = 0; r < d->rows; r++) memset(d->data + (long)d->stride * r, 0, sizeof(float) * d->cols);
    if (xent_deriv)
      for (int r = 0; r < xent_deriv->rows; r++)
        memset(xent_deriv->data + (long)xent_deriv->stride * r, 0, sizeof(float) * xent_deriv->cols);
    *objf = -10.0 * (*weight);
    ok = 0;
  }
  if (l2_regularize == 0.0f) {
    *l2_term = 0.0;
  } else {
    double scale = (double)sup->weight * l2_regularize, tr = 0;
    for (int r = 0; r < nnet_output->rows; r++)
      for (int c = 0; c < nnet_output->cols; c++) {
        double y = AT(nnet_output, r, c);
        tr += y * y;
        if (d) AT(d, r, c) += (float)(-scale * y);
      }
    *l2_term = -0.5 * scale * tr;
  }
  return ok;
}
