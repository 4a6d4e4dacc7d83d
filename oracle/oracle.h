/*
 * oracle.h -- CPU restatement (plain C) of the LF-MMI TDNN-F / DARTS hot path of
 * skhu101/TDNN-F_NAS.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (tdnn-f_nas_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or buildable
 * Kaldi tree, so this restatement is pinned only by
 * self-consistency checks in tests/ (finite differences, brute-force HMM path
 * enumeration, independent PyTorch-CPU autograd).  Every function cites the
 * reference file:line it follows; "UPSTREAM" marks stock-Kaldi algorithms that
 * the reference calls but does not ship (restated from the published papers).
 *
 * All matrices are row-major float with a row stride in ELEMENTS, exactly the
 * (Data, NumRows, NumCols, Stride) view of Kaldi's CuMatrixBase<float>
 * (reference usage: src/nnet3/nnet-tdnn-component.cc:815-819).
 * Inner products accumulate in double when built with -DORACLE_F64ACC (the
 * parity build); the "fast" build (cpu_baseline) accumulates in float.
 */
#ifndef TDNNF_ORACLE_H_
#define TDNNF_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  float *data;
  int rows, cols, stride;
} omat;

/* ---- coefficient modes of TdnnDARTSV3Component (nnet-tdnn-component.cc:250-289) */
enum {
  ORACLE_DARTS_USE_GUMBEL = 1,
  ORACLE_DARTS_FREE_SELECT = 2,
  ORACLE_DARTS_UNIFORM_SAMPLE = 4,
  ORACLE_DARTS_USE_ENTROPY = 8,
  ORACLE_DARTS_UPDATE_ALPHA = 16
};

/* ------------------------------------------------------------------ A1 / A2 */
int oracle_tdnn_share_index(const int *time_offsets, int K);
void oracle_tdnn_darts_coef(const float *log_alpha, int K, int flags,
                            float temp_proportion, const float *gumbel_u,
                            float sample_u, float *coef);
void oracle_tdnn_darts_effective_coef(const float *coef, int K, int flags,
                                      int share_index, float *eff);
void oracle_tdnn_propagate(const omat *in, const float *W, int ldw, int Do,
                           int Di, int K, int row_stride,
                           const int *row_offsets, const float *bias,
                           const float *eff_coef, int init_mode, omat *out);
void oracle_tdnn_backprop_data(const omat *out_deriv, const float *W, int ldw,
                               int Do, int Di, int K, int row_stride,
                               const int *row_offsets, const float *eff_coef,
                               omat *in_deriv);
void oracle_tdnn_update_simple(const omat *in_value, const omat *out_deriv,
                               int Do, int Di, int K, int row_stride,
                               const int *row_offsets, const float *eff_coef,
                               float lr, float *W_acc, int ldw, float *bias_acc);
void oracle_tdnn_darts_tap_dots(const omat *in_value, const omat *out_deriv,
                                const float *W, int ldw, int Do, int Di, int K,
                                int row_stride, const int *row_offsets,
                                double *s);
void oracle_tdnn_darts_alpha_update(const double *s, const float *coef, int K,
                                    int flags, int share_index,
                                    float temp_proportion, float lr,
                                    float *alpha_acc);
void oracle_tdnn_splice(const omat *in_value, int N, int Di, int K,
                        int row_stride, const int *row_offsets,
                        const float *eff_coef, int append_ones, omat *spliced);

/* ------------------------------------------------------------------ A3 / A4 */
void oracle_batchnorm_propagate(const omat *in, float epsilon, float target_rms,
                                omat *out, float *memo /* 5 x D */);
void oracle_batchnorm_backprop(const omat *out_value, const omat *out_deriv,
                               float target_rms, float *memo, omat *in_deriv);
void oracle_batchnorm_store_stats(const float *memo, int D, int num_frames,
                                  double *count, double *stats_sum,
                                  double *stats_sumsq);
void oracle_batchnorm_compute_derived(double count, const double *stats_sum,
                                      const double *stats_sumsq, int D,
                                      float epsilon, float target_rms,
                                      float *scale, float *offset);
void oracle_batchnorm_test_propagate(const omat *in, const float *scale,
                                     const float *offset, omat *out);
void oracle_batchnorm_test_backprop(const omat *out_deriv, const float *scale,
                                    omat *in_deriv);

/* ----------------------------------------------------------------------- A5 */
void oracle_gumbel_noise(const float *u, int n, float *g);
void oracle_softmax_flops_propagate(const omat *in, const float *gumbel_u,
                                    float temp_proportion, omat *out);
void oracle_softmax_flops_backprop(const omat *out_value, omat *out_deriv,
                                   float scale, const float *flops, int dim,
                                   float temp_proportion, omat *in_deriv);
int oracle_onehot_index(float u, int C);
void oracle_onehot_propagate(float u, omat *out);
void oracle_copyn_propagate(const omat *in, float scale, omat *out);
void oracle_copyn_backprop(const omat *out_deriv, float scale, omat *in_deriv);
void oracle_constant_function_propagate(const float *output, omat *out);
void oracle_constant_function_backprop(const omat *out_deriv, float lr,
                                       float *output_acc);
void oracle_flops_constraint_backprop(const float *flops, float scale, int rows_in,
                                      int cols_in, omat *in_deriv);

/* ----------------------------------------------------------------------- A6 */
void oracle_elementwise_product_propagate(const omat *in, int output_dim,
                                          omat *out);
void oracle_elementwise_product_backprop(const omat *in_value,
                                         const omat *out_deriv, int output_dim,
                                         omat *in_deriv);
void oracle_relu_propagate(const omat *in, omat *out);
void oracle_relu_backprop(const omat *out_value, const omat *out_deriv,
                          omat *in_deriv);
void oracle_relu_repair(const double *deriv_sum, double count, int dim,
                        float self_repair_scale, float lower, float upper,
                        omat *in_deriv);
void oracle_relu_store_stats(const omat *out_value, double *value_sum,
                             double *deriv_sum, double *count);
void oracle_affine_propagate(const omat *in, const float *W, int ldw,
                             const float *bias, int Do, omat *out);
void oracle_affine_backprop(const omat *out_deriv, const float *W, int ldw,
                            int Di, omat *in_deriv);
void oracle_affine_update_simple(const omat *in_value, const omat *out_deriv,
                                 float lr, float *W_acc, int ldw,
                                 float *bias_acc);
void oracle_log_softmax_propagate(const omat *in, omat *out);
void oracle_log_softmax_backprop(const omat *out_value, const omat *out_deriv,
                                 omat *in_deriv);
void oracle_sum_scaled(const omat *a, float sa, const omat *b, float sb,
                       omat *out);
void oracle_general_dropout_propagate(const omat *in, const float *mask,
                                      int num_seq, omat *out);

/* ----------------------------------------------------------------------- A7 */
typedef struct {
  int num_states, num_arcs, num_pdfs;
  const int *arc_src, *arc_dst, *arc_pdf; /* arc lists, any order */
  const float *arc_prob;
  const float *initial_probs; /* H */
} oracle_den_graph;

typedef struct {
  /* per-sequence numerator graphs, concatenated.  Every arc consumes exactly one
     frame; state_time[s] is the frame index at which state s is entered; the
     unique start state has time 0 and final states have time T. */
  int num_sequences, frames_per_sequence;
  const int *seq_state_begin; /* num_sequences+1 */
  const int *seq_arc_begin;   /* num_sequences+1 */
  const int *state_time;
  const float *final_logprob; /* per state; -inf when not final */
  const int *arc_src, *arc_dst, *arc_pdf; /* global state ids; sorted by time */
  const float *arc_logprob;
  float weight;
} oracle_supervision;

void oracle_den_initial_probs(int H, int A, const int *src, const int *dst,
                              const float *prob, int start_state, int num_iters,
                              float *init);
int oracle_chain_denominator(const oracle_den_graph *g, const omat *nnet_output,
                             int num_sequences, float leaky_hmm,
                             float deriv_weight, double *tot_logprob,
                             omat *deriv /* += deriv_weight*gamma */);
double oracle_chain_numerator(const oracle_supervision *sup,
                              const omat *nnet_output, omat *post /* += w*gamma */);
int oracle_chain_objf_and_deriv(const oracle_den_graph *g,
                                const oracle_supervision *sup,
                                const omat *nnet_output, float leaky_hmm,
                                float l2_regularize, float xent_regularize,
                                double *objf, double *l2_term, double *weight,
                                omat *nnet_output_deriv, omat *xent_deriv);

/* ----------------------------------------------------------------------- A8 */
typedef struct oracle_ng oracle_ng;
oracle_ng *oracle_ng_create(int rank, int update_period,
                            float num_samples_history, float alpha);
void oracle_ng_destroy(oracle_ng *);
void oracle_ng_precondition(oracle_ng *, omat *X, float *scale);
int oracle_ng_state(const oracle_ng *, float *W /* R x D */, float *d,
                    float *rho, int *t);

/* ----------------------------------------------------------------------- A9 */
void oracle_constrain_orthonormal(float scale, float *M, int rows, int cols,
                                  int ld);
void oracle_apply_l2(const float *params, float *delta, long n, float scale);
void oracle_max_change_scales(const double *dot_prods, const float *max_change,
                              int n, float max_param_change,
                              float max_change_scale, float scale,
                              float *scale_factors, int *ok);

#ifdef __cplusplus
}
#endif
#endif
