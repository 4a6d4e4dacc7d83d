/*
 * tdnnf_hip.h -- C-ABI of libtdnnf_hip.so: the MI355X (gfx950) implementation of the
 * LF-MMI TDNN-F / DARTS hot path of skhu101/TDNN-F_NAS.
 *
 * Every entry point replaces one nnet3 Component virtual (or one optimizer helper)
 * of the reference; the reference symbol is cited next to it as
 * /root/reference-relative file:line.  INTEGRATION.md shows the Kaldi-side adapter
 * that forwards CuMatrixBase<float> views to these functions.
 *
 * Conventions
 *  - tdnnf_mat is exactly Kaldi's CuMatrixBase<float> view: device pointer, rows,
 *    cols, row stride in ELEMENTS, row-major (usage: src/nnet3/nnet-tdnn-component.cc:815-819).
 *  - Pointers named *_dev are device memory; "host" pointers are small index
 *    arrays read during the call.  The caller owns every buffer.
 *  - stream is a hipStream_t (NULL = default stream).  No call synchronises the
 *    stream or the device; scalars (objf, dot products, taps' s_i) are written
 *    to device memory.  Random draws are INPUTS (device buffers) so results are
 *    reproducible (SURVEY.md 7 "Randomness").
 *  - Return value: 0 = ok, TDNNF_EINVAL = bad argument (nothing launched),
 *    TDNNF_EHIP = a HIP runtime call failed.  tdnnf_last_error() gives the text.
 *    Nothing throws across this boundary (the reference's KALDI_ERR/KALDI_ASSERT
 *    become error codes; the C++ adapter turns them back into exceptions).
 *  - kBackpropAdds / kPropagateAdds semantics of the reference are kept: where
 *    the reference adds into its output, so do we.
 */
#ifndef TDNNF_HIP_H_
#define TDNNF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDNNF_OK 0
#define TDNNF_EINVAL 1
#define TDNNF_EHIP 2

#define TDNNF_MAX_OFFSETS 16

typedef struct {
  float *data;
  int rows, cols, stride;
} tdnnf_mat;

typedef void *tdnnf_stream;
typedef struct tdnnf_ng tdnnf_ng; /* OnlineNaturalGradient state (A8, below) */

const char *tdnnf_last_error(void);
int tdnnf_abi_version(void);
/* Tuning options: process-wide integers selecting between code paths that are all held to the same parity tests (the library reads
   no environment variable for them).  Unknown names fail with TDNNF_EINVAL.
     "ng_grouped"    1 (default) natural-gradient side chain of a gradient bucket as grouped launches, 0 per object   [read by tdnnf_net_create]
     "ng_fuse"       1 (default) output-side statistic inside the BatchNorm / ReLU backward sweep when that pays, 0 own GEMM, 2 always
     "ng_early_in"   1 (default) input-side statistics ahead of the backward pass, 0 with the component's backward call [tdnnf_net_create]
     "wgrad_stream"  -1 (default) parameter gradients on their own stream for small minibatches, 0 never, 1 always       [tdnnf_net_create]
     "gemm_ring"     1 (default) persistent LDS-DMA-ring rows GEMM where it applies, 0 the plain tile kernel
     "planes"        1 (default) gemm_precision 2 runs the pre-split bf16-plane GEMMs where they apply, 0 the in-kernel split
     "den_mw_test_abort", "planes_check_bound": test hooks (0 = off), see tests/test_gpu_parity.py, tests/test_gpu_net.py
     "den_split"     -1 (default) the trainer runs the denominator's two recursions side by side for <= 96 sequences, 0 never, 1 always
                     (set before the first tdnnf_net_forward_backward of a net: it sizes the chain workspace) */
int tdnnf_set_option(const char *name, int value);
int tdnnf_get_option(const char *name, int *value_out);

/* ---- TdnnDARTSV3Component coefficient flags (nnet-tdnn-component.cc:150-163) */
#define TDNNF_DARTS_USE_GUMBEL 1
#define TDNNF_DARTS_FREE_SELECT 2
#define TDNNF_DARTS_UNIFORM_SAMPLE 4
#define TDNNF_DARTS_USE_ENTROPY 8
#define TDNNF_DARTS_UPDATE_ALPHA 16

/* precomputed indexes: TdnnDARTSV3Component::PrecomputedIndexes
   (src/nnet3/nnet-convolutional-component.h:208-218) */
typedef struct {
  int row_stride;
  int num_offsets;
  int row_offsets[TDNNF_MAX_OFFSETS];
} tdnnf_tdnn_indexes;

/* ======================================================================= A1/A2
 * TdnnComponent / TdnnDARTSV3Component (src/nnet3/nnet-tdnn-component.cc).
 * linear_params_dev is Do x (K*Di) with row stride ldw (column block i = tap i).
 */

/* Coefficients of Propagate :250-289.  Reads log_alpha (bias_params_[0:K]),
   the K uniform draws for the Gumbel noise and ONE uniform draw for the tap
   sample; writes coef_memo_dev[K] (the memo of :330-332) and eff_coef_dev[K] (the
   weight each tap's GEMM gets in :292-328).  share_index per :232-240. */
int tdnnf_tdnn_darts_coef(const float *log_alpha_dev, int K, int flags, float temp_proportion,
                          const float *gumbel_u_dev, const float *sample_u_dev, int share_index,
                          float *coef_memo_dev, float *eff_coef_dev, tdnnf_stream stream);

/* Propagate :214-333 (plain TdnnComponent: eff_coef_dev == NULL, all ones).
   init_mode 0: out += ... (no bias, kPropagateAdds); 1: out = bias + ... (:233-235);
   2: out = 0 + ... (:238, reference quirk q1 for layers whose offsets[1] < 0). */
int tdnnf_tdnn_propagate(const tdnnf_tdnn_indexes *indexes, const tdnnf_mat *in,
                         const float *linear_params_dev, int ldw, int Do, int Di,
                         const float *bias_dev, const float *eff_coef_dev, int init_mode,
                         tdnnf_mat *out, tdnnf_stream stream);

/* Backprop, data part :366-416: in_deriv views += c_i * out_deriv * W_i (kBackpropAdds). */
int tdnnf_tdnn_backprop_data(const tdnnf_tdnn_indexes *indexes, const tdnnf_mat *out_deriv,
                             const float *linear_params_dev, int ldw, int Do, int Di,
                             const float *eff_coef_dev, tdnnf_mat *in_deriv, tdnnf_stream stream);

/* UpdateSimple :433-455 (raw gradient, is_gradient_ path): bias_acc += lr*colsum(dY),
   W_acc_i += lr * c_i * dY^T X_i.  workspace: tdnnf_tdnn_update_workspace_bytes(). */
size_t tdnnf_tdnn_update_workspace_bytes(int Do, int Di, int K, int num_rows);
int tdnnf_tdnn_update_simple(const tdnnf_tdnn_indexes *indexes, const tdnnf_mat *in_value,
                             const tdnnf_mat *out_deriv, int Do, int Di,
                             const float *eff_coef_dev, float lr, float *W_acc_dev, int ldw,
                             float *bias_acc_dev, void *workspace_dev, size_t workspace_bytes,
                             tdnnf_stream stream);

/* scratch of tdnnf_tdnn_darts_alpha_update: s_i (K doubles), then K x TDNNF_TAP_DOTS_SLABS row-slab partials (added in slab order) */
#define TDNNF_TAP_DOTS_SLABS 64
#define TDNNF_TAP_DOTS_DOUBLES(K) ((K) * (1 + TDNNF_TAP_DOTS_SLABS))

/* Architecture-logit update of UpdateNaturalGradient :490-590.  s_i = <X_i W_i^T, dY>
   is obtained as <dW_i, W_i> from tap_grad_dev (K blocks of dY^T X_i, Do x K*Di,
   UNSCALED by coefficients and lr), avoiding the reference's extra forward GEMM
   per tap (:534-539).  alpha_acc_dev = to_update->bias_params_[0:K]. */
int tdnnf_tdnn_darts_alpha_update(const float *tap_grad_dev, int ldg, const float *linear_params_dev,
                                  int ldw, int Do, int Di, int K, const float *coef_memo_dev,
                                  int flags, int share_index, float temp_proportion, float lr,
                                  float *alpha_acc_dev, double *tap_dots_dev /* TDNNF_TAP_DOTS_DOUBLES(K) doubles: the first K receive s_i */,
                                  tdnnf_stream stream);
/* (in TDNNF_DARTS_UNIFORM_SAMPLE mode no gradient is added -- the reference computes and discards it, :502-507 --
   only the trailing scalings :565-590 run; tap_grad_dev may then be NULL) */

/* UpdateNaturalGradient :457-626 -- the branch Backprop takes whenever use_natural_gradient_ is set and the component is
   not a gradient holder (:427-430), i.e. in every recipe.  Literal order of the reference: architecture-logit update
   (:490-590, as tdnnf_tdnn_darts_alpha_update), in_value_temp = [c_i X_i ..., 1] (:466-532; the share tap unscaled
   unless free_select, a zero-coefficient tap left zero), out_deriv_temp = out_deriv (:592), both preconditioned in
   place by the caller's OnlineNaturalGradient objects (:598-599), then with local_lrate = in_scale * out_scale *
   learning_rate: linear_params_ += local_lrate out_deriv_temp^T in_value_temp[:, :K Di] (:621-624) and
   bias_params_[K:] += local_lrate out_deriv_temp^T in_value_temp[:, K Di] (:606-616).
   Plain TdnnComponent (UPSTREAM): coef_memo_dev = eff_coef_dev = alpha_acc_dev = NULL (linear_params_dev unused).
   bias_acc_dev (Do floats = to_update->bias_params_ + K, or NULL when the component has no bias: no ones column),
   alpha_acc_dev (K floats = to_update->bias_params_).  coef_memo_dev / eff_coef_dev: the two halves of the memo
   tdnnf_tdnn_darts_coef wrote in Propagate.  learning_rate == 0 returns at once (:423-424).  Nothing synchronises:
   the two scales stay on the device (tdnnf_ng_scale_dev). */
size_t tdnnf_tdnn_update_natural_gradient_workspace_bytes(int Do, int Di, int K, int num_rows, int has_bias);
int tdnnf_tdnn_update_natural_gradient(const tdnnf_tdnn_indexes *indexes, const tdnnf_mat *in_value,
                                       const tdnnf_mat *out_deriv, int Do, int Di,
                                       const float *linear_params_dev, int ldw_params,
                                       const float *coef_memo_dev, const float *eff_coef_dev, int flags,
                                       int share_index, float temp_proportion,
                                       tdnnf_ng *preconditioner_in, tdnnf_ng *preconditioner_out,
                                       float learning_rate, float *W_acc_dev, int ldw, float *bias_acc_dev,
                                       float *alpha_acc_dev, void *workspace_dev, size_t workspace_bytes,
                                       tdnnf_stream stream);

/* ======================================================================= A3/A4
 * BatchNormComponent / BatchNormTestComponent (src/nnet3/nnet-normalize-component.cc) */

/* Column reductions (BatchNorm statistics, ReLU stats, bias/alpha column sums) are done in two
   deterministic stages through a caller-supplied scratch buffer of this size. */
size_t tdnnf_colreduce_workspace_bytes(int rows, int cols);

/* train-mode Propagate :421-452; memo_dev is the 5 x D Memo::mean_uvar_scale. */
int tdnnf_batchnorm_propagate(const tdnnf_mat *in, float epsilon, float target_rms, tdnnf_mat *out,
                              float *memo_dev, void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream);
/* train-mode Backprop :505-542 */
int tdnnf_batchnorm_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, float target_rms,
                             float *memo_dev, tdnnf_mat *in_deriv, void *workspace_dev, size_t workspace_bytes,
                             tdnnf_stream stream);
/* StoreStats :551-589; stats are double as in the reference (h:458-462); stats_dev = [count, sum[D], sumsq[D]] */
int tdnnf_batchnorm_store_stats(const float *memo_dev, int D, int num_frames, double *stats_dev,
                                tdnnf_stream stream);
/* ComputeDerived :682-715 -> scale_dev[D], offset_dev[D] */
int tdnnf_batchnorm_compute_derived(const double *stats_dev, int D, float epsilon, float target_rms,
                                    float *scale_dev, float *offset_dev, tdnnf_stream stream);
/* BatchNormTestComponent::Propagate :872-874 / Backprop :919-920 */
int tdnnf_batchnorm_test_propagate(const tdnnf_mat *in, const float *scale_dev, const float *offset_dev,
                                   tdnnf_mat *out, tdnnf_stream stream);
int tdnnf_batchnorm_test_backprop(const tdnnf_mat *out_deriv, const float *scale_dev, tdnnf_mat *in_deriv,
                                  tdnnf_stream stream);

/* ========================================================================== A5
 * DARTS mixing ops (src/nnet3/nnet-simple-component.cc) */

/* SoftmaxFlopsComponent::Propagate :9968-9981 (gumbel_u_dev NULL, temp 1) and
   GumbelSoftmax[Flops]Component::Propagate :10088-10113 / :9774-9799. */
int tdnnf_softmax_flops_propagate(const tdnnf_mat *in, const float *gumbel_u_dev, float temp_proportion,
                                  tdnnf_mat *out, tdnnf_stream stream);
/* Backprop :9984-10020 / :10116-10158: out_deriv[:, :dim] += scale/(rows*cols)*flops (IN PLACE, as the
   reference does), in_deriv = DiffSoftmax(out_value, out_deriv) / temp.  flops_dev NULL = no penalty. */
int tdnnf_softmax_flops_backprop(const tdnnf_mat *out_value, tdnnf_mat *out_deriv, float scale,
                                 const float *flops_dev, int dim, float temp_proportion,
                                 tdnnf_mat *in_deriv, tdnnf_stream stream);
/* OnehotFunctionComponent::Propagate :9504-9519 (one uniform draw, device) */
int tdnnf_onehot_propagate(const float *sample_u_dev, tdnnf_mat *out, tdnnf_stream stream);
/* OnehotFunctionComponent::Backprop :9521-9552, non-natural-gradient branch (the recipes' configuration
   "is-updatable=true use-natural-gradient=false"): output_ += lr * colsum(out_deriv); no input derivative.
   workspace: tdnnf_colreduce_workspace_bytes(rows, cols). */
int tdnnf_onehot_backprop(const tdnnf_mat *out_deriv, float lr, float *output_acc_dev, void *workspace_dev,
                          size_t workspace_bytes, tdnnf_stream stream);
/* CopyNComponent :4843-4867 (AddMatBlocks: both directions ADD) */
int tdnnf_copyn_propagate(const tdnnf_mat *in, float scale, tdnnf_mat *out, tdnnf_stream stream);
int tdnnf_copyn_backprop(const tdnnf_mat *out_deriv, float scale, tdnnf_mat *in_deriv, tdnnf_stream stream);
/* ConstantFunctionComponent :2602-2642 (non-NG branch: output += 5*lr*colsum) */
int tdnnf_constant_function_propagate(const float *output_dev, tdnnf_mat *out, tdnnf_stream stream);
int tdnnf_constant_function_backprop(const tdnnf_mat *out_deriv, float lr, float *output_acc_dev,
                                     void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream);
/* FlopsConstraintComponent::Backprop :9465-9478 */
int tdnnf_flops_constraint_backprop(const float *flops_dev, float scale, int rows_in, int cols_in,
                                    tdnnf_mat *in_deriv, tdnnf_stream stream);

/* ========================================================================== A6 */
/* ElementwiseProductComponent :256-299 */
int tdnnf_elementwise_product_propagate(const tdnnf_mat *in, int output_dim, tdnnf_mat *out, tdnnf_stream);
int tdnnf_elementwise_product_backprop(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int output_dim,
                                       tdnnf_mat *in_deriv, tdnnf_stream);
/* RectifiedLinearComponent :958-1091 */
int tdnnf_relu_propagate(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_stream);
int tdnnf_relu_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, tdnnf_mat *in_deriv, tdnnf_stream);
/* RepairGradients :990-1074 after the coin flip; stats_dev = [count, value_sum[D], deriv_sum[D]] doubles */
int tdnnf_relu_repair(const double *stats_dev, int dim, float self_repair_scale, float lower, float upper,
                      tdnnf_mat *in_deriv, tdnnf_stream);
int tdnnf_relu_store_stats(const tdnnf_mat *out_value, double *stats_dev, void *workspace_dev,
                           size_t workspace_bytes, tdnnf_stream);
/* AffineComponent :1235-1279 / LinearComponent :3211-3254 (bias_dev NULL).  Same MFMA kernels as Tdnn. */
int tdnnf_affine_propagate(const tdnnf_mat *in, const float *W_dev, int ldw, const float *bias_dev, int Do,
                           tdnnf_mat *out, tdnnf_stream);
int tdnnf_affine_backprop(const tdnnf_mat *out_deriv, const float *W_dev, int ldw, int Di, tdnnf_mat *in_deriv,
                          tdnnf_stream);
int tdnnf_affine_update_simple(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, float lr, float *W_acc_dev,
                               int ldw, float *bias_acc_dev, void *workspace_dev, size_t workspace_bytes,
                               tdnnf_stream);
/* NaturalGradientAffineComponent::Update :2980-3024 (bias_acc_dev != NULL) and the natural-gradient branch of
   LinearComponent::Backprop :3240-3243 (bias_acc_dev == NULL: no ones column): [X, 1] and a copy of out_deriv are
   preconditioned by the caller's two OnlineNaturalGradient objects, W += lr a b dY'^T X', bias += lr a b dY'^T 1'. */
size_t tdnnf_affine_update_natural_gradient_workspace_bytes(int Do, int Di, int num_rows, int has_bias);
int tdnnf_affine_update_natural_gradient(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv,
                                         tdnnf_ng *preconditioner_in, tdnnf_ng *preconditioner_out,
                                         float learning_rate, float *W_acc_dev, int ldw, float *bias_acc_dev,
                                         void *workspace_dev, size_t workspace_bytes, tdnnf_stream);
/* LogSoftmaxComponent :3607-3632 */
int tdnnf_log_softmax_propagate(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_stream);
int tdnnf_log_softmax_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, tdnnf_mat *in_deriv,
                               tdnnf_stream);
/* out = sa*a + sb*b  (descriptor Sum(Scale(..), ..) into NoOpComponent, composite_layers.py:205-213);
   b may be NULL (plain scaled copy).  In-place (out aliasing a or b) is allowed. */
int tdnnf_sum_scaled(const tdnnf_mat *a, float sa, const tdnnf_mat *b, float sb, tdnnf_mat *out, tdnnf_stream);
/* out += s * a */
int tdnnf_add_scaled(const tdnnf_mat *a, float s, tdnnf_mat *out, tdnnf_stream);
/* GeneralDropoutComponent (UPSTREAM), continuous mask shared over time: row r uses mask row r % num_seq */
int tdnnf_general_dropout(const tdnnf_mat *in, const float *mask_dev, int num_seq, tdnnf_mat *out, tdnnf_stream);
/* GeneralDropoutComponent::GetMemo (UPSTREAM; factory /root/reference/src/nnet3/nnet-component-itf.cc:194): the mask from n uniform
   draws -- continuous: 1 - 2p + 4p U (expected value 1); otherwise Heaviside(U - p) / (1 - p) */
int tdnnf_general_dropout_mask(const float *uniform_dev, long long n, float proportion, int continuous, float *mask_dev, tdnnf_stream);

/* ========================================================================== A7
 * chain::ComputeChainObjfAndDeriv (UPSTREAM; options from
 * local/chain_NAS/run_TDNN_DARTSV3_fbk_stride_pretrain.sh:185-195). */
typedef struct tdnnf_den_graph tdnnf_den_graph;       /* device-resident denominator HMM */
typedef struct tdnnf_supervision tdnnf_supervision;   /* device-resident numerator graphs of one minibatch */

/* host arrays in, device copy out.  initial_probs may be NULL (computed as 100-step average occupancy
   from start_state, DenominatorGraph::SetInitialProbs). */
int tdnnf_den_graph_create(int num_states, int num_arcs, int num_pdfs, const int *arc_src, const int *arc_dst,
                           const int *arc_pdf, const float *arc_prob, const float *initial_probs, int start_state,
                           tdnnf_den_graph **out);
void tdnnf_den_graph_destroy(tdnnf_den_graph *);
int tdnnf_supervision_create(int num_sequences, int frames_per_sequence, const int *seq_state_begin,
                             const int *seq_arc_begin, const int *state_time, const float *final_logprob,
                             const int *arc_src, const int *arc_dst, const int *arc_pdf, const float *arc_logprob,
                             float weight, tdnnf_supervision **out);
void tdnnf_supervision_destroy(tdnnf_supervision *);

/* The denominator recursion has two forms: one persistent workgroup per sequence with the state vectors in LDS (graphs up
   to ~10 000 states), and one launch per frame over all sequences on sequence-minor arrays (larger graphs).  0 = chosen by
   graph size (default), 1 / 2 force one (tests, experiments); 3 = the persistent form with ONE workgroup per sequence even for
   few sequences (otherwise up to 32 sequences take four workgroups each, which must be co-resident: checked against the
   occupancy calculator before the launch, a bounded poll behind it, and the one-workgroup kernels redo a minibatch whose
   multi-workgroup launch gave up -- reported on stderr once, not used again in the process).  Affects the workspace size:
   set it before tdnnf_chain_workspace_bytes / tdnnf_net_create. */
int tdnnf_chain_set_denominator_mode(int mode);
/* diagnostics: minibatches the one-workgroup kernels redid behind a multi-workgroup launch that gave up; whether that form is switched off for
   the process; reset != 0 clears both (synchronises the device) */
int tdnnf_chain_den_mw_status(int *fallbacks, int *disabled, int reset);
size_t tdnnf_chain_workspace_bytes(const tdnnf_den_graph *, int num_sequences, int frames_per_sequence);
/* results_dev (device doubles): [0] objf, [1] l2_term, [2] weight, [3] num logprob (weighted),
   [4] den logprob (weighted), [5] ok flag (1.0 / 0.0), [6] xent objf when xent_output given.
   nnet_output_deriv is OVERWRITTEN; xent_deriv (may be NULL) receives the numerator posteriors
   already multiplied by xent_regularize (NnetChainTrainer::ProcessOutputs). */
int tdnnf_chain_objf_and_deriv(const tdnnf_den_graph *, const tdnnf_supervision *, const tdnnf_mat *nnet_output,
                               const tdnnf_mat *xent_output /* may be NULL */, float leaky_hmm_coefficient,
                               float l2_regularize, float xent_regularize, double *results_dev,
                               tdnnf_mat *nnet_output_deriv, tdnnf_mat *xent_deriv, void *workspace_dev,
                               size_t workspace_bytes, tdnnf_stream);

/* ========================================================================== A8
 * OnlineNaturalGradient::PreconditionDirections (UPSTREAM; call sites
 * src/nnet3/nnet-tdnn-component.cc:598-599, nnet-simple-component.cc:3001-3002). */
int tdnnf_ng_create(int rank, int update_period, float num_samples_history, float alpha, tdnnf_ng **out);
void tdnnf_ng_destroy(tdnnf_ng *);
/* OnlineNaturalGradient::Freeze (UpdatableComponent::FreezeNaturalGradient, nnet-tdnn-component.cc:979-982): freeze != 0 keeps the
   current low-rank state (no refresh steps) */
int tdnnf_ng_freeze(tdnnf_ng *, int freeze);
/* X is modified in place; *scale_host (may be NULL) receives the scalar, which costs a stream synchronisation.
   The R x R eigen-problem of a refresh step is solved on a host worker thread and W_{t+1} is installed at the
   next call on the same object, so a call with scale_host == NULL never blocks.  rank <= 128. */
int tdnnf_ng_precondition(tdnnf_ng *, tdnnf_mat *X, float *scale_host, tdnnf_stream);
/* The N-sized statistics pass of one PreconditionDirections call by itself (what X W_t^T costs; tests, micro-benchmarks):
   H (N x rank, ld = H->stride) = X~ WT, row m of X~ = the concatenation over the taps of eff[i] * X[m * ix->row_stride + ix->row_offsets[i]]
   (Di columns each; eff_dev NULL = ones) [+ bias_dev: the row of W_t^T that meets the appended column of ones]; WT_dev is W_t^T, (taps * Di) x
   rank, k-major, followed by at least 64 finite rows; sumsq_dev (NULL or sumsq_cap doubles): their sum = ||X~||_F^2.  use_valu != 0: the
   form: 0 the MFMA rows GEMM, one tap after the other (needs W_dev = W_t, rank x ldw); 1 the vector-ALU kernel (csrc/ng_valu.hip; rank 20 / 40 / 80,
   TDNNF_EINVAL otherwise); 2 the one-pass form for taps that are row shifts of one matrix (csrc/ng.hip pform_pass: P = X [W_0^T | W_1^T ..], then
   H[m] = sum_i P[m + o_i][block i]; taps whole 128-row tiles apart, N % 128 == 0, N >= 32768, taps x rank <= 64, H dense; needs W_dev and
   workspace_dev of tdnnf_ng_stats_pass_workspace_bytes()). */
size_t tdnnf_ng_stats_pass_workspace_bytes(int rank, int Di, int num_taps, int N);
int tdnnf_ng_stats_pass(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *X, int Di, const float *eff_dev, const float *WT_dev,
                        const float *W_dev, int ldw, const float *bias_dev, tdnnf_mat *H, double *sumsq_dev, int sumsq_cap, int form,
                        void *workspace_dev, size_t workspace_bytes, tdnnf_stream);
/* device float holding the scale of the object's last tdnnf_ng_precondition call (NULL before the first call) */
const float *tdnnf_ng_scale_dev(const tdnnf_ng *);

/* ========================================================================== A9
 * src/nnet3/nnet-utils.cc */
/* ConstrainOrthonormalInternal :914-1032 on M (rows <= cols; pass the transpose otherwise, :1068-1075).
   workspace: (rows*rows + rows*cols + 8) floats. */
size_t tdnnf_constrain_orthonormal_workspace_bytes(int rows, int cols);
int tdnnf_constrain_orthonormal(float scale, float *M_dev, int rows, int cols, int ld, void *workspace_dev,
                                size_t workspace_bytes, tdnnf_stream);
/* ApplyL2Regularization :2223-2245 : delta += scale * params */
int tdnnf_axpy(const float *x_dev, float a, float *y_dev, size_t n, tdnnf_stream);
/* UpdatableComponent::DotProduct (nnet-tdnn-component.cc:949-958): <x, y> accumulated in double, result on the HOST -- synchronises the
   stream (model averaging / diagnostics, not the training step) */
int tdnnf_dot(const float *x_dev, const float *y_dev, size_t n, double *result_host, tdnnf_stream);
/* UpdateNnetWithMaxChange :2085-2175 on a flat parameter vector partitioned into num_comp components
   (comp_begin_host[num_comp+1] element offsets).  params += factor_i * delta_i with the per-component and
   global max-change factors computed ON DEVICE (no D2H of the dot products).  delta is zeroed afterwards
   when zero_delta != 0 (momentum 0).  info_dev (may be NULL): [num_comp] factors then [1] ok flag. */
size_t tdnnf_max_change_workspace_bytes(int num_comp);
int tdnnf_update_with_max_change(float *params_dev, float *delta_dev, int num_comp, const long long *comp_begin_host,
                                 const float *max_change_host, float max_param_change, float max_change_scale,
                                 float scale, int zero_delta, void *workspace_dev, size_t workspace_bytes,
                                 float *info_dev, tdnnf_stream);

/* ====================================================================== descriptors
 * Row plumbing the nnet3 compiler does with kCopyRows commands for the graphs of SURVEY.md 3.4. */
/* lda input Append(-1,0,1,ReplaceIndex(ivector,t,0)) (run_tdnn_fbk_40_iv_sp_7q.sh:164): out row (k,b) =
   [feats(k + j, b) for j in 0..num_splice-1, ivector(b)]; feats is t-major with num_seq sequences and
   out->rows/num_seq + num_splice - 1 time steps. */
int tdnnf_splice_input(const tdnnf_mat *feats, const tdnnf_mat *ivectors, int num_seq, int num_splice,
                       tdnnf_mat *out, tdnnf_stream);
/* Convert between plain t-major rows (row = tau*B + b) and the row order a Tdnn component with
   row_stride rho > 1 expects on its input (row = (tau/rho)*rho*B + b*rho + tau%rho,
   nnet-tdnn-component.cc:897-902).  to_rho != 0: t-major -> rho order; else the inverse. */
int tdnnf_reorder_rows(const tdnnf_mat *in, int num_seq, int rho, int to_rho, tdnnf_mat *out, tdnnf_stream);

/* ===================================================================== chain trainer
 * One minibatch of nnet3-chain-train (UPSTREAM NnetChainTrainer::TrainInternal; the shipped pieces are
 * ApplyL2Regularization, UpdateNnetWithMaxChange, ConstrainOrthonormal in src/nnet3/nnet-utils.cc) for the
 * TDNN-F graphs of local/chain_NAS/run_tdnn_fbk_40_iv_sp_7q.sh:160-186 / run_tdnn_7q_fbk_40_manual.sh:138-151:
 * lda -> tdnn1 -> num_layers x tdnnf-layer -> prefinal-l -> {prefinal-chain -> output, prefinal-xent -> output-xent}.
 * Parameters and raw gradients live in two caller-owned flat device buffers (so a data-parallel caller can
 * all-reduce the gradient buffer between tdnnf_net_forward_backward and tdnnf_net_update). */
#define TDNNF_NET_MAX_LAYERS 32
typedef struct {
  int feat_dim, ivector_dim, num_pdfs;       /* 40, 100, 6034 */
  int hidden_dim, prefinal_small_dim;        /* 1536, 256 */
  int num_layers;                            /* tdnnf layers (14 for 7q) */
  int bottleneck_dim[TDNNF_NET_MAX_LAYERS];  /* 160 */
  int time_stride[TDNNF_NET_MAX_LAYERS];     /* 1,1,1,0,3 x 10 */
  float bypass_scale;                        /* 0.66 */
  int frames_per_chunk, num_sequences;       /* T (multiple of frame_subsampling), B */
  int frame_subsampling;                     /* 3 */
  float leaky_hmm, xent_regularize, chain_l2_regularize; /* 0.1, 0.1, 0.0 */
  float l2_hidden, l2_output;                /* 0.01, 0.002 (per-component l2-regularize) */
  float max_change_hidden, max_change_output, max_param_change; /* 0.75, 1.5, 2.0 */
  float relu_self_repair_scale;              /* 1e-5 */
  float batchnorm_stats_scale;               /* 0.8 (ScaleBatchnormStats, UPSTREAM trainer default) */
  /* DARTS offset-search supernet (run_TDNN_DARTSV3_fbk_stride_pretrain.sh:143-156 + scripts/generate_config.py:8-43):
     darts_num_offsets = K >= 2 turns every tdnnf layer's .linear / .affine into TdnnDARTSV3Components with offsets
     -(K-1)..0 / 0..K-1 (time_stride is then ignored), bias forced on, K architecture logits in front of the bias.
     darts_flags: TDNNF_DARTS_*; the pretrain recipe is TDNNF_DARTS_UNIFORM_SAMPLE (:124). 0 = plain TDNN-F. */
  int darts_num_offsets;
  int darts_flags;
  float darts_temp_proportion;
  /* != 0: weight gradients go through OnlineNaturalGradient exactly as UpdateNaturalGradient does
     (nnet-tdnn-component.cc:592-624, nnet-simple-component.cc:2980-3024): spliced input [c_i X_i ..., 1] and the
     output derivative are preconditioned (rank 20 / 80, alpha 4, history 2000, update period 4), the product of the
     two scale factors multiplies the update.  0: raw gradients (is_gradient_ / UpdateSimple semantics). */
  int use_natural_gradient;
  /* Bottleneck-dimension search supernet (run_TDNNf_DARTS_mod_fbk_bottleneckCBshare_95onehottrain.sh:151-176 +
     scripts/generate_bottleneckCB8share_onehottrain_config.py:8-102; cv-update: scripts/add_flopsconstraint.py:18-30).
     bn_num_choices = C in 2..8 turns every tdnnf layer into: X.linear (-> sum of bn_choice_dims, must equal
     bottleneck_dim[]) split into C column blocks of widths bn_choice_dims[k]; block k is multiplied (ElementwiseProduct)
     by CopyN(Sum(p_k..p_{C-1})) where p (C values, the same for every row) comes from
       bn_mode 0: OnehotFunctionComponent "X.softmax" (one uniform draw per layer and minibatch; its C-vector
                  output_ is updated with lr * colsum(deriv) although nothing reads it, nnet-simple-component.cc:9521-9552)
       bn_mode 1: ConstantFunctionComponent "X.alpha" (C logits, update 5 * lr * colsum, :2636) -> SoftmaxFlopsComponent
       bn_mode 2: the same through GumbelSoftmaxFlopsComponent (C uniform draws, bn_temp_proportion)
     and X.affine reads the masked blocks.  The FLOPs penalty of modes 1/2 adds bn_flops_scale / (rows * C) * flops_k to
     every row of d p with flops_k = -(bn_choice_dims[0] + ... + bn_choice_dims[k]) (the reference hard-codes
     -25 ... -240, :10006-10017).  0 = off.  Not combinable with darts_num_offsets. */
  int bn_num_choices;
  int bn_choice_dims[8];
  int bn_mode;
  float bn_flops_scale;
  float bn_temp_proportion;
  /* != 0: cross-validation architecture update (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142,
     run_TDNNf_DARTS_mod_fbk_bottleneckCBshare_cvupdate_flopsconstraint.sh:136-139): every BatchNormComponent becomes a
     BatchNormTestComponent (scale / offset from the stored statistics: load them with tdnnf_net_set_stats, nothing is
     accumulated or rescaled), learning-rate factor 0 on every component (no model derivative is computed for them)
     except 1e-4 on the TdnnDARTSV3Components and 1 on the X.alpha vectors of the bottleneck supernet. */
  int cv_update;
  /* Arithmetic of the trainer's GEMMs.  0: exact f32 MFMA (v_mfma_f32_32x32x2_f32), the reference's BaseFloat.
     1: split-bf16: every f32 operand is a_hi + a_lo in bf16 and a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi on
     v_mfma_f32_32x32x16_bf16 with f32 accumulation (products to ~2^-16 relative; BASELINE configs[4] names
     "fp32 objf / bf16 MFMA GEMM").  The objective, BatchNorm, the optimizer step and all reductions stay f32 / f64.
     2: the same with three bf16 planes per operand and the six products a_i b_j, i + j <= 2 (24 mantissa bits per
     operand, products to ~2^-24 relative: f32-equivalent results from the bf16 matrix cores).  With option "planes" (default) and
     a minibatch large enough to run on one stream, operands are split ONCE into planes in HBM (tdnnf_planes_*) instead of in the kernels.
     3: "f16x3": every GEMM operand is split once into two f16 planes after scaling by the power of two its Frobenius norm allows
     (no element can overflow), a b ~ h h' + h l' + l h' on v_mfma_f32_32x32x16_f16 with f32 accumulation: 22-23 operand bits,
     f32-equivalent results in norm (tests/test_gpu_planes_gemm.py) at 3/16 of the f32 MFMA's cycles and 4 bytes per operand element;
     GEMMs the planes do not cover (tap coefficients, row strides, small minibatches) run exact f32. */
  int gemm_precision;
  /* Derived child networks (local/chain_NAS/scripts/generate_top_list.py:97-141, generate_optimal_stride.py): when
     use_layer_offsets != 0, layer l's X.linear has time-offsets {-offset_left[l], 0} and its X.affine {0, offset_right[l]}
     (a single tap where the offset is 0) instead of the symmetric time_stride[l]; any values in 0..64.  Ignored by the
     offset supernet (darts_num_offsets >= 2). */
  int use_layer_offsets;
  int offset_left[TDNNF_NET_MAX_LAYERS];
  int offset_right[TDNNF_NET_MAX_LAYERS];
  /* GeneralDropoutComponent of tdnn1 and of every tdnnf layer (UPSTREAM; emitted with "continuous=true" and no
     time-period by composite_layers.py:186-199: one mask row per sequence, shared over time, scale uniform on
     [1 - 2p, 1 + 2p]).  The recipes train with --trainer.dropout-schedule '0,0@0.20,0.5@0.50,0'
     (run_tdnn_fbk_40_iv_sp_7q.sh:48,216).  use_dropout != 0 reserves the masks and (num_layers + 1) * num_sequences *
     hidden_dim further random draws per step (after the others); the proportion p of the moment is set with
     tdnnf_net_set_dropout_proportion (0 = identity, the initial value).  Off in cv-update mode (test mode). */
  int use_dropout;
} tdnnf_net_config;
typedef struct tdnnf_net tdnnf_net;

/* ========================================================================== egs
 * Chain examples ("cegs" archives of NnetChainExample) in and out, and their merge into the buffers
 * tdnnf_net_forward_backward / tdnnf_supervision_create take.  Host code, no GPU needed.  The formats are upstream
 * Kaldi / OpenFst (nnet3/nnet-chain-example.cc, nnet-example.cc, chain/chain-supervision.cc, matrix/compressed-matrix.cc,
 * OpenFst compact-fst.h), restated; the reference only passes flags for them (run_TDNN_DARTSV3_fbk_stride_pretrain.sh:192-199,
 * steps/nnet3/chain/train.py:373-406).  Binary archives only. */
typedef struct tdnnf_egs_reader tdnnf_egs_reader;
typedef struct tdnnf_egs_writer tdnnf_egs_writer;
typedef struct tdnnf_eg tdnnf_eg;
int tdnnf_egs_reader_open(const char *path, tdnnf_egs_reader **out);
void tdnnf_egs_reader_close(tdnnf_egs_reader *);
/* *out = the next example, or NULL at the end of the archive (still TDNNF_OK) */
int tdnnf_egs_reader_next(tdnnf_egs_reader *, tdnnf_eg **out);
void tdnnf_eg_destroy(tdnnf_eg *);
const char *tdnnf_eg_key(const tdnnf_eg *);
/* input "input" / "ivector": dimensions and the time of its first row; copy of the (decompressed) rows x cols values */
int tdnnf_eg_input_info(const tdnnf_eg *, const char *name, int *rows, int *cols, int *first_t);
int tdnnf_eg_input_copy(const tdnnf_eg *, const char *name, float *out);
int tdnnf_eg_supervision_info(const tdnnf_eg *, float *weight, int *num_sequences, int *frames_per_seq, int *label_dim, int *num_states,
                              int *num_arcs, int *first_t, int *t_step);
/* nnet3-chain-merge-egs for n single-sequence examples (+ nnet3-chain-copy-egs --frame-shift): features t-major
   (row = (t - first_t) * n + b, t = first_t .. first_t + num_t - 1 as tdnnf_net_input_frames reports), ivectors n x dim
   (NULL if unused), and the arrays of tdnnf_supervision_create (sizes from tdnnf_egs_merge_sizes; labels are pdf-id + 1,
   weights costs: arc_pdf = label - 1, log-probs = -cost). */
int tdnnf_egs_merge_sizes(const tdnnf_eg *const *egs, int n, int *num_states, int *num_arcs, int *frames_per_seq, int *feat_dim, int *ivector_dim);
int tdnnf_egs_merge(const tdnnf_eg *const *egs, int n, int first_t, int num_t, int frame_shift, float *feats, float *ivectors, int *seq_state_begin,
                    int *seq_arc_begin, int *state_time, float *final_logprob, int *arc_src, int *arc_dst, int *arc_pdf, float *arc_logprob,
                    float *weight_out);
int tdnnf_egs_writer_open(const char *path, tdnnf_egs_writer **out);
int tdnnf_egs_writer_close(tdnnf_egs_writer *);
/* one single-sequence example; the supervision as ONE sequence's arrays of tdnnf_supervision_create (state 0 = start);
   compress != 0 writes the features as a 16-bit CompressedMatrix ("CM2"), as nnet3-chain-get-egs does by default */
int tdnnf_egs_writer_write(tdnnf_egs_writer *, const char *key, const float *feats, int rows, int feat_dim, int first_t, const float *ivector,
                           int ivector_dim, int compress, float weight, int frames, int t_step, int label_dim, int num_states, int num_arcs,
                           const float *final_logprob, const int *arc_src, const int *arc_dst, const int *arc_pdf, const float *arc_logprob);

int tdnnf_net_create(const tdnnf_net_config *cfg, tdnnf_net **out);
/* A second net for ANOTHER MINIBATCH SHAPE of the same model (cfg differing from the primary's only in frames_per_chunk /
   num_sequences): the recipes cut examples of several chunk widths (--egs.chunk-width 150,110,100) and merge each width
   into its own minibatches.  It shares the primary's natural-gradient preconditioners and model statistics (BatchNorm /
   ReLU); give it the primary's parameter and gradient buffers with tdnnf_net_set_buffers.  Use the nets one after the
   other on one stream; destroy the primary last. */
int tdnnf_net_create_shared(const tdnnf_net_config *cfg, const tdnnf_net *primary, tdnnf_net **out);
void tdnnf_net_destroy(tdnnf_net *);
long long tdnnf_net_num_params(const tdnnf_net *);
int tdnnf_net_num_components(const tdnnf_net *);
/* name_out: >= 64 bytes.  Weights are rows x cols at [begin, begin + rows*cols), bias (if has_bias) follows. */
int tdnnf_net_component_info(const tdnnf_net *, int index, char *name_out, long long *begin, int *rows, int *cols,
                             int *has_bias, float *lr_factor, float *l2, float *max_change, float *orthonormal);
/* DARTS components: number K of architecture logits stored between the weights and the bias
   (bias_params_[0:K] of the reference, nnet-tdnn-component.cc:172-176); 0 for every other component. */
int tdnnf_net_component_num_alpha(const tdnnf_net *, int index);
/* Uniform(0,1) draws consumed by one forward pass of a DARTS net: (K + 1) per DARTS component, in component
   order (K Gumbel uniforms, then the tap-sample uniform).  The caller fills a device buffer before every step. */
int tdnnf_net_num_random_draws(const tdnnf_net *);
int tdnnf_net_set_random_draws(tdnnf_net *, const float *draws_dev);
/* rows of the t-major feature matrix the net consumes: (frames_per_chunk + left + right context) * B */
int tdnnf_net_input_frames(const tdnnf_net *, int *num_t_in, int *first_t);
int tdnnf_net_set_buffers(tdnnf_net *, float *params_dev, float *grads_dev);
/* forward + chain objective + backward; ACCUMULATES raw gradients (is_gradient_ semantics) into grads.
   results_dev: as tdnnf_chain_objf_and_deriv.  step selects the pseudo-random decisions of this minibatch
   (ReLU stats / self-repair coin flips). */
int tdnnf_net_forward_backward(tdnnf_net *, const tdnnf_mat *feats, const tdnnf_mat *ivectors, const tdnnf_den_graph *,
                               const tdnnf_supervision *, double *results_dev, long long step, tdnnf_stream);
/* Data-parallel callers may overlap the gradient all-reduce with the backward pass: the flat gradient buffer is final
   bucket by bucket -- contiguous ranges, whole layers, >= 16 MB where the model allows, in the order backward finishes
   them (heads + prefinal-l first, then tdnnf layers from the top down, tdnn1 last; together they cover the buffer).
   tdnnf_net_forward_backward records an event per bucket once grads[begin, end) holds this minibatch's contribution;
   tdnnf_net_wait_grad_bucket makes `stream` wait for it (hipStreamWaitEvent), so a collective enqueued there afterwards
   runs beside the rest of the backward pass.  (forward_backward's own stream is complete, as before, when it returns.) */
int tdnnf_net_num_grad_buckets(const tdnnf_net *);
int tdnnf_net_grad_bucket(const tdnnf_net *, int index, long long *begin, long long *end);
int tdnnf_net_wait_grad_bucket(const tdnnf_net *, int index, tdnnf_stream stream);
/* delta = lr_c*(grad) - 2*l2_scale*lr_c*l2_c*params; max-change; params += delta; grads = 0; orthonormal
   constraint on the scheduled quarter of the constrained matrices; batchnorm stats *= batchnorm_stats_scale. */
int tdnnf_net_update(tdnnf_net *, float learning_rate, float l2_regularize_scale, long long step, tdnnf_stream);
/* The nnet edit "set-temperature-proportion name=* proportion=p" of the temperature schedule
   (steps/libs/nnet3/train/temperature_schedule.py:51-60, applied by train.py:527-531 before every iteration): sets the
   Temp-Proportion of every TdnnDARTSV3Component and GumbelSoftmax(Flops)Component of the net.  p > 0. */
/* "nnet3-copy --edits='set-dropout-proportion name=* proportion=p'" of train.py's dropout schedule */
int tdnnf_net_set_dropout_proportion(tdnnf_net *, float proportion);
int tdnnf_net_set_temperature_proportion(tdnnf_net *, float proportion);
/* ---- f32-equivalent GEMMs on the 16-bit matrix cores from pre-split operands (csrc/planes_gemm.hip).
   An operand is split ONCE into 16-bit planes laid out for the consumer ("P16": [K block of 16][plane][row][16], 32-byte row
   records, zero rows in front of and behind the matrix so that row-shifted tap views and tile overhang read zeros):
     num_planes 3 ("bf16x6", gemm_precision 2): x = p0 + p1 + p2, three bf16 planes, the six products p_i q_j with i + j <= 2;
     num_planes 2 ("f16x3",  gemm_precision 3): x s = h + l, two f16 planes of the operand scaled by the power of two s that its
       Frobenius norm allows (s ||X||_F <= 65504: no element can overflow, whatever the data), the three products h h', h l', l h';
       the split writes the record [s, 1 / s, ||X||_F] (room for 4 floats) to scale_dev and the GEMM multiplies its result by 1 / (s s').
   Both accumulate in f32.  tdnnf_planes_split writes the row-major planes (k = column; `planes`, rows_total = lead_rows + rows +
   zero tail rows) and / or the planes of the TRANSPOSE (k = row; `planes_t`, t_rows_total >= cols), for the products that reduce over
   the matrix's rows.  Sizes: tdnnf_planes_bytes(num_planes, rows_total, k_blocks) with k_blocks = ceil(cols / 16), resp.
   4 ceil(rows / 64) for the transposed planes.  workspace_dev: tdnnf_planes_split_workspace_bytes() (num_planes 2 only).
   tdnnf_planes_gemm: C (rows x cols) (op)= sum over segments s of  A[a_row[s] + m][a_first_col[s] .. + seg_cols[s]) .
   B[b_row[s] + n][b_first_col[s] .. + seg_cols[s])  -- segments = the taps of TdnnComponent::Propagate / Backprop
   (/root/reference/src/nnet3/nnet-tdnn-component.cc:302-324, :378-411): row-shifted views of one activation matrix against column
   blocks of the weight matrix (B: one row per output column, k contiguous).  a_row / b_row count rows of the plane buffers (lead rows
   included; b_row may be NULL = zeros); first columns must be multiples of 16; the A buffer needs zero rows up to the next multiple of
   256 output rows, the B buffer its rows padded to a multiple of 160 (cols nearer a multiple of 160), 256 or 128.
   init_mode 0: C += , 1: C = bias + , 2: C = . */
size_t tdnnf_planes_bytes(int num_planes, long long rows_total, long long k_blocks);
size_t tdnnf_planes_split_workspace_bytes(void);
/* how many GEMMs / weight gradients of the trainer ran on the plane kernels so far in this process (either pointer may be NULL) */
void tdnnf_planes_routed(long long *rows_gemms, long long *weight_gradients);
/* option "planes_check_bound": how many bound-derived scales were compared with the measured Frobenius norm, and how many bounds were too small */
void tdnnf_planes_bound_checks(long long *checks, long long *violations);
int tdnnf_planes_split(int num_planes, const tdnnf_mat *x, int lead_rows, long long rows_total, void *planes, long long t_rows_total, void *planes_t,
                       float *scale_dev, void *workspace_dev, tdnnf_stream);
int tdnnf_planes_gemm(int num_planes, const void *a_planes, long long a_rows_total, const float *a_scale_dev, const void *b_planes, long long b_rows_total,
                      const float *b_scale_dev, int num_segments, const long long *a_row, const long long *b_row, const int *a_first_col,
                      const int *b_first_col, const int *seg_cols, const float *bias, int init_mode, int relu, tdnnf_mat *c, tdnnf_stream);
/* The same product with the trainer's epilogue options: `add` (may be NULL): C[m] += add_scale * add[m - add_first_row] for the output rows the
   addend covers (the bypass term of the TDNN-F layers' Sum(Scale(0.66, .), .), fused into Backprop's data GEMM); `colstats` (may be NULL):
   column sums and sums of squares of the STORED output, one partial row per row tile of the launch -- colstats[t * cols + n] and
   colstats[(*colstats_rows + t) * cols + n], t < *colstats_rows (written by the call: the tile height depends on the shape; at most
   ceil(rows / 128) rows each) -- what BatchNormComponent::Propagate needs of its input (nnet-normalize-component.cc:433-445). */
int tdnnf_planes_gemm_epilogue(int num_planes, const void *a_planes, long long a_rows_total, const float *a_scale_dev, const void *b_planes, long long b_rows_total,
                               const float *b_scale_dev, int num_segments, const long long *a_row, const long long *b_row, const int *a_first_col,
                               const int *b_first_col, const int *seg_cols, const float *bias, int init_mode, int relu, const tdnnf_mat *add, float add_scale,
                               int add_first_row, float *colstats, int *colstats_rows, tdnnf_mat *c, tdnnf_stream stream);
/* Synchronised BatchNorm for a data-parallel caller ("BN [sum x, sum x^2] (2 D floats per BN) for exact single-GPU
   equivalence").  The reference's BatchNormComponent takes its statistics over ALL rows of the minibatch
   (/root/reference/src/nnet3/nnet-normalize-component.cc:433-445); when the minibatch is sharded over world_size ranks, every
   train-mode BatchNorm of the net then all-reduces its column sums -- forward [sum x, sum x^2], backward [sum z dz, sum dz, sum dz^2],
   as doubles -- through `allreduce(ctx, buf, count, stream)`: an in-place SUM over the ranks of `count` doubles at device pointer
   `buf`, enqueued on `stream` (the trainer's compute stream; must not synchronise it), returning 0.  Mean, scale and the backward
   terms are then formed with the global row count: the sharded step equals the unsharded one.  Every rank must run the same net on
   the same number of rows.  allreduce == NULL switches it off (the default: statistics per shard, as Kaldi's per-job statistics).
   The typedef is spelled in capitals on purpose (not an exported symbol). */
typedef int TDNNF_ALLREDUCE_FN(void *ctx, double *buf, long long count, tdnnf_stream stream);
int tdnnf_net_set_batchnorm_sync(tdnnf_net *, TDNNF_ALLREDUCE_FN *allreduce, void *ctx, int world_size);

/* ---- The data-parallel exchanges on RCCL, issued from C++ on the library's streams (csrc/rccl_sync.hip).
   RCCL is resolved on first use (no link-time dependency): the copy ALREADY MAPPED in the process if there is one (a launcher that
   called torch.distributed's "nccl" backend has torch/lib/librccl.so live: a second copy must not be loaded beside it), else the
   global symbol scope, else dlopen by name; tdnnf_rccl_library_path reports "<path> [<how>]".  One process per GPU: rank 0 takes a unique id
   (128 bytes), the launcher's own channel distributes it (e.g. a torch.distributed broadcast), every rank creates its communicator
   on its current device.
   tdnnf_net_set_batchnorm_sync_rccl: the synchronised BatchNorm above with ncclAllReduce (doubles, in place) on the compute stream
     as the collective -- nothing host-side between the two finalize launches; comm NULL switches it off.
   tdnnf_net_allreduce_grads_rccl: the gradient exchange of one minibatch, call right after tdnnf_net_forward_backward: one
     ncclAllReduce (sum, f32) per gradient bucket (tdnnf_net_grad_bucket) on comm_stream, each behind the event recorded when that
     bucket became final, so the upper layers' reductions run under the lower layers' backward pass; `stream` then waits for the last.
   The two run on DIFFERENT streams inside one step, so they must be given DIFFERENT communicators (RCCL serialises the operations of
   one communicator in issue order: the buckets would queue behind every BatchNorm collective of the backward pass). */
int tdnnf_rccl_available(void);
int tdnnf_rccl_library_path(char *out, int out_bytes);
int tdnnf_rccl_unique_id(void *out_128_bytes);
int tdnnf_rccl_comm_create(const void *id_128_bytes, int world_size, int rank, void **comm_out);
void tdnnf_rccl_comm_destroy(void *comm);
int tdnnf_rccl_allreduce_sum(void *comm, void *buf_dev, long long count, int is_double, tdnnf_stream stream);
int tdnnf_net_set_batchnorm_sync_rccl(tdnnf_net *, void *comm, int world_size);
int tdnnf_net_allreduce_grads_rccl(tdnnf_net *, void *comm, tdnnf_stream comm_stream, tdnnf_stream stream);
/* The nnet edit "set-learning-rate-factor name=<pattern> learning-rate-factor=f" (ReadEditConfig,
   /root/reference/src/nnet3/nnet-utils.cc:1232-1256): SetLearningRateFactor(f) on every UPDATABLE component whose name matches
   the pattern ('*' matches any run of characters, as NameMatchesPattern; the fixed lda layer is not updatable).  The cv-update
   recipes freeze the parent model with it (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:129).  *num_set (optional): how many
   components matched ("Set learning rate factors for N components").  A factor of 0 stops the component's model derivative
   from being formed at all, as UpdatableComponent's "learning_rate_ != 0" guards do. */
int tdnnf_net_set_learning_rate_factor(tdnnf_net *, const char *name_pattern, float factor, int *num_set);
/* Model state outside the parameter vector, as doubles in network order (tdnn1, tdnnf2.., prefinal-chain, prefinal-xent):
   per BatchNorm [count, stats_sum[D], stats_sumsq[D]] (BatchNormComponent::StoreStats, nnet-normalize-component.cc:551-589)
   and per ReLU [count, value_sum[D], deriv_sum[D], oderiv_count, oderiv_sumsq[D]] (NonlinearComponent::StoreStatsInternal and
   StoreBackpropStats, nnet-component-itf.cc:433-480: the latter on three minibatches in four, always on the first);
   block order per unit: batchnorm, relu (heads: batchnorm1, relu, batchnorm2).  Host buffers; both calls synchronise. */
long long tdnnf_net_stats_size(const tdnnf_net *);
int tdnnf_net_get_stats(const tdnnf_net *, double *stats_host, tdnnf_stream);
int tdnnf_net_set_stats(tdnnf_net *, const double *stats_host, tdnnf_stream);
/* nnet3 "raw" model files of the network (what nnet3-copy reads and writes): "<Nnet3>", the config lines of the graph,
   "<NumComponents>" and every component in the token order of the reference's Write() functions (csrc/model_io.hip
   cites each one); binary != 0 writes Kaldi's binary encoding ("\0B" header, "FM"/"FV" blobs).  Parameters come from /
   go to the buffers of tdnnf_net_set_buffers, BatchNorm / ReLU statistics from / to the net (as tdnnf_net_get_stats).
   learning_rate is what the "<LearningRate>" tokens record (times each component's factor).  read_model matches
   components by name, checks dimensions and fails if a parameterised component of the net is missing from the file;
   both text and binary files are accepted.  Both calls synchronise the stream. */
int tdnnf_net_write_model(const tdnnf_net *, const char *path, int binary, float learning_rate, tdnnf_stream);
int tdnnf_net_read_model(tdnnf_net *, const char *path, tdnnf_stream);
/* The trainer configuration for the graph a model file holds (no GPU needed): dimensions from the component blocks,
   time strides / DARTS taps and flags from the Tdnn components, bypass scale and input dimensions from the config
   lines, l2 / max-change / self-repair from the tokens, BatchNormTestComponent => cv_update, the bottleneck supernet's
   blocks from its CopyN components.  What a model file does not record (chunk size, minibatch size, leaky-hmm,
   max-param-change ...) is set to the recipe's values (run_tdnn_fbk_40_iv_sp_7q.sh:149-203).  Fails for graphs other
   than the ones the trainer runs. */
int tdnnf_net_config_from_model(const char *path, int frames_per_chunk, int num_sequences, tdnnf_net_config *out);
/* The graph of a configuration as nnet3 config lines (input-node / component-node / dim-range-node / output-node: the
   part of a final.config that Nnet::Write keeps in a model file), newline-terminated, as tdnnf_net_write_model writes them.
   Mirrors the node lines of steps/libs/nnet3/xconfig/composite_layers.py:135-215,1283-1331 and of
   local/chain_NAS/scripts/generate_bottleneckCB8share_onehottrain_config.py:10-120, add_flopsconstraint.py:18-30.
   No GPU needed.  *needed (optional) receives the size including the terminating 0; out may be null with capacity 0 to
   query it. */
int tdnnf_net_config_text(const tdnnf_net_config *cfg, char *out, size_t capacity, size_t *needed);
/* diagnostics (tdnnf_set_option("phase_events", 1)): the last step's time between phase boundaries on the caller's stream, in ms:
   [0] forward trunk, [1] heads forward + objective issue, [2] heads backward (incl. the wait for the denominator), [3] trunk backward,
   [4] join of the side streams, [5] between forward_backward and update (collectives), [6] update.  *count = 7, or 0 when the option was
   off.  Synchronises the last event. */
int tdnnf_net_phase_times(tdnnf_net *, double *ms_out, int capacity, int *count);
/* debugging / parity: on != 0 makes the following forward_backward calls keep copies of the backward pass's derivative
   matrices (which otherwise live in recycled scratch): "<layer>.noop.deriv" (w.r.t. the layer output), "<layer>.affine.deriv",
   "<layer>.linear.deriv" (w.r.t. those components' outputs), "prefinal-l.deriv", "prefinal-{chain,xent}.{batchnorm2,linear,
   batchnorm1,affine}.deriv", "output-xent.deriv"; served by tdnnf_net_get_activation.  Costs memory and copies: tests only. */
int tdnnf_net_set_capture(tdnnf_net *, int on);
/* debugging / parity: copy an internal activation by name ("tdnnf2.linear", "output", ...) into out */
int tdnnf_net_get_activation(const tdnnf_net *, const char *name, tdnnf_mat *out, tdnnf_stream);
int tdnnf_net_activation_dims(const tdnnf_net *, const char *name, int *rows, int *cols);

/* ======================================================================== profiling
 * Optional per-launch timing of the MFMA GEMM kernels with HIP events recorded on the launch stream
 * (bench.py's live roofline measurement).  Classes: 0 = rows_gemm 128x128 tile, 1 = rows_gemm 128x160 tile,
 * 2 = wgrad, 3 = the skinny GEMMs of the natural-gradient statistics.  tdnnf_profile_read synchronises the recorded events and returns totals since enable. */
int tdnnf_profile_enable(int on);
int tdnnf_profile_read(int kernel_class, double *launches, double *total_ms, double *total_flops);
/* Algorithmic HBM bytes of the class's launches since enable, computed from the launch shapes: every operand element
   touched once -- 4 (distinct input rows x K-length + weight block + output rows x N [x 2 when added to]) per GEMM
   (e (N_in Di + K Di Do + N Do)); measured PMC traffic divided by this is the wasted-re-read ratio. */
int tdnnf_profile_read_bytes(int kernel_class, double *algorithmic_bytes);
const char *tdnnf_profile_class_name(int kernel_class);

#ifdef __cplusplus
}
#endif
#endif /* TDNNF_HIP_H_ */
