// tdnnf_nnet3_adapter.h -- the Kaldi-side binding a maintainer adds to kaldi/src/nnet3 to route the
// hot-path components of skhu101/TDNN-F_NAS through libtdnnf_hip.so (C-ABI: tdnnf_hip.h).
//
// It is written against the only three things it needs from Kaldi's CuMatrixBase<float> -- Data(),
// NumRows()/NumCols(), Stride() (usage in the reference: src/nnet3/nnet-tdnn-component.cc:815-819) -- so it
// compiles stand-alone (tests/test_adapter_compile.py builds it against a 10-line stub) and inside a Kaldi
// tree unchanged.  Each function body is what replaces the body of the reference method cited above it;
// the surrounding class (config parsing, Read/Write, parameter storage) stays Kaldi's.
//
// Error behaviour: the C-ABI never throws; Check() turns a non-zero status into the exception/abort the
// reference uses (KALDI_ERR throws std::runtime_error; define TDNNF_ADAPTER_FAIL to KALDI_ERR inside Kaldi).
#ifndef TDNNF_NNET3_ADAPTER_H_
#define TDNNF_NNET3_ADAPTER_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "tdnnf_hip.h"

#ifndef TDNNF_ADAPTER_FAIL
#define TDNNF_ADAPTER_FAIL(msg) throw std::runtime_error(msg)
#endif

namespace tdnnf_adapter {

inline void Check(int status) {
  if (status != TDNNF_OK) TDNNF_ADAPTER_FAIL(std::string("tdnnf: ") + tdnnf_last_error());
}

// CuMatrixBase<float> (or CuSubMatrix) -> tdnnf_mat.  Data() is the device pointer when Kaldi runs with a GPU.
template <class CuMat>
inline tdnnf_mat View(const CuMat &m) {
  tdnnf_mat v;
  v.data = const_cast<float *>(m.Data());
  v.rows = m.NumRows();
  v.cols = m.NumCols();
  v.stride = m.Stride();
  return v;
}

// TdnnDARTSV3Component::PrecomputedIndexes (nnet-convolutional-component.h:208-218) -> tdnnf_tdnn_indexes
inline tdnnf_tdnn_indexes Indexes(int row_stride, const std::vector<int> &row_offsets) {
  if (row_offsets.empty() || row_offsets.size() > TDNNF_MAX_OFFSETS) TDNNF_ADAPTER_FAIL("tdnnf: bad number of time offsets");
  tdnnf_tdnn_indexes ix;
  ix.row_stride = row_stride;
  ix.num_offsets = static_cast<int>(row_offsets.size());
  for (int i = 0; i < ix.num_offsets; i++) ix.row_offsets[i] = row_offsets[i];
  return ix;
}

// ---------------------------------------------------------------------------------------------------
// TdnnDARTSV3Component (src/nnet3/nnet-tdnn-component.cc).  State the Kaldi class already owns:
//   linear_params_ (Do x K*Di), bias_params_ (K + Do: first K = log-alpha), time_offsets_, the mode flags.
// The memo is a device buffer of 2K floats [coef | effective coef] instead of a heap CuVector (:330-332).
struct TdnnDartsState {
  int K, Di, Do, ldw;
  const float *linear_params;   // device
  const float *bias_params;     // device, K + Do (or null when use-bias=false)
  int flags;                    // TDNNF_DARTS_* from use_gumbel_/free_select_/uniform_sample_/use_entropy_/update_alpha_
  float temp_proportion;
  int share_index;              // (time_offsets_[1] > 0 ? 0 : K-1), :232-240
  bool offsets1_positive;       // time_offsets_[1] > 0
};

// Propagate :214-333.  uniform_draws_dev: K Gumbel uniforms then 1 sample uniform (the reference draws them
// with SetRandUniform(); pass a freshly filled CuVector<float>(K+1)).  memo_dev: 2K floats.
template <class CuMat>
inline void TdnnDartsPropagate(const TdnnDartsState &c, const tdnnf_tdnn_indexes &ix, const CuMat &in, CuMat *out,
                               const float *uniform_draws_dev, float *memo_dev, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_tdnn_darts_coef(c.bias_params, c.K, c.flags, c.temp_proportion, uniform_draws_dev, uniform_draws_dev + c.K,
                              c.share_index, memo_dev, memo_dev + c.K, stream));
  // :230-241: bias rows only when offsets[1] > 0, zero otherwise (quirk q1); no bias at all -> kPropagateAdds
  const int init_mode = c.bias_params == nullptr ? 0 : (c.offsets1_positive ? 1 : 2);
  const float *bias = c.bias_params ? c.bias_params + c.K : nullptr;
  Check(tdnnf_tdnn_propagate(&ix, &vin, c.linear_params, c.ldw, c.Do, c.Di, bias, memo_dev + c.K, init_mode, &vout, stream));
}

// The component's two OnlineNaturalGradient objects (preconditioner_in_ / preconditioner_out_ of the Kaldi class become
// two tdnnf_ng handles created once with the component's rank / update-period / history / alpha, nnet-tdnn-component.cc:183-210).
struct NaturalGradient {
  tdnnf_ng *in;
  tdnnf_ng *out;
};

// Backprop :335-431: data part :366-416, then (:418-430) "if (to_update->is_gradient_ || !to_update->use_natural_gradient_)
// UpdateSimple(...) else UpdateNaturalGradient(...)".  ng == nullptr selects UpdateSimple (:433-455); otherwise
// UpdateNaturalGradient :457-626 with the component's preconditioners.  to_update_bias is to_update->bias_params_ (K + Do).
// workspace: tdnnf_tdnn_update_workspace_bytes (simple) / tdnnf_tdnn_update_natural_gradient_workspace_bytes (natural gradient).
template <class CuMat>
inline void TdnnDartsBackprop(const TdnnDartsState &c, const tdnnf_tdnn_indexes &ix, const CuMat &in_value,
                              const CuMat &out_deriv, const float *memo_dev, CuMat *in_deriv /* may be null */,
                              float learning_rate, float *to_update_linear /* null: no update */, float *to_update_bias,
                              void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream, const NaturalGradient *ng = nullptr) {
  tdnnf_mat vdy = View(out_deriv), vx = View(in_value);
  if (in_deriv) {
    tdnnf_mat vdx = View(*in_deriv);
    Check(tdnnf_tdnn_backprop_data(&ix, &vdy, c.linear_params, c.ldw, c.Do, c.Di, memo_dev + c.K, &vdx, stream));
  }
  if (!to_update_linear || learning_rate == 0.0f) return;  // :418-424
  if (ng)
    Check(tdnnf_tdnn_update_natural_gradient(&ix, &vx, &vdy, c.Do, c.Di, c.linear_params, c.ldw, memo_dev, memo_dev + c.K, c.flags,
                                             c.share_index, c.temp_proportion, ng->in, ng->out, learning_rate, to_update_linear, c.ldw,
                                             to_update_bias ? to_update_bias + c.K : nullptr, to_update_bias, workspace_dev,
                                             workspace_bytes, stream));
  else
    Check(tdnnf_tdnn_update_simple(&ix, &vx, &vdy, c.Do, c.Di, memo_dev + c.K, learning_rate, to_update_linear, c.ldw,
                                   to_update_bias ? to_update_bias + c.K : nullptr, workspace_dev, workspace_bytes, stream));
}

// Plain TdnnComponent (UPSTREAM): all coefficients one, bias always added.
template <class CuMat>
inline void TdnnPropagate(const tdnnf_tdnn_indexes &ix, const CuMat &in, const float *linear_params, int ldw, int Do, int Di,
                          const float *bias /* Do or null */, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_tdnn_propagate(&ix, &vin, linear_params, ldw, Do, Di, bias, nullptr, bias ? 1 : 0, &vout, stream));
}
// TdnnComponent::Backprop (UPSTREAM): in_deriv += out_deriv W_i on the tap views, then UpdateSimple or UpdateNaturalGradient
template <class CuMat>
inline void TdnnBackprop(const tdnnf_tdnn_indexes &ix, const CuMat &in_value, const CuMat &out_deriv, const float *linear_params, int ldw,
                         int Do, int Di, CuMat *in_deriv /* may be null */, float learning_rate, float *to_update_linear /* null: no update */,
                         float *to_update_bias /* Do or null */, void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream,
                         const NaturalGradient *ng = nullptr) {
  tdnnf_mat vdy = View(out_deriv), vx = View(in_value);
  if (in_deriv) {
    tdnnf_mat vdx = View(*in_deriv);
    Check(tdnnf_tdnn_backprop_data(&ix, &vdy, linear_params, ldw, Do, Di, nullptr, &vdx, stream));
  }
  if (!to_update_linear || learning_rate == 0.0f) return;
  if (ng)
    Check(tdnnf_tdnn_update_natural_gradient(&ix, &vx, &vdy, Do, Di, nullptr, 0, nullptr, nullptr, 0, 0, 1.0f, ng->in, ng->out, learning_rate,
                                             to_update_linear, ldw, to_update_bias, nullptr, workspace_dev, workspace_bytes, stream));
  else
    Check(tdnnf_tdnn_update_simple(&ix, &vx, &vdy, Do, Di, nullptr, learning_rate, to_update_linear, ldw, to_update_bias, workspace_dev,
                                   workspace_bytes, stream));
}

// ---------------------------------------------------------------------------------------------------
// BatchNormComponent (src/nnet3/nnet-normalize-component.cc:401-589).  memo_dev = Memo::mean_uvar_scale (5 x D).
template <class CuMat>
inline void BatchNormPropagate(const CuMat &in, float epsilon, float target_rms, CuMat *out, float *memo_dev, void *ws,
                               size_t ws_bytes, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_batchnorm_propagate(&vin, epsilon, target_rms, &vout, memo_dev, ws, ws_bytes, stream));
}
template <class CuMat>
inline void BatchNormBackprop(const CuMat &out_value, const CuMat &out_deriv, float target_rms, float *memo_dev,
                              CuMat *in_deriv, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  tdnnf_mat vz = View(out_value), vdz = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_batchnorm_backprop(&vz, &vdz, target_rms, memo_dev, &vdx, ws, ws_bytes, stream));
}
// BatchNormTestComponent :843-922 (scale_/offset_ from ComputeDerived :682-715)
template <class CuMat>
inline void BatchNormTestPropagate(const CuMat &in, const float *scale_dev, const float *offset_dev, CuMat *out,
                                   tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_batchnorm_test_propagate(&vin, scale_dev, offset_dev, &vout, stream));
}
template <class CuMat>
inline void BatchNormTestBackprop(const CuMat &out_deriv, const float *scale_dev, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vdz = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_batchnorm_test_backprop(&vdz, scale_dev, &vdx, stream));
}

// ---------------------------------------------------------------------------------------------------
// (Gumbel)SoftmaxFlopsComponent (src/nnet3/nnet-simple-component.cc:9968-10020, :10088-10158)
template <class CuMat>
inline void SoftmaxFlopsPropagate(const CuMat &in, const float *gumbel_uniform_dev /* null: plain softmax */,
                                  float temp_proportion, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_softmax_flops_propagate(&vin, gumbel_uniform_dev, temp_proportion, &vout, stream));
}
template <class CuMat>
inline void SoftmaxFlopsBackprop(const CuMat &out_value, CuMat *out_deriv /* mutated, as in the reference */, float scale,
                                 const float *flops_dev, int dim, float temp_proportion, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vp = View(out_value), vdp = View(*out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_softmax_flops_backprop(&vp, &vdp, scale, flops_dev, dim, temp_proportion, &vdx, stream));
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm statistics: StoreStats nnet-normalize-component.cc:551-589 (stats_dev = [count, sum[D], sumsq[D]] doubles),
// BatchNormTestComponent::ComputeDerived :682-715
inline void BatchNormStoreStats(const float *memo_dev, int dim, int num_frames, double *stats_dev, tdnnf_stream stream) {
  Check(tdnnf_batchnorm_store_stats(memo_dev, dim, num_frames, stats_dev, stream));
}
inline void BatchNormComputeDerived(const double *stats_dev, int dim, float epsilon, float target_rms, float *scale_dev, float *offset_dev,
                                    tdnnf_stream stream) {
  Check(tdnnf_batchnorm_compute_derived(stats_dev, dim, epsilon, target_rms, scale_dev, offset_dev, stream));
}

// ---------------------------------------------------------------------------------------------------
// OnehotFunctionComponent (nnet-simple-component.cc:9504-9552): Propagate draws ONE uniform and replicates the one-hot row;
// Backprop has no input derivative and (is-updatable=true use-natural-gradient=false) adds lr * colsum(out_deriv) to output_.
template <class CuMat>
inline void OnehotPropagate(const float *uniform_draw_dev, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vout = View(*out);
  Check(tdnnf_onehot_propagate(uniform_draw_dev, &vout, stream));
}
template <class CuMat>
inline void OnehotBackprop(const CuMat &out_deriv, float learning_rate, float *to_update_output, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  tdnnf_mat vd = View(out_deriv);
  Check(tdnnf_onehot_backprop(&vd, learning_rate, to_update_output, ws, ws_bytes, stream));
}
// CopyNComponent :4843-4867 (both directions add)
template <class CuMat>
inline void CopyNPropagate(const CuMat &in, float scale, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_copyn_propagate(&vin, scale, &vout, stream));
}
template <class CuMat>
inline void CopyNBackprop(const CuMat &out_deriv, float scale, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vd = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_copyn_backprop(&vd, scale, &vdx, stream));
}
// ConstantFunctionComponent :2602-2642 (the NAS-modified non-natural-gradient branch: output_ += 5 lr colsum(out_deriv))
template <class CuMat>
inline void ConstantFunctionPropagate(const float *output_dev, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vout = View(*out);
  Check(tdnnf_constant_function_propagate(output_dev, &vout, stream));
}
template <class CuMat>
inline void ConstantFunctionBackprop(const CuMat &out_deriv, float learning_rate, float *to_update_output, void *ws, size_t ws_bytes,
                                     tdnnf_stream stream) {
  tdnnf_mat vd = View(out_deriv);
  Check(tdnnf_constant_function_backprop(&vd, learning_rate, to_update_output, ws, ws_bytes, stream));
}
// ElementwiseProductComponent :256-299
template <class CuMat>
inline void ElementwiseProductPropagate(const CuMat &in, int output_dim, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_elementwise_product_propagate(&vin, output_dim, &vout, stream));
}
template <class CuMat>
inline void ElementwiseProductBackprop(const CuMat &in_value, const CuMat &out_deriv, int output_dim, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vx = View(in_value), vd = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_elementwise_product_backprop(&vx, &vd, output_dim, &vdx, stream));
}

// ---------------------------------------------------------------------------------------------------
// RectifiedLinearComponent :958-1091.  stats_dev = [count, value_sum[D], deriv_sum[D]] doubles (NonlinearComponent's
// value_sum_ / deriv_sum_ / count_); the caller keeps the reference's coin flips (StoreStats w.p. 1/2 :1084, RepairGradients
// w.p. self_repair probability :1017) and calls Repair / StoreStats accordingly.
template <class CuMat>
inline void ReluPropagate(const CuMat &in, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_relu_propagate(&vin, &vout, stream));
}
template <class CuMat>
inline void ReluBackprop(const CuMat &out_value, const CuMat &out_deriv, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vy = View(out_value), vd = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_relu_backprop(&vy, &vd, &vdx, stream));
}
template <class CuMat>
inline void ReluRepairGradients(const double *stats_dev, int dim, float self_repair_scale, float lower, float upper, CuMat *in_deriv,
                                tdnnf_stream stream) {
  tdnnf_mat vdx = View(*in_deriv);
  Check(tdnnf_relu_repair(stats_dev, dim, self_repair_scale, lower, upper, &vdx, stream));
}
template <class CuMat>
inline void ReluStoreStats(const CuMat &out_value, double *stats_dev, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  tdnnf_mat vy = View(out_value);
  Check(tdnnf_relu_store_stats(&vy, stats_dev, ws, ws_bytes, stream));
}

// ---------------------------------------------------------------------------------------------------
// AffineComponent :1235-1279 / NaturalGradientAffineComponent::Update :2980-3024 / LinearComponent :3211-3254 (bias == null)
template <class CuMat>
inline void AffinePropagate(const CuMat &in, const float *linear_params, int ldw, const float *bias /* null: LinearComponent */, int output_dim,
                            CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_affine_propagate(&vin, linear_params, ldw, bias, output_dim, &vout, stream));
}
// Backprop :1253-1279: in_deriv += out_deriv W (kBackpropAdds; may be null), then the update of `to_update` -- UpdateSimple :1246-1251
// when ng == nullptr (is_gradient_ or use-natural-gradient=false), else the natural-gradient Update.
// workspace: tdnnf_tdnn_update_workspace_bytes(Do, Di, 1, rows) / tdnnf_affine_update_natural_gradient_workspace_bytes.
template <class CuMat>
inline void AffineBackprop(const CuMat &in_value, const CuMat &out_deriv, const float *linear_params, int ldw, CuMat *in_deriv,
                           float learning_rate, float *to_update_linear /* null: no update */, float *to_update_bias /* null: Linear */,
                           void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream, const NaturalGradient *ng = nullptr) {
  tdnnf_mat vx = View(in_value), vdy = View(out_deriv);
  if (in_deriv) {  // "add with coefficient 1.0 since property kBackpropAdds is true" (:1264-1269): the one-tap gather form adds
    tdnnf_mat vdx = View(*in_deriv);
    tdnnf_tdnn_indexes ix;
    ix.row_stride = 1;
    ix.num_offsets = 1;
    ix.row_offsets[0] = 0;
    Check(tdnnf_tdnn_backprop_data(&ix, &vdy, linear_params, ldw, vdy.cols, vx.cols, nullptr, &vdx, stream));
  }
  if (!to_update_linear || learning_rate == 0.0f) return;
  if (ng)
    Check(tdnnf_affine_update_natural_gradient(&vx, &vdy, ng->in, ng->out, learning_rate, to_update_linear, ldw, to_update_bias, workspace_dev,
                                               workspace_bytes, stream));
  else
    Check(tdnnf_affine_update_simple(&vx, &vdy, learning_rate, to_update_linear, ldw, to_update_bias, workspace_dev, workspace_bytes, stream));
}

// FixedAffineComponent :3378-3399: out = bias + in W^T; in_deriv += out_deriv W (kBackpropAdds), nothing to update
template <class CuMat>
inline void FixedAffineBackprop(const CuMat &out_deriv, const float *linear_params, int ldw, int input_dim, CuMat *in_deriv, tdnnf_stream stream) {
  if (!in_deriv) return;
  tdnnf_mat vdy = View(out_deriv), vdx = View(*in_deriv);
  tdnnf_tdnn_indexes ix;
  ix.row_stride = 1;
  ix.num_offsets = 1;
  ix.row_offsets[0] = 0;
  Check(tdnnf_tdnn_backprop_data(&ix, &vdy, linear_params, ldw, vdy.cols, input_dim, nullptr, &vdx, stream));
}

// NoOpComponent :437-456: out = in; in_deriv = backprop_scale * out_deriv (in place allowed)
template <class CuMat>
inline void NoOpPropagate(const CuMat &in, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  if (vin.data != vout.data) Check(tdnnf_sum_scaled(&vin, 1.0f, nullptr, 0.f, &vout, stream));
}
template <class CuMat>
inline void NoOpBackprop(const CuMat &out_deriv, float backprop_scale, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vd = View(out_deriv), vdx = View(*in_deriv);
  if (vd.data != vdx.data || backprop_scale != 1.0f) Check(tdnnf_sum_scaled(&vd, backprop_scale, nullptr, 0.f, &vdx, stream));
}

// GeneralDropoutComponent (UPSTREAM; factory nnet-component-itf.cc:194, edits nnet-utils.cc:1315): mask_dev = num_seq x dim
// from tdnnf_general_dropout_mask, shared over time (row r uses mask row r % num_seq); Backprop applies the same mask
template <class CuMat>
inline void GeneralDropoutApply(const CuMat &in, const float *mask_dev, int num_seq, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_general_dropout(&vin, mask_dev, num_seq, &vout, stream));
}

// FlopsConstraintComponent :9454-9478: Propagate copies; Backprop: in_deriv = rows of flops * scale / (rows * cols of in_value)
template <class CuMat>
inline void FlopsConstraintBackprop(const float *flops_dev, float scale, const CuMat &in_value, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vdx = View(*in_deriv);
  Check(tdnnf_flops_constraint_backprop(flops_dev, scale, in_value.NumRows(), in_value.NumCols(), &vdx, stream));
}

// LogSoftmaxComponent :3607-3632
template <class CuMat>
inline void LogSoftmaxPropagate(const CuMat &in, CuMat *out, tdnnf_stream stream) {
  tdnnf_mat vin = View(in), vout = View(*out);
  Check(tdnnf_log_softmax_propagate(&vin, &vout, stream));
}
template <class CuMat>
inline void LogSoftmaxBackprop(const CuMat &out_value, const CuMat &out_deriv, CuMat *in_deriv, tdnnf_stream stream) {
  tdnnf_mat vy = View(out_value), vd = View(out_deriv), vdx = View(*in_deriv);
  Check(tdnnf_log_softmax_backprop(&vy, &vd, &vdx, stream));
}

// ---------------------------------------------------------------------------------------------------
// chain::ComputeChainObjfAndDeriv (UPSTREAM; what NnetChainTrainer::ProcessOutputs calls).  results_dev: 8 device doubles
// (tdnnf_hip.h); xent_output / xent_deriv may be null.  workspace: tdnnf_chain_workspace_bytes(graph, num_sequences, frames).
template <class CuMat>
inline void ChainObjfAndDeriv(const tdnnf_den_graph *den_graph, const tdnnf_supervision *supervision, const CuMat &nnet_output,
                              const CuMat *xent_output, float leaky_hmm_coefficient, float l2_regularize, float xent_regularize,
                              double *results_dev, CuMat *nnet_output_deriv, CuMat *xent_deriv, void *workspace_dev, size_t workspace_bytes,
                              tdnnf_stream stream) {
  tdnnf_mat vy = View(nnet_output), vd = View(*nnet_output_deriv), vx, vdx;
  if (xent_output) vx = View(*xent_output);
  if (xent_deriv) vdx = View(*xent_deriv);
  Check(tdnnf_chain_objf_and_deriv(den_graph, supervision, &vy, xent_output ? &vx : nullptr, leaky_hmm_coefficient, l2_regularize,
                                   xent_regularize, results_dev, &vd, xent_deriv ? &vdx : nullptr, workspace_dev, workspace_bytes, stream));
}

// ConstrainOrthonormalInternal nnet-utils.cc:914-1032 on a CuMatrixBase with rows <= cols (pass the transpose otherwise, :1068-1075)
template <class CuMat>
inline void ConstrainOrthonormal(float scale, CuMat *M, void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream) {
  tdnnf_mat vm = View(*M);
  Check(tdnnf_constrain_orthonormal(scale, vm.data, vm.rows, vm.cols, vm.stride, workspace_dev, workspace_bytes, stream));
}

}  // namespace tdnnf_adapter
#endif  // TDNNF_NNET3_ADAPTER_H_
