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

// Backprop :335-431 (data part) and UpdateSimple :433-455 into `to_update`'s accumulators.
template <class CuMat>
inline void TdnnDartsBackprop(const TdnnDartsState &c, const tdnnf_tdnn_indexes &ix, const CuMat &in_value,
                              const CuMat &out_deriv, const float *memo_dev, CuMat *in_deriv /* may be null */,
                              float learning_rate, float *to_update_linear /* null: no update */, float *to_update_bias,
                              void *workspace_dev, size_t workspace_bytes, tdnnf_stream stream) {
  tdnnf_mat vdy = View(out_deriv), vx = View(in_value);
  if (in_deriv) {
    tdnnf_mat vdx = View(*in_deriv);
    Check(tdnnf_tdnn_backprop_data(&ix, &vdy, c.linear_params, c.ldw, c.Do, c.Di, memo_dev + c.K, &vdx, stream));
  }
  if (to_update_linear && learning_rate != 0.0f)  // :423-427
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

}  // namespace tdnnf_adapter
#endif  // TDNNF_NNET3_ADAPTER_H_
