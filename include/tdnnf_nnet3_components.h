// tdnnf_nnet3_components.h -- the hot-path nnet3 components of skhu101/TDNN-F_NAS as Component CLASSES over libtdnnf_hip.so,
// registered under the factory names of the reference (Component::NewComponentOfType,
// /root/reference/src/nnet3/nnet-component-itf.cc:120-281).
//
// tdnnf_nnet3_adapter.h holds the edited method BODIES as free functions; this header wraps them in classes with the virtual
// interface every shipped subclass shows (e.g. src/nnet3/nnet-convolutional-component.h:121-191): Type(), InputDim(), OutputDim(),
// Properties() (the reference's own flag expressions), Propagate() returning the memo, Backprop() with `to_update`, DeleteMemo(),
// StoreStats().  What stays Kaldi's inside a Kaldi tree -- config parsing, Read/Write, parameter storage, the nnet3 compiler's
// index bookkeeping -- is reduced here to plain setters, so that the file compiles and runs WITHOUT Kaldi (tests/
// test_adapter_compile.py builds it with g++ against the three-accessor matrix stub; tests/adapter_driver.cc runs every class on
// the GPU against the free functions).  Inside Kaldi: derive from kaldi::nnet3::Component instead of tdnnf_nnet3::Component
// (same signatures), keep the members in CuMatrix / CuVector and hand their Data() to the setters.
//
// Device memory the classes need beyond the caller's matrices (memos, workspaces, random draws) comes from two hooks the host
// program installs once: DeviceHooks::alloc / free (Kaldi: CuAllocator, nnet-utils.cc:1086 g_cuda_allocator) and
// DeviceHooks::fill_uniform (Kaldi: CuRand<BaseFloat>::RandUniform, as nnet-tdnn-component.cc:258-259).  No call synchronises.
#ifndef TDNNF_NNET3_COMPONENTS_H_
#define TDNNF_NNET3_COMPONENTS_H_

#include <algorithm>
#include <string>
#include <vector>

#include "tdnnf_nnet3_adapter.h"

namespace tdnnf_nnet3 {

typedef float BaseFloat;
typedef int int32;

// Properties() bits (UPSTREAM nnet-component-itf.h; uses: nnet-convolutional-component.h:130-134, nnet-normalize-component.h:182-190,
// :359-366, nnet-simple-component.h:2124-2126, :2750-2755, :2994-2997)
enum ComponentProperties {
  kSimpleComponent = 0x001, kUpdatableComponent = 0x002, kPropagateInPlace = 0x004, kPropagateAdds = 0x008, kReordersIndexes = 0x010,
  kBackpropAdds = 0x020, kBackpropNeedsInput = 0x040, kBackpropNeedsOutput = 0x080, kBackpropInPlace = 0x100, kStoresStats = 0x200,
  kInputContiguous = 0x400, kOutputContiguous = 0x800, kUsesMemo = 0x1000, kRandomComponent = 0x2000
};

// the part of CuMatrixBase<BaseFloat> the components touch (usage: nnet-tdnn-component.cc:815-819)
class CuMatrixBase {
 public:
  CuMatrixBase(float *data, int32 rows, int32 cols, int32 stride) : data_(data), rows_(rows), cols_(cols), stride_(stride) {}
  const float *Data() const { return data_; }
  float *Data() { return data_; }
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  int32 Stride() const { return stride_; }

 private:
  float *data_;
  int32 rows_, cols_, stride_;
};

struct DeviceHooks {
  void *(*alloc)(size_t bytes);
  void (*free)(void *p);
  void (*fill_uniform)(float *dev, int n);  // n uniform draws on (0, 1) into device memory, ordered on `stream`
  tdnnf_stream stream;
  // host-side control decisions (Kaldi: RandUniform() / RandInt(), nnet-simple-component.cc:1017, :1084); optional -- without them
  // RectifiedLinearComponent repairs / stores on every call unless told otherwise (SetRepairNow / SetStoreNow)
  float (*rand_uniform)();
  int (*rand_int)(int lo, int hi);
};
inline DeviceHooks &Hooks() {
  static DeviceHooks h = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  return h;
}
inline void *DeviceAlloc(size_t bytes) {
  if (!Hooks().alloc) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::alloc is not installed");
  return Hooks().alloc(bytes);
}
// a device buffer that grows on demand (workspaces)
class Scratch {
 public:
  Scratch() : p_(nullptr), bytes_(0) {}
  ~Scratch() { if (p_ && Hooks().free) Hooks().free(p_); }
  void *Get(size_t bytes) {
    if (bytes > bytes_) {
      if (p_ && Hooks().free) Hooks().free(p_);
      p_ = DeviceAlloc(bytes);
      bytes_ = bytes;
    }
    return p_;
  }
  size_t Bytes() const { return bytes_; }

 private:
  Scratch(const Scratch &);
  void *p_;
  size_t bytes_;
};

// the uniform draws of a kRandomComponent: kept by the component (grown on demand) instead of allocated and freed around the
// launch that reads them -- freeing right behind an asynchronous launch is only safe with a stream-ordered allocator
class DrawBuffer {
 public:
  float *Get(int n) { return static_cast<float *>(buf_.Get(sizeof(float) * (size_t)std::max(n, 1))); }

 private:
  Scratch buf_;
};

class ComponentPrecomputedIndexes {
 public:
  virtual ~ComponentPrecomputedIndexes() {}
};

class Component {
 public:
  virtual ~Component() {}
  virtual std::string Type() const = 0;
  virtual int32 InputDim() const = 0;
  virtual int32 OutputDim() const = 0;
  virtual int32 Properties() const = 0;
  virtual void *Propagate(const ComponentPrecomputedIndexes *indexes, const CuMatrixBase &in, CuMatrixBase *out) const = 0;
  virtual void Backprop(const std::string &debug_info, const ComponentPrecomputedIndexes *indexes, const CuMatrixBase &in_value,
                        const CuMatrixBase &out_value, const CuMatrixBase &out_deriv, void *memo, Component *to_update,
                        CuMatrixBase *in_deriv) const = 0;
  virtual void DeleteMemo(void *memo) const { (void)memo; }
  virtual void StoreStats(const CuMatrixBase &in_value, const CuMatrixBase &out_value, void *memo) { (void)in_value; (void)out_value; (void)memo; }
  static Component *NewComponentOfType(const std::string &type);
};

// UpdatableComponent: learning rate, is_gradient_, natural gradient switch (nnet-component-itf.cc:347-414)
class UpdatableComponent : public Component {
 public:
  UpdatableComponent() : learning_rate_(0.001f), learning_rate_factor_(1.0f), is_gradient_(false), use_natural_gradient_(true) {
    ng_.in = ng_.out = nullptr;
  }
  ~UpdatableComponent() {
    tdnnf_ng_destroy(ng_.in);
    tdnnf_ng_destroy(ng_.out);
  }
  void SetUnderlyingLearningRate(BaseFloat lr) { learning_rate_ = lr * learning_rate_factor_; }
  void SetLearningRateFactor(BaseFloat f) { learning_rate_factor_ = f; }
  void SetAsGradient() { is_gradient_ = true; learning_rate_ = 1.0f; }
  void SetUseNaturalGradient(bool b) { use_natural_gradient_ = b; }
  BaseFloat LearningRate() const { return learning_rate_; }

 protected:
  // the two OnlineNaturalGradient objects with the reference's defaults (nnet-tdnn-component.cc:183-210): created on first use
  const tdnnf_adapter::NaturalGradient *Preconditioners(int spliced_input_dim, int output_dim) const;
  BaseFloat learning_rate_, learning_rate_factor_;
  bool is_gradient_, use_natural_gradient_;
  mutable tdnnf_adapter::NaturalGradient ng_;
  mutable Scratch ws_;
};
inline const tdnnf_adapter::NaturalGradient *UpdatableComponent::Preconditioners(int spliced, int out_dim) const {
  if (is_gradient_ || !use_natural_gradient_) return nullptr;  // "if (to_update->is_gradient_ || !to_update->use_natural_gradient_) UpdateSimple"
  if (!ng_.in) {
    const int rank_in = std::min(20, (spliced + 1) / 2), rank_out = std::min(80, (out_dim + 1) / 2);
    tdnnf_adapter::Check(tdnnf_ng_create(rank_in, 4, 2000.0f, 4.0f, &ng_.in));
    tdnnf_adapter::Check(tdnnf_ng_create(rank_out, 4, 2000.0f, 4.0f, &ng_.out));
  }
  return &ng_;
}

// ------------------------------------------------------------------------------------------------ Tdnn / TdnnDARTSV3
class TdnnPrecomputedIndexes : public ComponentPrecomputedIndexes {  // nnet-convolutional-component.h:208-218
 public:
  int32 row_stride;
  std::vector<int32> row_offsets;
};

// TdnnDARTSV3Component (src/nnet3/nnet-tdnn-component.cc:214-626); with darts == false the plain TdnnComponent (UPSTREAM)
class TdnnComponentBase : public UpdatableComponent {
 public:
  explicit TdnnComponentBase(bool darts) : darts_(darts), K_(0), Di_(0), Do_(0), linear_(nullptr), bias_(nullptr), flags_(0), temp_(1.0f), offsets1_positive_(true) {}
  // linear_params_ (Do x K Di, dense) and bias_params_ (DARTS: K logits then Do biases; plain: Do or null) on the device
  void SetParams(int32 K, int32 Di, int32 Do, float *linear, float *bias, bool offsets1_positive) {
    K_ = K; Di_ = Di; Do_ = Do; linear_ = linear; bias_ = bias; offsets1_positive_ = offsets1_positive;
  }
  void SetDartsFlags(int flags, float temp_proportion) { flags_ = flags; temp_ = temp_proportion; }
  void SetTempProportion(BaseFloat p) { temp_ = p; }  // nnet-convolutional-component.h:229
  virtual std::string Type() const { return darts_ ? "TdnnDARTSV3Component" : "TdnnComponent"; }
  virtual int32 InputDim() const { return Di_; }
  virtual int32 OutputDim() const { return Do_; }
  virtual int32 Properties() const {  // h:130-134
    return kUpdatableComponent | kReordersIndexes | kBackpropAdds | (bias_ == nullptr ? kPropagateAdds : 0) | kBackpropNeedsInput | (darts_ ? kUsesMemo : 0);
  }
  virtual void *Propagate(const ComponentPrecomputedIndexes *indexes_in, const CuMatrixBase &in, CuMatrixBase *out) const {
    const TdnnPrecomputedIndexes *ix = static_cast<const TdnnPrecomputedIndexes *>(indexes_in);
    const tdnnf_tdnn_indexes tix = tdnnf_adapter::Indexes(ix->row_stride, ix->row_offsets);
    if (!darts_) {
      tdnnf_adapter::TdnnPropagate(tix, in, linear_, K_ * Di_, Do_, Di_, bias_, out, Hooks().stream);
      return nullptr;
    }
    float *memo = static_cast<float *>(DeviceAlloc(sizeof(float) * (3 * K_ + 1)));  // [coef | effective coef | K + 1 uniform draws]
    if (!Hooks().fill_uniform) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::fill_uniform is not installed");
    Hooks().fill_uniform(memo + 2 * K_, K_ + 1);
    tdnnf_adapter::TdnnDartsPropagate(State(), tix, in, out, memo + 2 * K_, memo, Hooks().stream);
    return memo;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *indexes_in, const CuMatrixBase &in_value, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *memo, Component *to_update_in, CuMatrixBase *in_deriv) const {
    const TdnnPrecomputedIndexes *ix = static_cast<const TdnnPrecomputedIndexes *>(indexes_in);
    const tdnnf_tdnn_indexes tix = tdnnf_adapter::Indexes(ix->row_stride, ix->row_offsets);
    TdnnComponentBase *to_update = static_cast<TdnnComponentBase *>(to_update_in);
    const int spliced = K_ * Di_ + (bias_ ? 1 : 0);
    const tdnnf_adapter::NaturalGradient *ng = to_update ? to_update->Preconditioners(spliced, Do_) : nullptr;
    const size_t wsb = ng ? tdnnf_tdnn_update_natural_gradient_workspace_bytes(Do_, Di_, K_, out_deriv.NumRows(), bias_ ? 1 : 0)
                          : tdnnf_tdnn_update_workspace_bytes(Do_, Di_, K_, out_deriv.NumRows());
    void *ws = to_update ? to_update->ws_.Get(wsb) : nullptr;
    if (darts_)
      tdnnf_adapter::TdnnDartsBackprop(State(), tix, in_value, out_deriv, static_cast<const float *>(memo), in_deriv,
                                       to_update ? to_update->learning_rate_ : 0.0f, to_update ? to_update->linear_ : nullptr,
                                       to_update ? to_update->bias_ : nullptr, ws, wsb, Hooks().stream, ng);
    else
      tdnnf_adapter::TdnnBackprop(tix, in_value, out_deriv, linear_, K_ * Di_, Do_, Di_, in_deriv, to_update ? to_update->learning_rate_ : 0.0f,
                                  to_update ? to_update->linear_ : nullptr, to_update ? to_update->bias_ : nullptr, ws, wsb, Hooks().stream, ng);
  }
  virtual void DeleteMemo(void *memo) const { if (memo && Hooks().free) Hooks().free(memo); }  // (h:147-149 deletes a CuVector through a CuMatrix*)

 private:
  tdnnf_adapter::TdnnDartsState State() const {
    tdnnf_adapter::TdnnDartsState s;
    s.K = K_; s.Di = Di_; s.Do = Do_; s.ldw = K_ * Di_; s.linear_params = linear_; s.bias_params = bias_; s.flags = flags_;
    s.temp_proportion = temp_; s.share_index = offsets1_positive_ ? 0 : K_ - 1; s.offsets1_positive = offsets1_positive_;
    return s;
  }
  bool darts_;
  int32 K_, Di_, Do_;
  float *linear_, *bias_;
  int flags_;
  float temp_;
  bool offsets1_positive_;
};
class TdnnDARTSV3Component : public TdnnComponentBase {
 public:
  TdnnDARTSV3Component() : TdnnComponentBase(true) {}
};
class TdnnComponent : public TdnnComponentBase {
 public:
  TdnnComponent() : TdnnComponentBase(false) {}
};

// ------------------------------------------------------------------------------------------------ BatchNorm / BatchNormTest
// nnet-normalize-component.cc:401-589 (train mode; memo = 5 x D floats, StoreStats from the memo :551-589)
class BatchNormComponent : public Component {
 public:
  BatchNormComponent() : dim_(0), epsilon_(1.0e-3f), target_rms_(1.0f), stats_(nullptr) {}
  void Init(int32 dim, float epsilon, float target_rms, double *stats_dev /* [count, sum[D], sumsq[D]] */) {
    dim_ = dim; epsilon_ = epsilon; target_rms_ = target_rms; stats_ = stats_dev;
  }
  virtual std::string Type() const { return "BatchNormComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const {  // nnet-normalize-component.h:182-190 with block-dim == dim, training mode
    return kSimpleComponent | kBackpropNeedsOutput | kPropagateInPlace | kBackpropInPlace | kUsesMemo | kStoresStats;
  }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    float *memo = static_cast<float *>(DeviceAlloc(sizeof(float) * 5 * dim_));
    const size_t wsb = tdnnf_colreduce_workspace_bytes(in.NumRows(), dim_);
    tdnnf_adapter::BatchNormPropagate(in, epsilon_, target_rms_, out, memo, ws_.Get(wsb), wsb, Hooks().stream);
    return memo;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &out_value,
                        const CuMatrixBase &out_deriv, void *memo, Component *, CuMatrixBase *in_deriv) const {
    const size_t wsb = tdnnf_colreduce_workspace_bytes(out_value.NumRows(), dim_);
    tdnnf_adapter::BatchNormBackprop(out_value, out_deriv, target_rms_, static_cast<float *>(memo), in_deriv, ws_.Get(wsb), wsb, Hooks().stream);
  }
  virtual void StoreStats(const CuMatrixBase &in_value, const CuMatrixBase &, void *memo) {
    tdnnf_adapter::BatchNormStoreStats(static_cast<const float *>(memo), dim_, in_value.NumRows(), stats_, Hooks().stream);
  }
  virtual void DeleteMemo(void *memo) const { if (memo && Hooks().free) Hooks().free(memo); }

 private:
  int32 dim_;
  float epsilon_, target_rms_;
  double *stats_;
  mutable Scratch ws_;
};
// nnet-normalize-component.cc:682-922: frozen statistics; scale_ / offset_ from ComputeDerived
class BatchNormTestComponent : public Component {
 public:
  BatchNormTestComponent() : dim_(0), scale_(nullptr), offset_(nullptr) {}
  void Init(int32 dim, float epsilon, float target_rms, const double *stats_dev) {
    dim_ = dim;
    scale_ = static_cast<float *>(DeviceAlloc(sizeof(float) * 2 * dim));
    offset_ = scale_ + dim;
    tdnnf_adapter::BatchNormComputeDerived(stats_dev, dim, epsilon, target_rms, scale_, offset_, Hooks().stream);
  }
  ~BatchNormTestComponent() { if (scale_ && Hooks().free) Hooks().free(scale_); }
  virtual std::string Type() const { return "BatchNormTestComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kBackpropNeedsOutput | kPropagateInPlace | kBackpropInPlace; }  // h:359-366
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::BatchNormTestPropagate(in, scale_, offset_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    tdnnf_adapter::BatchNormTestBackprop(out_deriv, scale_, in_deriv, Hooks().stream);
  }

 private:
  int32 dim_;
  float *scale_, *offset_;
};

// ------------------------------------------------------------------------------------------------ the DARTS mixing operators
// (Gumbel)SoftmaxFlopsComponent nnet-simple-component.cc:9968-10020, :10088-10158 (dim must be 8: the hard-coded FLOPs vector)
class SoftmaxFlopsComponentBase : public Component {
 public:
  explicit SoftmaxFlopsComponentBase(bool gumbel) : gumbel_(gumbel), dim_(8), scale_(0.f), temp_(1.0f), flops_(nullptr) {}
  void Init(int32 dim, float scale, const float *flops_dev /* dim floats: -(25, 50, 80, 100, 120, 160, 200, 240) */) { dim_ = dim; scale_ = scale; flops_ = flops_dev; }
  void SetTempProportion(BaseFloat p) { temp_ = p; }
  virtual std::string Type() const { return gumbel_ ? "GumbelSoftmaxFlopsComponent" : "SoftmaxFlopsComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kBackpropInPlace | kSimpleComponent | kBackpropNeedsInput | kBackpropNeedsOutput | kRandomComponent; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    float *u = nullptr;
    if (gumbel_) {  // one Gumbel vector shared by all rows (:10095-10104), in a buffer the component keeps
      u = draws_.Get(dim_);
      if (!Hooks().fill_uniform) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::fill_uniform is not installed");
      Hooks().fill_uniform(u, dim_);
    }
    tdnnf_adapter::SoftmaxFlopsPropagate(in, u, temp_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &out_value,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    CuMatrixBase d = out_deriv;  // the reference mutates the const out_deriv in place (:10006-10016, :10144-10154)
    tdnnf_adapter::SoftmaxFlopsBackprop(out_value, &d, scale_, flops_, dim_, gumbel_ ? temp_ : 1.0f, in_deriv, Hooks().stream);
  }

 private:
  bool gumbel_;
  int32 dim_;
  float scale_, temp_;
  const float *flops_;
  mutable DrawBuffer draws_;
};
class SoftmaxFlopsComponent : public SoftmaxFlopsComponentBase {
 public:
  SoftmaxFlopsComponent() : SoftmaxFlopsComponentBase(false) {}
};
class GumbelSoftmaxFlopsComponent : public SoftmaxFlopsComponentBase {
 public:
  GumbelSoftmaxFlopsComponent() : SoftmaxFlopsComponentBase(true) {}
};

// OnehotFunctionComponent :9504-9552 and the NAS-modified ConstantFunctionComponent :2602-2642: an updatable output_ vector
class OutputVectorComponent : public UpdatableComponent {
 public:
  explicit OutputVectorComponent(bool onehot) : onehot_(onehot), in_dim_(0), out_dim_(0), output_(nullptr), is_updatable_(true) { use_natural_gradient_ = false; }
  void Init(int32 input_dim, int32 output_dim, float *output_dev, bool is_updatable) { in_dim_ = input_dim; out_dim_ = output_dim; output_ = output_dev; is_updatable_ = is_updatable; }
  virtual std::string Type() const { return onehot_ ? "OnehotFunctionComponent" : "ConstantFunctionComponent"; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  virtual int32 Properties() const {
    return kSimpleComponent | (is_updatable_ ? kUpdatableComponent : 0) | (in_dim_ == out_dim_ ? kPropagateInPlace : 0) | kBackpropAdds;
  }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &, CuMatrixBase *out) const {
    if (!onehot_) {
      tdnnf_adapter::ConstantFunctionPropagate(output_, out, Hooks().stream);
      return nullptr;
    }
    float *u = draws_.Get(1);
    if (!Hooks().fill_uniform) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::fill_uniform is not installed");
    Hooks().fill_uniform(u, 1);
    tdnnf_adapter::OnehotPropagate(u, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *to_update_in, CuMatrixBase *) const {
    OutputVectorComponent *to_update = static_cast<OutputVectorComponent *>(to_update_in);
    if (!to_update || !to_update->is_updatable_) return;
    const size_t wsb = tdnnf_colreduce_workspace_bytes(out_deriv.NumRows(), out_dim_);
    if (onehot_) tdnnf_adapter::OnehotBackprop(out_deriv, to_update->learning_rate_, to_update->output_, to_update->ws_.Get(wsb), wsb, Hooks().stream);
    else tdnnf_adapter::ConstantFunctionBackprop(out_deriv, to_update->learning_rate_, to_update->output_, to_update->ws_.Get(wsb), wsb, Hooks().stream);
  }

 private:
  bool onehot_;
  int32 in_dim_, out_dim_;
  float *output_;
  bool is_updatable_;
  mutable DrawBuffer draws_;
};
class OnehotFunctionComponent : public OutputVectorComponent {
 public:
  OnehotFunctionComponent() : OutputVectorComponent(true) {}
};
class ConstantFunctionComponent : public OutputVectorComponent {
 public:
  ConstantFunctionComponent() : OutputVectorComponent(false) {}
};

class CopyNComponent : public Component {  // :4843-4867
 public:
  CopyNComponent() : in_dim_(1), out_dim_(1), scale_(1.0f) {}
  void Init(int32 input_dim, int32 output_dim, float scale) { in_dim_ = input_dim; out_dim_ = output_dim; scale_ = scale; }
  virtual std::string Type() const { return "CopyNComponent"; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kPropagateAdds | kBackpropAdds; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::CopyNPropagate(in, scale_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    if (in_deriv) tdnnf_adapter::CopyNBackprop(out_deriv, scale_, in_deriv, Hooks().stream);
  }

 private:
  int32 in_dim_, out_dim_;
  float scale_;
};

class ElementwiseProductComponent : public Component {  // :256-299
 public:
  ElementwiseProductComponent() : in_dim_(0), out_dim_(0) {}
  void Init(int32 input_dim, int32 output_dim) { in_dim_ = input_dim; out_dim_ = output_dim; }
  virtual std::string Type() const { return "ElementwiseProductComponent"; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kBackpropNeedsInput; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::ElementwiseProductPropagate(in, out_dim_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &in_value, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    if (in_deriv) tdnnf_adapter::ElementwiseProductBackprop(in_value, out_deriv, out_dim_, in_deriv, Hooks().stream);
  }

 private:
  int32 in_dim_, out_dim_;
};

// ------------------------------------------------------------------------------------------------ the standard layers around them
class RectifiedLinearComponent : public Component {  // :958-1091; statistics [count, value_sum[D], deriv_sum[D]] on the device
 public:
  RectifiedLinearComponent() : dim_(0), self_repair_scale_(0.f), stats_(nullptr) {}
  void Init(int32 dim, float self_repair_scale, double *stats_dev) { dim_ = dim; self_repair_scale_ = self_repair_scale; stats_ = stats_dev; }
  virtual std::string Type() const { return "RectifiedLinearComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kBackpropNeedsOutput | kPropagateInPlace | kStoresStats; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::ReluPropagate(in, out, Hooks().stream);
    return nullptr;
  }
  // RepairGradients runs "on about half of the minibatches" (:997-1018: `if (RandUniform() > repair_probability) return`): the
  // coin comes from DeviceHooks::rand_uniform when the host installed it, else from SetRepairNow()
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &out_value,
                        const CuMatrixBase &out_deriv, void *, Component *to_update, CuMatrixBase *in_deriv) const {
    if (!in_deriv) return;
    tdnnf_adapter::ReluBackprop(out_value, out_deriv, in_deriv, Hooks().stream);
    RectifiedLinearComponent *tu = static_cast<RectifiedLinearComponent *>(to_update);
    if (!tu || tu->self_repair_scale_ <= 0.f || !tu->stored_once_) return;  // "self_repair_scale_ == 0.0 || count_ == 0.0" :1001
    const bool repair = Hooks().rand_uniform ? !(Hooks().rand_uniform() > 0.5f) : tu->repair_now_;
    if (repair) tdnnf_adapter::ReluRepairGradients(tu->stats_, dim_, tu->self_repair_scale_, 0.05f, 0.95f, in_deriv, Hooks().stream);
  }
  // "Only store stats about every other minibatch (but on the first minibatch, always store it)" :1081-1085
  virtual void StoreStats(const CuMatrixBase &, const CuMatrixBase &out_value, void *) {
    const bool skip = Hooks().rand_int ? (Hooks().rand_int(0, 1) == 0 && stored_once_) : !store_now_;
    if (skip) return;
    const size_t wsb = tdnnf_colreduce_workspace_bytes(out_value.NumRows(), dim_);
    tdnnf_adapter::ReluStoreStats(out_value, stats_, ws_.Get(wsb), wsb, Hooks().stream);
    stored_once_ = true;
  }
  void SetRepairNow(bool b) { repair_now_ = b; }
  void SetStoreNow(bool b) { store_now_ = b; }
  void SetStatsPresent(bool b) { stored_once_ = b; }  // a model read from disk with count_ != 0

 private:
  int32 dim_;
  float self_repair_scale_;
  double *stats_;
  bool repair_now_ = true, store_now_ = true, stored_once_ = false;
  Scratch ws_;
};

// AffineComponent :1235-1279 with the natural-gradient Update of NaturalGradientAffineComponent :2980-3024; bias == null: LinearComponent :3211-3254
class AffineComponentBase : public UpdatableComponent {
 public:
  explicit AffineComponentBase(const char *type) : type_(type), in_dim_(0), out_dim_(0), linear_(nullptr), bias_(nullptr) {}
  void SetParams(int32 input_dim, int32 output_dim, float *linear_dev, float *bias_dev) { in_dim_ = input_dim; out_dim_ = output_dim; linear_ = linear_dev; bias_ = bias_dev; }
  virtual std::string Type() const { return type_; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kUpdatableComponent | kBackpropNeedsInput | (bias_ ? 0 : kPropagateAdds) | kBackpropAdds; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::AffinePropagate(in, linear_, in_dim_, bias_, out_dim_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &in_value, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *to_update_in, CuMatrixBase *in_deriv) const {
    AffineComponentBase *to_update = static_cast<AffineComponentBase *>(to_update_in);
    const tdnnf_adapter::NaturalGradient *ng = to_update ? to_update->Preconditioners(in_dim_ + (bias_ ? 1 : 0), out_dim_) : nullptr;
    const size_t wsb = ng ? tdnnf_affine_update_natural_gradient_workspace_bytes(out_dim_, in_dim_, out_deriv.NumRows(), bias_ ? 1 : 0)
                          : tdnnf_tdnn_update_workspace_bytes(out_dim_, in_dim_, 1, out_deriv.NumRows());
    tdnnf_adapter::AffineBackprop(in_value, out_deriv, linear_, in_dim_, in_deriv, to_update ? to_update->learning_rate_ : 0.0f,
                                  to_update ? to_update->linear_ : nullptr, to_update ? to_update->bias_ : nullptr,
                                  to_update ? to_update->ws_.Get(wsb) : nullptr, wsb, Hooks().stream, ng);
  }

 private:
  const char *type_;
  int32 in_dim_, out_dim_;
  float *linear_, *bias_;
};
class NaturalGradientAffineComponent : public AffineComponentBase {
 public:
  NaturalGradientAffineComponent() : AffineComponentBase("NaturalGradientAffineComponent") {}
};
class LinearComponent : public AffineComponentBase {
 public:
  LinearComponent() : AffineComponentBase("LinearComponent") {}
};

// AffineComponent :1235-1279: Update() is UpdateSimple (no preconditioning), whatever use-natural-gradient says
class AffineComponent : public AffineComponentBase {
 public:
  AffineComponent() : AffineComponentBase("AffineComponent") { use_natural_gradient_ = false; }
  void SetUseNaturalGradient(bool) {}
};

// FixedAffineComponent :3378-3399 (the lda layer): not updatable, kBackpropAdds
class FixedAffineComponent : public Component {
 public:
  FixedAffineComponent() : in_dim_(0), out_dim_(0), linear_(nullptr), bias_(nullptr) {}
  void SetParams(int32 input_dim, int32 output_dim, const float *linear_dev, const float *bias_dev) { in_dim_ = input_dim; out_dim_ = output_dim; linear_ = linear_dev; bias_ = bias_dev; }
  virtual std::string Type() const { return "FixedAffineComponent"; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kBackpropAdds; }  // nnet-simple-component.h:1010
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::AffinePropagate(in, linear_, in_dim_, bias_, out_dim_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    tdnnf_adapter::FixedAffineBackprop(out_deriv, linear_, in_dim_, in_dim_, in_deriv, Hooks().stream);
  }

 private:
  int32 in_dim_, out_dim_;
  const float *linear_, *bias_;
};

class NoOpComponent : public Component {  // :437-456
 public:
  NoOpComponent() : dim_(0), backprop_scale_(1.0f) {}
  void Init(int32 dim, float backprop_scale) { dim_ = dim; backprop_scale_ = backprop_scale; }
  virtual std::string Type() const { return "NoOpComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kPropagateInPlace | kBackpropInPlace; }  // nnet-simple-component.h:1192-1194
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::NoOpPropagate(in, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    if (in_deriv) tdnnf_adapter::NoOpBackprop(out_deriv, backprop_scale_, in_deriv, Hooks().stream);
  }

 private:
  int32 dim_;
  float backprop_scale_;
};

// GeneralDropoutComponent (UPSTREAM; factory nnet-component-itf.cc:194, `set-dropout-proportion` nnet-utils.cc:1315-1321) in the
// recipes' configuration: one mask row per sequence shared over time (time-period 0), block-dim == dim.  The precomputed indexes
// reduce to the number of sequences: row r of a t-major matrix belongs to sequence r % num_seq.
class GeneralDropoutPrecomputedIndexes : public ComponentPrecomputedIndexes {
 public:
  int32 num_mask_rows;
};
class GeneralDropoutComponent : public Component {
 public:
  GeneralDropoutComponent() : dim_(0), proportion_(0.5f), continuous_(false), test_mode_(false) {}
  void Init(int32 dim, float dropout_proportion, bool continuous) { dim_ = dim; proportion_ = dropout_proportion; continuous_ = continuous; }
  void SetDropoutProportion(BaseFloat p) { proportion_ = p; }
  void SetTestMode(bool b) { test_mode_ = b; }
  virtual std::string Type() const { return "GeneralDropoutComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kRandomComponent | kPropagateInPlace | kBackpropInPlace | kUsesMemo; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *indexes_in, const CuMatrixBase &in, CuMatrixBase *out) const {
    if (test_mode_ || proportion_ == 0.0f) {  // "if (test_mode_ || dropout_proportion_ == 0.0) return NULL" after the copy
      tdnnf_adapter::NoOpPropagate(in, out, Hooks().stream);
      return nullptr;
    }
    const int32 S = static_cast<const GeneralDropoutPrecomputedIndexes *>(indexes_in)->num_mask_rows;
    const long long n = (long long)S * dim_;
    float *memo = static_cast<float *>(DeviceAlloc(sizeof(float) * 2 * n));  // [mask | the uniform draws it was made from]
    if (!Hooks().fill_uniform) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::fill_uniform is not installed");
    Hooks().fill_uniform(memo + n, (int)n);
    tdnnf_adapter::Check(tdnnf_general_dropout_mask(memo + n, n, proportion_, continuous_ ? 1 : 0, memo, Hooks().stream));
    tdnnf_adapter::GeneralDropoutApply(in, memo, S, out, Hooks().stream);
    return memo;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *indexes_in, const CuMatrixBase &, const CuMatrixBase &,
                        const CuMatrixBase &out_deriv, void *memo, Component *, CuMatrixBase *in_deriv) const {
    if (!in_deriv) return;
    if (!memo) {
      tdnnf_adapter::NoOpBackprop(out_deriv, 1.0f, in_deriv, Hooks().stream);
      return;
    }
    const int32 S = static_cast<const GeneralDropoutPrecomputedIndexes *>(indexes_in)->num_mask_rows;
    tdnnf_adapter::GeneralDropoutApply(out_deriv, static_cast<const float *>(memo), S, in_deriv, Hooks().stream);
  }
  virtual void DeleteMemo(void *memo) const { if (memo && Hooks().free) Hooks().free(memo); }

 private:
  int32 dim_;
  float proportion_;
  bool continuous_, test_mode_;
};

// FlopsConstraintComponent :9454-9478 (add_flopsconstraint.py:18-30 puts it behind the choice softmax)
class FlopsConstraintComponent : public Component {
 public:
  FlopsConstraintComponent() : in_dim_(0), out_dim_(0), scale_(1.0f), flops_(nullptr) {}
  void Init(int32 input_dim, int32 output_dim, float scale, const float *flops_dev /* input_dim floats */) { in_dim_ = input_dim; out_dim_ = output_dim; scale_ = scale; flops_ = flops_dev; }
  virtual std::string Type() const { return "FlopsConstraintComponent"; }
  virtual int32 InputDim() const { return in_dim_; }
  virtual int32 OutputDim() const { return out_dim_; }
  // nnet-simple-component.h:2697-2699 -- the flags say "adds", yet both bodies overwrite (CopyFromMat :9458, CopyRowsFromVec :9475): reproduced
  virtual int32 Properties() const { return kSimpleComponent | kPropagateAdds | kBackpropAdds | kBackpropNeedsInput; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::NoOpPropagate(in, out, Hooks().stream);  // "out->CopyFromMat(in)" :9458
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &in_value, const CuMatrixBase &,
                        const CuMatrixBase &, void *, Component *, CuMatrixBase *in_deriv) const {
    if (in_deriv) tdnnf_adapter::FlopsConstraintBackprop(flops_, scale_, in_value, in_deriv, Hooks().stream);
  }

 private:
  int32 in_dim_, out_dim_;
  float scale_;
  const float *flops_;
};

// GumbelSoftmaxComponent :9774-9831: any width, no FLOPs penalty; the Gumbel draws are the memo-less part of Propagate
class GumbelSoftmaxComponent : public Component {
 public:
  GumbelSoftmaxComponent() : dim_(0), temp_(1.0f) {}
  void Init(int32 dim) { dim_ = dim; }
  void SetTempProportion(BaseFloat p) { temp_ = p; }
  virtual std::string Type() const { return "GumbelSoftmaxComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kBackpropInPlace | kSimpleComponent | kBackpropNeedsInput | kBackpropNeedsOutput | kRandomComponent; }  // h:2878-2881
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    float *u = draws_.Get(dim_);  // kept by the component: the kernel reads it after Propagate returns
    if (!Hooks().fill_uniform) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::fill_uniform is not installed");
    Hooks().fill_uniform(u, dim_);
    tdnnf_adapter::SoftmaxFlopsPropagate(in, u, temp_, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &out_value,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    if (!in_deriv) return;  // :9817-9818
    CuMatrixBase d = out_deriv;  // (not mutated: no FLOPs vector)
    tdnnf_adapter::SoftmaxFlopsBackprop(out_value, &d, 0.0f, (const float *)nullptr, 0, temp_, in_deriv, Hooks().stream);
  }

 private:
  int32 dim_;
  float temp_;
  mutable DrawBuffer draws_;
};

class LogSoftmaxComponent : public Component {  // :3607-3632
 public:
  LogSoftmaxComponent() : dim_(0) {}
  void Init(int32 dim) { dim_ = dim; }
  virtual std::string Type() const { return "LogSoftmaxComponent"; }
  virtual int32 InputDim() const { return dim_; }
  virtual int32 OutputDim() const { return dim_; }
  virtual int32 Properties() const { return kSimpleComponent | kBackpropNeedsOutput | kStoresStats; }
  virtual void *Propagate(const ComponentPrecomputedIndexes *, const CuMatrixBase &in, CuMatrixBase *out) const {
    tdnnf_adapter::LogSoftmaxPropagate(in, out, Hooks().stream);
    return nullptr;
  }
  virtual void Backprop(const std::string &, const ComponentPrecomputedIndexes *, const CuMatrixBase &, const CuMatrixBase &out_value,
                        const CuMatrixBase &out_deriv, void *, Component *, CuMatrixBase *in_deriv) const {
    if (in_deriv) tdnnf_adapter::LogSoftmaxBackprop(out_value, out_deriv, in_deriv, Hooks().stream);
  }

 private:
  int32 dim_;
};

// ------------------------------------------------------------------------------------------------ registration
// The factory names of Component::NewComponentOfType (nnet-component-itf.cc:120-281) this library serves.
inline const std::vector<std::string> &RegisteredTypes() {
  static const std::vector<std::string> names = {
      "TdnnDARTSV3Component", "TdnnComponent", "BatchNormComponent", "BatchNormTestComponent", "GumbelSoftmaxFlopsComponent",
      "SoftmaxFlopsComponent", "OnehotFunctionComponent", "ConstantFunctionComponent", "CopyNComponent", "ElementwiseProductComponent",
      "RectifiedLinearComponent", "NaturalGradientAffineComponent", "LinearComponent", "LogSoftmaxComponent", "AffineComponent",
      "FixedAffineComponent", "NoOpComponent", "GeneralDropoutComponent", "FlopsConstraintComponent", "GumbelSoftmaxComponent"};
  return names;
}
inline Component *Component::NewComponentOfType(const std::string &t) {
  if (t == "TdnnDARTSV3Component") return new TdnnDARTSV3Component();
  if (t == "TdnnComponent") return new TdnnComponent();
  if (t == "BatchNormComponent") return new BatchNormComponent();
  if (t == "BatchNormTestComponent") return new BatchNormTestComponent();
  if (t == "GumbelSoftmaxFlopsComponent") return new GumbelSoftmaxFlopsComponent();
  if (t == "SoftmaxFlopsComponent") return new SoftmaxFlopsComponent();
  if (t == "OnehotFunctionComponent") return new OnehotFunctionComponent();
  if (t == "ConstantFunctionComponent") return new ConstantFunctionComponent();
  if (t == "CopyNComponent") return new CopyNComponent();
  if (t == "ElementwiseProductComponent") return new ElementwiseProductComponent();
  if (t == "RectifiedLinearComponent") return new RectifiedLinearComponent();
  if (t == "NaturalGradientAffineComponent") return new NaturalGradientAffineComponent();
  if (t == "LinearComponent") return new LinearComponent();
  if (t == "LogSoftmaxComponent") return new LogSoftmaxComponent();
  if (t == "AffineComponent") return new AffineComponent();
  if (t == "FixedAffineComponent") return new FixedAffineComponent();
  if (t == "NoOpComponent") return new NoOpComponent();
  if (t == "GeneralDropoutComponent") return new GeneralDropoutComponent();
  if (t == "FlopsConstraintComponent") return new FlopsConstraintComponent();
  if (t == "GumbelSoftmaxComponent") return new GumbelSoftmaxComponent();
  return nullptr;  // (the reference returns NULL for unknown types as well, :279)
}

}  // namespace tdnnf_nnet3
#endif  // TDNNF_NNET3_COMPONENTS_H_
