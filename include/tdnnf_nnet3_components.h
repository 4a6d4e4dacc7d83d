// tdnnf_nnet3_components.h -- the hot-path nnet3 components of skhu101/TDNN-F_NAS as Component CLASSES over libtdnnf_hip.so,
// registered under the factory names of the reference (Component::NewComponentOfType,
// /root/reference/src/nnet3/nnet-component-itf.cc:120-281).
//
// tdnnf_nnet3_adapter.h holds the edited method BODIES as free functions; this header wraps them in classes with the virtual
// interface every shipped subclass shows (e.g. src/nnet3/nnet-convolutional-component.h:121-191): Type(), InputDim(), OutputDim(),
// Properties() (the reference's own flag expressions), Propagate() returning the memo, Backprop() with `to_update`, DeleteMemo(),
// StoreStats() -- and [round 5] the rest of what a Kaldi factory hands out: InitFromConfig(ConfigLine *) with the reference's keys and
// defaults, Read / Write in the reference's token order (text and binary, include/tdnnf_kaldi_io.h), Copy(), Info(), PrecomputeIndexes /
// ReorderIndexes for the Tdnn classes, and the UpdatableComponent virtuals Scale / Add / DotProduct / NumParameters / Vectorize /
// UnVectorize / PerturbParams / FreezeNaturalGradient (nnet-convolutional-component.h:121-191,229; nnet-component-itf.cc:332-414).  A
// component made by InitFromConfig / Read / Copy OWNS its parameters (device memory from DeviceHooks::alloc); SetParams / Init still
// bind caller-owned buffers (the trainer's flat parameter vector).  The file compiles and runs WITHOUT Kaldi (tests/
// test_adapter_compile.py builds it with g++ against the three-accessor matrix stub; tests/adapter_driver.cc runs every class on
// the GPU against the free functions).  Inside Kaldi: derive from kaldi::nnet3::Component instead of tdnnf_nnet3::Component
// (same signatures) and keep the members in CuMatrix / CuVector.
//
// Device memory the classes need beyond the caller's matrices (memos, workspaces, random draws) comes from two hooks the host
// program installs once: DeviceHooks::alloc / free (Kaldi: CuAllocator, nnet-utils.cc:1086 g_cuda_allocator) and
// DeviceHooks::fill_uniform (Kaldi: CuRand<BaseFloat>::RandUniform, as nnet-tdnn-component.cc:258-259).  No call synchronises.
#ifndef TDNNF_NNET3_COMPONENTS_H_
#define TDNNF_NNET3_COMPONENTS_H_

#include <math.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <limits>
#include <map>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "tdnnf_kaldi_io.h"
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
  // host <-> device copies ordered behind `stream` and complete on return (Kaldi: CuMatrix::CopyFromMat / CopyToMat): needed by
  // InitFromConfig, Read / Write, Vectorize / UnVectorize, Info -- never by Propagate / Backprop
  void (*h2d)(void *dst_dev, const void *src_host, size_t bytes);
  void (*d2h)(void *dst_host, const void *src_dev, size_t bytes);
};
inline DeviceHooks &Hooks() {
  static DeviceHooks h = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
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

// an owned device array with host upload / download (parameters and statistics of a component made by InitFromConfig / Read / Copy)
template <class T>
class DeviceArray {
 public:
  DeviceArray() : p_(nullptr), n_(0) {}
  ~DeviceArray() { Free(); }
  void Upload(const std::vector<T> &host) {
    if (host.size() != n_) {
      Free();
      n_ = host.size();
      p_ = n_ ? static_cast<T *>(DeviceAlloc(sizeof(T) * n_)) : nullptr;
    }
    if (n_) {
      if (!Hooks().h2d) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::h2d is not installed");
      Hooks().h2d(p_, host.data(), sizeof(T) * n_);
    }
  }
  T *Data() const { return p_; }
  size_t Size() const { return n_; }

 private:
  DeviceArray(const DeviceArray &);
  void Free() {
    if (p_ && Hooks().free) Hooks().free(p_);
    p_ = nullptr;
    n_ = 0;
  }
  T *p_;
  size_t n_;
};
template <class T>
inline std::vector<T> Download(const T *dev, size_t n) {
  std::vector<T> host(n);
  if (n) {
    if (!Hooks().d2h) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::d2h is not installed");
    Hooks().d2h(host.data(), dev, sizeof(T) * n);
  }
  return host;
}
// SetRandn (Kaldi: CuMatrix::SetRandn through CuRand): host-side normal draws from one process-wide generator (SetRandSeed for tests)
inline std::mt19937 &RandGen() {
  static std::mt19937 g(5489u);
  return g;
}
inline void SetRandSeed(unsigned seed) { RandGen().seed(seed); }
inline std::vector<float> Randn(size_t n, float stddev, float mean = 0.0f) {
  std::normal_distribution<float> d(0.0f, 1.0f);
  std::vector<float> v(n);
  for (size_t i = 0; i < n; i++) v[i] = mean + stddev * d(RandGen());
  return v;
}

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

// ConfigLine (UPSTREAM util/text-utils.h; used by every InitFromConfig, e.g. nnet-tdnn-component.cc:109-211): one line
// "key1=value1 key2=value2 ..." (a leading word without '=' is kept as FirstToken), values read by type and remembered as used.
class ConfigLine {
 public:
  bool ParseLine(const std::string &line) {
    whole_ = line;
    data_.clear();
    first_.clear();
    std::istringstream is(line);
    std::string w;
    bool first = true;
    while (is >> w) {
      const size_t eq = w.find('=');
      if (eq == std::string::npos) {
        if (!first) return false;
        first_ = w;
      } else {
        if (eq == 0) return false;
        data_[w.substr(0, eq)] = std::make_pair(w.substr(eq + 1), false);
      }
      first = false;
    }
    return true;
  }
  const std::string &FirstToken() const { return first_; }
  const std::string &WholeLine() const { return whole_; }
  bool GetValue(const std::string &key, std::string *v) {
    std::map<std::string, std::pair<std::string, bool> >::iterator it = data_.find(key);
    if (it == data_.end()) return false;
    it->second.second = true;
    *v = it->second.first;
    return true;
  }
  bool GetValue(const std::string &key, int32 *v) {
    std::string t;
    if (!GetValue(key, &t)) return false;
    char *end = nullptr;
    const long x = strtol(t.c_str(), &end, 10);
    if (t.empty() || *end) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: bad integer for " + key + " in: " + whole_);
    *v = (int32)x;
    return true;
  }
  bool GetValue(const std::string &key, BaseFloat *v) {
    std::string t;
    if (!GetValue(key, &t)) return false;
    char *end = nullptr;
    const double x = strtod(t.c_str(), &end);
    if (t.empty() || *end) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: bad number for " + key + " in: " + whole_);
    *v = (BaseFloat)x;
    return true;
  }
  bool GetValue(const std::string &key, bool *v) {
    std::string t;
    if (!GetValue(key, &t)) return false;
    if (t == "true" || t == "True" || t == "T" || t == "t") *v = true;
    else if (t == "false" || t == "False" || t == "F" || t == "f") *v = false;
    else TDNNF_ADAPTER_FAIL("tdnnf_nnet3: bad boolean for " + key + " in: " + whole_);
    return true;
  }
  bool HasUnusedValues() const {
    for (std::map<std::string, std::pair<std::string, bool> >::const_iterator it = data_.begin(); it != data_.end(); ++it)
      if (!it->second.second) return true;
    return false;
  }
  std::string UnusedValues() const {
    std::string r;
    for (std::map<std::string, std::pair<std::string, bool> >::const_iterator it = data_.begin(); it != data_.end(); ++it)
      if (!it->second.second) r += (r.empty() ? "" : " ") + it->first + "=" + it->second.first;
    return r;
  }

 private:
  std::string whole_, first_;
  std::map<std::string, std::pair<std::string, bool> > data_;
};
// the keys the nnet3 config reader strips before InitFromConfig sees the line ("component name=x type=Y ...")
inline void DropNameAndType(ConfigLine *cfl) {
  std::string dummy;
  cfl->GetValue("name", &dummy);
  cfl->GetValue("type", &dummy);
}
inline bool SplitStringToIntegers(const std::string &s, std::vector<int32> *out) {
  out->clear();
  std::istringstream is(s);
  std::string tok;
  while (std::getline(is, tok, ',')) {
    char *end = nullptr;
    const long v = strtol(tok.c_str(), &end, 10);
    if (tok.empty() || *end) return false;
    out->push_back((int32)v);
  }
  return true;
}

// Index / MiscComputationInfo (UPSTREAM nnet3/nnet-common.h): (n, t, x) of one matrix row
const int32 kNoTime = std::numeric_limits<int32>::min();
struct Index {
  int32 n, t, x;
  Index() : n(0), t(0), x(0) {}
  Index(int32 n_, int32 t_, int32 x_ = 0) : n(n_), t(t_), x(x_) {}
  bool operator==(const Index &o) const { return n == o.n && t == o.t && x == o.x; }
  bool operator<(const Index &o) const { return t != o.t ? t < o.t : (x != o.x ? x < o.x : n < o.n); }
};
struct MiscComputationInfo {};
// VectorBase<BaseFloat> (a HOST vector: Vectorize / UnVectorize exchange parameters with the CPU, nnet-tdnn-component.cc:960-977)
class VectorBase {
 public:
  VectorBase(float *data, int32 dim) : data_(data), dim_(dim) {}
  float *Data() { return data_; }
  const float *Data() const { return data_; }
  int32 Dim() const { return dim_; }

 private:
  float *data_;
  int32 dim_;
};

class Component {
 public:
  virtual ~Component() {}
  virtual std::string Type() const = 0;
  virtual int32 InputDim() const = 0;
  virtual int32 OutputDim() const = 0;
  virtual int32 Properties() const = 0;
  // ---- [round 5] construction, serialisation, index bookkeeping (nnet-component-itf.cc:105-118, :283-310)
  virtual void InitFromConfig(ConfigLine *cfl) = 0;
  virtual void Write(std::ostream &os, bool binary) const = 0;
  // Read: the tokens of this component up to and including its closing tag; the opening tag may or may not have been consumed
  // (ExpectOneOrTwoTokens in every reference Read).  Generic: tdnnf_kaldi_io::read_block, then the class takes what it needs.
  virtual void Read(std::istream &is, bool binary) {
    tdnnf_kaldi_io::In in{is, binary, std::string()};
    tdnnf_kaldi_io::Parsed p;
    if (!tdnnf_kaldi_io::read_block(in, Type(), &p)) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: reading " + Type() + ": " + in.err);
    FromParsed(p);
  }
  // Copy(): a new object with its own parameters (through the binary encoding: every reference copy constructor copies exactly
  // what Write writes, plus the preconditioners' configuration which Write carries too)
  virtual Component *Copy() const {
    std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
    Write(ss, true);
    Component *c = NewComponentOfType(Type());
    c->Read(ss, true);
    return c;
  }
  virtual std::string Info() const {  // Component::Info, nnet-component-itf.cc:283-288
    std::ostringstream os;
    os << Type() << ", input-dim=" << InputDim() << ", output-dim=" << OutputDim();
    return os.str();
  }
  // general components only (kReordersIndexes): simple components have none (nnet-convolutional-component.h:163-181)
  virtual void GetInputIndexes(const MiscComputationInfo &, const Index &output_index, std::vector<Index> *desired) const {
    desired->assign(1, output_index);  // nnet-component-itf.cc:290-295
  }
  virtual void ReorderIndexes(std::vector<Index> *, std::vector<Index> *) const {}
  virtual ComponentPrecomputedIndexes *PrecomputeIndexes(const MiscComputationInfo &, const std::vector<Index> &, const std::vector<Index> &,
                                                         bool /*need_backprop*/) const {
    return nullptr;
  }
  // ReadNew (nnet-component-itf.cc:105-118): "<Type>" then that type's Read
  static Component *ReadNew(std::istream &is, bool binary) {
    tdnnf_kaldi_io::In in{is, binary, std::string()};
    std::string tok;
    if (!in.token(&tok) || tok.size() < 3 || tok[0] != '<' || tok[tok.size() - 1] != '>') TDNNF_ADAPTER_FAIL("tdnnf_nnet3: ReadNew: expected <ComponentType>");
    Component *c = NewComponentOfType(tok.substr(1, tok.size() - 2));
    if (!c) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: ReadNew: unknown component type " + tok);
    c->Read(is, binary);
    return c;
  }

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) = 0;

 public:
  virtual void *Propagate(const ComponentPrecomputedIndexes *indexes, const CuMatrixBase &in, CuMatrixBase *out) const = 0;
  virtual void Backprop(const std::string &debug_info, const ComponentPrecomputedIndexes *indexes, const CuMatrixBase &in_value,
                        const CuMatrixBase &out_value, const CuMatrixBase &out_deriv, void *memo, Component *to_update,
                        CuMatrixBase *in_deriv) const = 0;
  virtual void DeleteMemo(void *memo) const { (void)memo; }
  virtual void StoreStats(const CuMatrixBase &in_value, const CuMatrixBase &out_value, void *memo) { (void)in_value; (void)out_value; (void)memo; }
  static Component *NewComponentOfType(const std::string &type);
};

// UpdatableComponent: learning rates, is_gradient_, the natural-gradient configuration, and the parameter-vector virtuals over the
// component's parameter blocks (nnet-component-itf.cc:310-414; per class e.g. nnet-tdnn-component.cc:907-982)
class UpdatableComponent : public Component {
 public:
  UpdatableComponent()
      : learning_rate_(0.001f), learning_rate_factor_(1.0f), l2_regularize_(0.0f), max_change_(0.0f), is_gradient_(false), use_natural_gradient_(true),
        rank_in_(-1), rank_out_(-1), update_period_(4), num_samples_history_(2000.0f), alpha_in_(4.0f), alpha_out_(4.0f), ng_frozen_(false) {
    ng_.in = ng_.out = nullptr;
  }
  ~UpdatableComponent() {
    tdnnf_ng_destroy(ng_.in);
    tdnnf_ng_destroy(ng_.out);
  }
  void SetUnderlyingLearningRate(BaseFloat lr) { learning_rate_ = lr * learning_rate_factor_; }
  void SetActualLearningRate(BaseFloat lr) { learning_rate_ = lr; }
  void SetLearningRateFactor(BaseFloat f) { learning_rate_factor_ = f; }
  void SetAsGradient() { is_gradient_ = true; learning_rate_ = 1.0f; }
  void SetUseNaturalGradient(bool b) { use_natural_gradient_ = b; }
  BaseFloat LearningRate() const { return learning_rate_; }
  BaseFloat LearningRateFactor() const { return learning_rate_factor_; }
  BaseFloat MaxChange() const { return max_change_; }
  BaseFloat L2Regularization() const { return l2_regularize_; }
  // ---- the parameter vector (blocks in Vectorize order: linear parameters row by row, then the bias)
  virtual int32 NumParameters() const {
    std::vector<Block> b;
    ParamBlocks(&b);
    size_t n = 0;
    for (size_t i = 0; i < b.size(); i++) n += b[i].n;
    return (int32)n;
  }
  virtual void Scale(BaseFloat scale) {  // scale == 0: SetZero (also of NaNs), as every reference Scale
    std::vector<Block> b;
    ParamBlocks(&b);
    for (size_t i = 0; i < b.size(); i++) {
      if (!b[i].n) continue;
      if (scale == 0.0f) {
        std::vector<float> z(b[i].n, 0.0f);
        if (!Hooks().h2d) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::h2d is not installed");
        Hooks().h2d(b[i].dev, z.data(), sizeof(float) * b[i].n);
      } else {
        tdnnf_mat m = {b[i].dev, 1, (int)b[i].n, (int)b[i].n};
        tdnnf_adapter::Check(tdnnf_sum_scaled(&m, scale, nullptr, 0.0f, &m, Hooks().stream));
      }
    }
  }
  virtual void Add(BaseFloat alpha, const Component &other_in) {
    const UpdatableComponent *other = dynamic_cast<const UpdatableComponent *>(&other_in);
    std::vector<Block> a, b;
    ParamBlocks(&a);
    if (!other || other->Type() != Type()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Add: component types differ");
    other->ParamBlocks(&b);
    if (a.size() != b.size()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Add: parameter layouts differ");
    for (size_t i = 0; i < a.size(); i++) {
      if (a[i].n != b[i].n) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Add: parameter dimensions differ");
      if (a[i].n) tdnnf_adapter::Check(tdnnf_axpy(b[i].dev, alpha, a[i].dev, a[i].n, Hooks().stream));
    }
  }
  virtual BaseFloat DotProduct(const UpdatableComponent &other) const {
    std::vector<Block> a, b;
    ParamBlocks(&a);
    other.ParamBlocks(&b);
    if (a.size() != b.size()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DotProduct: parameter layouts differ");
    double tot = 0;
    for (size_t i = 0; i < a.size(); i++) {
      if (a[i].n != b[i].n) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DotProduct: parameter dimensions differ");
      double d = 0;
      if (a[i].n) tdnnf_adapter::Check(tdnnf_dot(a[i].dev, b[i].dev, a[i].n, &d, Hooks().stream));
      tot += d;
    }
    return (BaseFloat)tot;
  }
  virtual void PerturbParams(BaseFloat stddev) {  // params += stddev * N(0, 1)
    std::vector<Block> b;
    ParamBlocks(&b);
    for (size_t i = 0; i < b.size(); i++) {
      if (!b[i].n) continue;
      DeviceArray<float> tmp;
      tmp.Upload(Randn(b[i].n, 1.0f));
      tdnnf_adapter::Check(tdnnf_axpy(tmp.Data(), stddev, b[i].dev, b[i].n, Hooks().stream));
      if (Hooks().d2h) {  // (the temporary is freed on return: wait for the launch that reads it)
        float probe;
        Hooks().d2h(&probe, b[i].dev, sizeof(float));
      }
    }
  }
  virtual void Vectorize(VectorBase *params) const {
    std::vector<Block> b;
    ParamBlocks(&b);
    if (params->Dim() != NumParameters()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Vectorize: dimension mismatch");
    size_t o = 0;
    for (size_t i = 0; i < b.size(); i++) {
      if (b[i].n) {
        if (!Hooks().d2h) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::d2h is not installed");
        Hooks().d2h(params->Data() + o, b[i].dev, sizeof(float) * b[i].n);
      }
      o += b[i].n;
    }
  }
  virtual void UnVectorize(const VectorBase &params) {
    std::vector<Block> b;
    ParamBlocks(&b);
    if (params.Dim() != NumParameters()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: UnVectorize: dimension mismatch");
    size_t o = 0;
    for (size_t i = 0; i < b.size(); i++) {
      if (b[i].n) {
        if (!Hooks().h2d) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: DeviceHooks::h2d is not installed");
        Hooks().h2d(b[i].dev, params.Data() + o, sizeof(float) * b[i].n);
      }
      o += b[i].n;
    }
  }
  virtual void FreezeNaturalGradient(bool freeze) {  // nnet-tdnn-component.cc:979-982
    ng_frozen_ = freeze;
    if (ng_.in) tdnnf_adapter::Check(tdnnf_ng_freeze(ng_.in, freeze ? 1 : 0));
    if (ng_.out) tdnnf_adapter::Check(tdnnf_ng_freeze(ng_.out, freeze ? 1 : 0));
  }
  virtual std::string Info() const {  // UpdatableComponent::Info, nnet-component-itf.cc:404-414
    std::ostringstream os;
    os << Type() << ", input-dim=" << InputDim() << ", output-dim=" << OutputDim() << ", learning-rate=" << learning_rate_;
    if (is_gradient_) os << ", is-gradient=true";
    if (l2_regularize_ != 0.0f) os << ", l2-regularize=" << l2_regularize_;
    if (learning_rate_factor_ != 1.0f) os << ", learning-rate-factor=" << learning_rate_factor_;
    if (max_change_ > 0.0f) os << ", max-change=" << max_change_;
    return os.str();
  }

 protected:
  struct Block {
    float *dev;
    size_t n;
  };
  virtual void ParamBlocks(std::vector<Block> *blocks) const = 0;
  // InitLearningRatesFromConfig, nnet-component-itf.cc:332-345
  void InitLearningRatesFromConfig(ConfigLine *cfl) {
    learning_rate_ = 0.001f;
    cfl->GetValue("learning-rate", &learning_rate_);
    learning_rate_factor_ = 1.0f;
    cfl->GetValue("learning-rate-factor", &learning_rate_factor_);
    max_change_ = 0.0f;
    cfl->GetValue("max-change", &max_change_);
    l2_regularize_ = 0.0f;
    cfl->GetValue("l2-regularize", &l2_regularize_);
    if (learning_rate_ < 0.0f || learning_rate_factor_ < 0.0f || max_change_ < 0.0f || l2_regularize_ < 0.0f)
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer " + cfl->WholeLine());
  }
  // WriteUpdatableCommon :385-402 / ReadUpdatableCommon :347-383
  void WriteUpdatableCommon(tdnnf_kaldi_io::Out &o) const {
    o.token("<" + Type() + ">");
    if (learning_rate_factor_ != 1.0f) { o.token("<LearningRateFactor>"); o.f32(learning_rate_factor_); }
    if (is_gradient_) { o.token("<IsGradient>"); o.boolean(is_gradient_); }
    if (max_change_ > 0.0f) { o.token("<MaxChange>"); o.f32(max_change_); }
    if (l2_regularize_ > 0.0f) { o.token("<L2Regularize>"); o.f32(l2_regularize_); }
    o.token("<LearningRate>");
    o.f32(learning_rate_);
  }
  static double Num(const tdnnf_kaldi_io::Parsed &p, const char *tok, double dflt) {
    std::map<std::string, double>::const_iterator it = p.num.find(tok);
    return it == p.num.end() ? dflt : it->second;
  }
  void ReadUpdatableCommon(const tdnnf_kaldi_io::Parsed &p) {
    learning_rate_factor_ = (float)Num(p, "<LearningRateFactor>", 1.0);
    is_gradient_ = Num(p, "<IsGradient>", 0.0) != 0.0;
    max_change_ = (float)Num(p, "<MaxChange>", 0.0);
    l2_regularize_ = (float)Num(p, "<L2Regularize>", 0.0);
    learning_rate_ = (float)Num(p, "<LearningRate>", 0.001);
  }
  // the two OnlineNaturalGradient objects (ranks / update period / history / alpha as configured, defaults nnet-tdnn-component.cc:183-210):
  // created on first use
  const tdnnf_adapter::NaturalGradient *Preconditioners(int spliced_input_dim, int output_dim) const;
  BaseFloat learning_rate_, learning_rate_factor_, l2_regularize_, max_change_;
  bool is_gradient_, use_natural_gradient_;
  int32 rank_in_, rank_out_, update_period_;
  BaseFloat num_samples_history_, alpha_in_, alpha_out_;
  bool ng_frozen_;
  mutable tdnnf_adapter::NaturalGradient ng_;
  mutable Scratch ws_;
};
inline const tdnnf_adapter::NaturalGradient *UpdatableComponent::Preconditioners(int spliced, int out_dim) const {
  if (is_gradient_ || !use_natural_gradient_) return nullptr;  // "if (to_update->is_gradient_ || !to_update->use_natural_gradient_) UpdateSimple"
  if (!ng_.in) {
    const int rank_in = rank_in_ >= 0 ? rank_in_ : std::min(20, (spliced + 1) / 2), rank_out = rank_out_ >= 0 ? rank_out_ : std::min(80, (out_dim + 1) / 2);
    tdnnf_adapter::Check(tdnnf_ng_create(rank_in, update_period_, num_samples_history_, alpha_in_, &ng_.in));
    tdnnf_adapter::Check(tdnnf_ng_create(rank_out, update_period_, num_samples_history_, alpha_out_, &ng_.out));
    if (ng_frozen_) {
      tdnnf_adapter::Check(tdnnf_ng_freeze(ng_.in, 1));
      tdnnf_adapter::Check(tdnnf_ng_freeze(ng_.out, 1));
    }
  }
  return &ng_;
}
inline std::string ParamRms(const char *name, const float *dev, size_t n) {  // (a short form of PrintParameterStats: the rms)
  if (!n || !Hooks().d2h) return std::string();
  const std::vector<float> h = Download(dev, n);
  double ss = 0;
  for (size_t i = 0; i < n; i++) ss += (double)h[i] * h[i];
  std::ostringstream os;
  os << ", " << name << "-rms=" << sqrt(ss / (double)n);
  return os.str();
}

// "matrix=<rxfilename>" of the affine components' InitFromConfig (ReadKaldiObject): a Kaldi matrix file, text or binary ("\0B" header)
inline void ReadKaldiMatrixFile(const std::string &path, std::vector<float> *m, int *rows, int *cols) {
  std::ifstream is(path.c_str(), std::ios::in | std::ios::binary);
  if (!is.good()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: cannot open matrix file " + path);
  bool binary = false;
  if (is.peek() == 0) {
    is.get();
    if (is.get() != 'B') TDNNF_ADAPTER_FAIL("tdnnf_nnet3: bad binary header in " + path);
    binary = true;
  }
  tdnnf_kaldi_io::In in{is, binary, std::string()};
  if (!in.mat(m, rows, cols)) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: reading " + path + ": " + in.err);
}
// NonlinearComponent::Write (nnet-component-itf.cc:630-686) for the classes that keep [count, value_sum[D], deriv_sum[D]] or nothing
inline void WriteNonlinear(std::ostream &os, bool binary, const std::string &type, int dim, const std::vector<double> &stats, float self_repair_scale) {
  tdnnf_kaldi_io::Out o{os, binary};
  o.token("<" + type + ">");
  o.token("<Dim>"); o.i32(dim);
  std::vector<float> va, da;
  double count = 0;
  if (stats.size() >= (size_t)(1 + 2 * dim) && stats[0] != 0) {
    count = stats[0];
    va.resize(dim);
    da.resize(dim);
    for (int d = 0; d < dim; d++) {
      va[d] = (float)(stats[1 + d] / count);
      da[d] = (float)(stats[1 + dim + d] / count);
    }
  }
  o.token("<ValueAvg>"); o.vec(va.data(), (int)va.size());
  o.token("<DerivAvg>"); o.vec(da.data(), (int)da.size());
  o.token("<Count>"); o.f64(count);
  o.token("<OderivRms>"); o.vec(nullptr, 0);
  o.token("<OderivCount>"); o.f64(0.0);
  o.token("<NumDimsSelfRepaired>"); o.f64(0.0);
  o.token("<NumDimsProcessed>"); o.f64(0.0);
  if (self_repair_scale != 0.0f) { o.token("<SelfRepairScale>"); o.f32(self_repair_scale); }
  o.token("</" + type + ">");
}
inline double PNum(const tdnnf_kaldi_io::Parsed &p, const char *tok, double dflt) {
  std::map<std::string, double>::const_iterator it = p.num.find(tok);
  return it == p.num.end() ? dflt : it->second;
}

// ------------------------------------------------------------------------------------------------ Tdnn / TdnnDARTSV3
class TdnnPrecomputedIndexes : public ComponentPrecomputedIndexes {  // nnet-convolutional-component.h:208-218
 public:
  int32 row_stride;
  std::vector<int32> row_offsets;
};

// ConvolutionComputationIo for a Tdnn component (UPSTREAM time_height_convolution::GetComputationIo, used at
// nnet-tdnn-component.cc:851-867): the regular (start, step, count) grids of the t values of the input and output indexes and the
// number of (n, x) "images"; t_step 0 = a single t value.
struct TdnnComputationIo {
  int32 start_t_in, t_step_in, num_t_in, start_t_out, t_step_out, num_t_out, num_images, reorder_t_in;
  std::vector<std::pair<int32, int32> > images;  // sorted (n, x)
};
inline int32 Gcd(int32 a, int32 b) {
  while (b) {
    const int32 t = a % b;
    a = b;
    b = t;
  }
  return a < 0 ? -a : a;
}
inline void TimeGrid(const std::vector<Index> &ix, int32 *start, int32 *step, int32 *num) {
  std::set<int32> ts;
  for (size_t i = 0; i < ix.size(); i++)
    if (ix[i].t != kNoTime) ts.insert(ix[i].t);
  if (ts.empty()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: no valid t values in the indexes");
  *start = *ts.begin();
  int32 g = 0;
  for (std::set<int32>::const_iterator it = ts.begin(); it != ts.end(); ++it) g = Gcd(g, *it - *start);
  *step = g;
  *num = g == 0 ? 1 : (*ts.rbegin() - *start) / g + 1;
}
inline void GetComputationIo(const std::vector<Index> &in, const std::vector<Index> &out, TdnnComputationIo *io) {
  TimeGrid(in, &io->start_t_in, &io->t_step_in, &io->num_t_in);
  TimeGrid(out, &io->start_t_out, &io->t_step_out, &io->num_t_out);
  std::set<std::pair<int32, int32> > im;
  for (size_t i = 0; i < out.size(); i++) im.insert(std::make_pair(out[i].n, out[i].x));
  io->images.assign(im.begin(), im.end());
  io->num_images = (int32)io->images.size();
  io->reorder_t_in = 1;
}
// TdnnDARTSV3Component::ModifyComputationIo, nnet-tdnn-component.cc:822-844
inline void ModifyComputationIo(TdnnComputationIo *io) {
  if (io->t_step_out == 0) {
    if (io->t_step_in == 0) io->t_step_in = 1;
    io->t_step_out = io->t_step_in;
  }
  if (io->t_step_in == 0) io->t_step_in = io->t_step_out;
  if (io->t_step_out % io->t_step_in != 0) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: output t step is not a multiple of the input t step");
  io->reorder_t_in = io->t_step_out / io->t_step_in;
  const int32 n = io->reorder_t_in;
  io->num_t_in = n * ((io->num_t_in + n - 1) / n);  // rounded up to a multiple of reorder_t_in (:841-843)
}

// TdnnDARTSV3Component (src/nnet3/nnet-tdnn-component.cc:214-626); with darts == false the plain TdnnComponent (UPSTREAM)
class TdnnComponentBase : public UpdatableComponent {
 public:
  explicit TdnnComponentBase(bool darts)
      : darts_(darts), K_(0), Di_(0), Do_(0), linear_(nullptr), bias_(nullptr), flags_(0), temp_(1.0f), offsets1_positive_(true), update_theta_(true),
        orthonormal_constraint_(0.0f) {}
  // linear_params_ (Do x K Di, dense) and bias_params_ (DARTS: K logits then Do biases; plain: Do or null) on the device
  void SetParams(int32 K, int32 Di, int32 Do, float *linear, float *bias, bool offsets1_positive) {
    K_ = K; Di_ = Di; Do_ = Do; linear_ = linear; bias_ = bias; offsets1_positive_ = offsets1_positive;
    if ((int32)time_offsets_.size() != K) {  // (callers that bind buffers give no offsets: only the sign of the second one matters to Propagate)
      time_offsets_.resize(K);
      for (int32 i = 0; i < K; i++) time_offsets_[i] = offsets1_positive ? i : i - (K - 1);
    }
  }
  const std::vector<int32> &TimeOffsets() const { return time_offsets_; }
  BaseFloat OrthonormalConstraint() const { return orthonormal_constraint_; }
  const float *LinearParams() const { return linear_; }
  const float *BiasParams() const { return bias_; }
  // InitFromConfig, nnet-tdnn-component.cc:109-211 (the plain TdnnComponent, UPSTREAM: the same without the mode flags and logits)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    InitLearningRatesFromConfig(cfl);
    std::string offs;
    int32 input_dim = -1, output_dim = -1;
    const bool ok = cfl->GetValue("time-offsets", &offs) && cfl->GetValue("input-dim", &input_dim) && cfl->GetValue("output-dim", &output_dim);
    if (!ok || input_dim <= 0 || output_dim <= 0 || !SplitStringToIntegers(offs, &time_offsets_) || time_offsets_.empty())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer: there is a problem with time-offsets, input-dim or output-dim (not defined?): " + cfl->WholeLine());
    if (std::set<int32>(time_offsets_.begin(), time_offsets_.end()).size() != time_offsets_.size())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer: repeated time-offsets: " + cfl->WholeLine());
    if (darts_ && time_offsets_.size() < 2) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: TdnnDARTSV3Component needs at least two time offsets (:232 reads time_offsets_[1])");
    orthonormal_constraint_ = 0.0f;
    BaseFloat param_stddev = -1.0f, bias_mean = 0.0f, bias_stddev = 1.0f;
    bool use_bias = true;
    cfl->GetValue("param-stddev", &param_stddev);
    cfl->GetValue("bias-stddev", &bias_stddev);
    cfl->GetValue("bias-mean", &bias_mean);
    cfl->GetValue("use-bias", &use_bias);
    cfl->GetValue("orthonormal-constraint", &orthonormal_constraint_);
    const int32 K = (int32)time_offsets_.size();
    if (param_stddev < 0.0f) param_stddev = 1.0f / sqrtf((float)(input_dim * K));
    bool use_gumbel = true, use_entropy = true, free_select = true, update_alpha = true, uniform_sample = true;  // (:154-160: "temp" defaults, all true)
    update_theta_ = true;
    temp_ = 1.0f;
    if (darts_) {
      cfl->GetValue("use-gumbel", &use_gumbel);
      cfl->GetValue("use-entropy", &use_entropy);
      cfl->GetValue("free-select", &free_select);
      cfl->GetValue("update-alpha", &update_alpha);
      cfl->GetValue("update-theta", &update_theta_);
      cfl->GetValue("uniform-sample", &uniform_sample);
      cfl->GetValue("Temp-Proportion", &temp_);
      flags_ = (use_gumbel ? TDNNF_DARTS_USE_GUMBEL : 0) | (use_entropy ? TDNNF_DARTS_USE_ENTROPY : 0) | (free_select ? TDNNF_DARTS_FREE_SELECT : 0) |
               (update_alpha ? TDNNF_DARTS_UPDATE_ALPHA : 0) | (uniform_sample ? TDNNF_DARTS_UNIFORM_SAMPLE : 0);
    }
    std::vector<float> lin = Randn((size_t)output_dim * input_dim * K, param_stddev), bias;
    if (use_bias) {
      bias = Randn((size_t)output_dim + (darts_ ? K : 0), bias_stddev, bias_mean);
      if (darts_)
        for (int32 i = 0; i < K; i++) bias[i] = 0.0f;  // the K architecture logits start at zero (:176)
    }
    use_natural_gradient_ = true;
    rank_in_ = rank_out_ = -1;
    alpha_in_ = alpha_out_ = 4.0f;
    num_samples_history_ = 2000.0f;
    cfl->GetValue("use-natural-gradient", &use_natural_gradient_);
    cfl->GetValue("rank-in", &rank_in_);
    cfl->GetValue("rank-out", &rank_out_);
    cfl->GetValue("alpha-in", &alpha_in_);
    cfl->GetValue("alpha-out", &alpha_out_);
    cfl->GetValue("num-samples-history", &num_samples_history_);
    update_period_ = 4;  // (:208-209: not configurable)
    if (cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Could not process these elements in initializer: " + cfl->UnusedValues());
    Own(K, input_dim, output_dim, lin, bias);
  }
  // Write, nnet-tdnn-component.cc:659-700
  virtual void Write(std::ostream &os, bool binary) const {
    tdnnf_kaldi_io::Out o{os, binary};
    WriteUpdatableCommon(o);
    if (darts_) {
      o.token("<use-gumbel>"); o.boolean(flags_ & TDNNF_DARTS_USE_GUMBEL);
      o.token("<use-entropy>"); o.boolean(flags_ & TDNNF_DARTS_USE_ENTROPY);
      o.token("<free-select>"); o.boolean(flags_ & TDNNF_DARTS_FREE_SELECT);
      o.token("<update-alpha>"); o.boolean(flags_ & TDNNF_DARTS_UPDATE_ALPHA);
      o.token("<update-theta>"); o.boolean(update_theta_);
      o.token("<uniform-sample>"); o.boolean(flags_ & TDNNF_DARTS_UNIFORM_SAMPLE);
      o.token("<Temp-Proportion>"); o.f32(temp_);
    }
    o.token("<TimeOffsets>");
    o.intvec(time_offsets_);
    const std::vector<float> lin = Download(linear_, (size_t)Do_ * K_ * Di_), bias = Download(bias_, BiasDim());
    o.token("<LinearParams>");
    o.mat(lin.data(), Do_, K_ * Di_, K_ * Di_);
    o.token("<BiasParams>");
    o.vec(bias.data(), (int)bias.size());
    o.token("<OrthonormalConstraint>"); o.f32(orthonormal_constraint_);
    o.token("<UseNaturalGradient>"); o.boolean(use_natural_gradient_);
    o.token("<NumSamplesHistory>"); o.f32(num_samples_history_);
    o.token("<AlphaInOut>"); o.f32(alpha_in_); o.f32(alpha_out_);
    o.token("<RankInOut>"); o.i32(RankIn()); o.i32(RankOut());
    o.token("</" + Type() + ">");
  }
  virtual std::string Info() const {  // :77-106 (parameter statistics reduced to the rms)
    std::ostringstream os;
    os << UpdatableComponent::Info();
    if (orthonormal_constraint_ != 0.0f) os << ", orthonormal-constraint=" << orthonormal_constraint_;
    os << ", time-offsets=";
    for (size_t i = 0; i < time_offsets_.size(); i++) os << (i ? "," : "") << time_offsets_[i];
    os << ParamRms("linear-params", linear_, (size_t)Do_ * K_ * Di_);
    if (!bias_) os << ", has-bias=false";
    else os << ParamRms("bias", bias_, BiasDim());
    if (!use_natural_gradient_) os << ", use-natural-gradient=false";
    else os << ", rank-in=" << RankIn() << ", rank-out=" << RankOut() << ", num-samples-history=" << num_samples_history_ << ", update-period=" << update_period_
            << ", alpha-in=" << alpha_in_ << ", alpha-out=" << alpha_out_;
    return os.str();
  }
  // GetInputIndexes :762-775, PrecomputeIndexes :846-905, ReorderIndexes (UPSTREAM: the regular grid of GetIndexesForComputation)
  virtual void GetInputIndexes(const MiscComputationInfo &, const Index &output_index, std::vector<Index> *desired) const {
    desired->resize(time_offsets_.size());
    for (size_t i = 0; i < time_offsets_.size(); i++) (*desired)[i] = Index(output_index.n, output_index.t + time_offsets_[i], output_index.x);
  }
  virtual ComponentPrecomputedIndexes *PrecomputeIndexes(const MiscComputationInfo &, const std::vector<Index> &input_indexes,
                                                         const std::vector<Index> &output_indexes, bool) const {
    TdnnComputationIo io;
    GetComputationIo(input_indexes, output_indexes, &io);
    ModifyComputationIo(&io);
    TdnnPrecomputedIndexes *ans = new TdnnPrecomputedIndexes();
    ans->row_stride = io.reorder_t_in;
    ans->row_offsets.resize(time_offsets_.size());
    for (size_t i = 0; i < time_offsets_.size(); i++) {
      const int32 required_input_t = io.start_t_out + time_offsets_[i], input_t = (required_input_t - io.start_t_in) / io.t_step_in;
      if (required_input_t != io.start_t_in + io.t_step_in * input_t || input_t < 0)
        TDNNF_ADAPTER_FAIL("tdnnf_nnet3: PrecomputeIndexes: a time offset does not land on the input grid");
      const int32 n = io.reorder_t_in, mult = n * (input_t / n), rem = input_t % n;
      ans->row_offsets[i] = mult * io.num_images + rem;  // :893-901
    }
    return ans;
  }
  virtual void ReorderIndexes(std::vector<Index> *input_indexes, std::vector<Index> *output_indexes) const {
    TdnnComputationIo io;
    GetComputationIo(*input_indexes, *output_indexes, &io);
    ModifyComputationIo(&io);
    std::set<Index> have_in(input_indexes->begin(), input_indexes->end()), have_out(output_indexes->begin(), output_indexes->end());
    const int32 B = io.num_images, rho = io.reorder_t_in;
    std::vector<Index> in((size_t)io.num_t_in * B), out((size_t)io.num_t_out * B);
    for (int32 tau = 0; tau < io.num_t_in; tau++)
      for (int32 b = 0; b < B; b++) {
        Index ix(io.images[b].first, io.start_t_in + tau * io.t_step_in, io.images[b].second);
        if (!have_in.count(ix)) ix.t = kNoTime;  // a blank the computation never reads
        in[(size_t)(tau / rho) * rho * B + (size_t)b * rho + tau % rho] = ix;  // blocks of rho time steps, image-major inside a block
      }
    for (int32 k = 0; k < io.num_t_out; k++)
      for (int32 b = 0; b < B; b++) {
        Index ix(io.images[b].first, io.start_t_out + k * io.t_step_out, io.images[b].second);
        if (!have_out.count(ix)) ix.t = kNoTime;
        out[(size_t)k * B + b] = ix;
      }
    input_indexes->swap(in);
    output_indexes->swap(out);
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

 protected:
  virtual void ParamBlocks(std::vector<Block> *b) const {  // Vectorize order :960-968: linear parameters, then the bias vector
    b->push_back(Block{linear_, (size_t)Do_ * K_ * Di_});
    b->push_back(Block{bias_, BiasDim()});
  }
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {  // Read :702-760
    ReadUpdatableCommon(p);
    if (p.offsets.empty() || p.rows <= 0 || p.cols <= 0 || p.cols % (int)p.offsets.size() != 0) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + "::Read: bad dimensions");
    time_offsets_ = p.offsets;
    const int32 K = (int32)time_offsets_.size();
    if (darts_) {
      flags_ = (Num(p, "<use-gumbel>", 0) != 0 ? TDNNF_DARTS_USE_GUMBEL : 0) | (Num(p, "<use-entropy>", 0) != 0 ? TDNNF_DARTS_USE_ENTROPY : 0) |
               (Num(p, "<free-select>", 0) != 0 ? TDNNF_DARTS_FREE_SELECT : 0) | (Num(p, "<update-alpha>", 0) != 0 ? TDNNF_DARTS_UPDATE_ALPHA : 0) |
               (Num(p, "<uniform-sample>", 0) != 0 ? TDNNF_DARTS_UNIFORM_SAMPLE : 0);
      update_theta_ = Num(p, "<update-theta>", 1) != 0;
      temp_ = (float)Num(p, "<Temp-Proportion>", 1.0);
    }
    orthonormal_constraint_ = (float)Num(p, "<OrthonormalConstraint>", 0.0);
    use_natural_gradient_ = Num(p, "<UseNaturalGradient>", 1) != 0;
    num_samples_history_ = (float)Num(p, "<NumSamplesHistory>", 2000.0);
    if (p.num.count("<AlphaInOut>")) {
      alpha_in_ = (float)Num(p, "<AlphaInOut>", 4.0);
      alpha_out_ = (float)Num(p, "<AlphaInOut>#2", 4.0);
    } else {  // (the older format :733-737)
      alpha_in_ = alpha_out_ = (float)Num(p, "<Alpha>", 4.0);
    }
    rank_in_ = (int32)Num(p, "<RankInOut>", -1);
    rank_out_ = (int32)Num(p, "<RankInOut>#2", -1);
    update_period_ = 4;
    const size_t want_bias = (size_t)p.rows + (darts_ ? K : 0);
    if (!p.b.empty() && p.b.size() != want_bias) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + "::Read: bias dimension mismatch (Check(), :64-74)");
    Own(K, p.cols / K, p.rows, p.W, p.b);
  }

 private:
  size_t BiasDim() const { return bias_ ? (size_t)Do_ + (darts_ ? K_ : 0) : 0; }
  int32 RankIn() const { return rank_in_ >= 0 ? rank_in_ : std::min(20, (K_ * Di_ + 1) / 2); }
  int32 RankOut() const { return rank_out_ >= 0 ? rank_out_ : std::min(80, (Do_ + 1) / 2); }
  void Own(int32 K, int32 Di, int32 Do, const std::vector<float> &lin, const std::vector<float> &bias) {
    std::vector<float> all(lin);
    all.insert(all.end(), bias.begin(), bias.end());
    own_.Upload(all);
    K_ = K; Di_ = Di; Do_ = Do;
    linear_ = own_.Data();
    bias_ = bias.empty() ? nullptr : own_.Data() + lin.size();
    offsets1_positive_ = K >= 2 ? time_offsets_[1] > 0 : true;
  }
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
  bool update_theta_;  // stored and serialised, never read
  std::vector<int32> time_offsets_;
  BaseFloat orthonormal_constraint_;
  DeviceArray<float> own_;
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
  BatchNormComponent() : dim_(0), epsilon_(1.0e-3f), target_rms_(1.0f), stats_(nullptr), test_mode_(false) {}
  void Init(int32 dim, float epsilon, float target_rms, double *stats_dev /* [count, sum[D], sumsq[D]] */) {
    dim_ = dim; epsilon_ = epsilon; target_rms_ = target_rms; stats_ = stats_dev;
  }
  // InitFromConfig, nnet-normalize-component.cc:288-310 (block-dim == dim only: the recipes' use)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    int32 dim = -1, block_dim = -1;
    epsilon_ = 1.0e-3f;
    target_rms_ = 1.0f;
    test_mode_ = false;
    const bool ok = cfl->GetValue("dim", &dim);
    cfl->GetValue("block-dim", &block_dim);
    cfl->GetValue("epsilon", &epsilon_);
    cfl->GetValue("target-rms", &target_rms_);
    cfl->GetValue("test-mode", &test_mode_);
    if (!ok || dim <= 0) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: BatchNormComponent must have 'dim' specified, and > 0");
    if (block_dim == -1) block_dim = dim;
    if (block_dim != dim || epsilon_ <= 0.0f || target_rms_ <= 0.0f) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid configuration in BatchNormComponent (block-dim must equal dim here)");
    if (test_mode_) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: BatchNormComponent test-mode=true: use BatchNormTestComponent (the recipes' sed, ...cvupdate.sh:133)");
    if (cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Could not process these elements in initializer: " + cfl->UnusedValues());
    OwnStats(dim, std::vector<double>((size_t)1 + 2 * dim, 0.0));
  }
  // Write :616-642: <Count> and the statistics as mean / variance
  virtual void Write(std::ostream &os, bool binary) const { WriteBatchNorm(os, binary, Type(), dim_, epsilon_, target_rms_, false, Download(stats_, (size_t)1 + 2 * dim_)); }
  static void WriteBatchNorm(std::ostream &os, bool binary, const std::string &type, int D, float eps, float rms, bool test, const std::vector<double> &st) {
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<" + type + ">");
    o.token("<Dim>"); o.i32(D);
    o.token("<BlockDim>"); o.i32(D);
    o.token("<Epsilon>"); o.f32(eps);
    o.token("<TargetRms>"); o.f32(rms);
    o.token("<TestMode>"); o.boolean(test);
    o.token("<Count>"); o.f64(st[0]);
    std::vector<float> mean(D), var(D);
    for (int d = 0; d < D; d++) {
      const double m = st[0] != 0 ? st[1 + d] / st[0] : st[1 + d];
      mean[d] = (float)m;
      var[d] = (float)(st[0] != 0 ? st[1 + D + d] / st[0] - m * m : st[1 + D + d]);
    }
    o.token("<StatsMean>"); o.vec(mean.data(), D);
    o.token("<StatsVar>"); o.vec(var.data(), D);
    o.token("</" + type + ">");
  }
  // Read :591-614: stats_sum = mean * count, stats_sumsq = (var + mean^2) * count
  static std::vector<double> StatsFromParsed(const tdnnf_kaldi_io::Parsed &p, int *dim) {
    const int D = (int)PNum(p, "<Dim>", 0);
    if (D <= 0 || (int)PNum(p, "<BlockDim>", D) != D) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: BatchNorm Read: bad <Dim> / <BlockDim>");
    if (!p.mean.empty() && ((int)p.mean.size() != D || (int)p.var.size() != D)) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: BatchNorm Read: statistics of the wrong dimension");
    std::vector<double> st((size_t)1 + 2 * D, 0.0);
    st[0] = p.count;
    for (int d = 0; d < D && !p.mean.empty(); d++) {
      st[1 + d] = (double)p.mean[d] * p.count;
      st[1 + D + d] = ((double)p.var[d] + (double)p.mean[d] * p.mean[d]) * p.count;
    }
    *dim = D;
    return st;
  }
  double *Stats() const { return stats_; }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    int D = 0;
    const std::vector<double> st = StatsFromParsed(p, &D);
    epsilon_ = (float)PNum(p, "<Epsilon>", 1.0e-3);
    target_rms_ = (float)PNum(p, "<TargetRms>", 1.0);
    test_mode_ = PNum(p, "<TestMode>", 0) != 0;
    OwnStats(D, st);
  }

 private:
  void OwnStats(int32 dim, const std::vector<double> &st) {
    own_stats_.Upload(st);
    dim_ = dim;
    stats_ = own_stats_.Data();
  }
  int32 dim_;
  float epsilon_, target_rms_;
  double *stats_;
  bool test_mode_;
  DeviceArray<double> own_stats_;
  mutable Scratch ws_;
};
// nnet-normalize-component.cc:682-922: frozen statistics; scale_ / offset_ from ComputeDerived
class BatchNormTestComponent : public Component {
 public:
  BatchNormTestComponent() : dim_(0), scale_(nullptr), offset_(nullptr), epsilon_(1.0e-3f), target_rms_(1.0f) {}
  void Init(int32 dim, float epsilon, float target_rms, const double *stats_dev) {
    if (scale_ && Hooks().free) Hooks().free(scale_);
    dim_ = dim;
    epsilon_ = epsilon;
    target_rms_ = target_rms;
    host_stats_ = Hooks().d2h ? Download(stats_dev, (size_t)1 + 2 * dim) : std::vector<double>();  // (kept for Write)
    scale_ = static_cast<float *>(DeviceAlloc(sizeof(float) * 2 * dim));
    offset_ = scale_ + dim;
    tdnnf_adapter::BatchNormComputeDerived(stats_dev, dim, epsilon, target_rms, scale_, offset_, Hooks().stream);
  }
  // InitFromConfig is EMPTY in the reference (nnet-normalize-component.cc:757-759): the component only comes into being by reading a
  // trained BatchNormComponent whose type name a sed has changed (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:133)
  virtual void InitFromConfig(ConfigLine *cfl) { DropNameAndType(cfl); }
  virtual void Write(std::ostream &os, bool binary) const {  // :956-982
    if (host_stats_.empty()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: BatchNormTestComponent::Write before any statistics were given");
    BatchNormComponent::WriteBatchNorm(os, binary, Type(), dim_, epsilon_, target_rms_, true, host_stats_);
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {  // Read :931-955, then ComputeDerived :682-715
    int D = 0;
    const std::vector<double> st = BatchNormComponent::StatsFromParsed(p, &D);
    DeviceArray<double> dev;
    dev.Upload(st);
    Init(D, (float)PNum(p, "<Epsilon>", 1.0e-3), (float)PNum(p, "<TargetRms>", 1.0), dev.Data());
    host_stats_ = st;
    if (Hooks().d2h) {  // (the upload is freed on return: wait for ComputeDerived)
      float probe;
      Hooks().d2h(&probe, scale_, sizeof(float));
    }
  }

 private:
  int32 dim_;
  float *scale_, *offset_;
  float epsilon_, target_rms_;
  std::vector<double> host_stats_;
};

// ------------------------------------------------------------------------------------------------ the DARTS mixing operators
// (Gumbel)SoftmaxFlopsComponent nnet-simple-component.cc:9968-10020, :10088-10158 (dim must be 8: the hard-coded FLOPs vector)
class SoftmaxFlopsComponentBase : public Component {
 public:
  explicit SoftmaxFlopsComponentBase(bool gumbel) : gumbel_(gumbel), dim_(8), scale_(0.f), temp_(1.0f), flops_(nullptr) {}
  void Init(int32 dim, float scale, const float *flops_dev /* dim floats: -(25, 50, 80, 100, 120, 160, 200, 240) */) { dim_ = dim; scale_ = scale; flops_ = flops_dev; }
  void SetTempProportion(BaseFloat p) { temp_ = p; }
  // InitFromConfig :9932-9950 / :10040-10060: dim, scale [, temp-proportion]; the FLOPs vector is hard-coded for dim == 8 (:10006-10016)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    int32 dim = -1;
    float scale = 0.0f;
    bool ok = cfl->GetValue("dim", &dim) && cfl->GetValue("scale", &scale);
    if (gumbel_) ok = ok && cfl->GetValue("temp-proportion", &temp_);
    if (!ok || dim <= 0 || cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
    OwnFlops(dim, scale);
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :9951-9960 / :10160-10187
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<" + Type() + ">");
    o.token("<Dim>"); o.i32(dim_);
    o.token("<Scale>"); o.f32(scale_);
    if (gumbel_) { o.token("<TempProportion>"); o.f32(temp_); }
    o.token("</" + Type() + ">");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    temp_ = (float)PNum(p, "<TempProportion>", 1.0);
    OwnFlops((int32)PNum(p, "<Dim>", 0), (float)PNum(p, "<Scale>", 0.0));
  }

 private:
  void OwnFlops(int32 dim, float scale) {
    if (dim != 8) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + ": dim must be 8 (the hard-coded FLOPs vector, nnet-simple-component.cc:10006-10016)");
    const float f[8] = {-25.f, -50.f, -80.f, -100.f, -120.f, -160.f, -200.f, -240.f};
    own_flops_.Upload(std::vector<float>(f, f + 8));
    dim_ = dim;
    scale_ = scale;
    flops_ = own_flops_.Data();
  }
  bool gumbel_;
  int32 dim_;
  float scale_, temp_;
  const float *flops_;
  DeviceArray<float> own_flops_;
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
  // InitFromConfig :2734-2760 (Onehot: the same keys, :9554-9590)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    InitLearningRatesFromConfig(cfl);
    int32 output_dim = 0, input_dim = 0;
    const bool ok = cfl->GetValue("output-dim", &output_dim) && cfl->GetValue("input-dim", &input_dim);
    is_updatable_ = true;
    use_natural_gradient_ = true;
    cfl->GetValue("is-updatable", &is_updatable_);
    cfl->GetValue("use-natural-gradient", &use_natural_gradient_);
    BaseFloat output_mean = 0.0f, output_stddev = 0.0f;
    cfl->GetValue("output-mean", &output_mean);
    cfl->GetValue("output-stddev", &output_stddev);
    if (!ok || cfl->HasUnusedValues() || input_dim <= 0 || output_dim <= 0)
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer " + cfl->WholeLine());
    if (use_natural_gradient_) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + ": use-natural-gradient=true is not served (the recipes write false, add_flopsconstraint.py:20)");
    in_dim_ = input_dim;
    OwnOutput(Randn((size_t)output_dim, output_stddev, output_mean));
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :2683-2694 / :9593-9604
    tdnnf_kaldi_io::Out o{os, binary};
    WriteUpdatableCommon(o);
    o.token("<InputDim>"); o.i32(in_dim_);
    const std::vector<float> out = Download(output_, (size_t)out_dim_);
    o.token("<Output>"); o.vec(out.data(), out_dim_);
    o.token("<IsUpdatable>"); o.boolean(is_updatable_);
    o.token("<UseNaturalGradient>"); o.boolean(use_natural_gradient_);
    o.token("</" + Type() + ">");
  }
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

 protected:
  virtual void ParamBlocks(std::vector<Block> *b) const { b->push_back(Block{output_, (size_t)out_dim_}); }
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    ReadUpdatableCommon(p);
    in_dim_ = (int32)PNum(p, "<InputDim>", 0);
    is_updatable_ = PNum(p, "<IsUpdatable>", 1) != 0;
    use_natural_gradient_ = PNum(p, "<UseNaturalGradient>", 0) != 0;
    if (p.out_vec.empty() || in_dim_ <= 0) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + "::Read: missing <Output> or <InputDim>");
    OwnOutput(p.out_vec);
  }

 private:
  void OwnOutput(const std::vector<float> &v) {
    own_.Upload(v);
    out_dim_ = (int32)v.size();
    output_ = own_.Data();
  }
  bool onehot_;
  int32 in_dim_, out_dim_;
  float *output_;
  bool is_updatable_;
  DeviceArray<float> own_;
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
  virtual void InitFromConfig(ConfigLine *cfl) {  // :4799-4815
    DropNameAndType(cfl);
    scale_ = 1.0f;
    const bool ok = cfl->GetValue("input-dim", &in_dim_) && cfl->GetValue("output-dim", &out_dim_);
    cfl->GetValue("scale", &scale_);
    if (!ok || in_dim_ <= 0 || out_dim_ <= 0 || out_dim_ % in_dim_ != 0 || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid values in: " + cfl->WholeLine());
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :4824-4833
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<CopyNComponent>");
    o.token("<InputDim>"); o.i32(in_dim_);
    o.token("<OutputDim>"); o.i32(out_dim_);
    o.token("<Scale>"); o.f32(scale_);
    o.token("</CopyNComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    in_dim_ = (int32)PNum(p, "<InputDim>", 1);
    out_dim_ = (int32)PNum(p, "<OutputDim>", 1);
    scale_ = (float)PNum(p, "<Scale>", 1.0);
  }

 private:
  int32 in_dim_, out_dim_;
  float scale_;
};

class ElementwiseProductComponent : public Component {  // :256-299
 public:
  ElementwiseProductComponent() : in_dim_(0), out_dim_(0) {}
  void Init(int32 input_dim, int32 output_dim) { in_dim_ = input_dim; out_dim_ = output_dim; }
  virtual void InitFromConfig(ConfigLine *cfl) {  // :245-254
    DropNameAndType(cfl);
    const bool ok = cfl->GetValue("output-dim", &out_dim_) && cfl->GetValue("input-dim", &in_dim_);
    if (!ok || out_dim_ <= 0 || in_dim_ <= out_dim_ || in_dim_ % out_dim_ != 0 || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :310-317
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<ElementwiseProductComponent>");
    o.token("<InputDim>"); o.i32(in_dim_);
    o.token("<OutputDim>"); o.i32(out_dim_);
    o.token("</ElementwiseProductComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    in_dim_ = (int32)PNum(p, "<InputDim>", 0);
    out_dim_ = (int32)PNum(p, "<OutputDim>", 0);
  }

 private:
  int32 in_dim_, out_dim_;
};

// ------------------------------------------------------------------------------------------------ the standard layers around them
class RectifiedLinearComponent : public Component {  // :958-1091; statistics [count, value_sum[D], deriv_sum[D]] on the device
 public:
  RectifiedLinearComponent() : dim_(0), self_repair_scale_(0.f), stats_(nullptr) {}
  void Init(int32 dim, float self_repair_scale, double *stats_dev) { dim_ = dim; self_repair_scale_ = self_repair_scale; stats_ = stats_dev; }
  // NonlinearComponent::InitFromConfig, nnet-component-itf.cc:704-722 (the thresholds keep the ReLU's 0.05 / 0.95)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    int32 dim = -1, block_dim = -1;
    float lower = -1000.0f, upper = -1000.0f;
    self_repair_scale_ = 0.0f;
    const bool ok = cfl->GetValue("dim", &dim);
    cfl->GetValue("block-dim", &block_dim);
    cfl->GetValue("self-repair-lower-threshold", &lower);
    cfl->GetValue("self-repair-upper-threshold", &upper);
    cfl->GetValue("self-repair-scale", &self_repair_scale_);
    if (!ok || cfl->HasUnusedValues() || dim <= 0 || (block_dim != -1 && block_dim != dim))
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
    OwnStats(dim, std::vector<double>((size_t)1 + 2 * dim, 0.0));
    stored_once_ = false;
  }
  virtual void Write(std::ostream &os, bool binary) const { WriteNonlinear(os, binary, Type(), dim_, Download(stats_, (size_t)1 + 2 * dim_), self_repair_scale_); }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {  // NonlinearComponent::Read :580-628: averages back to sums
    const int D = (int)PNum(p, "<Dim>", 0);
    if (D <= 0) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: RectifiedLinearComponent::Read: bad <Dim>");
    std::vector<double> st((size_t)1 + 2 * D, 0.0);
    if ((int)p.value_avg.size() == D && (int)p.deriv_avg.size() == D) {
      st[0] = p.count;
      for (int d = 0; d < D; d++) {
        st[1 + d] = (double)p.value_avg[d] * p.count;
        st[1 + D + d] = (double)p.deriv_avg[d] * p.count;
      }
    }
    self_repair_scale_ = (float)PNum(p, "<SelfRepairScale>", 0.0);
    OwnStats(D, st);
    stored_once_ = st[0] != 0;
  }

 private:
  void OwnStats(int32 dim, const std::vector<double> &st) {
    own_stats_.Upload(st);
    dim_ = dim;
    stats_ = own_stats_.Data();
  }
  int32 dim_;
  float self_repair_scale_;
  double *stats_;
  bool repair_now_ = true, store_now_ = true, stored_once_ = false;
  DeviceArray<double> own_stats_;
  Scratch ws_;
};

// AffineComponent :1235-1279 with the natural-gradient Update of NaturalGradientAffineComponent :2980-3024; bias == null: LinearComponent :3211-3254
class AffineComponentBase : public UpdatableComponent {
 public:
  explicit AffineComponentBase(const char *type) : type_(type), in_dim_(0), out_dim_(0), linear_(nullptr), bias_(nullptr), orthonormal_constraint_(0.0f) {}
  void SetParams(int32 input_dim, int32 output_dim, float *linear_dev, float *bias_dev) { in_dim_ = input_dim; out_dim_ = output_dim; linear_ = linear_dev; bias_ = bias_dev; }
  BaseFloat OrthonormalConstraint() const { return orthonormal_constraint_; }
  const float *LinearParams() const { return linear_; }
  const float *BiasParams() const { return bias_; }
  // InitFromConfig: NaturalGradientAffineComponent :2854-2933, LinearComponent :3095-3153, AffineComponent :1200-1230
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    const bool is_linear = type_ == std::string("LinearComponent"), is_plain = type_ == std::string("AffineComponent");
    is_gradient_ = false;
    InitLearningRatesFromConfig(cfl);
    std::string matrix_filename;
    std::vector<float> lin, bias;
    int32 input_dim = -1, output_dim = -1;
    if (cfl->GetValue("matrix", &matrix_filename)) {
      std::vector<float> m;
      int rows = 0, cols = 0;
      ReadKaldiMatrixFile(matrix_filename, &m, &rows, &cols);
      if (rows <= 0 || cols < (is_linear ? 1 : 2)) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + ": bad matrix in " + matrix_filename);
      const int id = is_linear ? cols : cols - 1;
      lin.resize((size_t)rows * id);
      if (!is_linear) bias.resize(rows);
      for (int r = 0; r < rows; r++) {
        for (int c = 0; c < id; c++) lin[(size_t)r * id + c] = m[(size_t)r * cols + c];
        if (!is_linear) bias[r] = m[(size_t)r * cols + id];  // the last column is the bias (:2865-2869)
      }
      if (cfl->GetValue("input-dim", &input_dim) && input_dim != id) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: input-dim mismatch vs. matrix.");
      if (cfl->GetValue("output-dim", &output_dim) && output_dim != rows) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: output-dim mismatch vs. matrix.");
      input_dim = id;
      output_dim = rows;
    } else {
      if (!(cfl->GetValue("input-dim", &input_dim) && cfl->GetValue("output-dim", &output_dim)) || input_dim <= 0 || output_dim <= 0)
        TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer " + cfl->WholeLine());
      BaseFloat param_stddev = 1.0f / sqrtf((float)input_dim), bias_stddev = 1.0f, bias_mean = 0.0f;
      cfl->GetValue("param-stddev", &param_stddev);
      if (!is_linear) cfl->GetValue("bias-stddev", &bias_stddev);
      if (!is_linear && !is_plain) cfl->GetValue("bias-mean", &bias_mean);
      if (param_stddev < 0.0f || bias_stddev < 0.0f) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer " + cfl->WholeLine());
      lin = Randn((size_t)output_dim * input_dim, param_stddev);
      if (!is_linear) bias = Randn((size_t)output_dim, bias_stddev, bias_mean);
    }
    orthonormal_constraint_ = 0.0f;
    cfl->GetValue("orthonormal-constraint", &orthonormal_constraint_);
    if (!is_plain) {
      num_samples_history_ = 2000.0f;
      alpha_in_ = 4.0f;
      rank_in_ = rank_out_ = -1;
      update_period_ = 4;
      cfl->GetValue("num-samples-history", &num_samples_history_);
      cfl->GetValue("alpha", &alpha_in_);
      alpha_out_ = alpha_in_;
      cfl->GetValue("rank-in", &rank_in_);
      cfl->GetValue("rank-out", &rank_out_);
      cfl->GetValue("update-period", &update_period_);
      if (is_linear) {
        use_natural_gradient_ = true;
        cfl->GetValue("use-natural-gradient", &use_natural_gradient_);
      }
    }
    if (cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Could not process these elements in initializer: " + cfl->UnusedValues());
    Own(input_dim, output_dim, lin, bias);
  }
  // Write: NaturalGradientAffineComponent :2935-2958, LinearComponent :3161-3188, AffineComponent :1302-1313
  virtual void Write(std::ostream &os, bool binary) const {
    const bool is_linear = type_ == std::string("LinearComponent"), is_plain = type_ == std::string("AffineComponent");
    tdnnf_kaldi_io::Out o{os, binary};
    WriteUpdatableCommon(o);
    const std::vector<float> lin = Download(linear_, (size_t)out_dim_ * in_dim_), bias = Download(bias_, bias_ ? (size_t)out_dim_ : 0);
    if (is_linear) {
      o.token("<Params>"); o.mat(lin.data(), out_dim_, in_dim_, in_dim_);
      if (orthonormal_constraint_ != 0.0f) { o.token("<OrthonormalConstraint>"); o.f32(orthonormal_constraint_); }
      o.token("<UseNaturalGradient>"); o.boolean(use_natural_gradient_);
      o.token("<RankInOut>"); o.i32(RankIn()); o.i32(RankOut());
      o.token("<Alpha>"); o.f32(alpha_in_);
      o.token("<NumSamplesHistory>"); o.f32(num_samples_history_);
      o.token("<UpdatePeriod>"); o.i32(update_period_);
    } else {
      o.token("<LinearParams>"); o.mat(lin.data(), out_dim_, in_dim_, in_dim_);
      o.token("<BiasParams>"); o.vec(bias.data(), (int)bias.size());
      if (is_plain) {
        if (orthonormal_constraint_ != 0.0f) { o.token("<OrthonormalConstraint>"); o.f32(orthonormal_constraint_); }
      } else {
        o.token("<RankIn>"); o.i32(RankIn());
        o.token("<RankOut>"); o.i32(RankOut());
        if (orthonormal_constraint_ != 0.0f) { o.token("<OrthonormalConstraint>"); o.f32(orthonormal_constraint_); }
        o.token("<UpdatePeriod>"); o.i32(update_period_);
        o.token("<NumSamplesHistory>"); o.f32(num_samples_history_);
        o.token("<Alpha>"); o.f32(alpha_in_);
      }
    }
    o.token("</" + Type() + ">");
  }
  virtual std::string Info() const {
    std::ostringstream os;
    os << UpdatableComponent::Info();
    if (orthonormal_constraint_ != 0.0f) os << ", orthonormal-constraint=" << orthonormal_constraint_;
    os << ParamRms("linear-params", linear_, (size_t)out_dim_ * in_dim_);
    if (bias_) os << ParamRms("bias", bias_, (size_t)out_dim_);
    if (type_ != std::string("AffineComponent"))
      os << ", rank-in=" << RankIn() << ", rank-out=" << RankOut() << ", num-samples-history=" << num_samples_history_ << ", update-period=" << update_period_
         << ", alpha=" << alpha_in_;
    return os.str();
  }
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

 protected:
  virtual void ParamBlocks(std::vector<Block> *b) const {
    b->push_back(Block{linear_, (size_t)out_dim_ * in_dim_});
    b->push_back(Block{bias_, bias_ ? (size_t)out_dim_ : 0});
  }
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    const bool is_linear = type_ == std::string("LinearComponent");
    ReadUpdatableCommon(p);
    if (p.rows <= 0 || p.cols <= 0 || (!is_linear && (int)p.b.size() != p.rows)) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: " + Type() + "::Read: bad parameter dimensions");
    orthonormal_constraint_ = (float)PNum(p, "<OrthonormalConstraint>", 0.0);
    if (is_linear) {
      use_natural_gradient_ = PNum(p, "<UseNaturalGradient>", 1) != 0;
      rank_in_ = (int32)PNum(p, "<RankInOut>", -1);
      rank_out_ = (int32)PNum(p, "<RankInOut>#2", -1);
    } else {
      rank_in_ = (int32)PNum(p, "<RankIn>", -1);
      rank_out_ = (int32)PNum(p, "<RankOut>", -1);
    }
    alpha_in_ = alpha_out_ = (float)PNum(p, "<Alpha>", 4.0);
    num_samples_history_ = (float)PNum(p, "<NumSamplesHistory>", 2000.0);
    update_period_ = (int32)PNum(p, "<UpdatePeriod>", 4);
    Own(p.cols, p.rows, p.W, is_linear ? std::vector<float>() : p.b);
  }

 private:
  int32 RankIn() const { return rank_in_ >= 0 ? rank_in_ : std::min(20, (in_dim_ + 1) / 2); }
  int32 RankOut() const { return rank_out_ >= 0 ? rank_out_ : std::min(80, (out_dim_ + 1) / 2); }
  void Own(int32 input_dim, int32 output_dim, const std::vector<float> &lin, const std::vector<float> &bias) {
    std::vector<float> all(lin);
    all.insert(all.end(), bias.begin(), bias.end());
    own_.Upload(all);
    in_dim_ = input_dim;
    out_dim_ = output_dim;
    linear_ = own_.Data();
    bias_ = bias.empty() ? nullptr : own_.Data() + lin.size();
  }
  const char *type_;
  int32 in_dim_, out_dim_;
  float *linear_, *bias_;
  BaseFloat orthonormal_constraint_;
  DeviceArray<float> own_;
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
  // InitFromConfig :3345-3371: "matrix=<rxfilename>" ([W | b]) or, for tests, input-dim / output-dim with random values
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    std::string filename;
    std::vector<float> m;
    int rows = 0, cols = 0;
    if (cfl->GetValue("matrix", &filename)) {
      if (cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
      ReadKaldiMatrixFile(filename, &m, &rows, &cols);
    } else {
      int32 input_dim = -1, output_dim = -1;
      if (!cfl->GetValue("input-dim", &input_dim) || !cfl->GetValue("output-dim", &output_dim) || cfl->HasUnusedValues() || input_dim <= 0 || output_dim <= 0)
        TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
      rows = output_dim;
      cols = input_dim + 1;
      m = Randn((size_t)rows * cols, 1.0f);
    }
    if (rows <= 0 || cols < 2) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: FixedAffineComponent: bad matrix");
    std::vector<float> lin((size_t)rows * (cols - 1)), bias(rows);
    for (int r = 0; r < rows; r++) {
      for (int c = 0; c + 1 < cols; c++) lin[(size_t)r * (cols - 1) + c] = m[(size_t)r * cols + c];
      bias[r] = m[(size_t)r * cols + cols - 1];
    }
    Own(cols - 1, rows, lin, bias);
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :3408-3415
    tdnnf_kaldi_io::Out o{os, binary};
    const std::vector<float> lin = Download(linear_, (size_t)out_dim_ * in_dim_), bias = Download(bias_, (size_t)out_dim_);
    o.token("<FixedAffineComponent>");
    o.token("<LinearParams>"); o.mat(lin.data(), out_dim_, in_dim_, in_dim_);
    o.token("<BiasParams>"); o.vec(bias.data(), out_dim_);
    o.token("</FixedAffineComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    if (p.rows <= 0 || p.cols <= 0 || (int)p.b.size() != p.rows) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: FixedAffineComponent::Read: bad parameter dimensions");
    Own(p.cols, p.rows, p.W, p.b);
  }

 private:
  void Own(int32 input_dim, int32 output_dim, const std::vector<float> &lin, const std::vector<float> &bias) {
    std::vector<float> all(lin);
    all.insert(all.end(), bias.begin(), bias.end());
    own_.Upload(all);
    in_dim_ = input_dim;
    out_dim_ = output_dim;
    linear_ = own_.Data();
    bias_ = own_.Data() + lin.size();
  }
  int32 in_dim_, out_dim_;
  const float *linear_, *bias_;
  DeviceArray<float> own_;
};

class NoOpComponent : public Component {  // :437-456
 public:
  NoOpComponent() : dim_(0), backprop_scale_(1.0f) {}
  void Init(int32 dim, float backprop_scale) { dim_ = dim; backprop_scale_ = backprop_scale; }
  virtual void InitFromConfig(ConfigLine *cfl) {  // :458-466
    DropNameAndType(cfl);
    backprop_scale_ = 1.0f;
    cfl->GetValue("backprop-scale", &backprop_scale_);
    if (!cfl->GetValue("dim", &dim_) || dim_ <= 0 || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :476-483
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<NoOpComponent>");
    o.token("<Dim>"); o.i32(dim_);
    o.token("<BackpropScale>"); o.f32(backprop_scale_);
    o.token("</NoOpComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    dim_ = (int32)PNum(p, "<Dim>", 0);
    backprop_scale_ = (float)PNum(p, "<BackpropScale>", 1.0);
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
  // InitFromConfig (UPSTREAM GeneralDropoutComponent; the recipes write dim, dropout-proportion, continuous: composite_layers.py:183-189)
  virtual void InitFromConfig(ConfigLine *cfl) {
    DropNameAndType(cfl);
    int32 block_dim = -1, time_period = 0;
    dim_ = -1;
    proportion_ = 0.5f;
    continuous_ = false;
    test_mode_ = false;
    const bool ok = cfl->GetValue("dim", &dim_);
    cfl->GetValue("block-dim", &block_dim);
    cfl->GetValue("dropout-proportion", &proportion_);
    cfl->GetValue("time-period", &time_period);
    cfl->GetValue("continuous", &continuous_);
    if (!ok || dim_ <= 0 || (block_dim != -1 && block_dim != dim_) || time_period != 0 || proportion_ < 0.0f || proportion_ > 1.0f || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + " (block-dim == dim, time-period 0 only): \"" + cfl->WholeLine() + "\"");
  }
  virtual void Write(std::ostream &os, bool binary) const {
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<GeneralDropoutComponent>");
    o.token("<Dim>"); o.i32(dim_);
    o.token("<BlockDim>"); o.i32(dim_);
    o.token("<TimePeriod>"); o.i32(0);
    o.token("<DropoutProportion>"); o.f32(proportion_);
    if (test_mode_) o.token("<TestMode>");
    if (continuous_) o.token("<Continuous>");
    o.token("</GeneralDropoutComponent>");
  }
  // the precomputed indexes are the number of sequences: one mask row per n (time-period 0)
  virtual ComponentPrecomputedIndexes *PrecomputeIndexes(const MiscComputationInfo &, const std::vector<Index> &, const std::vector<Index> &output_indexes,
                                                         bool) const {
    std::set<int32> ns;
    for (size_t i = 0; i < output_indexes.size(); i++) ns.insert(output_indexes[i].n);
    GeneralDropoutPrecomputedIndexes *ans = new GeneralDropoutPrecomputedIndexes();
    ans->num_mask_rows = (int32)ns.size();
    return ans;
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    dim_ = (int32)PNum(p, "<Dim>", 0);
    proportion_ = (float)PNum(p, "<DropoutProportion>", 0.5);
    continuous_ = p.num.count("<Continuous>") != 0;
    test_mode_ = p.num.count("<TestMode>") != 0;
  }

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
  virtual void InitFromConfig(ConfigLine *cfl) {  // :9384-9413: input-dim, output-dim, flops=a,b,c,... [scale]
    DropNameAndType(cfl);
    scale_ = 1.0f;
    std::string flops;
    const bool ok = cfl->GetValue("input-dim", &in_dim_) && cfl->GetValue("output-dim", &out_dim_) && cfl->GetValue("flops", &flops);
    if (!ok || in_dim_ <= 0 || out_dim_ <= 0 || !SplitStringToIntegers(flops, &flops_int_) || (int32)flops_int_.size() != in_dim_)
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Bad initializer: there is a problem with flops, input-dim and output-dim (not defined?): " + cfl->WholeLine());
    cfl->GetValue("scale", &scale_);
    if (cfl->HasUnusedValues()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Could not process these elements in initializer: " + cfl->UnusedValues());
    OwnFlops();
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :9427-9438
    if (flops_int_.empty()) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: FlopsConstraintComponent::Write needs the integer FLOPs vector (InitFromConfig / Read)");
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<FlopsConstraintComponent>");
    o.token("<InputDim>"); o.i32(in_dim_);
    o.token("<OutputDim>"); o.i32(out_dim_);
    o.token("<Flops>"); o.intvec(flops_int_);
    o.token("<Scale>"); o.f32(scale_);
    o.token("</FlopsConstraintComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {  // :9415-9425
    in_dim_ = (int32)PNum(p, "<InputDim>", 0);
    out_dim_ = (int32)PNum(p, "<OutputDim>", 0);
    scale_ = (float)PNum(p, "<Scale>", 1.0);
    flops_int_ = p.flops;
    if ((int32)flops_int_.size() != in_dim_) TDNNF_ADAPTER_FAIL("tdnnf_nnet3: FlopsConstraintComponent::Read: <Flops> has the wrong dimension");
    OwnFlops();
  }

 private:
  void OwnFlops() {
    own_.Upload(std::vector<float>(flops_int_.begin(), flops_int_.end()));
    flops_ = own_.Data();
  }
  int32 in_dim_, out_dim_;
  float scale_;
  const float *flops_;
  std::vector<int32> flops_int_;
  DeviceArray<float> own_;
};

// GumbelSoftmaxComponent :9774-9831: any width, no FLOPs penalty; the Gumbel draws are the memo-less part of Propagate
class GumbelSoftmaxComponent : public Component {
 public:
  GumbelSoftmaxComponent() : dim_(0), temp_(1.0f) {}
  void Init(int32 dim) { dim_ = dim; }
  void SetTempProportion(BaseFloat p) { temp_ = p; }
  virtual void InitFromConfig(ConfigLine *cfl) {  // :9753-9764
    DropNameAndType(cfl);
    const bool ok = cfl->GetValue("dim", &dim_) && cfl->GetValue("temp-proportion", &temp_);
    if (!ok || dim_ <= 0 || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
  }
  virtual void Write(std::ostream &os, bool binary) const {  // :9848-9855
    tdnnf_kaldi_io::Out o{os, binary};
    o.token("<GumbelSoftmaxComponent>");
    o.token("<Dim>"); o.i32(dim_);
    o.token("<TempProportion>"); o.f32(temp_);
    o.token("</GumbelSoftmaxComponent>");
  }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) {
    dim_ = (int32)PNum(p, "<Dim>", 0);
    temp_ = (float)PNum(p, "<TempProportion>", 1.0);
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
  virtual void InitFromConfig(ConfigLine *cfl) {  // NonlinearComponent::InitFromConfig, nnet-component-itf.cc:704-722
    DropNameAndType(cfl);
    int32 block_dim = -1;
    float f = 0.0f;
    const bool ok = cfl->GetValue("dim", &dim_);
    cfl->GetValue("block-dim", &block_dim);
    cfl->GetValue("self-repair-lower-threshold", &f);
    cfl->GetValue("self-repair-upper-threshold", &f);
    cfl->GetValue("self-repair-scale", &f);
    if (!ok || dim_ <= 0 || cfl->HasUnusedValues())
      TDNNF_ADAPTER_FAIL("tdnnf_nnet3: Invalid initializer for layer of type " + Type() + ": \"" + cfl->WholeLine() + "\"");
  }
  virtual void Write(std::ostream &os, bool binary) const { WriteNonlinear(os, binary, Type(), dim_, std::vector<double>(), 0.0f); }
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

 protected:
  virtual void FromParsed(const tdnnf_kaldi_io::Parsed &p) { dim_ = (int32)PNum(p, "<Dim>", 0); }

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
