// adapter_driver.cc -- a stand-in for the Kaldi side of the boundary, used by tests/test_gpu_adapter.py.
//
// It drives the nnet3 components of the hot path ONLY through include/tdnnf_nnet3_adapter.h, the way the edited
// Component::Propagate / Backprop bodies of INTEGRATION.md would, over a CuMatrixBase<float> stand-in that owns hipMalloc
// memory (Data / NumRows / NumCols / Stride, row stride padded like Kaldi's pitched allocations).  Inputs and outputs travel
// as files of named float matrices; the test compares the outputs with the CPU oracle.
// Build: hipcc -std=c++17 -I include tests/adapter_driver.cc -L tdnn-f_nas_amd -ltdnnf_hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "tdnnf_nnet3_adapter.h"
#include "tdnnf_nnet3_components.h"

#define HIPCK(e)                                                                      \
  do {                                                                                \
    hipError_t err__ = (e);                                                           \
    if (err__ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err__));                 \
      std::exit(3);                                                                   \
    }                                                                                 \
  } while (0)

struct HostMat {
  int rows = 0, cols = 0;
  std::vector<float> v;
  float at(int i) const { return v[i]; }
};
typedef std::map<std::string, HostMat> Blob;

static Blob read_blob(const char *path) {
  Blob b;
  FILE *f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  int n = 0;
  while (std::fread(&n, 4, 1, f) == 1) {
    std::string name(n, ' ');
    HostMat m;
    if (std::fread(&name[0], 1, n, f) != (size_t)n || std::fread(&m.rows, 4, 1, f) != 1 || std::fread(&m.cols, 4, 1, f) != 1) std::exit(2);
    m.v.resize((size_t)m.rows * m.cols);
    if (!m.v.empty() && std::fread(m.v.data(), 4, m.v.size(), f) != m.v.size()) std::exit(2);
    b[name] = m;
  }
  std::fclose(f);
  return b;
}
static void write_mat(FILE *f, const std::string &name, const HostMat &m) {
  int n = (int)name.size();
  std::fwrite(&n, 4, 1, f);
  std::fwrite(name.data(), 1, n, f);
  std::fwrite(&m.rows, 4, 1, f);
  std::fwrite(&m.cols, 4, 1, f);
  if (!m.v.empty()) std::fwrite(m.v.data(), 4, m.v.size(), f);
}

// kaldi::CuMatrix<float> stand-in: pitched device memory, the three accessors the adapter uses
class CuMatrixStub {
 public:
  CuMatrixStub(int rows, int cols, float fill = 0.f) : r_(rows), c_(cols), s_(((cols + 3) & ~3) + 4) {
    HIPCK(hipMalloc((void **)&d_, sizeof(float) * (size_t)std::max(1, r_) * s_));
    std::vector<float> h((size_t)std::max(1, r_) * s_, fill);
    HIPCK(hipMemcpy(d_, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  }
  explicit CuMatrixStub(const HostMat &m) : CuMatrixStub(m.rows, m.cols) {
    if (!m.v.empty()) HIPCK(hipMemcpy2D(d_, sizeof(float) * s_, m.v.data(), sizeof(float) * c_, sizeof(float) * c_, r_, hipMemcpyHostToDevice));
  }
  ~CuMatrixStub() { (void)hipFree(d_); }
  CuMatrixStub(const CuMatrixStub &) = delete;
  const float *Data() const { return d_; }
  float *Data() { return d_; }
  int NumRows() const { return r_; }
  int NumCols() const { return c_; }
  int Stride() const { return s_; }
  HostMat Host() const {
    HostMat m;
    m.rows = r_;
    m.cols = c_;
    m.v.resize((size_t)r_ * c_);
    HIPCK(hipDeviceSynchronize());
    if (!m.v.empty()) HIPCK(hipMemcpy2D(m.v.data(), sizeof(float) * c_, d_, sizeof(float) * s_, sizeof(float) * c_, r_, hipMemcpyDeviceToHost));
    return m;
  }

 private:
  float *d_ = nullptr;
  int r_, c_, s_;
};
// kaldi::CuVector<float> stand-in (dense)
struct CuVectorStub {
  float *d = nullptr;
  int n;
  explicit CuVectorStub(int dim, float fill = 0.f) : n(dim) {
    HIPCK(hipMalloc((void **)&d, sizeof(float) * std::max(1, n)));
    std::vector<float> h(std::max(1, n), fill);
    HIPCK(hipMemcpy(d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  }
  explicit CuVectorStub(const HostMat &m) : CuVectorStub((int)m.v.size()) {
    if (n) HIPCK(hipMemcpy(d, m.v.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  }
  ~CuVectorStub() { (void)hipFree(d); }
  CuVectorStub(const CuVectorStub &) = delete;
  HostMat Host() const {
    HostMat m;
    m.rows = 1;
    m.cols = n;
    m.v.resize(n);
    HIPCK(hipDeviceSynchronize());
    if (n) HIPCK(hipMemcpy(m.v.data(), d, sizeof(float) * n, hipMemcpyDeviceToHost));
    return m;
  }
};
struct DevBytes {
  void *p = nullptr;
  size_t n;
  explicit DevBytes(size_t bytes) : n(bytes) {
    HIPCK(hipMalloc(&p, std::max<size_t>(n, 256)));
    HIPCK(hipMemset(p, 0, std::max<size_t>(n, 256)));
  }
  ~DevBytes() { (void)hipFree(p); }
};
static HostMat doubles_to_host(const double *dev, int n) {
  std::vector<double> h(n);
  HIPCK(hipDeviceSynchronize());
  HIPCK(hipMemcpy(h.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost));
  HostMat m;
  m.rows = 1;
  m.cols = n;
  m.v.assign(h.begin(), h.end());
  return m;
}

using namespace tdnnf_adapter;

static NaturalGradient make_ng(int dim_in, int dim_out) {  // ranks / period / history / alpha of nnet-tdnn-component.cc:183-210
  NaturalGradient ng;
  Check(tdnnf_ng_create(std::min(20, (dim_in + 1) / 2), 4, 2000.0f, 4.0f, &ng.in));
  Check(tdnnf_ng_create(std::min(80, (dim_out + 1) / 2), 4, 2000.0f, 4.0f, &ng.out));
  return ng;
}

// A Tdnn[DARTSV3]Component over `steps` minibatches: Propagate, Backprop (data part + UpdateNaturalGradient into a zeroed
// delta-nnet component), the way NnetComputer would call them.  cfg = [K, Di, Do, row_stride, flags (-1: plain TdnnComponent),
// temp_proportion, lr, steps, offsets1_positive, row_offsets...]
static void run_tdnn(const Blob &in, FILE *out) {
  const HostMat &cfg = in.at("cfg");
  const int K = (int)cfg.at(0), Di = (int)cfg.at(1), Do = (int)cfg.at(2), rho = (int)cfg.at(3), flags = (int)cfg.at(4), steps = (int)cfg.at(7);
  const float temp = cfg.at(5), lr = cfg.at(6);
  const bool darts = flags >= 0, off1pos = cfg.at(8) != 0.f;
  std::vector<int> ro(K);
  for (int i = 0; i < K; i++) ro[i] = (int)cfg.at(9 + i);
  tdnnf_tdnn_indexes ix = Indexes(rho, ro);
  CuMatrixStub W(in.at("W"));
  CuVectorStub bias(in.at("bias"));  // DARTS: K + Do (logits first); plain: Do
  NaturalGradient ng = make_ng(K * Di + 1, Do);
  for (int t = 0; t < steps; t++) {
    const std::string sfx = std::to_string(t);
    CuMatrixStub x(in.at("x" + sfx)), dy(in.at("dy" + sfx));
    const int N = dy.NumRows();
    CuMatrixStub y(N, Do), dx(x.NumRows(), Di);
    CuMatrixStub W_acc(Do, K * Di);
    CuVectorStub b_acc(bias.n);
    DevBytes ws(tdnnf_tdnn_update_natural_gradient_workspace_bytes(Do, Di, K, N, 1));
    if (darts) {
      TdnnDartsState st = {K, Di, Do, W.Stride(), W.Data(), bias.d, flags, temp, off1pos ? 0 : K - 1, off1pos};
      CuVectorStub draws(in.at("draws" + sfx)), memo(2 * K);
      TdnnDartsPropagate(st, ix, x, &y, draws.d, memo.d, nullptr);
      TdnnDartsBackprop(st, ix, x, dy, memo.d, &dx, lr, W_acc.Data(), b_acc.d, ws.p, ws.n, nullptr, &ng);
      // the accumulator's row stride is its own (to_update->linear_params_.Stride()): the adapter passes c.ldw = W.Stride()
      write_mat(out, "memo" + sfx, memo.Host());
    } else {
      TdnnPropagate(ix, x, W.Data(), W.Stride(), Do, Di, bias.d, &y, nullptr);
      TdnnBackprop(ix, x, dy, W.Data(), W.Stride(), Do, Di, &dx, lr, W_acc.Data(), b_acc.d, ws.p, ws.n, nullptr, &ng);
    }
    write_mat(out, "y" + sfx, y.Host());
    write_mat(out, "dx" + sfx, dx.Host());
    write_mat(out, "W_acc" + sfx, W_acc.Host());
    write_mat(out, "b_acc" + sfx, b_acc.Host());
  }
  tdnnf_ng_destroy(ng.in);
  tdnnf_ng_destroy(ng.out);
}

// NaturalGradientAffine -> RectifiedLinear -> BatchNorm -> Linear -> LogSoftmax and back, `steps` minibatches.
// cfg = [steps, lr, self_repair_scale]
static void run_stack(const Blob &in, FILE *out) {
  const HostMat &cfg = in.at("cfg");
  const int steps = (int)cfg.at(0);
  const float lr = cfg.at(1), repair = cfg.at(2);
  CuMatrixStub Wa(in.at("Wa")), Wl(in.at("Wl"));
  CuVectorStub ba(in.at("ba"));
  const int Di = Wa.NumCols(), H = Wa.NumRows(), P = Wl.NumRows();
  NaturalGradient nga = make_ng(Di + 1, H), ngl = make_ng(H, P);
  DevBytes relu_stats(sizeof(double) * (1 + 2 * H)), bn_stats(sizeof(double) * (1 + 2 * H));
  for (int t = 0; t < steps; t++) {
    const std::string sfx = std::to_string(t);
    CuMatrixStub x(in.at("x" + sfx)), dlsm(in.at("d" + sfx));
    const int N = x.NumRows();
    CuMatrixStub a(N, H), r(N, H), z(N, H), l(N, P), lsm(N, P);
    CuVectorStub memo(5 * H);
    DevBytes cws(tdnnf_colreduce_workspace_bytes(N, std::max(H, P)));
    AffinePropagate(x, Wa.Data(), Wa.Stride(), ba.d, H, &a, nullptr);
    ReluPropagate(a, &r, nullptr);
    ReluStoreStats(r, (double *)relu_stats.p, cws.p, cws.n, nullptr);
    BatchNormPropagate(r, 1.0e-3f, 1.0f, &z, memo.d, cws.p, cws.n, nullptr);
    BatchNormStoreStats(memo.d, H, N, (double *)bn_stats.p, nullptr);
    AffinePropagate(z, Wl.Data(), Wl.Stride(), (const float *)nullptr, P, &l, nullptr);
    LogSoftmaxPropagate(l, &lsm, nullptr);
    // backward
    CuMatrixStub dl(N, P), dz(N, H), dr(N, H), da(N, H), dxm(N, Di);
    CuMatrixStub Wl_acc(P, H), Wa_acc(H, Di);
    CuVectorStub ba_acc(H);
    LogSoftmaxBackprop(lsm, dlsm, &dl, nullptr);
    DevBytes wsl(tdnnf_affine_update_natural_gradient_workspace_bytes(P, H, N, 0)), wsa(tdnnf_affine_update_natural_gradient_workspace_bytes(H, Di, N, 1));
    AffineBackprop(z, dl, Wl.Data(), Wl.Stride(), &dz, lr, Wl_acc.Data(), (float *)nullptr, wsl.p, wsl.n, nullptr, &ngl);
    BatchNormBackprop(z, dz, 1.0f, memo.d, &dr, cws.p, cws.n, nullptr);
    ReluBackprop(r, dr, &da, nullptr);
    ReluRepairGradients((const double *)relu_stats.p, H, repair, 0.05f, 0.95f, &da, nullptr);
    AffineBackprop(x, da, Wa.Data(), Wa.Stride(), &dxm, lr, Wa_acc.Data(), ba_acc.d, wsa.p, wsa.n, nullptr, &nga);
    write_mat(out, "lsm" + sfx, lsm.Host());
    write_mat(out, "z" + sfx, z.Host());
    write_mat(out, "dx" + sfx, dxm.Host());
    write_mat(out, "da" + sfx, da.Host());
    write_mat(out, "Wl_acc" + sfx, Wl_acc.Host());
    write_mat(out, "Wa_acc" + sfx, Wa_acc.Host());
    write_mat(out, "ba_acc" + sfx, ba_acc.Host());
  }
  write_mat(out, "relu_stats", doubles_to_host((const double *)relu_stats.p, 1 + 2 * H));
  write_mat(out, "bn_stats", doubles_to_host((const double *)bn_stats.p, 1 + 2 * H));
  for (tdnnf_ng *g : {nga.in, nga.out, ngl.in, ngl.out}) tdnnf_ng_destroy(g);
}

// The DARTS mixing components of one bottleneck-supernet layer, wired as generate_bottleneckCB8share_onehottrain_config.py /
// add_flopsconstraint.py do for ONE block: alpha -> ConstantFunction -> GumbelSoftmaxFlops -> (column k) CopyN ->
// ElementwiseProduct with the block of the linear output; and the Onehot variant.  cfg = [N, C, d, k, flops_scale, temp, lr]
static void run_mixing(const Blob &in, FILE *out) {
  const HostMat &cfg = in.at("cfg");
  const int N = (int)cfg.at(0), Cn = (int)cfg.at(1), d = (int)cfg.at(2);
  const float fscale = cfg.at(4), temp = cfg.at(5), lr = cfg.at(6);
  CuVectorStub alpha(in.at("alpha")), u(in.at("u")), flops(in.at("flops")), draw(in.at("draw"));
  CuMatrixStub lin(in.at("lin")), dmask(in.at("dmasked")), sk(in.at("sk")), dP_in(in.at("dP"));
  DevBytes cws(tdnnf_colreduce_workspace_bytes(N, std::max(Cn, 2 * d)));
  CuMatrixStub A(N, Cn), P(N, Cn), cop(N, d), ew_in(N, 2 * d), masked(N, d);
  ConstantFunctionPropagate(alpha.d, &A, nullptr);
  SoftmaxFlopsPropagate(A, u.d, temp, &P, nullptr);
  CopyNPropagate(sk, 1.0f, &cop, nullptr);
  {  // Append(copyn, linear block) is descriptor plumbing: two strided copies on the caller's side
    HIPCK(hipMemcpy2D(ew_in.Data(), sizeof(float) * ew_in.Stride(), cop.Data(), sizeof(float) * cop.Stride(), sizeof(float) * d, N, hipMemcpyDeviceToDevice));
    HIPCK(hipMemcpy2D(ew_in.Data() + d, sizeof(float) * ew_in.Stride(), lin.Data(), sizeof(float) * lin.Stride(), sizeof(float) * d, N, hipMemcpyDeviceToDevice));
  }
  ElementwiseProductPropagate(ew_in, d, &masked, nullptr);
  CuMatrixStub d_ew(N, 2 * d), d_sk(N, 1), dA(N, Cn);
  ElementwiseProductBackprop(ew_in, dmask, d, &d_ew, nullptr);
  CuMatrixStub d_cop(N, d);
  HIPCK(hipMemcpy2D(d_cop.Data(), sizeof(float) * d_cop.Stride(), d_ew.Data(), sizeof(float) * d_ew.Stride(), sizeof(float) * d, N, hipMemcpyDeviceToDevice));
  CopyNBackprop(d_cop, 1.0f, &d_sk, nullptr);
  SoftmaxFlopsBackprop(P, &dP_in, fscale, flops.d, Cn, temp, &dA, nullptr);
  CuVectorStub alpha_acc(Cn), onehot_acc(Cn);
  ConstantFunctionBackprop(dA, lr, alpha_acc.d, cws.p, cws.n, nullptr);
  CuMatrixStub oh(N, Cn);
  OnehotPropagate(draw.d, &oh, nullptr);
  OnehotBackprop(dP_in, lr, onehot_acc.d, cws.p, cws.n, nullptr);
  write_mat(out, "P", P.Host());
  write_mat(out, "masked", masked.Host());
  write_mat(out, "d_ew", d_ew.Host());
  write_mat(out, "d_sk", d_sk.Host());
  write_mat(out, "dA", dA.Host());
  write_mat(out, "dP_after", dP_in.Host());
  write_mat(out, "alpha_acc", alpha_acc.Host());
  write_mat(out, "onehot", oh.Host());
  write_mat(out, "onehot_acc", onehot_acc.Host());
}

// ---- the same two scenarios through the Component CLASSES of tdnnf_nnet3_components.h, created by factory name and driven
// through the virtual interface only (Propagate / StoreStats / Backprop with a second instance as the delta-nnet `to_update`),
// the way NnetComputer drives kaldi::nnet3::Component.  Outputs carry the names of run_stack / run_tdnn: the test holds them equal.
namespace n3 = tdnnf_nnet3;
static void *hook_alloc(size_t bytes) {
  void *p = nullptr;
  HIPCK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
  HIPCK(hipMemset(p, 0, std::max<size_t>(bytes, 256)));
  return p;
}
static void hook_free(void *p) { (void)hipFree(p); }
static std::vector<float> g_draws;  // what the components' RandUniform calls return, in call order
static size_t g_draw_pos = 0;
static void hook_uniform(float *dev, int n) {
  if (g_draw_pos + n > g_draws.size()) { std::fprintf(stderr, "out of draws\n"); std::exit(4); }
  HIPCK(hipMemcpy(dev, g_draws.data() + g_draw_pos, sizeof(float) * n, hipMemcpyHostToDevice));
  g_draw_pos += n;
}
static void install_hooks() {
  n3::Hooks().alloc = hook_alloc;
  n3::Hooks().free = hook_free;
  n3::Hooks().fill_uniform = hook_uniform;
  n3::Hooks().stream = nullptr;
}
static n3::CuMatrixBase V(CuMatrixStub &m) { return n3::CuMatrixBase(m.Data(), m.NumRows(), m.NumCols(), m.Stride()); }
struct DenseParams {  // a component's parameter matrix, dense on the device
  float *d = nullptr;
  int rows, cols;
  DenseParams(int r, int c, const HostMat *init) : rows(r), cols(c) {
    d = (float *)hook_alloc(sizeof(float) * (size_t)std::max(1, r * c));
    if (init && !init->v.empty()) HIPCK(hipMemcpy(d, init->v.data(), sizeof(float) * init->v.size(), hipMemcpyHostToDevice));
  }
  ~DenseParams() { hook_free(d); }
  void Zero() { HIPCK(hipMemset(d, 0, sizeof(float) * (size_t)std::max(1, rows * cols))); }
  HostMat Host() const {
    HostMat m;
    m.rows = rows;
    m.cols = cols;
    m.v.resize((size_t)rows * cols);
    HIPCK(hipDeviceSynchronize());
    if (!m.v.empty()) HIPCK(hipMemcpy(m.v.data(), d, sizeof(float) * m.v.size(), hipMemcpyDeviceToHost));
    return m;
  }
};
template <class T>
static T *make(const char *type) {
  n3::Component *c = n3::Component::NewComponentOfType(type);
  T *t = dynamic_cast<T *>(c);
  if (!t || c->Type() != type) { std::fprintf(stderr, "factory: %s\n", type); std::exit(5); }
  return t;
}

static void run_stack_classes(const Blob &in, FILE *out) {
  install_hooks();
  const HostMat &cfg = in.at("cfg");
  const int steps = (int)cfg.at(0);
  const float lr = cfg.at(1), repair = cfg.at(2);
  const HostMat &hWa = in.at("Wa"), &hWl = in.at("Wl"), &hba = in.at("ba");
  const int Di = hWa.cols, H = hWa.rows, P = hWl.rows;
  DenseParams Wa(H, Di, &hWa), Wl(P, H, &hWl), ba(1, H, &hba), Wa_acc(H, Di, nullptr), Wl_acc(P, H, nullptr), ba_acc(1, H, nullptr);
  DevBytes relu_stats(sizeof(double) * (1 + 2 * H)), bn_stats(sizeof(double) * (1 + 2 * H));
  auto *aff = make<n3::NaturalGradientAffineComponent>("NaturalGradientAffineComponent");
  auto *aff_upd = make<n3::NaturalGradientAffineComponent>("NaturalGradientAffineComponent");
  auto *lin = make<n3::LinearComponent>("LinearComponent");
  auto *lin_upd = make<n3::LinearComponent>("LinearComponent");
  auto *relu = make<n3::RectifiedLinearComponent>("RectifiedLinearComponent");
  auto *bn = make<n3::BatchNormComponent>("BatchNormComponent");
  auto *lsmc = make<n3::LogSoftmaxComponent>("LogSoftmaxComponent");
  aff->SetParams(Di, H, Wa.d, ba.d);
  aff_upd->SetParams(Di, H, Wa_acc.d, ba_acc.d);
  aff_upd->SetUnderlyingLearningRate(lr);
  lin->SetParams(H, P, Wl.d, nullptr);
  lin_upd->SetParams(H, P, Wl_acc.d, nullptr);
  lin_upd->SetUnderlyingLearningRate(lr);
  relu->Init(H, repair, (double *)relu_stats.p);
  bn->Init(H, 1.0e-3f, 1.0f, (double *)bn_stats.p);
  lsmc->Init(P);
  for (int t = 0; t < steps; t++) {
    const std::string sfx = std::to_string(t);
    CuMatrixStub x(in.at("x" + sfx)), dlsm(in.at("d" + sfx));
    const int N = x.NumRows();
    CuMatrixStub a(N, H), r(N, H), z(N, H), l(N, P), lsm(N, P);
    n3::CuMatrixBase vx = V(x), va = V(a), vr = V(r), vz = V(z), vl = V(l), vlsm = V(lsm), vdlsm = V(dlsm);
    aff->Propagate(nullptr, vx, &va);
    relu->Propagate(nullptr, va, &vr);
    relu->StoreStats(va, vr, nullptr);
    void *memo = bn->Propagate(nullptr, vr, &vz);
    bn->StoreStats(vr, vz, memo);
    lin->Propagate(nullptr, vz, &vl);
    lsmc->Propagate(nullptr, vl, &vlsm);
    CuMatrixStub dl(N, P), dz(N, H), dr(N, H), da(N, H), dxm(N, Di);
    n3::CuMatrixBase vdl = V(dl), vdz = V(dz), vdr = V(dr), vda = V(da), vdx = V(dxm);
    Wa_acc.Zero(); Wl_acc.Zero(); ba_acc.Zero();
    lsmc->Backprop("", nullptr, vl, vlsm, vdlsm, nullptr, nullptr, &vdl);
    lin->Backprop("", nullptr, vz, vl, vdl, nullptr, lin_upd, &vdz);
    bn->Backprop("", nullptr, vr, vz, vdz, memo, nullptr, &vdr);
    relu->Backprop("", nullptr, va, vr, vdr, nullptr, relu, &vda);
    aff->Backprop("", nullptr, vx, va, vda, nullptr, aff_upd, &vdx);
    HIPCK(hipDeviceSynchronize());
    bn->DeleteMemo(memo);
    write_mat(out, "lsm" + sfx, lsm.Host());
    write_mat(out, "z" + sfx, z.Host());
    write_mat(out, "dx" + sfx, dxm.Host());
    write_mat(out, "da" + sfx, da.Host());
    write_mat(out, "Wl_acc" + sfx, Wl_acc.Host());
    write_mat(out, "Wa_acc" + sfx, Wa_acc.Host());
    write_mat(out, "ba_acc" + sfx, ba_acc.Host());
  }
  write_mat(out, "relu_stats", doubles_to_host((const double *)relu_stats.p, 1 + 2 * H));
  write_mat(out, "bn_stats", doubles_to_host((const double *)bn_stats.p, 1 + 2 * H));
  HIPCK(hipDeviceSynchronize());
  for (n3::Component *c : std::vector<n3::Component *>{aff, aff_upd, lin, lin_upd, relu, bn, lsmc}) delete c;
}

static void run_tdnn_classes(const Blob &in, FILE *out) {
  install_hooks();
  const HostMat &cfg = in.at("cfg");
  const int K = (int)cfg.at(0), Di = (int)cfg.at(1), Do = (int)cfg.at(2), rho = (int)cfg.at(3), flags = (int)cfg.at(4), steps = (int)cfg.at(7);
  const float temp = cfg.at(5), lr = cfg.at(6);
  const bool darts = flags >= 0, off1pos = cfg.at(8) != 0.f;
  n3::TdnnPrecomputedIndexes ix;
  ix.row_stride = rho;
  for (int i = 0; i < K; i++) ix.row_offsets.push_back((int)cfg.at(9 + i));
  const HostMat &hW = in.at("W"), &hb = in.at("bias");
  DenseParams W(Do, K * Di, &hW), bias(1, (int)hb.v.size(), &hb), W_acc(Do, K * Di, nullptr), b_acc(1, (int)hb.v.size(), nullptr);
  n3::TdnnComponentBase *c = darts ? (n3::TdnnComponentBase *)make<n3::TdnnDARTSV3Component>("TdnnDARTSV3Component") : make<n3::TdnnComponent>("TdnnComponent");
  n3::TdnnComponentBase *upd = darts ? (n3::TdnnComponentBase *)make<n3::TdnnDARTSV3Component>("TdnnDARTSV3Component") : make<n3::TdnnComponent>("TdnnComponent");
  c->SetParams(K, Di, Do, W.d, bias.d, off1pos);
  upd->SetParams(K, Di, Do, W_acc.d, b_acc.d, off1pos);
  upd->SetUnderlyingLearningRate(lr);
  if (darts) {
    c->SetDartsFlags(flags, temp);
    upd->SetDartsFlags(flags, temp);
  }
  for (int t = 0; t < steps; t++) {
    const std::string sfx = std::to_string(t);
    CuMatrixStub x(in.at("x" + sfx)), dy(in.at("dy" + sfx));
    const int N = dy.NumRows();
    CuMatrixStub y(N, Do), dx(x.NumRows(), Di);
    if (darts) {
      g_draws = in.at("draws" + sfx).v;
      g_draw_pos = 0;
    }
    W_acc.Zero();
    b_acc.Zero();
    n3::CuMatrixBase vx = V(x), vy = V(y), vdy = V(dy), vdx = V(dx);
    void *memo = c->Propagate(&ix, vx, &vy);
    c->Backprop("", &ix, vx, vy, vdy, memo, upd, &vdx);
    HIPCK(hipDeviceSynchronize());
    if (darts) {  // [coef | effective coef]
      HostMat m;
      m.rows = 1;
      m.cols = 2 * K;
      m.v.resize(2 * K);
      HIPCK(hipMemcpy(m.v.data(), memo, sizeof(float) * 2 * K, hipMemcpyDeviceToHost));
      write_mat(out, "memo" + sfx, m);
    }
    c->DeleteMemo(memo);
    write_mat(out, "y" + sfx, y.Host());
    write_mat(out, "dx" + sfx, dx.Host());
    write_mat(out, "W_acc" + sfx, W_acc.Host());
    write_mat(out, "b_acc" + sfx, b_acc.Host());
  }
  delete c;
  delete upd;
}

// The remaining factory names of the 7q / supernet graphs (nnet-component-itf.cc:136,150,156,194,260,266), each created by name and
// driven through the virtual interface: FixedAffine -> Affine (UpdateSimple) -> NoOp (backprop-scale) -> GeneralDropout and back,
// with in_deriv matrices pre-filled to show kBackpropAdds; GumbelSoftmax (any width) -> FlopsConstraint.
// cfg = [num_seq, dropout_proportion, continuous, temp, flops_scale, lr, noop_backprop_scale]
static void run_rest_classes(const Blob &in, FILE *out) {
  install_hooks();
  const HostMat &cfg = in.at("cfg");
  const int S = (int)cfg.at(0);
  const float p = cfg.at(1), temp = cfg.at(3), fscale = cfg.at(4), lr = cfg.at(5), nbs = cfg.at(6);
  const bool continuous = cfg.at(2) != 0.f;
  const HostMat &hWf = in.at("Wf"), &hbf = in.at("bf"), &hWa = in.at("Wa"), &hba = in.at("ba"), &hfl = in.at("flops");
  const int Di = hWf.cols, H = hWf.rows, Cn = (int)hfl.v.size();
  DenseParams Wf(H, Di, &hWf), bf(1, H, &hbf), Wa(H, H, &hWa), ba(1, H, &hba), Wa_acc(H, H, nullptr), ba_acc(1, H, nullptr), flops(1, Cn, &hfl);
  auto *fixed = make<n3::FixedAffineComponent>("FixedAffineComponent");
  auto *aff = make<n3::AffineComponent>("AffineComponent");
  auto *aff_upd = make<n3::AffineComponent>("AffineComponent");
  auto *noop = make<n3::NoOpComponent>("NoOpComponent");
  auto *drop = make<n3::GeneralDropoutComponent>("GeneralDropoutComponent");
  auto *gs = make<n3::GumbelSoftmaxComponent>("GumbelSoftmaxComponent");
  auto *fc = make<n3::FlopsConstraintComponent>("FlopsConstraintComponent");
  fixed->SetParams(Di, H, Wf.d, bf.d);
  aff->SetParams(H, H, Wa.d, ba.d);
  aff_upd->SetParams(H, H, Wa_acc.d, ba_acc.d);
  aff_upd->SetUseNaturalGradient(true);  // an AffineComponent still updates with UpdateSimple
  aff_upd->SetUnderlyingLearningRate(lr);
  noop->Init(H, nbs);
  drop->Init(H, p, continuous);
  gs->Init(Cn);
  gs->SetTempProportion(temp);
  fc->Init(Cn, Cn, fscale, flops.d);
  g_draws = in.at("draws").v;
  g_draw_pos = 0;
  CuMatrixStub x(in.at("x")), d(in.at("d")), d0(in.at("d0_init")), dx(in.at("dx_init")), xs(in.at("xs")), dgs(in.at("dgs"));
  const int N = x.NumRows();
  CuMatrixStub y0(N, H), y1(N, H), y2(N, H), y3(N, H), d2(N, H), d1(N, H), P(N, Cn), Fo(N, Cn), dxs(N, Cn), dP(N, Cn);
  n3::CuMatrixBase vx = V(x), vd = V(d), vd0 = V(d0), vdx = V(dx), vy0 = V(y0), vy1 = V(y1), vy2 = V(y2), vy3 = V(y3), vd2 = V(d2), vd1 = V(d1);
  n3::GeneralDropoutPrecomputedIndexes dix;
  dix.num_mask_rows = S;
  fixed->Propagate(nullptr, vx, &vy0);
  aff->Propagate(nullptr, vy0, &vy1);
  noop->Propagate(nullptr, vy1, &vy2);
  void *memo = drop->Propagate(&dix, vy2, &vy3);
  drop->Backprop("", &dix, vy2, vy3, vd, memo, nullptr, &vd2);
  noop->Backprop("", nullptr, vy1, vy2, vd2, nullptr, nullptr, &vd1);
  aff->Backprop("", nullptr, vy0, vy1, vd1, nullptr, aff_upd, &vd0);
  fixed->Backprop("", nullptr, vx, vy0, vd0, nullptr, nullptr, &vdx);
  n3::CuMatrixBase vxs = V(xs), vP = V(P), vF = V(Fo), vdgs = V(dgs), vdxs = V(dxs), vdP = V(dP);
  gs->Propagate(nullptr, vxs, &vP);
  fc->Propagate(nullptr, vP, &vF);
  gs->Backprop("", nullptr, vxs, vP, vdgs, nullptr, nullptr, &vdxs);
  fc->Backprop("", nullptr, vP, vF, vdgs, nullptr, nullptr, &vdP);
  HIPCK(hipDeviceSynchronize());
  drop->DeleteMemo(memo);
  // test mode / proportion 0: a copy and no memo
  drop->SetTestMode(true);
  CuMatrixStub y3t(N, H);
  n3::CuMatrixBase vy3t = V(y3t);
  if (drop->Propagate(&dix, vy2, &vy3t) != nullptr) { std::fprintf(stderr, "test-mode dropout returned a memo\n"); std::exit(6); }
  write_mat(out, "y0", y0.Host());
  write_mat(out, "y1", y1.Host());
  write_mat(out, "y2", y2.Host());
  write_mat(out, "y3", y3.Host());
  write_mat(out, "y3_test_mode", y3t.Host());
  write_mat(out, "d2", d2.Host());
  write_mat(out, "d1", d1.Host());
  write_mat(out, "d0", d0.Host());
  write_mat(out, "dx", dx.Host());
  write_mat(out, "Wa_acc", Wa_acc.Host());
  write_mat(out, "ba_acc", ba_acc.Host());
  write_mat(out, "P", P.Host());
  write_mat(out, "F", Fo.Host());
  write_mat(out, "dxs", dxs.Host());
  write_mat(out, "dP", dP.Host());
  write_mat(out, "dgs_after", dgs.Host());
  HIPCK(hipDeviceSynchronize());
  for (n3::Component *c : std::vector<n3::Component *>{fixed, aff, aff_upd, noop, drop, gs, fc}) delete c;
  // every registered name comes out of the factory with its own Type()
  for (const std::string &t : n3::RegisteredTypes()) {
    n3::Component *c = n3::Component::NewComponentOfType(t);
    if (!c || c->Type() != t) { std::fprintf(stderr, "factory: %s\n", t.c_str()); std::exit(5); }
    delete c;
  }
  if (n3::RegisteredTypes().size() != 20 || n3::Component::NewComponentOfType("NoSuchComponent") != nullptr) std::exit(7);
}

int main(int argc, char **argv) {
  if (argc != 4) {
    std::fprintf(stderr, "usage: adapter_driver tdnn|stack|mixing|tdnn_classes|stack_classes|rest_classes in.bin out.bin\n");
    return 2;
  }
  try {
    Blob in = read_blob(argv[2]);
    FILE *out = std::fopen(argv[3], "wb");
    if (!out) { std::perror(argv[3]); return 2; }
    const std::string what = argv[1];
    if (what == "tdnn") run_tdnn(in, out);
    else if (what == "stack") run_stack(in, out);
    else if (what == "mixing") run_mixing(in, out);
    else if (what == "stack_classes") run_stack_classes(in, out);
    else if (what == "tdnn_classes") run_tdnn_classes(in, out);
    else if (what == "rest_classes") run_rest_classes(in, out);
    else return 2;
    std::fclose(out);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "adapter_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
