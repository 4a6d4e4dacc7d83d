// surface_driver.cc -- drives the WHOLE virtual surface of the Component classes in include/tdnnf_nnet3_components.h the way Kaldi's
// nnet3-init / nnet3-copy / nnet3-am-copy --edits / model combination do, on the GPU, through the C-ABI:
//   for each factory name:  NewComponentOfType -> InitFromConfig(the config line the reference's scripts emit) -> Write (text and binary)
//   -> ReadNew into a fresh object -> byte-identical Write again -> Copy() -> identical Propagate on the device; for the updatable ones
//   NumParameters / Vectorize / UnVectorize / Scale / Add / DotProduct / PerturbParams / FreezeNaturalGradient; for the Tdnn classes
//   PrecomputeIndexes / ReorderIndexes / GetInputIndexes; and the cv-update `sed` edits (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-134)
//   on the text form of a TdnnDARTSV3Component and a BatchNormComponent.
// usage: surface_driver LINES.txt   (one "Type<TAB>config line" per row); prints "OK <Type>" per class, exits non-zero on the first failure.
// Build: hipcc -std=c++17 -I include tests/surface_driver.cc -L tdnn-f_nas_amd -ltdnnf_hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "tdnnf_nnet3_components.h"

namespace n3 = tdnnf_nnet3;

#define HIPCK(e)                                                      \
  do {                                                                \
    hipError_t err__ = (e);                                           \
    if (err__ != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err__)); \
      std::exit(3);                                                   \
    }                                                                 \
  } while (0)
#define REQUIRE(cond, ...)                                  \
  do {                                                      \
    if (!(cond)) {                                          \
      std::fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
      std::fprintf(stderr, __VA_ARGS__);                    \
      std::fprintf(stderr, "\n");                           \
      std::exit(1);                                         \
    }                                                       \
  } while (0)

static void *hook_alloc(size_t bytes) {
  void *p = nullptr;
  HIPCK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
  HIPCK(hipMemset(p, 0, std::max<size_t>(bytes, 256)));
  return p;
}
static void hook_free(void *p) {
  HIPCK(hipDeviceSynchronize());  // (no stream-ordered allocator here: nothing in flight may still read the block)
  (void)hipFree(p);
}
static std::vector<float> g_draws;
static size_t g_draw_pos = 0;
static void hook_uniform(float *dev, int n) {
  std::vector<float> h(n);
  for (int i = 0; i < n; i++) h[i] = g_draws[(g_draw_pos + i) % g_draws.size()];
  g_draw_pos += n;
  HIPCK(hipMemcpy(dev, h.data(), sizeof(float) * n, hipMemcpyHostToDevice));
}
static void hook_h2d(void *d, const void *h, size_t b) {
  HIPCK(hipDeviceSynchronize());
  HIPCK(hipMemcpy(d, h, b, hipMemcpyHostToDevice));
}
static void hook_d2h(void *h, const void *d, size_t b) {
  HIPCK(hipDeviceSynchronize());
  HIPCK(hipMemcpy(h, d, b, hipMemcpyDeviceToHost));
}

struct DevMat {  // CuMatrix stand-in: pitched device memory
  float *d = nullptr;
  int rows, cols, stride;
  DevMat(int r, int c, const std::vector<float> *init = nullptr) : rows(r), cols(c), stride(((c + 3) & ~3) + 4) {
    HIPCK(hipMalloc((void **)&d, sizeof(float) * (size_t)std::max(1, r) * stride));
    HIPCK(hipMemset(d, 0, sizeof(float) * (size_t)std::max(1, r) * stride));
    if (init) HIPCK(hipMemcpy2D(d, sizeof(float) * stride, init->data(), sizeof(float) * c, sizeof(float) * c, r, hipMemcpyHostToDevice));
  }
  ~DevMat() { (void)hipFree(d); }
  n3::CuMatrixBase View() { return n3::CuMatrixBase(d, rows, cols, stride); }
  std::vector<float> Host() const {
    std::vector<float> h((size_t)rows * cols);
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy2D(h.data(), sizeof(float) * cols, d, sizeof(float) * stride, sizeof(float) * cols, rows, hipMemcpyDeviceToHost));
    return h;
  }
};

static std::string WriteStr(const n3::Component &c, bool binary) {
  std::ostringstream os(std::ios::out | std::ios::binary);
  c.Write(os, binary);
  return os.str();
}
static n3::Component *ReadStr(const std::string &s, bool binary) {
  std::istringstream is(s, std::ios::in | std::ios::binary);
  return n3::Component::ReadNew(is, binary);
}
static void replace_all(std::string *s, const std::string &a, const std::string &b) {
  for (size_t p = 0; (p = s->find(a, p)) != std::string::npos; p += b.size()) s->replace(p, a.size(), b);
}

// the index lists of a t-major minibatch: B sequences, output frames 0 .. T - 1 (step `out_step`), input frames covering the offsets at step 1
static void tdnn_indexes(const std::vector<int> &offsets, int B, int T, int out_step, std::vector<n3::Index> *in, std::vector<n3::Index> *out) {
  const int lo = *std::min_element(offsets.begin(), offsets.end()), hi = *std::max_element(offsets.begin(), offsets.end());
  in->clear();
  out->clear();
  for (int t = lo; t <= (T - 1) * out_step + hi; t++)
    for (int n = 0; n < B; n++) in->push_back(n3::Index(n, t));
  for (int k = 0; k < T; k++)
    for (int n = 0; n < B; n++) out->push_back(n3::Index(n, k * out_step));
}

// Propagate of `c` on a seeded input; the output values (and, for a Tdnn class, its precomputed indexes checked against the formula)
static std::vector<float> propagate(n3::Component *c, int B, int T) {
  g_draw_pos = 0;  // the same "random" draws for every object that is compared
  n3::ComponentPrecomputedIndexes *ix = nullptr;
  int rows_in = B * T, rows_out = B * T;
  if (n3::TdnnComponentBase *td = dynamic_cast<n3::TdnnComponentBase *>(c)) {
    std::vector<n3::Index> in, out;
    tdnn_indexes(td->TimeOffsets(), B, T, 1, &in, &out);
    n3::MiscComputationInfo misc;
    ix = c->PrecomputeIndexes(misc, in, out, true);
    rows_in = (int)in.size();
    rows_out = (int)out.size();
  } else if (dynamic_cast<n3::GeneralDropoutComponent *>(c)) {
    std::vector<n3::Index> in, out;
    tdnn_indexes(std::vector<int>(1, 0), B, T, 1, &in, &out);
    n3::MiscComputationInfo misc;
    ix = c->PrecomputeIndexes(misc, in, out, true);
    REQUIRE(static_cast<n3::GeneralDropoutPrecomputedIndexes *>(ix)->num_mask_rows == B, "dropout mask rows");
  }
  std::vector<float> x((size_t)rows_in * c->InputDim());
  unsigned lcg = 12345u;
  for (size_t i = 0; i < x.size(); i++) {
    lcg = lcg * 1664525u + 1013904223u;
    x[i] = ((lcg >> 8) & 0xffff) / 32768.0f - 1.0f;
  }
  DevMat in(rows_in, c->InputDim(), &x), out(rows_out, c->OutputDim());
  n3::CuMatrixBase vi = in.View(), vo = out.View();
  void *memo = c->Propagate(ix, vi, &vo);
  if (c->Properties() & n3::kStoresStats) c->StoreStats(vi, vo, memo);
  std::vector<float> y = out.Host();
  c->DeleteMemo(memo);
  delete ix;
  for (size_t i = 0; i < y.size(); i++) REQUIRE(std::isfinite(y[i]), "%s: non-finite output", c->Type().c_str());
  return y;
}

static void check_updatable(n3::UpdatableComponent *u, n3::Component *same) {
  const int n = u->NumParameters();
  REQUIRE(n > 0, "%s: NumParameters", u->Type().c_str());
  std::vector<float> v(n), w(n);
  n3::VectorBase vv(v.data(), n), ww(w.data(), n);
  u->Vectorize(&vv);
  double nv = 0;
  for (int i = 0; i < n; i++) nv += (double)v[i] * v[i];
  u->Scale(2.0f);
  u->Vectorize(&ww);
  for (int i = 0; i < n; i++) REQUIRE(w[i] == 2.0f * v[i], "%s: Scale", u->Type().c_str());
  u->Add(-0.5f, *same);  // `same` still holds v
  u->Vectorize(&ww);
  for (int i = 0; i < n; i++) REQUIRE(std::fabs(w[i] - 1.5f * v[i]) <= 1e-6f * std::fabs(v[i]) + 1e-30f, "%s: Add", u->Type().c_str());
  const float dot = u->DotProduct(*static_cast<n3::UpdatableComponent *>(same));
  REQUIRE(std::fabs(dot - 1.5 * nv) <= 1e-4 * (1.5 * nv) + 1e-12, "%s: DotProduct %g vs %g", u->Type().c_str(), (double)dot, 1.5 * nv);
  u->UnVectorize(vv);
  u->Vectorize(&ww);
  REQUIRE(v == w, "%s: UnVectorize", u->Type().c_str());
  u->PerturbParams(0.1f);
  u->Vectorize(&ww);
  double diff = 0;
  for (int i = 0; i < n; i++) diff += ((double)w[i] - v[i]) * ((double)w[i] - v[i]);
  REQUIRE(diff > 0 && (n < 200 || std::fabs(std::sqrt(diff / n) - 0.1) < 0.03), "%s: PerturbParams rms %g", u->Type().c_str(), std::sqrt(diff / n));
  u->UnVectorize(vv);
  u->Scale(0.0f);
  u->Vectorize(&ww);
  for (int i = 0; i < n; i++) REQUIRE(w[i] == 0.0f, "%s: Scale(0)", u->Type().c_str());
  u->UnVectorize(vv);
  u->FreezeNaturalGradient(true);
  u->FreezeNaturalGradient(false);
}

int main(int argc, char **argv) {
  if (argc != 2) {
    std::fprintf(stderr, "usage: surface_driver LINES.txt\n");
    return 2;
  }
  n3::Hooks().alloc = hook_alloc;
  n3::Hooks().free = hook_free;
  n3::Hooks().fill_uniform = hook_uniform;
  n3::Hooks().h2d = hook_h2d;
  n3::Hooks().d2h = hook_d2h;
  n3::Hooks().stream = nullptr;
  g_draws.resize(4096);
  for (size_t i = 0; i < g_draws.size(); i++) g_draws[i] = 0.05f + 0.9f * (float)((i * 2654435761u) % 1000) / 1000.0f;
  const int B = 4, T = 12;
  std::ifstream lines(argv[1]);
  std::string row;
  int done = 0;
  std::string bn_text, darts_text;
  try {
    while (std::getline(lines, row)) {
      if (row.empty()) continue;
      const size_t tab = row.find('\t');
      REQUIRE(tab != std::string::npos, "bad row: %s", row.c_str());
      const std::string type = row.substr(0, tab), line = row.substr(tab + 1);
      n3::SetRandSeed(1234u + (unsigned)done);
      n3::Component *c = n3::Component::NewComponentOfType(type);
      REQUIRE(c && c->Type() == type, "factory: %s", type.c_str());
      n3::ConfigLine cfl;
      REQUIRE(cfl.ParseLine(line), "cannot parse: %s", line.c_str());
      if (type == "BatchNormTestComponent") {
        // InitFromConfig is empty in the reference: the object comes from a trained BatchNormComponent's text with the type name changed
        c->InitFromConfig(&cfl);
        delete c;
        REQUIRE(!bn_text.empty(), "BatchNormComponent must come before BatchNormTestComponent in the list");
        std::string t = bn_text;
        replace_all(&t, "<TestMode> F", "<TestMode> T");             // ...cvupdate.sh:133
        replace_all(&t, "BatchNormComponent", "BatchNormTestComponent");
        c = ReadStr(t, false);
        REQUIRE(c->Type() == type, "sed to BatchNormTestComponent");
      } else {
        c->InitFromConfig(&cfl);
        REQUIRE(!cfl.HasUnusedValues(), "%s: unused config values: %s", type.c_str(), cfl.UnusedValues().c_str());
      }
      const std::vector<float> y = propagate(c, B, T);  // (also fills BatchNorm / ReLU statistics: StoreStats)
      const std::string text = WriteStr(*c, false), bin = WriteStr(*c, true);
      REQUIRE(text.compare(0, type.size() + 2, "<" + type + ">") == 0, "%s: Write does not begin with the opening tag", type.c_str());
      if (type == "BatchNormComponent") bn_text = text;
      if (type == "TdnnDARTSV3Component" && darts_text.empty()) darts_text = text;
      n3::Component *ct = ReadStr(text, false), *cb = ReadStr(bin, true), *cc = c->Copy();
      REQUIRE(WriteStr(*ct, false) == text, "%s: text Write -> Read -> Write differs", type.c_str());
      REQUIRE(WriteStr(*cb, true) == bin, "%s: binary Write -> Read -> Write differs", type.c_str());
      REQUIRE(WriteStr(*cb, false) == text, "%s: binary and text forms disagree", type.c_str());
      REQUIRE(WriteStr(*cc, false) == text, "%s: Copy() differs", type.c_str());
      REQUIRE(ct->InputDim() == c->InputDim() && ct->OutputDim() == c->OutputDim() && ct->Properties() == c->Properties(), "%s: dims / properties after Read", type.c_str());
      // identical Propagate (statistics-dependent classes: the copies carry the statistics the original had when it was written)
      const std::vector<float> y0 = propagate(c, B, T), yt = propagate(ct, B, T), yb = propagate(cb, B, T), yc = propagate(cc, B, T);
      REQUIRE(y0 == yt && y0 == yb && y0 == yc, "%s: Propagate differs after Write / Read / Copy", type.c_str());
      if (type != "BatchNormComponent" && type != "RectifiedLinearComponent") REQUIRE(y == y0, "%s: Propagate is not repeatable", type.c_str());
      REQUIRE(c->Info().find(type) == 0, "%s: Info()", type.c_str());
      if (c->Properties() & n3::kUpdatableComponent) {
        n3::UpdatableComponent *u = dynamic_cast<n3::UpdatableComponent *>(c);
        REQUIRE(u, "%s: updatable flag without the class", type.c_str());
        check_updatable(u, ct);
        REQUIRE(WriteStr(*c, false) == text, "%s: parameters not restored", type.c_str());
      }
      if (n3::TdnnComponentBase *td = dynamic_cast<n3::TdnnComponentBase *>(c)) {
        // PrecomputeIndexes against the closed form of nnet-tdnn-component.cc:878-903 on a subsampled output grid, ReorderIndexes idempotent
        for (int step : {1, 3}) {
          std::vector<n3::Index> in, out;
          tdnn_indexes(td->TimeOffsets(), B, T, step, &in, &out);
          n3::MiscComputationInfo misc;
          std::vector<n3::Index> in2 = in, out2 = out;
          c->ReorderIndexes(&in2, &out2);
          std::vector<n3::Index> in3 = in2, out3 = out2;
          c->ReorderIndexes(&in3, &out3);
          REQUIRE(in3 == in2 && out3 == out2, "%s: ReorderIndexes is not idempotent", type.c_str());
          REQUIRE(out2 == out && (step > 1 || in2 == in), "%s: ReorderIndexes changed a regular t-major grid", type.c_str());
          n3::TdnnPrecomputedIndexes *ix = static_cast<n3::TdnnPrecomputedIndexes *>(c->PrecomputeIndexes(misc, in2, out2, true));
          REQUIRE(ix->row_stride == step, "%s: row_stride %d", type.c_str(), ix->row_stride);
          const int t0 = *std::min_element(td->TimeOffsets().begin(), td->TimeOffsets().end());
          for (size_t i = 0; i < td->TimeOffsets().size(); i++) {
            const int input_t = td->TimeOffsets()[i] - t0;
            REQUIRE(ix->row_offsets[i] == step * (input_t / step) * B + input_t % step, "%s: row offset %zu", type.c_str(), i);
          }
          delete ix;
          std::vector<n3::Index> want;
          c->GetInputIndexes(misc, n3::Index(2, 7), &want);
          REQUIRE(want.size() == td->TimeOffsets().size() && want[0].t == 7 + td->TimeOffsets()[0] && want[0].n == 2, "%s: GetInputIndexes", type.c_str());
        }
      }
      delete c;
      delete ct;
      delete cb;
      delete cc;
      std::printf("OK %s\n", type.c_str());
      done++;
    }
    // the cv-update edits on a pretrain-mode TdnnDARTSV3Component (..._cvupdate.sh:128-134): frozen by learning-rate factor, Gumbel on
    REQUIRE(!darts_text.empty(), "no TdnnDARTSV3Component in the list");
    {
      n3::Component *c0 = ReadStr(darts_text, false);
      n3::UpdatableComponent *u0 = dynamic_cast<n3::UpdatableComponent *>(c0);
      u0->SetLearningRateFactor(0.0f);  // nnet3-am-copy --edits="set-learning-rate-factor learning-rate-factor=0"
      u0->SetUnderlyingLearningRate(0.001f);
      std::string t = WriteStr(*c0, false);
      REQUIRE(t.find("<TdnnDARTSV3Component> <LearningRateFactor> 0 ") == 0, "text form of a zero learning-rate factor: %.80s", t.c_str());
      replace_all(&t, "<TdnnDARTSV3Component> <LearningRateFactor> 0", "<TdnnDARTSV3Component> <LearningRateFactor> 0.0001");
      replace_all(&t, "<use-gumbel> F", "<use-gumbel> T");
      replace_all(&t, "<update-alpha> F", "<update-alpha> T");
      replace_all(&t, "<update-theta> T", "<update-theta> F");
      replace_all(&t, "<uniform-sample> T", "<uniform-sample> F");
      n3::Component *c1 = ReadStr(t, false);
      const std::string t1 = WriteStr(*c1, false);
      // (the factor itself is checked by value below: this writer prints floats with nine significant digits so that text files round-trip exactly)
      REQUIRE(t1.find("<LearningRateFactor> ") != std::string::npos && t1.find("<use-gumbel> T") != std::string::npos &&
                  t1.find("<update-alpha> T") != std::string::npos && t1.find("<update-theta> F") != std::string::npos &&
                  t1.find("<uniform-sample> F") != std::string::npos,
              "the cv-update edits did not survive Read -> Write");
      REQUIRE(std::fabs(dynamic_cast<n3::UpdatableComponent *>(c1)->LearningRateFactor() - 1.0e-4f) < 1e-10f, "learning-rate factor after the edit");
      const std::vector<float> ya = propagate(c0, B, T), yb = propagate(c1, B, T);
      REQUIRE(ya != yb, "Gumbel over all taps must differ from the uniform one-hot sample");
      delete c0;
      delete c1;
      std::printf("OK cv-update edits\n");
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "surface_driver: %s\n", e.what());
    return 1;
  }
  std::printf("DONE %d\n", done);
  return done == 20 ? 0 : 1;
}
