// abi_tdnn.hip -- C-ABI of the TDNN / affine layers (include/tdnnf_hip.h, rows A1/A2/A6) on top
// of the f32 MFMA GEMM kernels, plus error plumbing.
//
// Reference: /root/reference/src/nnet3/nnet-tdnn-component.cc (Propagate :214-333, Backprop
// :335-431, UpdateSimple :433-455, GetInputPart :806-820), nnet-simple-component.cc
// (AffineComponent :1235-1279).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "common.h"
#include "gemm_f32.h"
#include "planes_gemm.h"

namespace tdnnf {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
int hip_status(hipError_t e, const char *what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return TDNNF_EHIP;
}

Options &options() {
  static Options o;
  return o;
}

namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)();
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
bool roctx_ready() {
  static const bool ok = [] {
    const char *e = getenv("TDNNF_ROCTX");
    if (e && atoi(e) == 0) return false;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
      g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
      if (g_roctx_push && g_roctx_pop) return true;
    }
    return false;
  }();
  return ok;
}
}  // namespace
TraceRange::TraceRange(const char *name) : on(roctx_ready()) {
  if (on) g_roctx_push(name);
}
TraceRange::~TraceRange() {
  if (on) g_roctx_pop();
}

// rows of `in` needed by tap views: GetInputPart's assert, nnet-tdnn-component.cc:811-813
static bool tdnn_rows_ok(const tdnnf_tdnn_indexes *ix, int rows_in, int N) {
  if (!ix || ix->num_offsets < 1 || ix->num_offsets > TDNNF_MAX_OFFSETS || ix->row_stride < 1) return false;
  for (int i = 0; i < ix->num_offsets; i++) {
    const long long need = (long long)ix->row_offsets[i] + (long long)ix->row_stride * N - (ix->row_stride - 1);
    if (ix->row_offsets[i] < 0 || (N > 0 && rows_in < need)) return false;
  }
  return true;
}

}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

const char *tdnnf_last_error(void) { return g_last_error.c_str(); }
int tdnnf_abi_version(void) { return 1; }

static int *option_slot(const char *name) {
  Options &o = options();
  if (!name) return nullptr;
  const struct { const char *n; int *p; } table[] = {{"ng_grouped", &o.ng_grouped}, {"ng_fuse", &o.ng_fuse}, {"ng_early_in", &o.ng_early_in},
                                                     {"wgrad_stream", &o.wgrad_stream}, {"gemm_ring", &o.gemm_ring}, {"planes", &o.planes}, {"den_split", &o.den_split}, {"phase_events", &o.phase_events}, {"wgrad_lag", &o.wgrad_lag}, {"ng_bk", &o.ng_bk}, {"ng_valu", &o.ng_valu}, {"ng_pform", &o.ng_pform}, {"planes_group", &o.planes_group}, {"xent_behind_den", &o.xent_behind_den}, {"ng_early_fork", &o.ng_early_fork}, {"ng_diag_skip", &o.ng_diag_skip}, {"wgrad_small", &o.wgrad_small}, {"splitk_per_cu", &o.splitk_per_cu}, {"gemm_alt_taps", &o.gemm_alt_taps}, {"reverse_passes", &o.reverse_passes}, {"splitk_partial_round", &o.splitk_partial_round}, {"wgrad_on_caller", &o.wgrad_on_caller}, {"den_mw_test_abort", &o.den_mw_test_abort}, {"planes_check_bound", &o.planes_check_bound}};
  for (auto &e : table)
    if (strcmp(e.n, name) == 0) return e.p;
  return nullptr;
}
int tdnnf_set_option(const char *name, int value) {
  int *p = option_slot(name);
  TDNNF_REQUIRE(p, "set_option: unknown option '%s' (ng_grouped, ng_fuse, ng_early_in, wgrad_stream, gemm_ring, planes, den_split)", name ? name : "(null)");
  *p = value;
  return TDNNF_OK;
}
int tdnnf_get_option(const char *name, int *value) {
  int *p = option_slot(name);
  TDNNF_REQUIRE(p && value, "get_option: unknown option '%s'", name ? name : "(null)");
  *value = *p;
  return TDNNF_OK;
}

}  // extern "C"

namespace tdnnf {
// Propagate with an optional fused ReLU in the GEMM epilogue (trainer-internal; the C-ABI entry has relu = 0)
int tdnn_propagate_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in, const float *W, int ldw, int Do, int Di,
                        const float *bias, const float *eff_coef, int init_mode, int relu, tdnnf_mat *out, tdnnf_stream stream,
                        float *colstats, int *colstats_rows) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && W, "tdnn_propagate: bad matrices");
  TDNNF_REQUIRE(Do > 0 && Di > 0 && in->cols == Di && out->cols == Do, "tdnn_propagate: dims: in.cols=%d Di=%d out.cols=%d Do=%d",
                in ? in->cols : -1, Di, out ? out->cols : -1, Do);
  TDNNF_REQUIRE(tdnn_rows_ok(ix, in->rows, out->rows), "tdnn_propagate: input has too few rows for the time offsets");
  TDNNF_REQUIRE(ldw >= ix->num_offsets * Di, "tdnn_propagate: ldw < K*Di");
  TDNNF_REQUIRE(init_mode >= 0 && init_mode <= 2 && (init_mode != 1 || bias), "tdnn_propagate: init_mode 1 needs a bias");
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = in->data;
  a.lda = (long long)in->stride * ix->row_stride;
  a.B = W;
  a.ldb = ldw;
  a.C = out->data;
  a.ldc = out->stride;
  a.M = out->rows;
  a.N = Do;
  a.bias = bias;
  a.coef = eff_coef;
  a.init_mode = init_mode;
  a.relu = relu;
  a.colstats = colstats;  // BatchNorm statistics of the output, when the launch can form them (gemm_f32.h)
  a.colstats_rows = colstats_rows;
  a.nseg = ix->num_offsets;
  for (int i = 0; i < a.nseg; i++) {
    a.seg[i].a_off = (long long)ix->row_offsets[i] * in->stride;
    a.seg[i].b_off = (long long)i * Di;
    a.seg[i].klen = Di;
    a.seg[i].m_lo = 0;
    a.seg[i].m_hi = out->rows;
  }
  TDNNF_HIP(rows_gemm(a, true, (hipStream_t)stream));
  return TDNNF_OK;
}
}  // namespace tdnnf

extern "C" {

int tdnnf_tdnn_propagate(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in, const float *W, int ldw, int Do, int Di,
                         const float *bias, const float *eff_coef, int init_mode, tdnnf_mat *out, tdnnf_stream stream) {
  return tdnn_propagate_impl(ix, in, W, ldw, Do, Di, bias, eff_coef, init_mode, 0, out, stream);
}

// Gather form of Backprop :366-416.  in_deriv row q receives sum over taps i with (q - off_i) % rho == 0 of
// dY[(q - off_i)/rho] W_i, so every in_deriv element is written by exactly one thread (no atomics for
// the overlapping taps).  One launch per residue class of q mod rho that has at least one tap.
}  // extern "C"

namespace tdnnf {
// overwrite != 0: in_deriv = (instead of +=) the gathered sum, legal only when row_stride == 1 (every row of
// in_deriv up to the last tap's range is produced by the single launch; rows beyond are zeroed by the caller's
// contract that in_deriv->rows == max row reached).  add (may be null): in_deriv[m] += add_scale * add[m - add_lo].
int tdnn_backprop_data_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *out_deriv, const float *W, int ldw, int Do,
                            int Di, const float *eff_coef, int overwrite, const tdnnf_mat *add, float add_scale, int add_lo,
                            tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in_deriv) && mat_ok(out_deriv) && W, "tdnn_backprop_data: bad matrices");
  TDNNF_REQUIRE(Do > 0 && Di > 0 && in_deriv->cols == Di && out_deriv->cols == Do, "tdnn_backprop_data: bad dims");
  TDNNF_REQUIRE(tdnn_rows_ok(ix, in_deriv->rows, out_deriv->rows), "tdnn_backprop_data: in_deriv has too few rows for the time offsets");
  TDNNF_REQUIRE(ldw >= ix->num_offsets * Di, "tdnn_backprop_data: ldw < K*Di");
  const int rho = ix->row_stride, N = out_deriv->rows, K = ix->num_offsets;
  if (N == 0) return TDNNF_OK;
  for (int cls = 0; cls < rho; cls++) {
    RowsGemmArgs a;
    memset(&a, 0, sizeof(a));
    float cf_dummy = 0;
    (void)cf_dummy;
    int nseg = 0, u_max = 0;
    int seg_tap[kMaxSeg];
    for (int i = 0; i < K; i++) {
      if (ix->row_offsets[i] % rho != cls) continue;
      const int shift = (ix->row_offsets[i] - cls) / rho;  // u = r + shift
      a.seg[nseg].a_off = -(long long)shift * out_deriv->stride;
      a.seg[nseg].b_off = (long long)i * Di;
      a.seg[nseg].klen = Do;
      a.seg[nseg].m_lo = shift;
      a.seg[nseg].m_hi = shift + N;
      if (shift + N > u_max) u_max = shift + N;
      seg_tap[nseg] = i;
      nseg++;
    }
    if (nseg == 0) continue;
    // the per-segment coefficients must be contiguous for the kernel: taps of one class are not, so when
    // coefficients are present and the class skips taps, launch one GEMM per tap (rho > 1 with K taps is rare).
    bool contiguous = true;
    for (int s = 1; s < nseg; s++) contiguous = contiguous && seg_tap[s] == seg_tap[s - 1] + 1;
    a.A = out_deriv->data;
    a.lda = out_deriv->stride;
    a.B = W;
    a.ldb = ldw;
    // split-bf16 default arithmetic with a transposed copy of W registered (the trainer): B becomes k-contiguous,
    // B[n = i][k = o] = WT[(tap Di + i) Do + o]
    const float *WT = (ldw == K * Di && !planes_hint_b()) ? transposed_weights(W) : nullptr;  // (pre-split planes hold both orientations)
    if (WT) {
      a.B = WT;
      a.ldb = Do;
      for (int sg = 0; sg < nseg; sg++) a.seg[sg].b_off = (long long)seg_tap[sg] * Di * Do;
    }
    a.C = in_deriv->data + (long long)cls * in_deriv->stride;
    a.ldc = (long long)in_deriv->stride * rho;
    a.M = u_max;
    a.N = Di;
    a.init_mode = 0;  // kBackpropAdds
    if (overwrite) {
      TDNNF_REQUIRE(rho == 1 && u_max == in_deriv->rows, "tdnn_backprop_data: overwrite needs row_stride 1 and full row coverage");
      a.init_mode = 2;
    }
    if (add) {
      TDNNF_REQUIRE(rho == 1 && add->cols == Di && add_lo >= 0 && add_lo + add->rows <= in_deriv->rows, "tdnn_backprop_data: bad addend");
      a.add = add->data;
      a.ldadd = add->stride;
      a.add_scale = add_scale;
      a.add_lo = add_lo;
      a.add_hi = add_lo + add->rows;
    }
    TDNNF_REQUIRE(!(overwrite || add) || !eff_coef || contiguous, "tdnn_backprop_data: fused overwrite/addend needs a single launch");
    if (!eff_coef || contiguous) {
      a.coef = eff_coef ? eff_coef + seg_tap[0] : nullptr;
      a.nseg = nseg;
      TDNNF_HIP(rows_gemm(a, WT != nullptr, (hipStream_t)stream));
    } else {
      for (int s = 0; s < nseg; s++) {
        RowsGemmArgs b = a;
        b.seg[0] = a.seg[s];
        b.nseg = 1;
        b.coef = eff_coef + seg_tap[s];
        b.M = a.seg[s].m_hi;
        TDNNF_HIP(rows_gemm(b, WT != nullptr, (hipStream_t)stream));
      }
    }
  }
  return TDNNF_OK;
}
}  // namespace tdnnf

extern "C" {

int tdnnf_tdnn_backprop_data(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *out_deriv, const float *W, int ldw, int Do,
                             int Di, const float *eff_coef, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  return tdnn_backprop_data_impl(ix, out_deriv, W, ldw, Do, Di, eff_coef, 0, nullptr, 0.f, 0, in_deriv, stream);
}

size_t tdnnf_tdnn_update_workspace_bytes(int Do, int Di, int K, int num_rows) {
  return wgrad_workspace_bytes(Do, Di, K, num_rows);
}

}  // extern "C"

namespace tdnnf {
// active_dev / max_active: optional compacted list of the taps with a non-zero coefficient (gemm_f32.h)
int tdnn_update_simple_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do,
                            int Di, const float *eff_coef, float lr, float *W_acc, int ldw, float *bias_acc, void *ws,
                            size_t ws_bytes, const int *active_dev, int max_active, tdnnf_stream stream, bool overwrite) {
  TDNNF_REQUIRE(mat_ok(in_value) && mat_ok(out_deriv) && W_acc, "tdnn_update_simple: bad matrices");
  TDNNF_REQUIRE(Do > 0 && Di > 0 && in_value->cols == Di && out_deriv->cols == Do, "tdnn_update_simple: bad dims");
  TDNNF_REQUIRE(tdnn_rows_ok(ix, in_value->rows, out_deriv->rows), "tdnn_update_simple: in_value has too few rows for the time offsets");
  TDNNF_REQUIRE(ldw >= ix->num_offsets * Di, "tdnn_update_simple: ldw < K*Di");
  if (out_deriv->rows == 0) return TDNNF_OK;
  TDNNF_REQUIRE(ws && ws_bytes >= wgrad_workspace_bytes(Do, Di, ix->num_offsets, out_deriv->rows), "tdnn_update_simple: workspace too small");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dY = out_deriv->data;
  a.lddy = out_deriv->stride;
  a.X = in_value->data;
  a.ldx = in_value->stride;
  a.Do = Do;
  a.Di = Di;
  a.K = ix->num_offsets;
  a.N = out_deriv->rows;
  a.row_stride = ix->row_stride;
  for (int i = 0; i < a.K; i++) a.row_offsets[i] = ix->row_offsets[i];
  a.coef = eff_coef;
  a.scale = lr;
  a.G = W_acc;
  a.ldg = ldw;
  a.accumulate = overwrite ? 0 : 1;  // overwrite: W_acc[:, :K Di] = the gradient (every tap computed: no compacted launch)
  a.bias_acc = bias_acc;
  a.active = active_dev;
  a.max_active = max_active;
  TDNNF_HIP(wgrad(a, ws, ws_bytes, (hipStream_t)stream));
  return TDNNF_OK;
}
}  // namespace tdnnf

extern "C" {

int tdnnf_tdnn_update_simple(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do,
                             int Di, const float *eff_coef, float lr, float *W_acc, int ldw, float *bias_acc, void *ws,
                             size_t ws_bytes, tdnnf_stream stream) {
  return tdnn_update_simple_impl(ix, in_value, out_deriv, Do, Di, eff_coef, lr, W_acc, ldw, bias_acc, ws, ws_bytes, nullptr, 0, stream);
}

// AffineComponent::Propagate (nnet-simple-component.cc:1235-1244); bias NULL = LinearComponent :3211-3216
int tdnnf_affine_propagate(const tdnnf_mat *in, const float *W, int ldw, const float *bias, int Do, tdnnf_mat *out,
                           tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && W && Do > 0 && out->cols == Do && in->rows == out->rows && ldw >= in->cols,
                "affine_propagate: bad arguments");
  tdnnf_tdnn_indexes ix;
  memset(&ix, 0, sizeof(ix));
  ix.row_stride = 1;
  ix.num_offsets = 1;
  return tdnnf_tdnn_propagate(&ix, in, W, ldw, Do, in->cols, bias, nullptr, bias ? 1 : 2, out, stream);
}

// Backprop :1262-1264: in_deriv = out_deriv * W (overwrites)
int tdnnf_affine_backprop(const tdnnf_mat *out_deriv, const float *W, int ldw, int Di, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_deriv) && mat_ok(in_deriv) && W && Di > 0 && in_deriv->cols == Di && in_deriv->rows == out_deriv->rows && ldw >= Di,
                "affine_backprop: bad arguments");
  if (out_deriv->rows == 0) return TDNNF_OK;
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = out_deriv->data;
  a.lda = out_deriv->stride;
  a.B = W;
  a.ldb = ldw;
  const float *WT = (ldw == Di && !planes_hint_b()) ? transposed_weights(W) : nullptr;  // (see tdnn_backprop_data_impl)
  if (WT) {
    a.B = WT;
    a.ldb = out_deriv->cols;
  }
  a.C = in_deriv->data;
  a.ldc = in_deriv->stride;
  a.M = out_deriv->rows;
  a.N = Di;
  a.init_mode = 2;
  a.nseg = 1;
  a.seg[0].klen = out_deriv->cols;
  a.seg[0].m_lo = 0;
  a.seg[0].m_hi = out_deriv->rows;
  TDNNF_HIP(rows_gemm(a, WT != nullptr, (hipStream_t)stream));
  return TDNNF_OK;
}

int tdnnf_affine_update_simple(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, float lr, float *W_acc, int ldw,
                               float *bias_acc, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in_value) && mat_ok(out_deriv) && in_value->rows == out_deriv->rows, "affine_update_simple: bad matrices");
  tdnnf_tdnn_indexes ix;
  memset(&ix, 0, sizeof(ix));
  ix.row_stride = 1;
  ix.num_offsets = 1;
  return tdnnf_tdnn_update_simple(&ix, in_value, out_deriv, out_deriv->cols, in_value->cols, nullptr, lr, W_acc, ldw,
                                  bias_acc, ws, ws_bytes, stream);
}

}  // extern "C"
