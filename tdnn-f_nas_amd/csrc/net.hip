// net.hip -- one minibatch of nnet3-chain-train for the TDNN-F graphs of the reference recipes, as a
// host-side C++ executor over the component kernels of this library.
//
// Mirrors (UPSTREAM) NnetChainTrainer::TrainInternal:  forward through every component's Propagate,
// chain::ComputeChainObjfAndDeriv, Backprop through every component (raw-gradient / is_gradient_
// UpdateSimple path), then the shipped optimizer helpers ApplyL2Regularization,
// UpdateNnetWithMaxChange, ConstrainOrthonormal (/root/reference/src/nnet3/nnet-utils.cc:2223-2245,
// :2085-2175, :1040-1077).  Graph: /root/reference/local/chain_NAS/run_tdnn_fbk_40_iv_sp_7q.sh:160-186;
// one tdnnf-layer = steps/libs/nnet3/xconfig/composite_layers.py:135-215, prefinal-layer :1283-1331.
//
// Time bookkeeping replaces the nnet3 compiler for these graphs: every layer's output lives on a regular
// grid (t0, step, n) in t-major row order (row = k*B + b), derived backwards from the output grid
// (0, 3, T/3) exactly as the compiler's dependency analysis would (tdnnf4.linear is computed on the
// padded step-1 grid, as TdnnComponent::ReorderIndexes pads it).
#include <math.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "common.h"
#include "fused.h"
#include "gemm_f32.h"
#include "gemm_ring.h"
#include "net.h"
#include "ng.h"
#include "optim_group.h"

namespace tdnnf {
namespace {

// per-component table for the update kernels
struct UpdTable {
  long long begin[129];
  float lr[128];
  float l2coef[128];
};

// W_acc[o][i*Di + d] += coef[i] * G[o][i*Di + d]   (DARTS: fold the unscaled tap gradients into the accumulator)
__global__ void add_scaled_taps_kernel(const float *G, const float *coef, float *acc, int Do, int KDi, int Di) {
  const long long total = (long long)Do * KDi;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) acc[e] += coef[(e % KDi) / Di] * G[e];
}
// T[o][c] = coef[c / Di] * G[o][c]   (raw gradient of the spliced, coefficient-scaled input from the unscaled tap gradients)
__global__ void scaled_taps_to_kernel(const float *G, const float *coef, int Do, int KDi, int Di, float *T, int ldT) {
  const long long total = (long long)Do * KDi;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int o = (int)(e / KDi), c = (int)(e % KDi);
    T[(size_t)o * ldT + c] = (coef ? coef[c / Di] : 1.0f) * G[e];
  }
}
__global__ void set_column_kernel(const float *v, int rows, float *T, int ldT, int col) {  // (and zeros in the row's padding)
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    T[(size_t)r * ldT + col] = v[r];
    for (int c = col + 1; c < ldT; c++) T[(size_t)r * ldT + c] = 0.f;
  }
}
// W_acc[o][c] += a b T[o][c] (c < ldw), bias_acc[o] += a b T[o][ldw]: "local_lrate = scale * learning_rate_"
// (nnet-tdnn-component.cc:604-624); a, b are the two preconditioners' scales, still on the device
__global__ void ng_commit_kernel(const float *T, int ldT, int Do, int ldw, const float *sa, const float *sb, float *W_acc, float *bias_acc) {
  const float sc = sa[0] * sb[0];
  const int C = ldw + (bias_acc ? 1 : 0);
  const long long total = (long long)Do * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int o = (int)(e / C), c = (int)(e % C);
    const float v = sc * T[(size_t)o * ldT + c];
    if (c < ldw) W_acc[(size_t)o * ldw + c] += v;
    else bias_acc[o] += v;
  }
}
// ---- bottleneck-dimension supernet (scripts/generate_bottleneckCB8share_onehottrain_config.py:10-85).
struct BnChoice {
  int C, mode;
  int cum[8];  // cumulative block widths: candidate bottleneck dims
  float flops_scale, temp;
};
// p (C): choice probabilities, identical for every row -- mode 0 OnehotFunctionComponent::Propagate
// (nnet-simple-component.cc:9504-9519), 1 SoftmaxFlops :9968-9981 on the ConstantFunction output, 2 GumbelSoftmaxFlops
// :10088-10113 (one noise vector shared by all rows).  mask[c] = sum_{j >= block(c)} p_j: CopyN of Sum(p_k..p_{C-1}).
__global__ void bn_choice_forward_kernel(BnChoice bc, const float *alpha, const float *u, float *p, float *mask) {
  __shared__ float sp[8];
  if (threadIdx.x == 0) {
    const int C = bc.C;
    if (bc.mode == 0) {
      for (int i = 0; i < C; i++) sp[i] = (u[0] >= (float)i / C && u[0] < (float)(i + 1) / C) ? 1.0f : 0.0f;
    } else {
      float v[8], mx = -INFINITY;
      for (int i = 0; i < C; i++) {
        v[i] = bc.mode == 2 ? (alpha[i] + -logf(-logf(u[i]))) * (1.0f / bc.temp) : alpha[i];
        mx = fmaxf(mx, v[i]);
      }
      double sum = 0;
      for (int i = 0; i < C; i++) sum += exp((double)v[i] - mx);
      for (int i = 0; i < C; i++) {
        const float q = (float)(exp((double)v[i] - mx) / sum);
        sp[i] = q < 1.0e-20f ? 1.0e-20f : q;  // ApplyFloor(1e-20)
      }
    }
    for (int i = 0; i < C; i++) p[i] = sp[i];
  }
  __syncthreads();
  const int bn = bc.cum[bc.C - 1];
  for (int c = threadIdx.x; c < bn; c += blockDim.x) {
    int k = 0;
    while (c >= bc.cum[k]) k++;
    float m = 0.f;
    for (int j = k; j < bc.C; j++) m += sp[j];  // Sum(softmax_k, ..., softmax_{C-1}) descriptor
    mask[c] = m;
  }
}
// out[r][c] = in[r][c] * mask[c]   (ElementwiseProductComponent :256-274 / :276-299 with a row-constant factor)
__global__ void col_scale_kernel(MatView in, const float *mask, MatView out) {
  const int C = in.cols;
  const long long total = (long long)in.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    out.data[(size_t)r * out.stride + c] = in.data[(size_t)r * in.stride + c] * mask[c];
  }
}
// Gradient of the C-vector.  partial[chunk][c] = sum over the chunk's rows of lin[r][c] * d_masked[r][c]
// (ElementwiseProduct backprop w.r.t. the CopyN factor), E_j = sum_{c < cum[j]} (CopyN backprop + Sum descriptor).
//   mode 0: grad_j += E_j                                                    (OnehotFunction :9539-9548)
//   mode 1/2: e_j = E_j + flops_scale / C * (-cum[j]);  grad_j += 5 * p_j (e_j - <p, e>) / temp
//             ((Gumbel)SoftmaxFlops backprop summed over rows, then ConstantFunction :2636)
__global__ void bn_choice_backward_kernel(BnChoice bc, const float *partial, int chunks, const float *p, float *grad) {
  __shared__ double dm[512];
  const int bn = bc.cum[bc.C - 1];
  for (int c = threadIdx.x; c < bn; c += blockDim.x) {
    double sacc = 0;
    for (int k = 0; k < chunks; k++) sacc += partial[(size_t)k * bn + c];
    dm[c] = sacc;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double E[8], run = 0;
  int c = 0;
  for (int j = 0; j < bc.C; j++) {
    for (; c < bc.cum[j]; c++) run += dm[c];
    E[j] = run;
  }
  if (bc.mode == 0) {
    for (int j = 0; j < bc.C; j++) grad[j] += (float)E[j];
    return;
  }
  double pe = 0;
  for (int j = 0; j < bc.C; j++) {
    E[j] += (double)bc.flops_scale / bc.C * -(double)bc.cum[j];
    pe += (double)p[j] * E[j];
  }
  for (int j = 0; j < bc.C; j++) grad[j] += 5.0f * (float)(p[j] * (E[j] - pe)) * (1.0f / bc.temp);
}
// partial[chunk][c] = sum_{r in chunk} a[r][c] * b[r][c]
__global__ __launch_bounds__(256) void colsum_prod_partial_kernel(MatView a, MatView b, int rows_per_chunk, float *partial) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(a.rows, r0 + rows_per_chunk);
  if (col >= a.cols) return;
  float sacc = 0.f;
  for (int r = r0; r < r1; r++) sacc += a.data[(size_t)r * a.stride + col] * b.data[(size_t)r * b.stride + col];
  partial[(size_t)blockIdx.y * a.cols + col] = sacc;
}
// active[0] = number of taps with a non-zero effective coefficient, active[1..] = their ids
__global__ void active_taps_kernel(const float *eff, int K, int *active) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int i = 0; i < K; i++)
    if (eff[i] != 0.f) active[1 + n++] = i;
  active[0] = n;
}
// grads += this minibatch's gradient, unless the chain objective failed (results[5] == 0): then, as in the
// reference (derivatives set to zero), the minibatch contributes nothing
__global__ void commit_grads_kernel(float *grads, const float *gtmp, long long n, const double *results) {
  if (results[5] == 0.0) return;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) grads[i] += gtmp[i];
}
// paramsT[begin_c + col * rows + row] = params[begin_c + row * cols + col] for every component c (blockIdx.y)
struct TransTable {
  long long begin[128];
  int rows[128], cols[128];
};
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float *params, float *paramsT, TransTable tb) {
  const int c = blockIdx.y, rows = tb.rows[c], cols = tb.cols[c];
  const float *W = params + tb.begin[c];
  float *WT = paramsT + tb.begin[c];
  const long long total = (long long)rows * cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int col = (int)(e / rows), row = (int)(e % rows);  // consecutive threads: consecutive rows of one column -> coalesced writes
    WT[e] = W[(long long)row * cols + col];
  }
}
// out (cols x rows) = in (rows x cols)^T, both dense
__global__ void transpose_kernel(const float *in, int rows, int cols, float *out) {
  const long long total = (long long)rows * cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    out[(size_t)c * rows + r] = in[e];
  }
}
// ScaleBatchnormStats: every BatchNorm's [count, sum[D], sumsq[D]] *= s in one launch (block row = one component)
struct ScaleTable {
  double *p[48];
  int n[48];
};
__global__ void scale_doubles_kernel(ScaleTable tb, double s) {
  double *x = tb.p[blockIdx.y];
  const int n = tb.n[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] *= s;
}
// lda input: [feats(k+j, b), j < S ; ivector(b)]
__global__ void splice_input_kernel(MatView feats, MatView iv, int B, int S, MatView out) {
  const int C = out.cols, fd = feats.cols;
  const long long total = (long long)out.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C), k = r / B, b = r % B;
    float v;
    if (c < S * fd) v = feats.data[(size_t)((k + c / fd) * B + b) * feats.stride + c % fd];
    else v = iv.data[(size_t)b * iv.stride + (c - S * fd)];
    out.data[(size_t)r * out.stride + c] = v;
  }
}
__global__ void reorder_rows_kernel(MatView in, int B, int rho, int to_rho, MatView out) {
  const int C = in.cols;
  const long long total = (long long)in.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    const int tau = r / B, b = r % B;  // plain t-major coordinates
    const int pr = (tau / rho) * rho * B + b * rho + tau % rho;
    if (to_rho) out.data[(size_t)pr * out.stride + c] = in.data[(size_t)r * in.stride + c];
    else out.data[(size_t)r * out.stride + c] = in.data[(size_t)pr * in.stride + c];
  }
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;


namespace {

int N_of(const Grid &g, int B) { return g.n * B; }

void make_tdnn(Tdnn *t, int comp, int Di, int Do, const std::vector<int> &offs, const Grid &in, const Grid &out, int B) {
  const int K = (int)offs.size();
  t->comp = comp;
  t->Di = Di;
  t->Do = Do;
  t->K = K;
  t->darts = false;
  t->share = 0;
  t->draw0 = 0;
  t->memo = nullptr;
  t->active = nullptr;
  for (int i = 0; i < K; i++) t->offsets[i] = offs[i];
  t->in = in;
  t->out = out;
  memset(&t->ix, 0, sizeof(t->ix));
  const int rho = out.step / in.step;
  t->ix.row_stride = rho;
  t->ix.num_offsets = K;
  for (int i = 0; i < K; i++) {  // PrecomputeIndexes, nnet-tdnn-component.cc:878-903
    const int req = out.t0 + t->offsets[i];
    const int input_t = (req - in.t0) / in.step;
    t->ix.row_offsets[i] = rho * (input_t / rho) * B + input_t % rho;
  }
  t->rows_in = in.n * B;
  t->rows_out = out.n * B;
}

// Row stride of every activation matrix of the trainer: a multiple of 32 floats, so that rows start on 128-byte lines.
// (With the 16-byte minimum, the 6034-wide output matrices ran their GEMMs at 80 instead of 120 TFLOP/s: every 128-byte
// row segment a tile load fetches straddled two cache lines.)
inline int ldpad(int cols) { return (cols + 31) & ~31; }
tdnnf_mat M(float *p, int rows, int cols) { return tdnnf_mat{p, rows, cols, ldpad(cols)}; }

struct Arena {
  size_t off = 0;
  char *base = nullptr;
  template <class T>
  T *take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += sizeof(T) * n;
    return p;
  }
  float *mat(int rows, int cols) { return take<float>((size_t)rows * ldpad(cols)); }
};

int add_comp(tdnnf_net *n, const std::string &name, int rows, int cols, int has_bias, float lr_factor, float l2, float mc,
             float ortho, int num_alpha = 0) {
  CompDesc c;
  c.num_alpha = num_alpha;
  c.name = name;
  c.begin = n->num_params;
  c.rows = rows;
  c.cols = cols;
  c.has_bias = has_bias;
  c.lr_factor = lr_factor;
  c.l2 = l2;
  c.max_change = mc;
  c.orthonormal = ortho;
  n->num_params += c.size();
  n->num_params = (n->num_params + 3) & ~3LL;  // keep every matrix 16-byte aligned
  n->comps.push_back(c);
  return (int)n->comps.size() - 1;
}

float *Wp(const tdnnf_net *n, int comp) { return n->params + n->comps[comp].begin; }
float *Bp(const tdnnf_net *n, int comp) {
  const CompDesc &c = n->comps[comp];
  return c.has_bias ? n->params + c.begin + (long long)c.rows * c.cols + c.num_alpha : nullptr;
}
float *Ap(const tdnnf_net *n, int comp) { return n->params + n->comps[comp].begin + (long long)n->comps[comp].rows * n->comps[comp].cols; }
float *Ag(const tdnnf_net *n, int comp) { return n->gtmp + n->comps[comp].begin + (long long)n->comps[comp].rows * n->comps[comp].cols; }
float *Wg(const tdnnf_net *n, int comp) { return n->gtmp + n->comps[comp].begin; }
float *Bg(const tdnnf_net *n, int comp) {
  const CompDesc &c = n->comps[comp];
  return c.has_bias ? n->gtmp + c.begin + (long long)c.rows * c.cols + c.num_alpha : nullptr;
}

// carve (or, with base == nullptr, just size) every activation buffer
void layout_arena(tdnnf_net *n, Arena &A) {
  const tdnnf_net_config &c = n->cfg;
  const int B = n->B, Hd = c.hidden_dim, S = c.prefinal_small_dim, P = c.num_pdfs;
  const int lda_dim = 3 * c.feat_dim + c.ivector_dim;
  const int N0 = N_of(n->g_lda, B);
  n->lda_in = A.mat(N0, lda_dim);
  n->lda_out = A.mat(N0, lda_dim);
  n->t1_relu = A.mat(N0, Hd);
  n->t1_bn = A.mat(N0, Hd);
  n->t1_bn_memo = A.take<float>(5 * Hd);
  n->t1_bn_stats = A.take<double>(1 + 2 * Hd);
  n->t1_relu_stats = A.take<double>(2 + 3 * Hd);  // [count, value_sum, deriv_sum, oderiv_count, oderiv_sumsq]
  int max_rows = N0, max_lin_rows = 0;
  for (auto &L : n->layers) {
    const int nl = N_of(L.lin.out, B), no = N_of(L.gout, B);
    L.lin_out = A.mat(nl, L.bn);
    L.lin_perm = L.perm ? A.mat(nl, L.bn) : nullptr;
    L.arch_p = L.arch_mask = L.lin_masked = nullptr;
    if (L.c_arch >= 0) {
      L.arch_p = A.take<float>(8);
      L.arch_mask = A.take<float>(L.bn + 4);
      L.lin_masked = A.mat(nl, L.bn);
    }
    L.relu_out = A.mat(no, Hd);
    L.noop_out = A.mat(no, Hd);
    L.bn_memo = A.take<float>(5 * Hd);
    L.lin.memo = L.lin.darts ? A.take<float>(2 * TDNNF_MAX_OFFSETS) : nullptr;
    L.aff.memo = L.aff.darts ? A.take<float>(2 * TDNNF_MAX_OFFSETS) : nullptr;
    L.lin.active = L.lin.darts ? A.take<int>(TDNNF_MAX_OFFSETS + 1) : nullptr;
    L.aff.active = L.aff.darts ? A.take<int>(TDNNF_MAX_OFFSETS + 1) : nullptr;
    L.bn_stats = A.take<double>(1 + 2 * Hd);
    L.relu_stats = A.take<double>(2 + 3 * Hd);
    max_rows = std::max(max_rows, std::max(no, N_of(L.gin, B)));
    max_lin_rows = std::max(max_lin_rows, nl);
  }
  const int No = n->Tout * B;
  n->prefinal_l_out = A.mat(No, S);
  for (int h = 0; h < 2; h++) {
    auto &H = n->head[h];
    H.aff_relu = A.mat(No, Hd);
    H.bn1_out = A.mat(No, Hd);
    H.lin_out = A.mat(No, S);
    H.bn2_out = A.mat(No, S);
    H.y = A.mat(No, P);
    H.bn1_memo = A.take<float>(5 * Hd);
    H.bn2_memo = A.take<float>(5 * S);
    H.bn1_stats = A.take<double>(1 + 2 * Hd);
    H.bn2_stats = A.take<double>(1 + 2 * S);
    H.relu_stats = A.take<double>(2 + 3 * Hd);
  }
  n->xent_logsoftmax = A.mat(No, P);
  n->d_y = A.mat(No, P);
  n->d_xent = A.mat(No, P);
  n->dA = A.mat(max_rows, Hd);
  n->dB = A.mat(max_rows, Hd);
  n->dC = A.mat(max_rows, Hd);
  n->d_small = A.mat(std::max(max_lin_rows, No), std::max(S, 512));
  n->d_small2 = A.mat(std::max(max_lin_rows, No), std::max(S, 512));
  // weight gradients three components behind the caller's stream (wg_lag 3, net.h): a second buffer for the derivative the affine's
  // gradient reads, two more for what the linear's reads -- layers alternate between them
  n->wg_lag = options().wgrad_lag == 1 ? 1 : 3;
  const bool wg_size = options().wgrad_stream >= 0 ? options().wgrad_stream != 0 : std::max(max_rows, N0) <= 32768;
  const bool lag3 = wg_size && n->wg_lag == 3;
  n->dC2 = lag3 ? A.mat(max_rows, Hd) : nullptr;
  n->dS[0] = lag3 ? A.mat(std::max(max_lin_rows, No), std::max(S, 512)) : nullptr;
  n->dS[1] = lag3 ? A.mat(std::max(max_lin_rows, No), std::max(S, 512)) : nullptr;
  size_t tg = 0;
  for (auto &L : n->layers)
    if (L.lin.darts) tg = std::max(tg, (size_t)L.bn * L.lin.K * Hd);
  n->tapgrad = tg ? A.take<float>(tg) : nullptr;
  n->tapdots = A.take<double>(TDNNF_TAP_DOTS_DOUBLES(TDNNF_MAX_OFFSETS));
  n->bn_sync.buf = A.take<double>(5 * (size_t)std::max(std::max(Hd, S), 1) + 8);  // (the ReLU backward sweep stages five column sums)
  n->dropout_masks = (c.use_dropout && !c.cv_update) ? A.take<float>((size_t)(c.num_layers + 1) * B * Hd) : nullptr;
  n->gtmp = A.take<float>((size_t)n->num_params + 16);
  // transposed copy of every weight matrix for the split-bf16 backward-data GEMMs (k-contiguous B operand).  (Measured for exact
  // f32 too, twice: no gain -- docs/experiments.md.)
  n->paramsT = (n->cfg.gemm_precision == 1 || n->cfg.gemm_precision == 2) ? A.take<float>((size_t)n->num_params + 16) : nullptr;
  // ---- pre-split plane operands (net.h): a slot per GEMM operand matrix, keyed by its base pointer
  n->planes_np = n->cfg.gemm_precision == 3 ? 2 : (n->cfg.gemm_precision == 2 && options().planes ? 3 : 0);
  n->plane_slots.clear();
  n->pw.assign(n->comps.size(), PlanesOperand());
  n->pw_scale.assign(n->comps.size(), nullptr);
  n->planes_ws = nullptr;
  n->fro_buf = nullptr;
  if (n->planes_np) {
    const int np = n->planes_np;
    int lead_cap = 0;  // the largest row shift of a backward-data view (taps of rho == 1 layers)
    for (auto &L : n->layers)
      for (const Tdnn *td : {&L.lin, &L.aff})
        for (int i = 0; i < td->K; i++) lead_cap = std::max(lead_cap, td->ix.row_offsets[i]);
    auto slot = [&](const float *key, int rows, int cols, bool with_lead) {
      const long long R = planes_rows_padded((long long)(with_lead ? 2 * ((lead_cap + 15) & ~15) : 0) + rows + 256);
      const long long Rt = planes_rows_padded(((cols + 255) / 256) * 256LL);
      tdnnf_net::PlaneSlot ps;
      ps.bytesP = planes_bytes(np, R, (planes_kblocks(cols) + 15) / 16 * 16);  // (whole 256-column tiles of K blocks: the rows-as-K reads of a weight gradient)
      ps.bytesPT = planes_bytes(np, Rt, planes_t_kblocks(rows));
      ps.P = A.take<char>(ps.bytesP + 64);
      ps.PT = A.take<char>(ps.bytesPT + 64);
      ps.scale = A.take<float>(4);
      if (A.base) n->plane_slots[key] = ps;
    };
    const int big_rows = std::max(max_rows, std::max(N0, No)), small_rows = std::max(max_lin_rows, No), small_cols = std::max(S, 512);
    slot(n->lda_out, N0, lda_dim, false);
    slot(n->t1_bn, N0, Hd, false);
    for (auto &L : n->layers) {
      slot(L.noop_out, N_of(L.gout, B), Hd, false);
      slot(L.lin_out, N_of(L.lin.out, B), L.bn, false);
      if (L.c_arch >= 0) slot(L.lin_masked, N_of(L.lin.out, B), L.bn, false);  // (bottleneck supernet: the affine reads the masked blocks)
    }
    slot(n->prefinal_l_out, No, S, false);
    for (int h = 0; h < 2; h++) {
      slot(n->head[h].bn1_out, No, Hd, false);
      slot(n->head[h].bn2_out, No, S, false);
    }
    slot(n->d_y, No, P, false);
    slot(n->d_xent, No, P, false);
    slot(n->dA, big_rows, Hd, true);
    slot(n->dB, big_rows, Hd, true);
    slot(n->dC, big_rows, Hd, true);
    slot(n->d_small, small_rows, small_cols, true);
    slot(n->d_small2, small_rows, small_cols, true);
    n->planes_ws = A.take<char>(planes_sumsq_ws_bytes() + 64);
    n->fro_buf = A.take<double>(finalize_grid(std::max(Hd, S)) + 8);
    // the weight matrices: row-major planes (forward: one row per output, k contiguous) and transposed planes (backward-data)
    for (size_t i = 0; i < n->comps.size(); i++) {
      const CompDesc &cd = n->comps[i];
      if (cd.plain || cd.rows < 2 || (int)i == n->c_lda) continue;
      PlanesOperand &o = n->pw[i];
      o.rows = cd.rows; o.cols = cd.cols; o.ld = cd.cols; o.np = np; o.lead = 0;
      o.R = planes_rows_padded(((cd.rows + 255) / 256) * 256LL);
      o.Rt = planes_rows_padded(((cd.cols + 255) / 256) * 256LL);
      o.P = A.take<char>(planes_bytes(np, o.R, planes_kblocks(cd.cols)) + 64);
      o.PT = A.take<char>(planes_bytes(np, o.Rt, planes_t_kblocks(cd.rows)) + 64);
      n->pw_scale[i] = A.take<float>(4);
      o.scale = n->pw_scale[i];
    }
  }
  n->s3_scratch = nullptr;
  n->s3_scratch_bytes = 0;
  n->ng_grouped = options().ng_grouped != 0;  // 0: the per-object side chain for every component
  size_t tall = 0, tall_ws = 0;
  for (auto &cd : n->comps)
    if (cd.orthonormal != 0.f && cd.rows > cd.cols) {
      tall = std::max(tall, (size_t)cd.rows * cd.cols);
      tall_ws = std::max(tall_ws, tdnnf_constrain_orthonormal_workspace_bytes(cd.cols, cd.rows));
    }
  n->orthoT = tall ? A.take<float>(tall + 16) : nullptr;
  size_t ng_ws = 0;
  n->ngc.assign(n->comps.size(), tdnnf_net::NgComp());
  if (n->cfg.use_natural_gradient) {
    size_t mtmp = 0;
    auto comp_ng = [&](int comp, int K, int rows) {  // rows = N of the component's output grid
      const CompDesc &cd = n->comps[comp];
      if (!cd.updatable || cd.plain) return;  // (sized whatever the learning-rate factor is: an edit may unfreeze a component)
      const int Dx = cd.cols + (cd.has_bias ? 1 : 0), ldT = (Dx + 3) & ~3;
      const int rank_in = std::min(20, (Dx + 1) / 2), rank_out = std::min(80, (cd.rows + 1) / 2);
      const int Rpi = (rank_in + 3) & ~3, Rpo = (rank_out + 3) & ~3;
      mtmp = std::max(mtmp, std::max((size_t)cd.rows * Rpi, (size_t)Rpo * ldT));
      ng_ws = std::max(ng_ws, std::max(ng_stats_workspace_bytes(rank_in, Dx, K, rows), ng_stats_workspace_bytes(rank_out, cd.rows, 1, rows)));
      auto &S = n->ngc[comp];
      S.N = rows;
      S.H_in = A.take<float>((size_t)rows * Rpi + 64);
      S.H_out = A.take<float>((size_t)rows * Rpo + 64);
      S.T = A.take<float>((size_t)cd.rows * ldT + 16);
      S.bsum = nullptr;  // (carved below: one block for all components, zeroed once per step)
      S.part_in = A.take<double>((size_t)rows_gemm_sumsq_blocks(rows) + 8);
      S.part_out = A.take<double>((size_t)rows_gemm_sumsq_blocks(rows) + 8);
    };
    const int No_ = n->Tout * B;
    comp_ng(n->tdnn1.comp, 1, N0);
    for (auto &L : n->layers) {
      comp_ng(L.lin.comp, L.lin.K, L.lin.rows_out);
      comp_ng(L.aff.comp, L.aff.K, L.aff.rows_out);
    }
    comp_ng(n->c_prefinal_l, 1, No_);
    for (int h = 0; h < 2; h++) {
      comp_ng(n->head[h].c_affine, 1, No_);
      comp_ng(n->head[h].c_linear, 1, No_);
      comp_ng(n->head[h].c_output, 1, No_);
    }
    {
      size_t tot = 0;
      for (size_t i = 0; i < n->comps.size(); i++)
        if (n->ngc[i].N > 0) tot += ((size_t)n->comps[i].rows + 15) & ~(size_t)15;  // (N, not the pointers: the sizing pass has none)
      n->ng_bsum_floats = tot;
      n->ng_bsum_all = A.take<float>(tot + 16);
      size_t o = 0;
      for (size_t i = 0; i < n->comps.size(); i++)
        if (n->ngc[i].N > 0) {
          n->ngc[i].bsum = n->ng_bsum_all ? n->ng_bsum_all + o : nullptr;
          o += ((size_t)n->comps[i].rows + 15) & ~(size_t)15;
        }
    }
    const size_t maxN = (size_t)std::max(std::max(max_rows, N0), No_);
    n->ngset_ws_bytes = wgrad_workspace_bytes(80, 80, 1, (int)maxN) + 256;
    n->ngTmp = A.take<float>(mtmp + 64);
    n->ng_side_ws = A.take<char>(n->ngset_ws_bytes);
    n->s3_scratch_bytes = 16u << 20;
    n->s3_scratch = A.take<float>(n->s3_scratch_bytes / sizeof(float));
  }
  // shared workspace: wgrad slabs, column reductions, orthonormal
  size_t ws = 0;
  auto upd = [&](size_t b) { ws = std::max(ws, b); };
  upd(wgrad_workspace_bytes(Hd, lda_dim, 1, N0));
  upd(colreduce_bytes(max_rows, Hd));
  upd(sizeof(float) * 2 * (size_t)Hd * rows_gemm_colstats_cap(std::max(max_rows, std::max(N0, No))));  // BatchNorm partials out of the GEMM epilogue
  upd(bn_relu_bwd_workspace_bytes(max_rows, Hd));
  for (auto &L : n->layers) {
    upd(wgrad_workspace_bytes(L.lin.Do, L.lin.Di, L.lin.K, L.lin.rows_out));
    upd(wgrad_workspace_bytes(L.aff.Do, L.aff.Di, L.aff.K, L.aff.rows_out));
    upd(tdnnf_constrain_orthonormal_workspace_bytes(L.bn, L.lin.K * Hd));
  }
  upd(wgrad_workspace_bytes(S, Hd, 1, No));
  upd(wgrad_workspace_bytes(Hd, S, 1, No));
  upd(wgrad_workspace_bytes(P, S, 1, No));
  upd(colreduce_bytes(No, P));
  upd(tdnnf_constrain_orthonormal_workspace_bytes(S, Hd));
  upd(tdnnf_max_change_workspace_bytes((int)n->comps.size()));
  upd(ng_ws);
  upd(tall_ws);
  if (n->cfg.bn_num_choices > 0) upd(sizeof(float) * (size_t)((max_lin_rows + 511) / 512 + 1) * 512);
  n->ws_bytes = ws + 256;
  n->ws = A.take<char>(n->ws_bytes);
  n->wg_on = options().wgrad_stream >= 0 ? options().wgrad_stream != 0 : std::max(max_rows, N0) <= 32768;
  // (input-side statistics ahead of the backward pass: for minibatches whose GEMMs fill the chip.  With the weight-gradient streams three
  // components behind the caller's stream the small minibatches lose by it -- 150 x 64 11.98 -> 11.25 ms, 1500 x 16 22.12 -> 21.60 on one box
  // with it off: the statistics then run with their component's gradient instead of in front of the heads' gradients.  Option ng_early_in 2 forces it.)
  // (option ng_early_in 3, weight-gradient streams on: the passes of ALL components as ONE grouped launch on s4 -- rows_gemm_group, 33 launches of
  // 26 .. 78 blocks each at 150 x 64 -- and J of a refresh step left to the component's own gradient call.  Measured 11.01 against 10.92 ms at
  // 150 x 64, 21.28 / 21.28 at 1500 x 16: fewer launches, the same work, no faster -- off.)
  n->early_on = n->cfg.use_natural_gradient && n->ng_grouped && (options().ng_early_in >= 2 || (options().ng_early_in != 0 && !n->wg_on));
  n->early_group = n->early_on && n->wg_on && options().ng_early_in == 3;
  const bool s4_used = n->wg_on || n->early_on;
  n->ws4 = s4_used ? A.take<char>(n->ws_bytes) : nullptr;
  n->s4_scratch_bytes = s4_used ? (32u << 20) : 0;
  n->s4_scratch = s4_used ? A.take<float>(n->s4_scratch_bytes / sizeof(float)) : nullptr;
  const bool two = n->wg_on && options().wgrad_stream != 1;  // (option wgrad_stream: 1 = one weight-gradient stream as rounds 2-3, 2 = two, -1 = by size, two)
  n->ws2 = two ? A.take<char>(n->ws_bytes) : nullptr;
  n->s2_scratch = two ? A.take<float>(n->s4_scratch_bytes / sizeof(float)) : nullptr;
  // (option wgrad_stream 3: a third weight-gradient stream, s5 -- beside s4 from the start of the backward pass, beside s4 and s2 once the denominator has joined)
  const bool three = two && options().wgrad_stream == 3;
  n->ws5 = three ? A.take<char>(n->ws_bytes) : nullptr;
  n->s5_scratch = three ? A.take<float>(n->s4_scratch_bytes / sizeof(float)) : nullptr;
}

// diagnostics: phase boundary k of the step (option phase_events)
static int phase_mark(tdnnf_net *n, int k, hipStream_t s) {
  if (!tdnnf::options().phase_events) return TDNNF_OK;
  if (!n->ev_phase[k]) TDNNF_HIP(hipEventCreate(&n->ev_phase[k]));
  TDNNF_HIP(hipEventRecord(n->ev_phase[k], s));
  n->phase_rec[k] = true;
  return TDNNF_OK;
}

#define CK(expr)             \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

// view of the rows of a t-major matrix (grid g, B sequences, `cols` wide) that lie on a coarser grid `sub`
tdnnf_mat sub_grid_view(float *data, const Grid &g, const Grid &sub, int B, int cols) {
  const int stride = ldpad(cols);
  const int tau0 = (sub.t0 - g.t0) / g.step, ratio = sub.step / g.step;
  if (ratio == 1) return tdnnf_mat{data + (size_t)tau0 * B * stride, sub.n * B, cols, stride};
  // every ratio-th block of B rows: n "super rows" of B*stride elements
  return tdnnf_mat{data + (size_t)tau0 * B * stride, sub.n, B * stride - (stride - cols), ratio * B * stride};
}

// GeneralDropoutComponent::GetMemo (UPSTREAM), continuous form: mask = 1 - 2p + 4p U, U uniform on (0, 1)
__global__ void dropout_mask_kernel(const float *u, float p, long long n, float *mask) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    mask[i] = 1.0f - 2.0f * p + 4.0f * p * u[i];
}

// BatchNormTestComponent (cv-update): memo rows 0 (mean) and 2 (scale) from the stored statistics, ComputeDerived
// nnet-normalize-component.cc:682-715; rows 3 and 4 (backward terms of the train-mode component) are zero
__global__ void bn_test_memo_kernel(const double *stats, int D, float epsilon, float target_rms, float *memo) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const double count = stats[0];
  const float off = (float)(stats[1 + d] * (-1.0 / count));
  float sc = (float)(stats[1 + D + d] * (1.0 / count));
  sc += -1.0f * off * off;
  memo[D + d] = sc;
  sc = floor_keep_nan(sc, 0.f) + epsilon;
  sc = target_rms / sqrtf(sc);
  memo[d] = -off;
  memo[2 * D + d] = sc;
  memo[3 * D + d] = 0.f;
  memo[4 * D + d] = 0.f;
}
int bn_test_memo(float *memo, const double *stats, int cols, hipStream_t s) {
  hipLaunchKernelGGL(bn_test_memo_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, stats, cols, 1.0e-3f, 1.0f, memo);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int bn_fwd(tdnnf_net *n, float *in, float *out, int rows, int cols, float *memo, double *stats, hipStream_t s) {
  tdnnf_mat a = M(in, rows, cols), o = M(out, rows, cols);
  if (n->cfg.cv_update) {
    CK(bn_test_memo(memo, stats, cols, s));
    const MatView none{nullptr, 0, 0, 0};
    TDNNF_HIP(bn_apply_bypass(view(&a), memo, cols, ldpad(cols), none, 0.f, view(&o), s));
    return TDNNF_OK;
  }
  CK(tdnnf_batchnorm_propagate(&a, 1.0e-3f, 1.0f, &o, memo, n->ws, n->ws_bytes, s));
  return tdnnf_batchnorm_store_stats(memo, cols, rows, stats, s);  // StoreStats runs on every minibatch
}
// Affine (+ bias) + ReLU into `out` and the BatchNorm statistics of that output; the GEMM's epilogue forms the column sums
// while it stores the tile when it can (exact-f32 128-wide tile), otherwise a pass over `out` does.  The normalisation itself
// is applied later by a fused pass.
int affine_relu_bn_stats(tdnnf_net *n, const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in, const float *W, int ldw, int Do, int Di, const float *bias,
                         const float *eff, tdnnf_mat *out, float *memo, double *stats, hipStream_t s) {
  if (n->cfg.cv_update) {
    CK(tdnn_propagate_impl(ix, in, W, ldw, Do, Di, bias, eff, 1, 1, out, s));
    return bn_test_memo(memo, stats, Do, s);
  }
  int prows = 0;
  const bool room = n->ws_bytes >= sizeof(float) * 2 * (size_t)Do * rows_gemm_colstats_cap(out->rows);
  CK(tdnn_propagate_impl(ix, in, W, ldw, Do, Di, bias, eff, 1, 1, out, s, room ? (float *)n->ws : nullptr, room ? &prows : nullptr));
  // (StoreStats runs on every minibatch: in the finalize launch)
  if (prows > 0) TDNNF_HIP(batchnorm_stats_from_partials((const float *)n->ws, prows, out->rows, Do, 1.0e-3f, 1.0f, memo, s, stats));
  else TDNNF_HIP(batchnorm_stats(view(out), 1.0e-3f, 1.0f, memo, n->ws, s, stats));
  return TDNNF_OK;
}
// statistics of BatchNorm(x) only; the normalisation itself is applied by a fused pass
int bn_stats(tdnnf_net *n, float *x, int rows, int cols, float *memo, double *stats, hipStream_t s) {
  if (n->cfg.cv_update) return bn_test_memo(memo, stats, cols, s);
  tdnnf_mat a = M(x, rows, cols);
  TDNNF_HIP(batchnorm_stats(view(&a), 1.0e-3f, 1.0f, memo, n->ws, s, stats));
  return TDNNF_OK;
}

}  // namespace

extern "C" {

int tdnnf_splice_input(const tdnnf_mat *feats, const tdnnf_mat *iv, int B, int S, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(feats) && mat_ok(iv) && mat_ok(out) && B > 0 && S > 0, "splice_input: bad arguments");
  TDNNF_REQUIRE(out->rows % B == 0 && feats->rows == (out->rows / B + S - 1) * B && iv->rows == B &&
                    out->cols == S * feats->cols + iv->cols,
                "splice_input: feats must have out_frames + num_splice - 1 time steps and out.cols = S*feat_dim + ivector_dim");
  if (out->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(splice_input_kernel, dim3(grid_for((long long)out->rows * out->cols, 256)), dim3(256), 0, (hipStream_t)stream,
                     view(feats), view(iv), B, S, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_reorder_rows(const tdnnf_mat *in, int B, int rho, int to_rho, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && B > 0 && rho >= 1 && in->rows % (B * rho) == 0 && in->data != out->data,
                "reorder_rows: rows must be a multiple of num_seq*rho and in != out");
  if (in->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(reorder_rows_kernel, dim3(grid_for((long long)in->rows * in->cols, 256)), dim3(256), 0, (hipStream_t)stream,
                     view(in), B, rho, to_rho, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

static int net_create_impl(const tdnnf_net_config *cfg, const tdnnf_net *share, tdnnf_net **out);

int tdnnf_net_create(const tdnnf_net_config *cfg, tdnnf_net **out) { return net_create_impl(cfg, nullptr, out); }

int tdnnf_net_create_shared(const tdnnf_net_config *cfg, const tdnnf_net *primary, tdnnf_net **out) {
  TDNNF_REQUIRE(primary, "net_create_shared: null primary net");
  return net_create_impl(cfg, primary, out);
}

static int net_create_impl(const tdnnf_net_config *cfg, const tdnnf_net *share, tdnnf_net **out) {
  TDNNF_REQUIRE(cfg && out, "net_create: null argument");
  const tdnnf_net_config &c = *cfg;
  TDNNF_REQUIRE(c.feat_dim > 0 && c.ivector_dim > 0 && c.num_pdfs > 0 && c.hidden_dim > 0 && c.prefinal_small_dim > 0,
                "net_create: dims must be positive");
  TDNNF_REQUIRE(c.num_layers >= 1 && c.num_layers <= TDNNF_NET_MAX_LAYERS, "net_create: 1..%d tdnnf layers", TDNNF_NET_MAX_LAYERS);
  TDNNF_REQUIRE(c.darts_num_offsets == 0 || (c.darts_num_offsets >= 2 && c.darts_num_offsets <= TDNNF_MAX_OFFSETS),
                "net_create: darts_num_offsets must be 0 or 2..%d (the reference assumes K >= 2, nnet-tdnn-component.cc:232)", TDNNF_MAX_OFFSETS);
  if (c.bn_num_choices != 0) {
    TDNNF_REQUIRE(c.bn_num_choices >= 2 && c.bn_num_choices <= 8 && c.bn_mode >= 0 && c.bn_mode <= 2, "net_create: bn_num_choices must be 2..8, bn_mode 0..2");
    TDNNF_REQUIRE(c.darts_num_offsets == 0, "net_create: the bottleneck and the offset supernet cannot be combined");
    TDNNF_REQUIRE(c.bn_mode != 2 || c.bn_temp_proportion > 0, "net_create: bn_temp_proportion must be > 0");
    int sum = 0;
    for (int k = 0; k < c.bn_num_choices; k++) {
      TDNNF_REQUIRE(c.bn_choice_dims[k] > 0, "net_create: bn_choice_dims must be positive");
      sum += c.bn_choice_dims[k];
    }
    TDNNF_REQUIRE(sum <= 512, "net_create: bottleneck supernet wider than 512");
    for (int l = 0; l < c.num_layers; l++)
      TDNNF_REQUIRE(c.bottleneck_dim[l] == sum, "net_create: bottleneck_dim[%d] = %d but the choice blocks sum to %d", l, c.bottleneck_dim[l], sum);
  }
  TDNNF_REQUIRE(c.gemm_precision >= 0 && c.gemm_precision <= 3,
                "net_create: gemm_precision must be 0 (f32), 1 (split-bf16, 3 products), 2 (split-bf16, 6 products) or 3 (pre-split scaled f16 pairs, 3 products)");
  TDNNF_REQUIRE(c.darts_num_offsets == 0 || !(c.darts_flags & TDNNF_DARTS_USE_GUMBEL) || c.darts_temp_proportion > 0,
                "net_create: gumbel mode needs temp-proportion > 0");
  TDNNF_REQUIRE(c.frame_subsampling >= 1 && c.frames_per_chunk > 0 && c.frames_per_chunk % c.frame_subsampling == 0 && c.num_sequences > 0,
                "net_create: frames_per_chunk must be a positive multiple of frame_subsampling");
  tdnnf_net *n = new tdnnf_net();
  n->cfg = c;
  n->num_params = 0;
  n->params = n->grads = nullptr;
  n->arena = nullptr;
  n->B = c.num_sequences;
  n->T = c.frames_per_chunk;
  n->Tout = c.frames_per_chunk / c.frame_subsampling;
  const int B = n->B, Hd = c.hidden_dim, S = c.prefinal_small_dim, P = c.num_pdfs, lda_dim = 3 * c.feat_dim + c.ivector_dim;
  // ---- grids, derived backwards from the output grid
  n->layers.resize(c.num_layers);
  Grid g{0, c.frame_subsampling, n->Tout};
  for (int l = c.num_layers - 1; l >= 0; l--) {
    TdnnfLayer &L = n->layers[l];
    L.stride = c.time_stride[l];
    L.left = c.use_layer_offsets ? c.offset_left[l] : L.stride;
    L.right = c.use_layer_offsets ? c.offset_right[l] : L.stride;
    L.bn = c.bottleneck_dim[l];
    TDNNF_REQUIRE(L.bn > 0 && L.bn <= 512 && L.left >= 0 && L.right >= 0 && L.left <= 64 && L.right <= 64,
                  "net_create: layer %d: bottleneck-dim must be in 1..512, time-stride / layer offsets in 0..64", l);
    L.gout = g;
    L.perm = false;
    Grid lin = g, in = g;
    const int Kd = c.darts_num_offsets;
    if (Kd >= 2) {
      // offset supernet: taps -(K-1)..0 / 0..K-1 at the input frame rate on every layer
      if (g.step == 1) {
        lin = Grid{g.t0, 1, g.n + Kd - 1};
      } else {
        const int rho = g.step;
        const int cnt = rho * (g.n - 1) + Kd;
        lin = Grid{g.t0, 1, ((cnt + rho - 1) / rho) * rho};  // padded to a multiple of rho (:841-843)
        L.perm = true;
      }
      in = Grid{lin.t0 - (Kd - 1), 1, lin.n + Kd - 1};
    } else if (L.left > 0 || L.right > 0) {
      // X.linear taps {-a, 0}, X.affine taps {0, b} (time-stride s: a = b = s; a derived child: any a, b >= 0).  The
      // linear runs on the coarsest regular grid that holds every frame the affine needs and whose own taps stay on the
      // input grid: step gcd(output step, a, b).  When that is finer than the output grid the affine has row_stride
      // rho > 1 and the grid is padded to a multiple of rho (nnet-tdnn-component.cc:841-843).
      const int a = L.left, b = L.right;
      auto gcd = [](int x, int y) {
        while (y) {
          const int t = x % y;
          x = y;
          y = t;
        }
        return x;
      };
      const int ls = gcd(gcd(g.step, a), b);
      if (ls == g.step) {
        lin = Grid{g.t0, g.step, g.n + b / g.step};
      } else {
        const int rho = g.step / ls, cnt = ((g.n - 1) * g.step + b) / ls + 1;
        lin = Grid{g.t0, ls, ((cnt + rho - 1) / rho) * rho};
        L.perm = true;
      }
      in = Grid{lin.t0 - a, ls, lin.n + a / ls};
    }
    L.glin = lin;
    L.gin = in;
    g = in;
  }
  n->g_lda = g;
  n->g_feat = Grid{g.t0 - 1, 1, g.n * g.step + 2};
  TDNNF_REQUIRE(g.step == 1, "net_create: the first tdnnf layers must run at the input frame rate");
  // ---- components, in nnet3 config order
  n->c_lda = add_comp(n, "lda", lda_dim, lda_dim, 1, 0.f, 0.f, 0.f, 0.f);
  n->comps[n->c_lda].updatable = false;
  const int c_t1 = add_comp(n, "tdnn1.affine", Hd, lda_dim, 1, 1.f, c.l2_hidden, c.max_change_hidden, 0.f);
  make_tdnn(&n->tdnn1, c_t1, lda_dim, Hd, std::vector<int>{0}, n->g_lda, n->g_lda, B);
  n->num_draws = 0;
  n->draws = nullptr;
  for (int l = 0; l < c.num_layers; l++) {
    TdnnfLayer &L = n->layers[l];
    const int Kd = c.darts_num_offsets;
    const bool darts = Kd >= 2;
    const int K = darts ? Kd : 0;  // taps of a searched component
    std::vector<int> lin_off, aff_off;
    if (darts) {
      for (int i = 0; i < K; i++) {
        lin_off.push_back(-(K - 1) + i);
        aff_off.push_back(i);
      }
    } else {  // a zero offset leaves a single tap ("time-offsets=0", composite_layers.py:145-150, generate_top_list.py:109-118)
      lin_off = L.left > 0 ? std::vector<int>{-L.left, 0} : std::vector<int>{0};
      aff_off = L.right > 0 ? std::vector<int>{0, L.right} : std::vector<int>{0};
    }
    const int Kl = (int)lin_off.size(), Ka = (int)aff_off.size();
    char nm[64];
    L.c_arch = -1;
    L.arch_draw0 = 0;
    if (c.bn_num_choices > 0) {
      // X.softmax (OnehotFunctionComponent, is-updatable=true use-natural-gradient=false) or X.alpha
      // (ConstantFunctionComponent, same flags): a C-vector, no l2, no per-component max-change
      snprintf(nm, sizeof(nm), c.bn_mode == 0 ? "tdnnf%d.softmax" : "tdnnf%d.alpha", l + 2);
      L.c_arch = add_comp(n, nm, c.bn_num_choices, 1, 0, 1.f, 0.f, 0.f, 0.f);
      n->comps[L.c_arch].plain = true;
      L.arch_draw0 = n->num_draws;
      n->num_draws += c.bn_mode == 0 ? 1 : (c.bn_mode == 2 ? c.bn_num_choices : 0);
    }
    snprintf(nm, sizeof(nm), "tdnnf%d.linear", l + 2);
    // DARTS: bias forced on (scripts/generate_config.py:25-26), K logits in front of it, and the orthonormal
    // constraint is inert because ConstrainOrthonormal does not match TdnnDARTSV3Component (nnet-utils.cc:1047-1061)
    const int cl = add_comp(n, nm, L.bn, Kl * Hd, darts ? 1 : 0, 1.f, c.l2_hidden, c.max_change_hidden, darts ? 0.f : -1.0f, darts ? K : 0);
    snprintf(nm, sizeof(nm), "tdnnf%d.affine", l + 2);
    const int ca = add_comp(n, nm, Hd, Ka * L.bn, 1, 1.f, c.l2_hidden, c.max_change_hidden, 0.f, darts ? K : 0);
    make_tdnn(&L.lin, cl, Hd, L.bn, lin_off, L.gin, L.glin, B);
    make_tdnn(&L.aff, ca, L.bn, Hd, aff_off, L.glin, L.gout, B);
    if (darts) {
      L.lin.darts = L.aff.darts = true;
      L.lin.share = K - 1;  // time_offsets_[1] < 0  (nnet-tdnn-component.cc:237-240)
      L.aff.share = 0;      // time_offsets_[1] > 0  (:232-236)
      L.lin.draw0 = n->num_draws;
      L.aff.draw0 = n->num_draws + K + 1;
      n->num_draws += 2 * (K + 1);
    }
  }
  n->dropout_draw0 = n->num_draws;
  n->dropout_proportion = 0.f;
  if (c.use_dropout && !c.cv_update) n->num_draws += (c.num_layers + 1) * B * Hd;  // one B x Hd mask per GeneralDropoutComponent
  n->c_prefinal_l = add_comp(n, "prefinal-l", S, Hd, 0, 1.f, c.l2_hidden, c.max_change_hidden, -1.0f);
  const char *hn[2] = {"chain", "xent"};
  for (int h = 0; h < 2; h++) {
    char nm[64];
    snprintf(nm, sizeof(nm), "prefinal-%s.affine", hn[h]);
    n->head[h].c_affine = add_comp(n, nm, Hd, S, 1, 1.f, c.l2_hidden, c.max_change_hidden, 0.f);
    snprintf(nm, sizeof(nm), "prefinal-%s.linear", hn[h]);
    n->head[h].c_linear = add_comp(n, nm, S, Hd, 0, 1.f, c.l2_hidden, c.max_change_hidden, -1.0f);
    // output-xent: learning-rate-factor = 0.5 / xent_regularize (run_tdnn_fbk_40_iv_sp_7q.sh:151,184)
    const float lrf = h == 1 && c.xent_regularize > 0 ? 0.5f / c.xent_regularize : 1.f;
    n->head[h].c_output = add_comp(n, h == 0 ? "output.affine" : "output-xent.affine", P, S, 1, lrf, c.l2_output,
                                   c.max_change_output, 0.f);
  }
  TDNNF_REQUIRE(n->comps.size() <= 128, "net_create: too many components");
  if (c.cv_update) {
    // cross-validation architecture update (run_TDNN_DARTSV3_fbk_stride_cvupdate.sh:128-142,
    // run_TDNNf_DARTS_mod_fbk_bottleneckCBshare_cvupdate_flopsconstraint.sh:136-139): "set-learning-rate-factor 0" on
    // everything, 1e-4 on the TdnnDARTSV3Components (theta is frozen only by that factor, the logits are compensated by
    // update-alpha's x10000), the freshly added X.alpha vectors keep factor 1.
    for (auto &cd : n->comps) cd.lr_factor = cd.plain ? 1.0f : (cd.num_alpha > 0 ? 1.0e-4f : 0.0f);
  }
  n->owns_ng = share == nullptr;
  if (share) {  // another minibatch shape of the same model: same components, the primary's preconditioners
    bool same = share->comps.size() == n->comps.size() && share->num_params == n->num_params &&
                (share->cfg.use_natural_gradient != 0) == (c.use_natural_gradient != 0) && share->cfg.cv_update == c.cv_update;
    for (size_t i = 0; same && i < n->comps.size(); i++)
      same = share->comps[i].name == n->comps[i].name && share->comps[i].rows == n->comps[i].rows && share->comps[i].cols == n->comps[i].cols &&
             share->comps[i].begin == n->comps[i].begin;
    if (!same) {
      delete n;
      TDNNF_REQUIRE(false, "net_create_shared: the configuration describes another model than the primary net's");
    }
    n->ng_in = share->ng_in;
    n->ng_out = share->ng_out;
    n->oderiv_nonzero = share->oderiv_nonzero;
  } else if (c.use_natural_gradient) {
    // one input-side and one output-side preconditioner per updatable component; configuration of
    // TdnnDARTSV3Component::InitFromConfig (nnet-tdnn-component.cc:183-210), the same defaults as
    // NaturalGradientAffineComponent / LinearComponent
    n->ng_in.assign(n->comps.size(), nullptr);
    n->ng_out.assign(n->comps.size(), nullptr);
    for (size_t i = 0; i < n->comps.size(); i++) {
      const CompDesc &cd = n->comps[i];
      if (!cd.updatable || cd.plain) continue;  // fixed lda layer; vectors updated without natural gradient
      const int spliced = cd.cols + (cd.has_bias ? 1 : 0);
      const int rank_in = std::min(20, (spliced + 1) / 2), rank_out = std::min(80, (cd.rows + 1) / 2);
      if (tdnnf_ng_create(rank_in, 4, 2000.0f, 4.0f, &n->ng_in[i]) || tdnnf_ng_create(rank_out, 4, 2000.0f, 4.0f, &n->ng_out[i])) {
        tdnnf_net_destroy(n);
        return TDNNF_EINVAL;
      }
    }
  }
  // ---- gradient buckets: whole layers, cut from the top of the flat buffer downwards (= the order backward finishes them)
  {
    const long long target = 4LL << 20;  // >= 16 MB of fp32 per collective (xGMI rings want large messages), layers kept whole
    auto first_comp_of_layer = [&](int l) { return n->layers[l].c_arch >= 0 ? n->layers[l].c_arch : n->layers[l].lin.comp; };
    long long end = n->num_params;
    tdnnf_net::GradBucket b{n->comps[n->c_prefinal_l].begin, end, -2, nullptr, nullptr};
    n->buckets.push_back(b);
    end = b.begin;
    for (int l = c.num_layers - 1; l >= 0; l--) {
      const long long begin = n->comps[first_comp_of_layer(l)].begin;
      if (end - begin >= target) {
        n->buckets.push_back(tdnnf_net::GradBucket{begin, end, l, nullptr, nullptr});
        end = begin;
      }
    }
    n->buckets.push_back(tdnnf_net::GradBucket{0, end, -1, nullptr, nullptr});
    for (auto &gb : n->buckets)
      if (hipEventCreateWithFlags(&gb.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&gb.handoff, hipEventDisableTiming) != hipSuccess) {
        set_error("net_create: cannot create events");
        delete n;
        return TDNNF_EHIP;
      }
  }
  // ---- activations
  Arena sizing;
  layout_arena(n, sizing);
  n->arena_bytes = sizing.off + 1024;
  n->chain_ws = nullptr;
  n->chain_ws_bytes = 0;
  n->s2 = nullptr;
  n->wg_two = false;
  n->ev_fork = n->ev_den = n->ev_num = nullptr;
  n->s3 = nullptr;
  n->ev_s3 = nullptr;
  n->ev_fin0 = n->ev_fin = nullptr;
  n->s4 = nullptr;
  n->s5 = nullptr;
  n->ev_pg[0] = n->ev_pg[1] = n->ev_pg[2] = n->ev_pg[3] = n->ev_pg_in = nullptr;
  n->pg_count = 0;
  if (hipMalloc((void **)&n->arena, n->arena_bytes) != hipSuccess) {
    set_error("net_create: cannot allocate %zu bytes of activations", n->arena_bytes);
    delete n;
    return TDNNF_EHIP;
  }
  hipMemset(n->arena, 0, n->arena_bytes);
  Arena real;
  real.base = n->arena;
  layout_arena(n, real);
  if (share) {  // the model's BatchNorm / ReLU statistics live in the primary net
    n->t1_bn_stats = share->t1_bn_stats;
    n->t1_relu_stats = share->t1_relu_stats;
    for (size_t l = 0; l < n->layers.size(); l++) {
      n->layers[l].bn_stats = share->layers[l].bn_stats;
      n->layers[l].relu_stats = share->layers[l].relu_stats;
    }
    for (int h = 0; h < 2; h++) {
      n->head[h].bn1_stats = share->head[h].bn1_stats;
      n->head[h].bn2_stats = share->head[h].bn2_stats;
      n->head[h].relu_stats = share->head[h].relu_stats;
    }
  }
  // named activations for parity tests
  auto name = [&](const std::string &s, float *p, int rows, int cols) { n->named.push_back({s, M(p, rows, cols)}); };
  name("lda", n->lda_out, N_of(n->g_lda, B), lda_dim);
  name("tdnn1.relu", n->t1_relu, N_of(n->g_lda, B), Hd);
  name("tdnn1.batchnorm", n->t1_bn, N_of(n->g_lda, B), Hd);
  for (int l = 0; l < c.num_layers; l++) {
    TdnnfLayer &L = n->layers[l];
    const std::string p = "tdnnf" + std::to_string(l + 2);
    name(p + ".linear", L.lin_out, L.lin.rows_out, L.bn);
    name(p + ".relu", L.relu_out, L.aff.rows_out, Hd);
    name(p + ".noop", L.noop_out, L.aff.rows_out, Hd);
  }
  if (!n->layers.empty() && !n->layers[0].perm)  // what the last backward step left: d objective / d tdnnf2.linear (debugging aid)
    name("tdnnf2.linear.deriv", n->dS[0] ? n->dS[(c.num_layers - 1) & 1] : n->d_small, n->layers[0].lin.rows_out, n->layers[0].bn);
  name("prefinal-l", n->prefinal_l_out, n->Tout * B, S);
  name("prefinal-chain.relu", n->head[0].aff_relu, n->Tout * B, Hd);
  name("prefinal-xent.relu", n->head[1].aff_relu, n->Tout * B, Hd);
  name("output", n->head[0].y, n->Tout * B, P);
  name("output-xent", n->xent_logsoftmax, n->Tout * B, P);
  name("output.deriv", n->d_y, n->Tout * B, P);
  *out = n;
  return TDNNF_OK;
}

void tdnnf_net_destroy(tdnnf_net *n) {
  if (!n) return;
  if (n->s3) hipStreamSynchronize(n->s3);  // its kernels use the preconditioners' buffers
  if (n->s4) hipStreamSynchronize(n->s4);
  if (n->s5) hipStreamSynchronize(n->s5);
  if (n->s2) hipStreamSynchronize(n->s2);
  for (auto &nb : n->ng_buckets) ng_group_destroy(nb.group);
  ng_fin_destroy(n->ngfin);
  if (n->ev_ngc) hipEventDestroy(n->ev_ngc);
  if (n->owns_ng) {
    for (auto *g : n->ng_in) tdnnf_ng_destroy(g);
    for (auto *g : n->ng_out) tdnnf_ng_destroy(g);
  }
  hipFree(n->arena);
  hipFree(n->chain_ws);
  for (float *p : n->captured) hipFree(p);
  for (auto &gb : n->buckets) {
    if (gb.ready) hipEventDestroy(gb.ready);
    if (gb.handoff) hipEventDestroy(gb.handoff);
  }
  if (n->s2) hipStreamDestroy(n->s2);
  if (n->ev_fork) hipEventDestroy(n->ev_fork);
  if (n->ev_den) hipEventDestroy(n->ev_den);
  if (n->ev_num) hipEventDestroy(n->ev_num);
  if (n->ev_s3) hipEventDestroy(n->ev_s3);
  if (n->ev_fin0) hipEventDestroy(n->ev_fin0);
  if (n->ev_fin) hipEventDestroy(n->ev_fin);
  if (n->s3) hipStreamDestroy(n->s3);
  for (hipEvent_t e : {n->ev_pg[0], n->ev_pg[1], n->ev_pg[2], n->ev_pg[3], n->ev_pg_in, n->ev_early_in, n->ev_early, n->ev_comm, n->ev_den_rec})
    if (e) hipEventDestroy(e);
  if (n->s4) hipStreamDestroy(n->s4);
  if (n->s5) hipStreamDestroy(n->s5);
  rows_gemm_group_destroy(n->early_launch);
  planes_split_group_destroy(n->wsplit_group);
  for (hipEvent_t e : n->ev_phase)
    if (e) hipEventDestroy(e);
  upd_group_destroy(n->upd);
  delete n;
}

long long tdnnf_net_num_params(const tdnnf_net *n) { return n ? n->num_params : 0; }
int tdnnf_net_num_components(const tdnnf_net *n) { return n ? (int)n->comps.size() : 0; }

int tdnnf_net_component_info(const tdnnf_net *n, int i, char *name_out, long long *begin, int *rows, int *cols, int *has_bias,
                             float *lr_factor, float *l2, float *max_change, float *orthonormal) {
  TDNNF_REQUIRE(n && i >= 0 && i < (int)n->comps.size(), "net_component_info: bad index");
  const CompDesc &c = n->comps[i];
  if (name_out) snprintf(name_out, 64, "%s", c.name.c_str());
  if (begin) *begin = c.begin;
  if (rows) *rows = c.rows;
  if (cols) *cols = c.cols;
  if (has_bias) *has_bias = c.has_bias;
  if (lr_factor) *lr_factor = c.lr_factor;
  if (l2) *l2 = c.l2;
  if (max_change) *max_change = c.max_change;
  if (orthonormal) *orthonormal = c.orthonormal;
  return TDNNF_OK;
}

// model statistics outside the parameter vector: [count, sum[D], sumsq[D]] of every BatchNorm and
// [count, value_sum[D], deriv_sum[D]] of every ReLU, in network order
static void stat_blocks(const tdnnf_net *n, std::vector<std::pair<double *, int>> &out) {  // (pointer, number of doubles)
  const int Hd = n->cfg.hidden_dim, S = n->cfg.prefinal_small_dim;
  auto bn = [&](double *p, int D) { out.push_back({p, 1 + 2 * D}); };
  auto relu = [&](double *p, int D) { out.push_back({p, 2 + 3 * D}); };
  bn(n->t1_bn_stats, Hd);
  relu(n->t1_relu_stats, Hd);
  for (auto &L : n->layers) {
    bn(L.bn_stats, Hd);
    relu(L.relu_stats, Hd);
  }
  for (int h = 0; h < 2; h++) {
    bn(n->head[h].bn1_stats, Hd);
    relu(n->head[h].relu_stats, Hd);
    bn(n->head[h].bn2_stats, S);
  }
}
long long tdnnf_net_stats_size(const tdnnf_net *n) {
  if (!n) return 0;
  std::vector<std::pair<double *, int>> b;
  stat_blocks(n, b);
  long long t = 0;
  for (auto &x : b) t += x.second;
  return t;
}
int tdnnf_net_get_stats(const tdnnf_net *n, double *host_out, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && host_out, "net_get_stats: null argument");
  std::vector<std::pair<double *, int>> b;
  stat_blocks(n, b);
  TDNNF_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (auto &x : b) {
    TDNNF_HIP(hipMemcpy(host_out, x.first, sizeof(double) * x.second, hipMemcpyDeviceToHost));
    host_out += x.second;
  }
  return TDNNF_OK;
}
int tdnnf_net_set_stats(tdnnf_net *n, const double *host_in, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && host_in, "net_set_stats: null argument");
  std::vector<std::pair<double *, int>> b;
  stat_blocks(n, b);
  {  // oderiv_count of every ReLU block ([count, value_sum[D], deriv_sum[D], oderiv_count, oderiv_sumsq[D]]); stat_blocks() order:
     // tdnn1 (bn, relu), every tdnnf layer (bn, relu), both heads (bn1, relu, bn2)
    std::vector<char> &nz = *n->oderiv_nonzero;
    nz.assign(n->cfg.num_layers + 3, 0);
    std::vector<int> is_relu = {0, 1};
    for (int l = 0; l < n->cfg.num_layers; l++) is_relu.insert(is_relu.end(), {0, 1});
    for (int h = 0; h < 2; h++) is_relu.insert(is_relu.end(), {0, 1, 0});
    const double *p = host_in;
    int k = 0;
    for (size_t i = 0; i < b.size(); i++) {
      if (i < is_relu.size() && is_relu[i]) {
        const int D = (b[i].second - 2) / 3;
        nz[k++] = p[1 + 2 * D] != 0.0;
      }
      p += b[i].second;
    }
  }
  TDNNF_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (auto &x : b) {
    TDNNF_HIP(hipMemcpy(x.first, host_in, sizeof(double) * x.second, hipMemcpyHostToDevice));
    host_in += x.second;
  }
  return TDNNF_OK;
}

int tdnnf_net_set_dropout_proportion(tdnnf_net *n, float proportion) {
  TDNNF_REQUIRE(n && proportion >= 0.f && proportion <= 0.5f, "net_set_dropout_proportion: proportion must be in [0, 0.5] (continuous masks: scale in [1 - 2p, 1 + 2p])");
  TDNNF_REQUIRE(proportion == 0.f || n->dropout_masks, "net_set_dropout_proportion: the net was created without use_dropout (or in cv-update mode)");
  n->dropout_proportion = proportion;
  return TDNNF_OK;
}

int tdnnf_net_set_temperature_proportion(tdnnf_net *n, float proportion) {
  TDNNF_REQUIRE(n && proportion > 0.f, "net_set_temperature_proportion: proportion must be > 0");
  n->cfg.darts_temp_proportion = proportion;
  n->cfg.bn_temp_proportion = proportion;
  return TDNNF_OK;
}

// NameMatchesPattern (UPSTREAM, used by every edit directive of nnet-utils.cc:1166-1415): '*' matches any run of characters
static bool name_matches(const char *name, const char *pat) {
  if (*pat == 0) return *name == 0;
  if (*pat == '*') {
    for (const char *p = name;; p++) {
      if (name_matches(p, pat + 1)) return true;
      if (*p == 0) return false;
    }
  }
  return *name == *pat && name_matches(name + 1, pat + 1);
}

int tdnnf_net_set_batchnorm_sync(tdnnf_net *n, TDNNF_ALLREDUCE_FN *allreduce, void *ctx, int world_size) {
  TDNNF_REQUIRE(n && world_size >= 1, "net_set_batchnorm_sync: bad arguments");
  n->bn_sync.fn = allreduce;
  n->bn_sync.ctx = ctx;
  n->bn_sync.world = world_size;
  return TDNNF_OK;
}

int tdnnf_net_set_learning_rate_factor(tdnnf_net *n, const char *name_pattern, float factor, int *num_set) {
  TDNNF_REQUIRE(n && name_pattern && factor >= 0.f, "net_set_learning_rate_factor: bad arguments (the factor must be >= 0)");
  int cnt = 0;
  for (auto &cd : n->comps)
    if (cd.updatable && name_matches(cd.name.c_str(), name_pattern)) {
      cd.lr_factor = factor;
      cnt++;
    }
  if (num_set) *num_set = cnt;
  return TDNNF_OK;
}

int tdnnf_net_component_num_alpha(const tdnnf_net *n, int i) {
  return n && i >= 0 && i < (int)n->comps.size() ? n->comps[i].num_alpha : 0;
}
int tdnnf_net_num_random_draws(const tdnnf_net *n) { return n ? n->num_draws : 0; }
int tdnnf_net_set_random_draws(tdnnf_net *n, const float *draws) {
  TDNNF_REQUIRE(n && (draws || n->num_draws == 0), "net_set_random_draws: null argument");
  n->draws = draws;
  return TDNNF_OK;
}

int tdnnf_net_input_frames(const tdnnf_net *n, int *num_t_in, int *first_t) {
  TDNNF_REQUIRE(n, "net_input_frames: null net");
  if (num_t_in) *num_t_in = n->g_feat.n;
  if (first_t) *first_t = n->g_feat.t0;
  return TDNNF_OK;
}

int tdnnf_net_set_buffers(tdnnf_net *n, float *params, float *grads) {
  TDNNF_REQUIRE(n && params && grads && ((uintptr_t)params & 15) == 0 && ((uintptr_t)grads & 15) == 0,
                "net_set_buffers: buffers must be non-null and 16-byte aligned");
  n->params = params;
  n->grads = grads;
  return TDNNF_OK;
}

int tdnnf_net_num_grad_buckets(const tdnnf_net *n) { return n ? (int)n->buckets.size() : 0; }
int tdnnf_net_grad_bucket(const tdnnf_net *n, int i, long long *begin, long long *end) {
  TDNNF_REQUIRE(n && i >= 0 && i < (int)n->buckets.size(), "net_grad_bucket: bad index");
  if (begin) *begin = n->buckets[i].begin;
  if (end) *end = n->buckets[i].end;
  return TDNNF_OK;
}
int tdnnf_net_wait_grad_bucket(const tdnnf_net *n, int i, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && i >= 0 && i < (int)n->buckets.size(), "net_wait_grad_bucket: bad index");
  TDNNF_HIP(hipStreamWaitEvent((hipStream_t)stream, n->buckets[i].ready, 0));
  return TDNNF_OK;
}

int tdnnf_net_set_capture(tdnnf_net *n, int on) {
  TDNNF_REQUIRE(n, "net_set_capture: null net");
  n->capture_on = on != 0;
  return TDNNF_OK;
}

int tdnnf_net_activation_dims(const tdnnf_net *n, const char *name, int *rows, int *cols) {
  TDNNF_REQUIRE(n && name, "net_activation_dims: null argument");
  for (auto &kv : n->named)
    if (kv.first == name) {
      if (rows) *rows = kv.second.rows;
      if (cols) *cols = kv.second.cols;
      return TDNNF_OK;
    }
  set_error("net_activation_dims: unknown activation '%s'", name);
  return TDNNF_EINVAL;
}

int tdnnf_net_get_activation(const tdnnf_net *n, const char *name, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && name && mat_ok(out), "net_get_activation: bad argument");
  for (auto &kv : n->named)
    if (kv.first == name) {
      TDNNF_REQUIRE(out->rows == kv.second.rows && out->cols == kv.second.cols, "net_get_activation: %s is %d x %d", name,
                    kv.second.rows, kv.second.cols);
      return tdnnf_sum_scaled(&kv.second, 1.0f, nullptr, 0.f, out, stream);
    }
  set_error("net_get_activation: unknown activation '%s'", name);
  return TDNNF_EINVAL;
}

static BnChoice bn_choice(const tdnnf_net_config &c) {
  BnChoice bc;
  memset(&bc, 0, sizeof(bc));
  bc.C = c.bn_num_choices;
  bc.mode = c.bn_mode;
  int run = 0;
  for (int k = 0; k < 8; k++) {
    if (k < bc.C) run += c.bn_choice_dims[k];
    bc.cum[k] = run;
  }
  bc.flops_scale = c.bn_flops_scale;
  bc.temp = c.bn_mode == 2 ? c.bn_temp_proportion : 1.0f;
  return bc;
}

namespace {
// Few sequences leave most CUs idle while one workgroup per sequence walks the frames: there the backward recursion of the denominator runs beside
// the forward one and the occupancies of all frames at once (the split form).  Measured (ms per step, one-kernel backward -> split): 1500 x 16
// 37.3 -> 28.5, x 32 53.1 -> 45.3, x 64 80.9 -> 75.2, x 128 127.6 -> 128.2; 150 x 64 14.2 -> 13.8.  The second recursion runs on the
// weight-gradient stream (or the natural-gradient side stream), idle until the backward pass (on a stream of its own -- a fifth in flight -- the
// step at 150 x 64 took 20.0 ms: they then share hardware queues).
// [r4] with the pre-split plane GEMMs (planes = true) the xent head no longer covers a 25 ms denominator at 128 sequences: side by side there too
// (same box, ms per step: f16x3 103.8 / 105.2 -> 102.1 / 102.4; exact f32 124.4 / 124.6 -> 125.4 / 125.0, so f32 keeps the one-kernel backward pass)
// [r4, later] with the recursions at 8.8 us per frame (chain.hip, FAST kernels) side by side wins for exact f32 at 128 sequences as well: 122.3 / 122.6 ms
// against 122.7 / 122.9 on one box, and the GEMM launches beside it are stretched less (event-timed 128 x 128 class 0.642 of the peak against 0.617)
bool den_uses_split() { return options().den_split >= 0 ? options().den_split != 0 : true; }
}  // namespace

int tdnnf_net_forward_backward(tdnnf_net *n, const tdnnf_mat *feats, const tdnnf_mat *ivectors, const tdnnf_den_graph *den,
                               const tdnnf_supervision *sup, double *results, long long step, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && n->params && n->grads, "net_forward_backward: call net_set_buffers first");
  TDNNF_REQUIRE(mat_ok(feats) && mat_ok(ivectors) && den && sup && results, "net_forward_backward: bad arguments");
  const tdnnf_net_config &c = n->cfg;
  const int B = n->B, Hd = c.hidden_dim, S = c.prefinal_small_dim, P = c.num_pdfs, lda_dim = 3 * c.feat_dim + c.ivector_dim;
  TDNNF_REQUIRE(feats->rows == n->g_feat.n * B && feats->cols == c.feat_dim, "net_forward_backward: feats must be %d x %d (t-major)",
                n->g_feat.n * B, c.feat_dim);
  TDNNF_REQUIRE(ivectors->rows == B && ivectors->cols == c.ivector_dim, "net_forward_backward: ivectors must be %d x %d", B, c.ivector_dim);
  hipStream_t s = (hipStream_t)stream;
  const bool cv = c.cv_update != 0;  // BatchNorm components are BatchNormTestComponents
  if (!n->chain_ws) {
    n->den_split = den_uses_split();  // latched: option den_split read once per net (the workspace is sized for it)
    n->chain_ws_bytes = tdnnf_chain_workspace_bytes(den, B, n->Tout) - (n->den_split ? 0 : chain_split_region_bytes(den, B, n->Tout));
    TDNNF_HIP(hipMalloc(&n->chain_ws, n->chain_ws_bytes));
    TDNNF_HIP(hipStreamCreateWithFlags(&n->s2, hipStreamNonBlocking));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_den, hipEventDisableTiming));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_num, hipEventDisableTiming));
    // The side stream (natural-gradient statistics, the denominator's second recursion) at the default priority, as every stream of the
    // library: rounds 2-3 gave it the lowest priority above 32 sequences, which is worth nothing measurable in the step any more
    // (124.0 / 124.0 ms at 128 sequences, 22.9 / 23.0 at 16) and made a job's step time depend on what had run before it in the
    // process -- once a stream of another priority has existed, HIP's hardware-queue pool maps the next net's streams differently (a
    // 16-sequence step took 35 ms instead of 23 after a 128-sequence job, a 128-sequence step 142 ms after a 16-sequence one;
    // docs/experiments.md r4-d).
    TDNNF_HIP(hipStreamCreateWithFlags(&n->s3, hipStreamNonBlocking));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_s3, hipEventDisableTiming));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_fin0, hipEventDisableTiming));
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_fin, hipEventDisableTiming));
    if (n->early_on) {
      TDNNF_HIP(hipEventCreateWithFlags(&n->ev_early_in, hipEventDisableTiming));
      TDNNF_HIP(hipEventCreateWithFlags(&n->ev_early, hipEventDisableTiming));
      n->early.assign(n->comps.size(), tdnnf_net::EarlyIn());
    }
    if (n->early_on && !n->wg_on) TDNNF_HIP(hipStreamCreateWithFlags(&n->s4, hipStreamNonBlocking));
    if (n->wg_on) {
      TDNNF_HIP(hipStreamCreateWithFlags(&n->s4, hipStreamNonBlocking));
      if (n->ws5) TDNNF_HIP(hipStreamCreateWithFlags(&n->s5, hipStreamNonBlocking));
      for (int i = 0; i < 4; i++) TDNNF_HIP(hipEventCreateWithFlags(&n->ev_pg[i], hipEventDisableTiming));
      TDNNF_HIP(hipEventCreateWithFlags(&n->ev_pg_in, hipEventDisableTiming));
    }
    TDNNF_HIP(hipEventCreateWithFlags(&n->ev_ngc, hipEventDisableTiming));
  }
  n->fb_count++;
  CK(phase_mark(n, 0, s));
  n->wg_two = false;  // (a step that failed half-way may have left it set)
  TDNNF_HIP(hipMemsetAsync(n->gtmp, 0, sizeof(float) * (size_t)n->num_params, s));
  if (n->ng_bsum_all && n->ng_bsum_floats) TDNNF_HIP(hipMemsetAsync(n->ng_bsum_all, 0, sizeof(float) * n->ng_bsum_floats, s));  // every component's raw bias gradient
  n->pg_count = 0;
  // Refreshes whose host part has finished: upload W_{t+1} now, on the side stream, which is idle during the forward pass --
  // otherwise the ~9 small launches of each (18 refreshes per step) sit in front of the component's statistics passes in the
  // backward pass.  Whatever is not ready yet stays with its next use.
  bool early_refresh = false;
  auto all_ng = [&]() {
    std::vector<tdnnf_ng *> v;
    for (auto *list : {&n->ng_in, &n->ng_out})
      for (tdnnf_ng *g : *list)
        if (g) v.push_back(g);
    return v;
  };
  if (c.use_natural_gradient && n->s3) {
    {
      TDNNF_HIP(hipEventRecord(n->ev_fin0, s));
      TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_fin0, 0));
      SplitKScratchOverride side_scratch(n->s3_scratch, n->s3_scratch_bytes);
      if (n->ngfin) {  // all of them in five grouped launches, if every host part has finished
        int did = 0;
        CK(ng_fin_run(n->ngfin, n->s3, false, &did));
        early_refresh = did != 0;
      } else {
        for (tdnnf_ng *g : all_ng()) {
          int did = 0;
          CK(ng_finalize_if_ready(g, n->s3, &did));
          early_refresh = early_refresh || did;
        }
      }
      if (early_refresh) TDNNF_HIP(hipEventRecord(n->ev_fin, n->s3));
    }
  }
  BnSyncScope bn_sync_scope(n->bn_sync.fn && !c.cv_update ? &n->bn_sync : nullptr);  // synchronised BatchNorm (data-parallel callers)
  // scope values: 1 two bf16 planes split in the kernel, 3 three; 4 pre-split f16 pairs (exact f32 where no planes are hinted)
  GemmPrecisionScope gemm_arith(c.gemm_precision == 2 ? 3 : c.gemm_precision == 3 ? 4 : c.gemm_precision);
  const int np = n->planes_np;
  if (np && n->wg_on) {
    // (the weight-gradient stream reads operands while the caller's stream moves on: plane slots are reused per layer -- small minibatches keep the f32 kernels)
  }
  const bool pl_on = np != 0 && !n->wg_on;
  // plane operands: split a matrix into its slot and describe it
  enum { kP = 1, kT = 2 };
  // (optional) norm bound a BatchNorm finalize launch left in n->fro_buf: blocks > 0 -> the split takes its scale from it (planes_gemm.h)
  struct FroBound {
    int blocks = 0;
    float mul = 1.0f, add_coef = 0.0f;
    const float *add_rec = nullptr;
  };
  auto split = [&](const tdnnf_mat &m, int lead, int want, PlanesOperand *o, hipStream_t st, const FroBound &fb = FroBound()) -> int {
    *o = PlanesOperand();
    if (!pl_on) return TDNNF_OK;
    auto it = n->plane_slots.find(m.data);
    if (it == n->plane_slots.end() || m.rows <= 0) return TDNNF_OK;  // (no slot: the GEMM runs its own kernels)
    tdnnf_net::PlaneSlot &ps = it->second;
    // a wide matrix (the 1536- / 6034-column activations and derivatives) is the tile-row operand of its weight gradient, which reads
    // the ROW-MAJOR planes through transposing LDS loads: no planes of the transpose for those
    if (m.cols >= 1024) want = kP;
    PlanesSplitArgs a;
    a.np = np; a.x = view(&m); a.lead = lead; a.scale = ps.scale; a.sumsq_ws = n->planes_ws;
    if (np == 2 && fb.blocks > 0 && (fb.add_coef == 0.f || fb.add_rec)) {
      a.fro2_bound = n->fro_buf; a.fro2_blocks = fb.blocks; a.fro_mul = fb.mul; a.add_coef = fb.add_coef; a.add_rec = fb.add_rec;
    }
    lead = (lead + 15) & ~15;  // (a weight gradient reads the row-major planes in K steps of 16 rows: the matrix starts on one)
    a.lead = lead;
    a.R = planes_rows_padded((long long)2 * lead + m.rows + 256);
    a.Rt = planes_rows_padded(((m.cols + 255) / 256) * 256LL);
    a.P = (want & kP) ? ps.P : nullptr;
    a.PT = (want & kT) ? ps.PT : nullptr;
    const long long kb_alloc = (planes_kblocks(m.cols) + 15) / 16 * 16;
    TDNNF_REQUIRE(planes_bytes(np, a.R, kb_alloc) <= ps.bytesP && planes_bytes(np, a.Rt, planes_t_kblocks(m.rows)) <= ps.bytesPT,
                  "net_forward_backward: plane slot too small for a %d x %d matrix", m.rows, m.cols);
    const long long cfg5[5] = {m.rows, m.cols, lead, a.R, want};
    a.pads_done = memcmp(cfg5, ps.last, sizeof(cfg5)) == 0;
    memcpy(ps.last, cfg5, sizeof(cfg5));
    TDNNF_HIP(planes_split(a, st));
    o->base = m.data; o->rows = m.rows; o->cols = m.cols; o->ld = m.stride; o->np = np;
    o->P = a.P; o->R = a.R; o->lead = lead; o->kb_alloc = kb_alloc; o->PT = a.PT; o->Rt = a.Rt; o->scale = np == 2 ? ps.scale : nullptr;
    return TDNNF_OK;
  };
  auto hint_of = [&](const PlanesOperand &o) -> const PlanesOperand * { return o.base ? &o : nullptr; };
  // a component's weight matrix as planes (row-major: forward; transposed: backward-data); coef: a TdnnDARTSV3Component's effective tap
  // coefficients (device, one per `period` columns), folded into the planes so that its GEMMs need none
  // (`group`: collect the split instead of launching it -- the plain components' weights of a step go as ONE grouped pair of launches)
  std::vector<PlanesSplitArgs> wsplits;
  bool darts_coef_done = false;  // (pl_on: the DARTS components' coefficients and planes were formed at the start of the step)
  auto split_weights = [&](int comp, const float *coef, int period, bool group = false) -> int {
    PlanesOperand &o = n->pw[comp];
    if (!o.P) return TDNNF_OK;
    o.base = Wp(n, comp);
    o.coef = coef;
    o.coef_period = period;
    PlanesSplitArgs a;
    a.np = np; a.x = MatView{Wp(n, comp), o.rows, o.cols, o.cols}; a.lead = 0; a.R = o.R; a.P = const_cast<void *>(o.P); a.Rt = o.Rt;
    a.PT = const_cast<void *>(o.PT); a.scale = n->pw_scale[comp]; a.sumsq_ws = n->planes_ws;
    a.col_coef = coef; a.col_coef_period = period;
    o.scale = np == 2 ? n->pw_scale[comp] : nullptr;
    a.pads_done = n->fb_count > 1;  // (fixed shapes: the zero rows written by the first step stay)
    if (group && options().planes_group && planes_split_group_ok(a)) wsplits.push_back(a);
    else TDNNF_HIP(planes_split(a, s));
    return TDNNF_OK;
  };
  if (pl_on) {  // this step's weights as planes (the DARTS components' again in the layer loop, once their coefficients are formed)
    // 36 components x (norm pass + split) were 72 launches of 6 - 12 us, serial on the caller's stream with nothing else in flight: two launches now
    for (size_t i = 0; i < n->comps.size(); i++)
      if (n->comps[i].num_alpha == 0 || n->comps[i].plain) CK(split_weights((int)i, nullptr, 0, true));
    // the TdnnDARTSV3Components' too: their tap coefficients depend on the architecture logits and this step's draws only, so they are formed
    // here, ahead of the layer loop, and the coefficient-folded weight planes join the same two launches
    if (options().planes_group && n->draws) {
      for (auto &L : n->layers) {
        if (!L.lin.darts) continue;
        for (Tdnn *td : {&L.lin, &L.aff}) {
          const float *u = n->draws + td->draw0;
          CK(tdnnf_tdnn_darts_coef(Ap(n, td->comp), td->K, c.darts_flags, c.darts_temp_proportion, u, u + td->K, td->share, td->memo,
                                   td->memo + TDNNF_MAX_OFFSETS, s));
          hipLaunchKernelGGL(active_taps_kernel, dim3(1), dim3(64), 0, s, td->memo + TDNNF_MAX_OFFSETS, td->K, td->active);
        }
        CK(split_weights(L.lin.comp, L.lin.memo + TDNNF_MAX_OFFSETS, c.hidden_dim, true));
        CK(split_weights(L.aff.comp, L.aff.memo + TDNNF_MAX_OFFSETS, L.bn, true));
      }
      darts_coef_done = true;
    }
    TDNNF_HIP(planes_split_group(wsplits, &n->wsplit_group, s));
  }
  auto wplanes = [&](int comp) -> const PlanesOperand * { return pl_on && n->pw[comp].P ? &n->pw[comp] : nullptr; };
  // where the fused BatchNorm / ReLU backward sweep may write the f16 planes of the derivative matrix `d` it produces (f16x3, 1536-wide
  // matrices with a slot): fills *bp for bn_relu_bwd and *po for the GEMMs that read `d` next; bp->P == null: not fused, split afterwards
  auto bwd_planes = [&](const tdnnf_mat &d, int lead, BwdPlanes *bp, PlanesOperand *po) -> int {
    *bp = BwdPlanes{nullptr, 0, 0, nullptr};
    *po = PlanesOperand();
    if (!(pl_on && np == 2 && d.cols % 16 == 0 && d.cols >= 1024)) return TDNNF_OK;
    auto it = n->plane_slots.find(d.data);
    if (it == n->plane_slots.end()) return TDNNF_OK;
    tdnnf_net::PlaneSlot &ps = it->second;
    lead = (lead + 15) & ~15;
    const long long R = planes_rows_padded((long long)2 * lead + d.rows + 256), kb_alloc = (planes_kblocks(d.cols) + 15) / 16 * 16;
    TDNNF_REQUIRE(planes_bytes(np, R, kb_alloc) <= ps.bytesP, "net_forward_backward: plane slot too small for a %d x %d matrix", d.rows, d.cols);
    const long long cfg5[5] = {d.rows, d.cols, lead, R, kP};
    if (memcmp(cfg5, ps.last, sizeof(cfg5)) != 0) TDNNF_HIP(planes_pad(np, ps.P, planes_kblocks(d.cols), R, lead, d.rows, s));
    memcpy(ps.last, cfg5, sizeof(cfg5));
    *bp = BwdPlanes{ps.P, R, lead, ps.scale};
    po->base = d.data; po->rows = d.rows; po->cols = d.cols; po->ld = d.stride; po->np = np;
    po->P = ps.P; po->R = R; po->lead = lead; po->kb_alloc = kb_alloc; po->scale = ps.scale;
    return TDNNF_OK;
  };
  // bn_apply_bypass that ALSO writes its output as f16 planes when the BatchNorm finalize left a norm bound (f16x3, plain views): the
  // scale record first (from the bound), then one pass writes the f32 matrix and its row-major planes -- the GEMM that reads `out`
  // next needs no split pass.  *po describes the planes (empty: not fused, the consumer splits).
  auto bn_apply_planes = [&](const tdnnf_mat &x, const float *memo, const MatView &byp, float bypass, const tdnnf_mat &out, const float *mask, const FroBound &fb,
                             PlanesOperand *po) -> int {
    *po = PlanesOperand();
    auto it = (pl_on && np == 2 && fb.blocks > 0 && (fb.add_coef == 0.f || fb.add_rec) && out.cols == Hd && out.stride == ldpad(Hd)) ? n->plane_slots.find(out.data)
                                                                                                                                         : n->plane_slots.end();
    if (it == n->plane_slots.end()) {
      TDNNF_HIP(bn_apply_bypass(view(&x), memo, Hd, ldpad(Hd), byp, bypass, view(&out), s, mask, B));
      return TDNNF_OK;
    }
    tdnnf_net::PlaneSlot &ps = it->second;
    const long long R = planes_rows_padded((long long)out.rows + 256), kb_alloc = (planes_kblocks(out.cols) + 15) / 16 * 16;
    TDNNF_REQUIRE(planes_bytes(np, R, kb_alloc) <= ps.bytesP, "net_forward_backward: plane slot too small for a %d x %d matrix", out.rows, out.cols);
    TDNNF_HIP(planes_scale_bound(n->fro_buf, fb.blocks, (double)out.rows * out.cols, fb.mul, fb.add_coef, fb.add_rec, ps.scale, s));
    const long long cfg5[5] = {out.rows, out.cols, 0, R, kP};
    if (memcmp(cfg5, ps.last, sizeof(cfg5)) != 0 && ps.last[0] >= 0) {  // (the slot last held another shape: zero rows behind the matrix again)
      TDNNF_HIP(planes_pad(np, ps.P, planes_kblocks(out.cols), R, 0, out.rows, s));
    }
    memcpy(ps.last, cfg5, sizeof(cfg5));
    const PlanesSink sink{ps.P, R, ps.scale};
    TDNNF_HIP(bn_apply_bypass(view(&x), memo, Hd, ldpad(Hd), byp, bypass, view(&out), s, mask, B, &sink));
    if (options().planes_check_bound) TDNNF_HIP(planes_check_bound(view(&out), ps.scale, n->planes_ws, s));
    po->base = out.data; po->rows = out.rows; po->cols = out.cols; po->ld = out.stride; po->np = np;
    po->P = ps.P; po->R = R; po->lead = 0; po->kb_alloc = kb_alloc; po->scale = ps.scale;
    return TDNNF_OK;
  };
  TransposedWeightsScope gemm_wt(n->params, n->paramsT, n->paramsT ? n->num_params : 0);
  if (n->paramsT) {  // split-bf16 backward-data GEMMs read W^T (k-contiguous B operand)
    TransTable tb;
    memset(&tb, 0, sizeof(tb));
    const int nc = (int)n->comps.size();
    for (int i = 0; i < nc; i++) {
      tb.begin[i] = n->comps[i].begin;
      tb.rows[i] = n->comps[i].rows;
      tb.cols[i] = n->comps[i].cols;
    }
    hipLaunchKernelGGL(transpose_weights_kernel, dim3(256, nc), dim3(256), 0, s, n->params, n->paramsT, tb);
  }
  TDNNF_REQUIRE(n->chain_ws_bytes >= tdnnf_chain_workspace_bytes(den, B, n->Tout) - (n->den_split ? 0 : chain_split_region_bytes(den, B, n->Tout)),
                "net_forward_backward: denominator graph changed size");
  // the reference's RandInt()/RandUniform() coin flips, made reproducible: k-th decision of this minibatch
  unsigned long long coin_k = 0;
  auto coin = [&]() { return (int)(::tdnnf::tdnnf_decision((unsigned long long)step, 2 * coin_k++) & 1); };

  // ================================================================= forward
  TraceRange trace_step("tdnnf_net_forward_backward");
  const int N0 = N_of(n->g_lda, B);
  tdnnf_mat lda_in = M(n->lda_in, N0, lda_dim), lda_out = M(n->lda_out, N0, lda_dim);
  CK(tdnnf_splice_input(feats, ivectors, B, 3, &lda_in, s));
  CK(tdnnf_affine_propagate(&lda_in, Wp(n, n->c_lda), lda_dim, Bp(n, n->c_lda), lda_dim, &lda_out, s));
  tdnnf_tdnn_indexes ix1;
  memset(&ix1, 0, sizeof(ix1));
  ix1.row_stride = 1;
  ix1.num_offsets = 1;
  tdnnf_mat t1r = M(n->t1_relu, N0, Hd), t1b = M(n->t1_bn, N0, Hd);
  const MatView none{nullptr, 0, 0, 0};
  // dropout masks of this minibatch (mask m: tdnn1 = 0, tdnnf layer l = l + 1); null = identity
  const bool drop = n->dropout_masks && n->dropout_proportion > 0.f;
  if (drop) {
    TDNNF_REQUIRE(n->draws, "net_forward_backward: dropout needs net_set_random_draws before every step");
    const long long nm = (long long)(c.num_layers + 1) * B * Hd;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(nm, 256)), dim3(256), 0, s, n->draws + n->dropout_draw0, n->dropout_proportion, nm, n->dropout_masks);
  }
  auto mask_of = [&](int m) -> const float * { return drop ? n->dropout_masks + (size_t)m * B * Hd : nullptr; };
  // tdnn1: affine (+bias, ReLU in the GEMM epilogue) -> BatchNorm
  // plane operands of this step, by role (empty = the GEMM runs its own kernels)
  std::vector<PlanesOperand> po_in(n->layers.size()), po_lin(n->layers.size());
  PlanesOperand po_lda, po_top, po_pl, po_b1[2], po_b2[2];
  CK(split(lda_out, 0, kP | kT, &po_lda, s));
  // the norm bound of the matrix the next bn_apply_bypass writes (= the next layer's input): from the BatchNorm finalize inside
  // affine_relu_bn_stats; carried to the split at the top of the next layer
  FroBound fb_next;
  const float mask_max = drop ? 1.0f + 2.0f * n->dropout_proportion : 1.0f;
  {
    PlanesHintScope ph(hint_of(po_lda), wplanes(n->tdnn1.comp));
    FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_next.blocks);
    CK(affine_relu_bn_stats(n, &ix1, &lda_out, Wp(n, n->tdnn1.comp), lda_dim, Hd, lda_dim, Bp(n, n->tdnn1.comp), nullptr, &t1r, n->t1_bn_memo, n->t1_bn_stats, s));
  }
  fb_next.mul = mask_max;
  PlanesOperand po_next;  // planes of the matrix just written by bn_apply_planes (= the next layer's input), when it wrote them itself
  CK(bn_apply_planes(t1r, n->t1_bn_memo, none, 0.f, t1b, mask_of(0), fb_next, &po_next));
  float *prev = n->t1_bn;
  int layer_no = 0;
  for (auto &L : n->layers) {
    layer_no++;
    TraceRange trace_layer(("forward tdnnf" + std::to_string(layer_no + 1)).c_str());
    tdnnf_mat in = M(prev, N_of(L.gin, B), Hd);
    tdnnf_mat lin = M(L.lin_out, L.lin.rows_out, L.bn);
    const float *lin_eff = nullptr, *aff_eff = nullptr;
    if (L.lin.darts) {  // TdnnDARTSV3Component::Propagate :250-289 for both components of the layer
      TDNNF_REQUIRE(n->draws, "net_forward_backward: a DARTS net needs net_set_random_draws before every step");
      if (!darts_coef_done)
        for (Tdnn *td : {&L.lin, &L.aff}) {
          const float *u = n->draws + td->draw0;
          CK(tdnnf_tdnn_darts_coef(Ap(n, td->comp), td->K, c.darts_flags, c.darts_temp_proportion, u, u + td->K, td->share, td->memo,
                                   td->memo + TDNNF_MAX_OFFSETS, s));
          hipLaunchKernelGGL(active_taps_kernel, dim3(1), dim3(64), 0, s, td->memo + TDNNF_MAX_OFFSETS, td->K, td->active);
        }
      lin_eff = L.lin.memo + TDNNF_MAX_OFFSETS;
      aff_eff = L.aff.memo + TDNNF_MAX_OFFSETS;
      if (pl_on && !darts_coef_done) {  // the effective coefficients folded into this step's weight planes
        CK(split_weights(L.lin.comp, lin_eff, Hd));
        CK(split_weights(L.aff.comp, aff_eff, L.bn));
      }
    }
    // uniform-sample mode runs at most two taps of K (share + sampled): tell the FLOP accounting of the profiler
    ProfFlopsScale taps_active(L.lin.darts && (c.darts_flags & TDNNF_DARTS_UNIFORM_SAMPLE) && L.lin.K > 2 ? 2.0 / L.lin.K : 1.0);
    // (DARTS .linear: bias present but offsets[1] < 0 -> out is zeroed and the bias never added, :237-240)
    const bool use_pl = pl_on;
    if (use_pl) {  // (the planes are also the tile-row operand of this layer's weight gradient)
      if (po_next.base == in.data && po_next.rows == in.rows) po_in[layer_no - 1] = po_next;
      else CK(split(in, 0, kP | kT, &po_in[layer_no - 1], s, fb_next));
    }
    fb_next = FroBound();
    po_next = PlanesOperand();
    {
      PlanesHintScope ph(hint_of(po_in[layer_no - 1]), wplanes(L.lin.comp));
      CK(tdnnf_tdnn_propagate(&L.lin.ix, &in, Wp(n, L.lin.comp), L.lin.K * Hd, L.bn, Hd, nullptr, lin_eff, 2, &lin, s));
    }
    tdnnf_mat aff_in = lin;
    if (L.c_arch >= 0) {  // bottleneck supernet: column blocks of the linear output times CopyN(Sum(p_k..))
      TDNNF_REQUIRE(n->draws || c.bn_mode == 1, "net_forward_backward: the bottleneck supernet needs net_set_random_draws before every step");
      hipLaunchKernelGGL(bn_choice_forward_kernel, dim3(1), dim3(256), 0, s, bn_choice(c), Wp(n, L.c_arch),
                         n->draws ? n->draws + L.arch_draw0 : nullptr, L.arch_p, L.arch_mask);
      aff_in = M(L.lin_masked, L.lin.rows_out, L.bn);
      hipLaunchKernelGGL(col_scale_kernel, dim3(grid_for((long long)lin.rows * L.bn, 256)), dim3(256), 0, s, view(&lin), L.arch_mask, view(&aff_in));
    }
    if (L.perm) {
      tdnnf_mat src = aff_in;
      aff_in = M(L.lin_perm, L.lin.rows_out, L.bn);
      CK(tdnnf_reorder_rows(&src, B, L.aff.ix.row_stride, 1, &aff_in, s));
    }
    tdnnf_mat relu = M(L.relu_out, L.aff.rows_out, Hd);
    if (use_pl && !L.perm) CK(split(aff_in, 0, kP | kT, &po_lin[layer_no - 1], s));  // (the linear output, or its masked blocks in the bottleneck supernet)
    {
      PlanesHintScope ph(hint_of(po_lin[layer_no - 1]), wplanes(L.aff.comp));
      FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_next.blocks);
      CK(affine_relu_bn_stats(n, &L.aff.ix, &aff_in, Wp(n, L.aff.comp), L.aff.K * L.bn, Hd, L.bn, Bp(n, L.aff.comp), aff_eff, &relu, L.bn_memo, L.bn_stats, s));
    }
    // noop = mask * batchnorm(relu) + bypass_scale * (rows of the layer input): its norm bound from the two parts
    fb_next.mul = mask_max;
    fb_next.add_coef = c.bypass_scale;
    fb_next.add_rec = po_in[layer_no - 1].base ? po_in[layer_no - 1].scale : nullptr;
    if (c.bypass_scale != 0.f && !fb_next.add_rec) fb_next.blocks = 0;  // (the input was not split: no bound for the sum)
    // noop = Sum(Scale(bypass, input), dropout(batchnorm(relu)))  in one pass
    tdnnf_mat byp = sub_grid_view(prev, L.gin, L.gout, B, Hd);
    tdnnf_mat x = relu, out = M(L.noop_out, L.aff.rows_out, Hd);
    if (byp.rows != out.rows) {  // strided bypass rows: view everything as (n, B*stride) super rows
      x = tdnnf_mat{L.relu_out, L.gout.n, byp.cols, B * ldpad(Hd)};
      out = tdnnf_mat{L.noop_out, L.gout.n, byp.cols, B * ldpad(Hd)};
    }
    if (byp.rows == relu.rows) CK(bn_apply_planes(x, L.bn_memo, view(&byp), c.bypass_scale, out, mask_of(layer_no), fb_next, &po_next));
    else TDNNF_HIP(bn_apply_bypass(view(&x), L.bn_memo, Hd, ldpad(Hd), view(&byp), c.bypass_scale, view(&out), s, mask_of(layer_no), B));
    prev = L.noop_out;
  }
  CK(phase_mark(n, 1, s));
  // ---- natural gradient: pending refreshes, and the input-side statistics that need nothing but forward activations
  const bool use_ng = c.use_natural_gradient != 0;
  auto finish_refreshes = [&]() -> int {
    if (use_ng && n->ng_grouped) {
      // refreshes still pending when the step began: W_{t+1} of ALL of them now, as grouped launches on the side stream (the host
      // waits for the eigen-decompositions here; the GPU has the forward pass and the denominator in its queues meanwhile)
      if (!n->ngfin) {
        bool ready = false;
        for (tdnnf_ng *g : all_ng()) ready = ready || ng_dim(g) != 0;
        if (ready) CK(ng_fin_create(all_ng(), &n->ngfin));
      }
      if (n->ngfin) {
        int did = 0;
        SplitKScratchOverride side_scratch(n->s3_scratch, n->s3_scratch_bytes);
        CK(ng_fin_run(n->ngfin, n->s3, true, &did));
        if (did) {
          TDNNF_HIP(hipEventRecord(n->ev_fin, n->s3));
          early_refresh = true;
        }
      }
    }
    if (early_refresh) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_fin, 0));  // the preconditioners refreshed on s3 (at the start of the step, or just now)
    return TDNNF_OK;
  };
  n->early_any = false;
  auto is_head_comp = [&](int comp) { return comp == n->c_prefinal_l || comp == n->head[0].c_affine || comp == n->head[0].c_linear || comp == n->head[0].c_output ||
                                             comp == n->head[1].c_affine || comp == n->head[1].c_linear || comp == n->head[1].c_output; };
  // which: 0 the trunk's components (their inputs exist once the trunk's forward pass is enqueued), 1 the heads' (and whatever was not launched yet), 2 all
  auto launch_early_in = [&](int which) -> int {
    if (!(use_ng && n->early_on)) return TDNNF_OK;
    // input-side statistics of every component whose backward call of the previous minibatch recorded its arguments (and has not come yet
    // in this one) and whose preconditioners exist (the grouped chain will take them): on s4, behind the forward pass and the refresh uploads
    bool forked = false;
    if (n->early_group) {  // one grouped launch on s4 (behind the numerator, in front of the heads' weight gradients; s3 carries the denominator's second recursion)
      std::vector<RowsGemmArgs> calls;
      std::vector<int> who;
      for (int comp = (int)n->comps.size() - 1; comp >= 0; comp--) {
        auto &E = n->early[comp];
        if (E.done == n->fb_count || (which == 0 && is_head_comp(comp))) continue;
        if (E.recorded != n->fb_count - 1 || !n->ng_in[comp] || !n->ng_out[comp] || ng_dim(n->ng_in[comp]) == 0 || ng_dim(n->ng_out[comp]) == 0) continue;
        if (n->comps[comp].lr_factor == 0.f || !n->ngc[comp].H_in) continue;
        NgInput xin;
        memcpy(&xin, E.xin, sizeof(xin));
        if (!forked) {
          TDNNF_HIP(hipEventRecord(n->ev_early_in, s));
          TDNNF_HIP(hipStreamWaitEvent(n->s4, n->ev_early_in, 0));
          if (early_refresh) TDNNF_HIP(hipStreamWaitEvent(n->s4, n->ev_fin, 0));
          forked = true;
        }
        RowsGemmArgs a;
        CK(ng_stats_main_prepare(n->ng_in[comp], xin, n->ngc[comp].H_in, n->ngc[comp].part_in, n->s4, &a));
        if (!rows_gemm_group_ok(a)) continue;  // (this one in its own gradient call, as without the early launch)
        calls.push_back(a);
        who.push_back(comp);
      }
      if (!calls.empty()) {
        TDNNF_HIP(rows_gemm_group(calls, &n->early_launch, n->s4));
        for (int comp : who) n->early[comp].done = n->fb_count;
        TDNNF_HIP(hipEventRecord(n->ev_early, n->s4));
        n->early_any = true;
      }
      return TDNNF_OK;
    }
    SplitKScratchOverride early_scratch(n->s4_scratch, n->s4_scratch_bytes);
    for (int comp = (int)n->comps.size() - 1; comp >= 0; comp--) {
      auto &E = n->early[comp];
      if (E.done == n->fb_count || (which == 0 && is_head_comp(comp))) continue;
      if (E.recorded != n->fb_count - 1 || !n->ng_in[comp] || !n->ng_out[comp] || ng_dim(n->ng_in[comp]) == 0 || ng_dim(n->ng_out[comp]) == 0) continue;
      if (n->comps[comp].lr_factor == 0.f || !n->ngc[comp].H_in) continue;
      if (!forked) {
        TDNNF_HIP(hipEventRecord(n->ev_early_in, s));
        TDNNF_HIP(hipStreamWaitEvent(n->s4, n->ev_early_in, 0));
        if (early_refresh) TDNNF_HIP(hipStreamWaitEvent(n->s4, n->ev_fin, 0));
        forked = true;
      }
      NgInput xin;
      memcpy(&xin, E.xin, sizeof(xin));
      CK(ng_stats_main(n->ng_in[comp], xin, n->ngc[comp].H_in, n->ngc[comp].part_in, n->ws4, n->ws_bytes, n->s4));
      E.done = n->fb_count;
    }
    if (forked) {
      TDNNF_HIP(hipEventRecord(n->ev_early, n->s4));
      n->early_any = true;
    }
    return TDNNF_OK;
  };
  // Minibatches whose GEMMs fill the chip (no weight-gradient streams): the TRUNK components' statistics start here, where the trunk's forward
  // pass ends -- the caller's stream is about to wait 4.4 ms for the denominator's two latency-bound recursions (xent_behind_den), with the matrix
  // cores and HBM idle; the heads' components follow where all of them used to start, behind the xent head's backward pass.
  const bool early_at_fork = n->early_on && !n->wg_on && !n->early_group && options().ng_early_fork != 0;
  if (early_at_fork) {
    CK(finish_refreshes());
    CK(launch_early_in(0));
  }
  const int No = n->Tout * B;
  tdnnf_mat top = M(prev, No, Hd), pl = M(n->prefinal_l_out, No, S);
  if (po_next.base == top.data && po_next.rows == top.rows) po_top = po_next;
  else CK(split(top, 0, kP | kT, &po_top, s, fb_next));
  {
    PlanesHintScope ph(hint_of(po_top), wplanes(n->c_prefinal_l));
    CK(tdnnf_affine_propagate(&top, Wp(n, n->c_prefinal_l), Hd, nullptr, S, &pl, s));
  }
  CK(split(pl, 0, kP | kT, &po_pl, s));
  tdnnf_mat y = M(n->head[0].y, No, P), dy = M(n->d_y, No, P), dx = M(n->d_xent, No, P);
  tdnnf_mat lsm = M(n->xent_logsoftmax, No, P);
  for (int h = 0; h < 2; h++) {
    auto &H = n->head[h];
    tdnnf_mat ar = M(H.aff_relu, No, Hd), lo = M(H.lin_out, No, S), b1 = M(H.bn1_out, No, Hd), b2 = M(H.bn2_out, No, S), yh = M(H.y, No, P);
    FroBound fb_b1;
    {
      PlanesHintScope ph(hint_of(po_pl), wplanes(H.c_affine));
      FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_b1.blocks);
      CK(affine_relu_bn_stats(n, &ix1, &pl, Wp(n, H.c_affine), S, Hd, S, Bp(n, H.c_affine), nullptr, &ar, H.bn1_memo, H.bn1_stats, s));
    }
    CK(bn_apply_planes(ar, H.bn1_memo, none, 0.f, b1, nullptr, fb_b1, &po_b1[h]));
    if (!po_b1[h].base) CK(split(b1, 0, kP | kT, &po_b1[h], s, fb_b1));
    {
      PlanesHintScope ph(hint_of(po_b1[h]), wplanes(H.c_linear));
      CK(tdnnf_affine_propagate(&b1, Wp(n, H.c_linear), Hd, nullptr, S, &lo, s));
    }
    CK(bn_fwd(n, H.lin_out, H.bn2_out, No, S, H.bn2_memo, H.bn2_stats, s));
    CK(split(b2, 0, kP | kT, &po_b2[h], s));
    {
      PlanesHintScope ph(hint_of(po_b2[h]), wplanes(H.c_output));
      CK(tdnnf_affine_propagate(&b2, Wp(n, H.c_output), S, Bp(n, H.c_output), P, &yh, s));
    }
    if (h == 0) {
      // ====================================================== objective, part 1 (second stream)
      // The denominator forward-backward (one workgroup per sequence) only needs the chain head's output: it
      // runs on n->s2 while this stream does the xent head forward, the numerator and the xent head backward.
      TraceRange trace_den("chain denominator forward-backward (second stream)");
      TDNNF_HIP(hipEventRecord(n->ev_fork, s));
      TDNNF_HIP(hipStreamWaitEvent(n->s2, n->ev_fork, 0));
      // few sequences leave most CUs idle while one workgroup per sequence walks the frames: there the backward recursion runs
      // beside the forward one (den_beta_kernel on a further stream) and the occupancies of all frames at once
      const bool den_split = n->den_split;
      TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_fork, 0));
      // (option xent_behind_den: the xent head's forward GEMMs start when the recursions are done.  Beside the recursions' 1024-thread workgroups,
      // which hold half the CUs for 4.4 ms, those GEMMs run at 94 instead of 103 TFLOP/s exact f32 (157 instead of 183 f32-equivalent on the plane
      // kernels) -- and the step takes as long either way: three interleaved pairs 120.15 / 120.32 ms exact f32, 82.66 / 82.72 f16x3.  Rounds 2-4
      // had this order by accident, through a 1024-thread BatchNorm finalize block that could not start beside the recursions.  Kept explicit:
      // the same step, GEMM launches that are not stretched by a neighbour.)
      const int xbd = options().xent_behind_den;
      const bool behind = den_split && (xbd > 0 || (xbd < 0 && !n->wg_on));
      if (behind && !n->ev_den_rec) TDNNF_HIP(hipEventCreateWithFlags(&n->ev_den_rec, hipEventDisableTiming));
      bool rec = false;
      CK(chain_den(den, sup, &y, c.leaky_hmm, &dy, n->chain_ws, n->s2, !den_split, n->s3, behind ? n->ev_den_rec : nullptr, &rec));
      if (rec) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_den_rec, 0));
      TDNNF_HIP(hipEventRecord(n->ev_den, n->s2));
      // ... and so does the numerator's forward-backward recursion (one wave per sequence): on a stream that is idle until the
      // backward pass -- the weight-gradient stream when there is one, else behind the second recursion on the side stream
      hipStream_t sn = n->s3;
      if (n->wg_on && den_split) {
        sn = n->s4;
        TDNNF_HIP(hipStreamWaitEvent(sn, n->ev_fork, 0));
      }
      CK(chain_num_recursion(sup, den, &y, n->chain_ws, sn));
      TDNNF_HIP(hipEventRecord(n->ev_num, sn));
    }
  }
  tdnnf_mat yx = M(n->head[1].y, No, P);
  // xent head: LogSoftmax, numerator posteriors, LogSoftmax backward.  The derivative handed to LogSoftmax is xent_regularize *
  // weight * (posteriors of a frame: they sum to 1), so its backward pass is -xent_regularize * weight * softmax -- written by the
  // forward kernel while the row is in registers -- plus the posteriors the numerator kernel adds on top.
  const bool dense_first = log_softmax_propagate_with_aux(&yx, &lsm, &dx, -c.xent_regularize * chain_supervision_weight(sup), s);
  if (!dense_first) CK(tdnnf_log_softmax_propagate(&yx, &lsm, s));
  // objective, part 2: numerator recursion -> xent_deriv (+)= xent_regularize * posteriors, xent objective
  TDNNF_HIP(hipStreamWaitEvent(s, n->ev_num, 0));
  CK(chain_num_xent(den, sup, &y, &lsm, c.xent_regularize, &dx, n->chain_ws, s, dense_first));
  if (!dense_first) CK(tdnnf_log_softmax_backprop(&lsm, &dx, &dx, s));  // in place into d_xent

  CK(phase_mark(n, 2, s));
  // ================================================================= backward
  // BatchNorm backward + ReLU backward (+ StoreStats / self-repair coin flips as in the reference:
  // RectifiedLinearComponent::StoreStats nnet-simple-component.cc:1084, RepairGradients :1017) in two fused
  // passes; also yields the bias gradient of the affine layer in front of the ReLU.
  // With natural gradient the same sweep also forms the output-side statistic H = dY Wy^T of the affine component in front
  // of the ReLU (fused.h NgFuse): into the buffer set the component's param_grad call, which must come next, will take.
  // parity aid: keep a copy of a derivative matrix under `name` (tdnnf_net_set_capture); the originals are recycled scratch
  auto capture = [&](const std::string &name, const tdnnf_mat &m) -> int {
    if (!n->capture_on) return TDNNF_OK;
    tdnnf_mat *dst = nullptr;
    for (auto &kv : n->named)
      if (kv.first == name) dst = &kv.second;
    if (!dst) {
      float *p = nullptr;
      TDNNF_HIP(hipMalloc((void **)&p, sizeof(float) * (size_t)std::max(1, m.rows) * ldpad(m.cols)));
      n->captured.push_back(p);
      n->named.push_back({name, M(p, m.rows, m.cols)});
      dst = &n->named.back().second;
    }
    TDNNF_REQUIRE(dst->rows == m.rows && dst->cols == m.cols, "net_forward_backward: captured %s changed shape", name.c_str());
    return tdnnf_sum_scaled(&m, 1.0f, nullptr, 0.f, dst, s);
  };
  int fused_comp = -1;
  auto out_stats_fuse = [&](int comp, MatView xv, MatView dzv, MatView dv, NgFuse &f) -> int {  // 1: f is to be passed on
    fused_comp = -1;
    if (!c.use_natural_gradient || n->ng_out.empty() || !n->ng_out[comp] || n->comps[comp].lr_factor == 0.f) return 0;
    if (options().ng_fuse == 0) return 0;  // the statistic by its own GEMM
    const bool fuse_always = options().ng_fuse == 2;  // fused whatever the row count
    auto &S = n->ngc[comp];
    const float *W = nullptr;
    int Rp = 0, ldw = 0;
    CK(ng_external_begin(n->ng_out[comp], dv.cols, &W, &Rp, &ldw, s));
    if (!W || !bn_relu_bwd_ng_ok(xv, dzv, dv, Rp) || !(fuse_always || bn_relu_bwd_ng_pays(dv.rows))) return 0;
    f.W = W; f.Rp = Rp; f.ldw = ldw; f.H = S.H_out; f.part = S.part_out; f.part_cap = rows_gemm_sumsq_blocks(dv.rows);
    fused_comp = comp;
    return 1;
  };
  // NonlinearComponent::StoreBackpropStats (nnet-component-itf.cc:461-480): "if (RandInt(0, 3) == 0 && oderiv_count_ != 0) return"
  // -- three minibatches in four, always the first; a decision stream of its own (the k-th ReLU of the backward pass)
  unsigned long long relu_k = 0;
  auto oderiv_of = [&](double *relu_stats, int relu_index) -> double * {  // relu_index: 0 tdnn1, 1 + l tdnnf layer l, num_layers + 1 + h head h
    std::vector<char> &nz = *n->oderiv_nonzero;
    if ((int)nz.size() < c.num_layers + 3) nz.resize(c.num_layers + 3, 0);
    const bool skip = nz[relu_index] && ::tdnnf::tdnnf_decision((unsigned long long)step, 2 * (4096 + relu_k)) % 4 == 0;
    relu_k++;
    if (!skip) nz[relu_index] = 1;
    return skip ? nullptr : relu_stats + 1 + 2 * Hd;
  };
  // po (may be null): receives the plane operand of the derivative when the sweep wrote its f16 planes itself (bwd_planes; the caller
  // has a FroBoundScope installed), else stays empty and the caller splits
  auto bn_relu_backward = [&](float *relu_out, float *d_io, int rows, float *memo, double *relu_stats, float *bias_acc, int comp, int relu_index,
                              const float *mask = nullptr, PlanesOperand *po = nullptr) -> int {
    const bool store = coin() || step == 0;
    const bool repair = c.relu_self_repair_scale > 0.f && coin();
    tdnnf_mat x = M(relu_out, rows, Hd), d = M(d_io, rows, Hd);
    NgFuse f;
    const int fuse = out_stats_fuse(comp, view(&x), view(&d), view(&d), f);
    if (fuse < 0) return TDNNF_EINVAL;
    BwdPlanes bp{nullptr, 0, 0, nullptr};
    PlanesOperand tmp;
    if (po && fro_bound_buf()) CK(bwd_planes(d, 0, &bp, &tmp));
    TDNNF_HIP(bn_relu_bwd(view(&x), view(&d), memo, 1.0f, cv, relu_stats, store, repair, c.relu_self_repair_scale, view(&d), bias_acc, 1.0f,
                          n->ws, n->ws_bytes, s, mask, B, fuse ? &f : nullptr, oderiv_of(relu_stats, relu_index), bp.P ? &bp : nullptr));
    if (po && bp.P) *po = tmp;
    if (bp.P && options().planes_check_bound) TDNNF_HIP(planes_check_bound(view(&d), bp.rec, n->planes_ws, s));
    return TDNNF_OK;
  };
  // Weight gradients up to three components behind the caller's stream (option wgrad_lag, default 3; 1 = rounds 2-4: one behind).  A
  // buffer that component k's gradient reads may be rewritten once the caller's stream has waited for k, i.e. from the hand-off of
  // component k + 3 on: the derivative matrices those gradients read alternate between two buffers per role (layout_arena).
  const bool lag3 = n->wg_on && n->wg_lag == 3 && n->dC2 != nullptr;
  const int Ltop = c.num_layers - 1;
  auto dC_of = [&](int l) -> float * { return !lag3 ? n->dC : (((Ltop - l) & 1) ? n->dC2 : n->dC); };      // d affine-out of tdnnf layer l
  auto dS_of = [&](int l) -> float * { return !lag3 ? nullptr : n->dS[(Ltop - l) & 1]; };                  // what layer l's .linear gradient reads
  // A bucket of the flat gradient buffer is final once the last of its components (in backward order) has been enqueued:
  // grads[range] += this minibatch's gradient (unless the objective failed), then the bucket's event.  With natural gradient
  // the components' commits run on the side stream, so the bucket's commit follows them there.
  // (forward declaration of the natural-gradient group chain of the components enqueued since the last bucket closed)
  std::function<int(int)> ng_close;
  auto side_streams = [&]() -> unsigned { return 1u + (n->wg_two ? 1u : 0u) + (n->s5 ? 1u : 0u); };  // weight-gradient streams in use now (<= 3 <= the event ring's lag)
  auto close_bucket = [&](int key) -> int {
    if (use_ng) CK(ng_close(key));
    for (auto &gb : n->buckets) {
      if (gb.close_key != key) continue;
      hipStream_t cs = s;
      if (!use_ng && n->wg_on && n->pg_count > 0) {  // the components' gradients were formed on s4 (and s2)
        for (unsigned b = 1; b <= side_streams() && b <= n->pg_count; b++) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_pg[(n->pg_count - b) & 3], 0));
      }
      if (use_ng) {
        TDNNF_HIP(hipEventRecord(gb.handoff, s));  // s-side writes of the range (bias sums, architecture parameters) are done
        TDNNF_HIP(hipStreamWaitEvent(n->s3, gb.handoff, 0));
        cs = n->s3;
      }
      const long long cnt = gb.end - gb.begin;
      if (cnt > 0)
        hipLaunchKernelGGL(commit_grads_kernel, dim3(grid_for(cnt, 256)), dim3(256), 0, cs, n->grads + gb.begin, n->gtmp + gb.begin, cnt, results);
      TDNNF_HIP(hipEventRecord(gb.ready, cs));
    }
    return TDNNF_OK;
  };
  // Gradient of one component's weights [+ bias] from its input (tap views) and output derivative.
  //   raw:  W += c_i dY^T X_i, bias += colsum(dY)                      (UpdateSimple, :433-455)
  //   NG :  X~ = [c_i X_i ..., 1], (X~', a) = NG_in(X~), (dY', b) = NG_out(dY), W += a b dY'^T X~'[:, :K Di],
  //         bias += a b dY'^T X~'[:, -1]                               (UpdateNaturalGradient, :592-624)
  //         computed as a b (I - Wy^T Wy) [raw gradient] (I - Wx^T Wx): see ng.h.
  // bias_done: the raw bias gradient was already produced by the fused BatchNorm/ReLU backward pass (into Bg(), or
  // into n->ngBias when natural gradient is on).  tapgrad: unscaled per-tap gradients already in n->tapgrad.
  auto bias_target = [&](int comp) -> float * {  // where a fused backward pass should accumulate the raw bias gradient
    if (n->comps[comp].lr_factor == 0.f) return nullptr;  // "if (to_update && learning_rate != 0)": no model derivative
    if (!use_ng) return Bg(n, comp);
    return n->ngc[comp].bsum;  // (zeroed with all the others at the start of the step)
  };
  bool den_joined = false;   // the caller's stream has waited for the denominator
  bool caller_used = false;  // some component of the open bucket formed its gradient on the caller's stream although wg_on
  auto param_grad = [&](int comp, const tdnnf_tdnn_indexes &ix, int K, int Di, int Do, tdnnf_mat *x, tdnnf_mat *dyv, const float *eff,
                        bool bias_done, const int *active, int max_active, bool from_tapgrad) -> int {
    const int ldw = K * Di;
    float *bias_acc = Bg(n, comp);
    if (n->comps[comp].lr_factor == 0.f) return TDNNF_OK;  // frozen component (cv-update): the reference skips its update
    // sw: where the gradient is formed -- the weight-gradient stream (its inputs are final on s now), or s itself
    hipStream_t sw = s;
    void *wsw = n->ws;
    // Small minibatches: the weight-gradient side (gradient GEMM + slab reduce + the natural-gradient passes of the component) is the longer
    // chain of the two -- 216 us per component against 145 on the caller's stream at 150 x 64 -- and the caller may only run one component
    // ahead of it.  From the denominator's join on its stream is idle: components alternate between the two (event parity = stream parity).
    // (option wgrad_on_caller: the xent head's components -- before the denominator's join, with the input-side statistics of the whole net
    // queued on s4 in front of them -- form their gradients on the caller's stream, which waits for the denominator anyway)
    const bool on_caller = n->wg_on && options().wgrad_on_caller != 0 && n->early_any && !den_joined;
    if (on_caller) caller_used = true;
    // which of the weight-gradient streams: s4 [s5] before the denominator's join, s4 s2 [s5] in turn after it
    float *scr = nullptr;
    if (n->wg_on && !on_caller) {
      const int ns = (n->wg_two ? 2 : 1) + (n->s5 ? 1 : 0), k = (int)(n->pg_count % (unsigned)ns);
      const int which = k == 0 ? 0 : (k == 1 && n->wg_two) ? 1 : 2;  // 0: s4, 1: s2, 2: s5
      sw = which == 0 ? n->s4 : which == 1 ? n->s2 : n->s5;
      wsw = which == 0 ? n->ws4 : which == 1 ? n->ws2 : n->ws5;
      scr = which == 0 ? n->s4_scratch : which == 1 ? n->s2_scratch : n->s5_scratch;
      TDNNF_HIP(hipEventRecord(n->ev_pg_in, s));
      TDNNF_HIP(hipStreamWaitEvent(sw, n->ev_pg_in, 0));
    }
    SplitKScratchOverride sw_scratch(scr, scr ? n->s4_scratch_bytes : 0);
    // after this component is enqueued the caller's stream may only run ahead of it, not of the one before: what that one reads
    // (derivative scratch, the bias sums) is rewritten from here on
    auto handed_off = [&]() -> int {
      if (!n->wg_on || on_caller) return TDNNF_OK;
      // (a ring of four events; wg_lag 1: the caller's stream waits for the component before this one, 3: for the one three back --
      // every component is waited for exactly once either way, so "waited for k" means every component up to k has finished)
      TDNNF_HIP(hipEventRecord(n->ev_pg[n->pg_count & 3], sw));
      const unsigned lag = lag3 ? 3u : 1u;
      if (n->pg_count >= lag) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_pg[(n->pg_count - lag) & 3], 0));
      n->pg_count++;
      return TDNNF_OK;
    };
    if (!use_ng) {
      CK(tdnn_update_simple_impl(&ix, x, dyv, Do, Di, eff, 1.0f, Wg(n, comp), ldw, bias_done ? nullptr : bias_acc, wsw, n->ws_bytes, active, max_active, sw));
      return handed_off();
    }
    const int N = dyv->rows, ones = bias_acc ? 1 : 0, Dx = ldw + ones, ldT = (Dx + 3) & ~3;
    auto &S = n->ngc[comp];
    TDNNF_REQUIRE(S.T && S.N == N, "net_forward_backward: component %s has no natural-gradient buffers for %d rows", n->comps[comp].name.c_str(), N);
    float *T = S.T;
    // grouped chain: once both preconditioners exist (from the second minibatch on)
    const bool grouped = n->ng_grouped && ng_dim(n->ng_in[comp]) != 0 && ng_dim(n->ng_out[comp]) != 0;
    // the gradient GEMM writes T[:, :K Di] itself when it computes every tap; the bias column and the row padding come with
    // set_column_kernel (the group's first launch) -- no zero fill of the 2-20 MB block first.  (With tap coefficients a zero one
    // makes the reduce kernel skip its columns: those launches start from zeros.)
    const bool overwrite = !from_tapgrad && !active && eff == nullptr && (ones || ldT == ldw);
    if (!overwrite) TDNNF_HIP(hipMemsetAsync(T, 0, sizeof(float) * (size_t)Do * ldT, sw));
    if (from_tapgrad)
      hipLaunchKernelGGL(scaled_taps_to_kernel, dim3(grid_for((long long)Do * ldw, 256)), dim3(256), 0, sw, n->tapgrad, eff, Do, ldw, Di, T, ldT);
    else
      CK(tdnn_update_simple_impl(&ix, x, dyv, Do, Di, eff, 1.0f, T, ldT, nullptr, wsw, n->ws_bytes, active, max_active, sw, overwrite));
    if (ones) {
      if (!bias_done) TDNNF_HIP(colsum_add(view(dyv), 1.0f, S.bsum, wsw, sw));  // (otherwise the fused ReLU backward pass on s filled it)
      if (!grouped) hipLaunchKernelGGL(set_column_kernel, dim3((Do + 255) / 256), dim3(256), 0, sw, S.bsum, Do, T, ldT, ldw);
    }
    // ---- the passes over the N-sized operands
    NgInput xin;
    memset(&xin, 0, sizeof(xin));
    xin.x = view(x); xin.ix = ix; xin.Di = Di; xin.ones = ones; xin.N = N; xin.eff = eff; xin.active = active; xin.max_active = max_active;
    static_assert(sizeof(NgInput) <= sizeof(tdnnf_net::EarlyIn::xin), "EarlyIn::xin too small");
    if (n->early_on && n->early[comp].done == n->fb_count) {
      // H_in (and J on a refresh) were formed ahead of the backward pass from the arguments recorded one minibatch ago: they must be these
      TDNNF_REQUIRE(memcmp(n->early[comp].xin, &xin, sizeof(xin)) == 0, "net_forward_backward: the input of component %s moved between minibatches",
                    n->comps[comp].name.c_str());
      if (n->early_group) {  // H came with the grouped launch on s3: the bookkeeping -- and J of a refresh step -- here, behind it
        TDNNF_HIP(hipStreamWaitEvent(sw, n->ev_early, 0));
        CK(ng_stats_main_finish(n->ng_in[comp], xin, S.H_in, wsw, n->ws_bytes, sw));
      }
    } else {
      CK(ng_stats_main(n->ng_in[comp], xin, S.H_in, S.part_in, wsw, n->ws_bytes, sw));
    }
    if (n->early_on) {
      memcpy(n->early[comp].xin, &xin, sizeof(xin));
      n->early[comp].recorded = n->fb_count;
    }
    NgInput yin;
    memset(&yin, 0, sizeof(yin));
    yin.x = view(dyv); yin.ix.row_stride = 1; yin.ix.num_offsets = 1; yin.Di = Do; yin.N = N;
    if (fused_comp == comp) {  // H_out and its partials came with the BatchNorm/ReLU backward sweep
      fused_comp = -1;
      CK(ng_external_end(n->ng_out[comp], yin, S.H_out, wsw, n->ws_bytes, sw));
    } else {
      CK(ng_stats_main(n->ng_out[comp], yin, S.H_out, S.part_out, wsw, n->ws_bytes, sw));
    }
    if (grouped) {  // the rest comes with the bucket (ng_close)
      n->ng_cur.push_back(comp);
      CK(handed_off());
      if (from_tapgrad && n->wg_on) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_pg[(n->pg_count - 1) & 3], 0));
      return TDNNF_OK;
    }
    // ---- per-object chain (first minibatch): the R x R work, the projections of the raw gradient and the commit, on the side stream
    TDNNF_HIP(hipEventRecord(n->ev_ngc, sw));
    TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_ngc, 0));
    {
      SplitKScratchOverride side_scratch(n->s3_scratch, n->s3_scratch_bytes);
      CK(ng_stats_side(n->ng_in[comp], S.H_in, S.part_in, n->ng_side_ws, n->ngset_ws_bytes, n->s3));
      CK(ng_stats_side(n->ng_out[comp], S.H_out, S.part_out, n->ng_side_ws, n->ngset_ws_bytes, n->s3));
      CK(ng_project(n->ng_in[comp], n->ng_out[comp], T, Do, Dx, ldT, n->ngTmp, n->s3));
    }
    hipLaunchKernelGGL(ng_commit_kernel, dim3(grid_for((long long)Do * Dx, 256)), dim3(256), 0, n->s3, T, ldT, Do, ldw, ng_scale_dev(n->ng_in[comp]),
                       ng_scale_dev(n->ng_out[comp]), Wg(n, comp), bias_acc);
    CK(handed_off());
    // (the unscaled tap gradients this one reads are rebuilt by the next DARTS component on the caller's stream)
    if (from_tapgrad && n->wg_on) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_pg[(n->pg_count - 1) & 3], 0));
    return TDNNF_OK;
  };
  // Natural gradient, grouped: the chains of the components enqueued since the last bucket closed, as one sequence of grouped
  // launches on the side stream behind their N-sized passes.
  ng_close = [&](int key) -> int {
    if (n->ng_cur.empty()) return TDNNF_OK;
    bool closes = false;  // (called after every layer: only those that end a gradient bucket run the chain)
    for (auto &gb : n->buckets) closes = closes || gb.close_key == key;
    if (!closes) return TDNNF_OK;
    tdnnf_net::NgBucket *nb = nullptr;
    for (auto &b : n->ng_buckets)
      if (b.key == key) nb = &b;
    if (!nb) {
      std::vector<NgGroupComp> gc;
      for (int comp : n->ng_cur) {
        const CompDesc &cd = n->comps[comp];
        auto &S = n->ngc[comp];
        NgGroupComp g;
        g.in = n->ng_in[comp]; g.out = n->ng_out[comp];
        g.T = S.T; g.Do = cd.rows; g.ldw = cd.cols; g.Dx = cd.cols + (cd.has_bias ? 1 : 0); g.ldT = (g.Dx + 3) & ~3;
        g.bsum = cd.has_bias ? S.bsum : nullptr;
        g.H_in = S.H_in; g.H_out = S.H_out; g.part_in = S.part_in; g.part_out = S.part_out; g.N = S.N;
        g.W_acc = Wg(n, comp); g.bias_acc = Bg(n, comp);
        gc.push_back(g);
      }
      NgGroup *grp = nullptr;
      CK(ng_group_create(gc, &grp));
      n->ng_buckets.push_back(tdnnf_net::NgBucket{key, n->ng_cur, grp});
      nb = &n->ng_buckets.back();
    }
    TDNNF_REQUIRE(nb->comps == n->ng_cur, "net_forward_backward: the components of gradient bucket %d changed between minibatches", key);
    // behind the last component's passes (on the weight-gradient stream when that is on)
    if (n->wg_on && n->pg_count > 0) {
      // (the last component of every weight-gradient stream in use)
      for (unsigned b = 1; b <= side_streams() && b <= n->pg_count; b++) TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_pg[(n->pg_count - b) & 3], 0));
      if (caller_used) {
        TDNNF_HIP(hipEventRecord(n->ev_ngc, s));
        TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_ngc, 0));
        caller_used = false;
      }
    } else {
      TDNNF_HIP(hipEventRecord(n->ev_ngc, s));
      TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_ngc, 0));
    }
    if (n->early_any) TDNNF_HIP(hipStreamWaitEvent(n->s3, n->ev_early, 0));  // the H_in of this step (all of them: one event)
    CK(ng_group_run(nb->group, n->s3));
    n->ng_cur.clear();
    return TDNNF_OK;
  };
  n->ng_cur.clear();
  tdnnf_mat d_pl = M(n->d_small, No, S);  // deriv w.r.t. prefinal-l output, summed over both heads
  if (!early_at_fork) CK(finish_refreshes());
  // Where in the host's order: minibatches whose GEMMs fill the chip start the statistics BEHIND the xent head's backward pass -- they then run
  // while the caller's stream waits for the denominator instead of beside the xent head's GEMMs (same step time, 124.0 against 124.1 ms, and the
  // 128 x 128 class is not slowed: 99.5 against 96.4 TFLOP/s, weight gradients 107 against 102); the small ones start them at once (behind the
  // xent head: 13.03 -> 13.10 ms at 150 x 64, 23.16 -> 23.31 at 1500 x 16).
  const bool early_after_xent = !n->wg_on;
  if (!early_after_xent) CK(launch_early_in(2));
  for (int h = 1; h >= 0; h--) {  // xent head first: it does not depend on the denominator
    auto &H = n->head[h];
    TraceRange trace_head(h == 0 ? "backward prefinal-chain / output" : "backward prefinal-xent / output-xent");
    if (h == 0) {
      // objective, part 3: join the denominator stream, d_y += posteriors, objf / failure handling
      TDNNF_HIP(hipStreamWaitEvent(s, n->ev_den, 0));
      den_joined = true;
      n->wg_two = n->wg_on && n->ws2 != nullptr;  // the denominator's stream is idle from here on
      CK(chain_finish(den, sup, &y, c.chain_l2_regularize, results, &dy, nullptr, n->chain_ws, s));
    }
    tdnnf_mat dout = h == 0 ? dy : dx;
    tdnnf_mat b2 = M(H.bn2_out, No, S), b1 = M(H.bn1_out, No, Hd);
    // (lag3: head h = 1 takes the buffers of the top layer (Ltop), h = 0 those of the layer below: free again by the time those layers write them)
    float *d_b1_buf = !lag3 ? n->dA : dC_of(h == 1 ? Ltop : Ltop - 1), *d_b2_buf = !lag3 ? n->d_small2 : n->dS[h == 1 ? 0 : 1];
    tdnnf_mat d_b2 = M(d_b2_buf, No, S), d_b1 = M(d_b1_buf, No, Hd);
    const std::string hname = h == 0 ? "prefinal-chain" : "prefinal-xent";
    if (h == 1) CK(capture("output-xent.deriv", dout));
    PlanesOperand po_d;  // the derivative matrix being propagated: planes where a GEMM reads it
    CK(split(dout, 0, kP | kT, &po_d, s));
    {
      PlanesHintScope ph(hint_of(po_d), hint_of(po_b2[h]));
      CK(param_grad(H.c_output, ix1, 1, S, P, &b2, &dout, nullptr, false, nullptr, 0, false));
    }
    {
      PlanesHintScope ph(hint_of(po_d), wplanes(H.c_output));
      CK(tdnnf_affine_backprop(&dout, Wp(n, H.c_output), S, S, &d_b2, s));
    }
    CK(capture(hname + ".batchnorm2.deriv", d_b2));
    if (cv) CK(tdnnf_batchnorm_test_backprop(&d_b2, H.bn2_memo + 2 * S, &d_b2, s));
    else CK(tdnnf_batchnorm_backprop(&b2, &d_b2, 1.0f, H.bn2_memo, &d_b2, n->ws, n->ws_bytes, s));  // -> d lin_out
    CK(capture(hname + ".linear.deriv", d_b2));
    CK(split(d_b2, 0, kP | kT, &po_d, s));
    {
      PlanesHintScope ph(hint_of(po_d), hint_of(po_b1[h]));
      CK(param_grad(H.c_linear, ix1, 1, Hd, S, &b1, &d_b2, nullptr, false, nullptr, 0, false));
    }
    {
      PlanesHintScope ph(hint_of(po_d), wplanes(H.c_linear));
      CK(tdnnf_affine_backprop(&d_b2, Wp(n, H.c_linear), Hd, Hd, &d_b1, s));
    }
    CK(capture(hname + ".batchnorm1.deriv", d_b1));
    FroBound fb_d;
    {
      FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_d.blocks);
      CK(bn_relu_backward(H.aff_relu, d_b1_buf, No, H.bn1_memo, H.relu_stats, bias_target(H.c_affine), H.c_affine, c.num_layers + 1 + h, nullptr, &po_d));  // in place -> d affine out
    }
    CK(capture(hname + ".affine.deriv", d_b1));
    if (!po_d.base || po_d.base != d_b1.data) CK(split(d_b1, 0, kP | kT, &po_d, s, fb_d));
    {
      PlanesHintScope ph(hint_of(po_d), hint_of(po_pl));
      CK(param_grad(H.c_affine, ix1, 1, S, Hd, &pl, &d_b1, nullptr, true, nullptr, 0, false));
    }
    PlanesHintScope ph_bp(hint_of(po_d), wplanes(H.c_affine));
    if (h == 1) {
      CK(tdnnf_affine_backprop(&d_b1, Wp(n, H.c_affine), S, S, &d_pl, s));
      if (early_after_xent) CK(launch_early_in(early_at_fork ? 1 : 2));
    } else {
      tdnnf_mat tmp = M(!lag3 ? n->d_small2 : n->dS[0], No, S);  // (lag3: the xent head's d_b2 buffer -- its reader, three components back, has been waited for)
      CK(tdnnf_affine_backprop(&d_b1, Wp(n, H.c_affine), S, S, &tmp, s));
      CK(tdnnf_add_scaled(&tmp, 1.0f, &d_pl, s));
    }
  }
  CK(capture("prefinal-l.deriv", d_pl));
  PlanesOperand po_dpl;
  CK(split(d_pl, 0, kP | kT, &po_dpl, s));
  {
    PlanesHintScope ph(hint_of(po_dpl), hint_of(po_top));
    CK(param_grad(n->c_prefinal_l, ix1, 1, Hd, S, &top, &d_pl, nullptr, false, nullptr, 0, false));
  }
  CK(close_bucket(-2));
  CK(phase_mark(n, 3, s));
  float *d_cur = n->dA, *d_next = n->dB;  // d_cur: deriv w.r.t. the current layer's output (noop)
  {
    tdnnf_mat d_top = M(d_cur, No, Hd);
    PlanesHintScope ph(hint_of(po_dpl), wplanes(n->c_prefinal_l));
    CK(tdnnf_affine_backprop(&d_pl, Wp(n, n->c_prefinal_l), Hd, Hd, &d_top, s));
  }
  for (int l = c.num_layers - 1; l >= 0; l--) {
    TdnnfLayer &L = n->layers[l];
    float *in_act = l > 0 ? n->layers[l - 1].noop_out : n->t1_bn;
    const int no = L.aff.rows_out, nl = L.lin.rows_out, ni = N_of(L.gin, B);
    // d_cur is needed again for the bypass term, so the derivative w.r.t. the affine output goes to dC
    tdnnf_mat d_out = M(d_cur, no, Hd), d_aff = M(dC_of(l), no, Hd);
    const std::string lname = "tdnnf" + std::to_string(l + 2);
    TraceRange trace_layer(("backward " + lname).c_str());
    CK(capture(lname + ".noop.deriv", d_out));
    FroBound fb_daff;
    auto max_off = [](const Tdnn &td) {
      int m = 0;
      for (int i = 0; i < td.K; i++) m = std::max(m, td.ix.row_offsets[i]);
      return m;
    };
    const bool use_pl = pl_on;
    PlanesOperand po_daff, po_dlin;
    BwdPlanes bp_daff{nullptr, 0, 0, nullptr};
    {
      const bool store = coin() || step == 0;
      const bool repair = c.relu_self_repair_scale > 0.f && coin();
      tdnnf_mat x = M(L.relu_out, no, Hd);
      NgFuse f;
      const int fuse = out_stats_fuse(L.aff.comp, view(&x), view(&d_out), view(&d_aff), f);
      if (fuse < 0) return TDNNF_EINVAL;
      FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_daff.blocks);
      PlanesOperand tmp;
      if (use_pl) CK(bwd_planes(d_aff, max_off(L.aff), &bp_daff, &tmp));
      TDNNF_HIP(bn_relu_bwd(view(&x), view(&d_out), L.bn_memo, 1.0f, cv, L.relu_stats, store, repair, c.relu_self_repair_scale,
                            view(&d_aff), bias_target(L.aff.comp), 1.0f, n->ws, n->ws_bytes, s, mask_of(l + 1), B, fuse ? &f : nullptr,
                            oderiv_of(L.relu_stats, 1 + l), bp_daff.P ? &bp_daff : nullptr));
      if (bp_daff.P) po_daff = tmp;
      if (bp_daff.P && options().planes_check_bound) TDNNF_HIP(planes_check_bound(view(&d_aff), bp_daff.rec, n->planes_ws, s));
    }
    CK(capture(lname + ".affine.deriv", d_aff));
    tdnnf_mat lin = M(L.lin_out, nl, L.bn);
    tdnnf_mat aff_in = L.perm ? M(L.lin_perm, nl, L.bn) : (L.c_arch >= 0 ? M(L.lin_masked, nl, L.bn) : lin);
    const float *lin_eff = L.lin.darts ? L.lin.memo + TDNNF_MAX_OFFSETS : nullptr;
    const float *aff_eff = L.aff.darts ? L.aff.memo + TDNNF_MAX_OFFSETS : nullptr;
    ProfFlopsScale taps_active(L.lin.darts && (c.darts_flags & TDNNF_DARTS_UNIFORM_SAMPLE) && L.lin.K > 2 ? 2.0 / L.lin.K : 1.0);
    // weight gradient of one Tdnn component.  DARTS in a non-sampling mode also needs the architecture-logit
    // gradient (UpdateNaturalGradient :516-590): tap gradients are formed unscaled once, s_i = <dW_i, W_i>
    // replaces the reference's extra forward GEMM per tap, then c_i * dW_i goes into the accumulator.
    auto tdnn_wgrad = [&](Tdnn &td, tdnnf_mat *x, tdnnf_mat *dyv, const float *eff, bool bias_done) -> int {
      const int ldw = td.K * td.Di;
      bool tap_ready = false;
      if (td.darts && !(c.darts_flags & TDNNF_DARTS_UNIFORM_SAMPLE)) {
        tap_ready = true;
        TDNNF_HIP(hipMemsetAsync(n->tapgrad, 0, sizeof(float) * (size_t)td.Do * ldw, s));
        CK(tdnnf_tdnn_update_simple(&td.ix, x, dyv, td.Do, td.Di, nullptr, 1.0f, n->tapgrad, ldw, nullptr, n->ws, n->ws_bytes, s));
        CK(tdnnf_tdnn_darts_alpha_update(n->tapgrad, ldw, Wp(n, td.comp), ldw, td.Do, td.Di, td.K, td.memo, c.darts_flags, td.share,
                                         c.darts_temp_proportion, 1.0f, Ag(n, td.comp), n->tapdots, s));
        if (!use_ng) {
          hipLaunchKernelGGL(add_scaled_taps_kernel, dim3(grid_for((long long)td.Do * ldw, 256)), dim3(256), 0, s, n->tapgrad, eff,
                             Wg(n, td.comp), td.Do, ldw, td.Di);
          if (!bias_done) TDNNF_HIP(colsum_add(view(dyv), 1.0f, Bg(n, td.comp), n->ws, s));
          return TDNNF_OK;
        }
      }
      // uniform-sample mode: only the share tap and the sampled tap are non-zero (:293-304) -> compacted launch
      const bool compact = td.darts && (c.darts_flags & TDNNF_DARTS_UNIFORM_SAMPLE) && td.K > 2;
      return param_grad(td.comp, td.ix, td.K, td.Di, td.Do, x, dyv, eff, bias_done, compact ? td.active : nullptr, compact ? 2 : 0, tap_ready);
    };
    if (use_pl && !po_daff.base) CK(split(d_aff, max_off(L.aff), kP | kT, &po_daff, s, fb_daff));
    {
      PlanesHintScope ph(hint_of(po_daff), hint_of(po_lin[l]));
      CK(tdnn_wgrad(L.aff, &aff_in, &d_aff, aff_eff, true));
    }
    PlanesHintScope ph_aff_bp(hint_of(po_daff), wplanes(L.aff.comp));  // (for the backward-data GEMM of the affine, either branch below)
    // (lag3: the matrix the .linear's gradient reads lives in this layer's dS buffer; with rho > 1 that is the un-permuted copy)
    // (and the permuted matrix is formed in d_small2: d_small still holds prefinal-l's output derivative, which its gradient may be reading)
    tdnnf_mat d_lin = M(lag3 ? (L.perm ? n->d_small2 : dS_of(l)) : n->d_small, nl, L.bn);
    if (L.perm) {  // rho > 1: some row classes receive no tap -> zero first, then add; un-permute afterwards
      TDNNF_HIP(hipMemsetAsync(d_lin.data, 0, sizeof(float) * (size_t)nl * d_lin.stride, s));
      CK(tdnnf_tdnn_backprop_data(&L.aff.ix, &d_aff, Wp(n, L.aff.comp), L.aff.K * L.bn, Hd, L.bn, aff_eff, &d_lin, s));
      tdnnf_mat un = M(lag3 ? dS_of(l) : n->d_small2, nl, L.bn);
      CK(tdnnf_reorder_rows(&d_lin, B, L.aff.ix.row_stride, 0, &un, s));
      d_lin = un;
    } else {
      CK(tdnn_backprop_data_impl(&L.aff.ix, &d_aff, Wp(n, L.aff.comp), L.aff.K * L.bn, Hd, L.bn, aff_eff, 1, nullptr, 0.f, 0, &d_lin, s));
    }
    if (L.c_arch >= 0) {
      // d_lin is the derivative w.r.t. the masked blocks: the CopyN factor receives colsum(lin * d) (-> gradient of the
      // C-vector), the linear output receives d * mask (ElementwiseProductComponent::Backprop :276-299)
      const int chunks = (nl + 511) / 512;
      hipLaunchKernelGGL(colsum_prod_partial_kernel, dim3((L.bn + 255) / 256, chunks), dim3(256), 0, s, view(&lin), view(&d_lin), 512, (float *)n->ws);
      hipLaunchKernelGGL(bn_choice_backward_kernel, dim3(1), dim3(256), 0, s, bn_choice(c), (const float *)n->ws, chunks, L.arch_p, Wg(n, L.c_arch));
      hipLaunchKernelGGL(col_scale_kernel, dim3(grid_for((long long)nl * L.bn, 256)), dim3(256), 0, s, view(&d_lin), L.arch_mask, view(&d_lin));
    }
    CK(capture(lname + ".linear.deriv", d_lin));
    tdnnf_mat in = M(in_act, ni, Hd);
    // (the never-added bias of a DARTS .linear is still updated by the reference, :614 -- Bg() is null for plain layers)
    if (use_pl) CK(split(d_lin, max_off(L.lin), kP | kT, &po_dlin, s));
    {
      PlanesHintScope ph(hint_of(po_dlin), hint_of(po_in[l]));
      CK(tdnn_wgrad(L.lin, &in, &d_lin, lin_eff, false));
    }
    CK(close_bucket(l));
    PlanesHintScope ph_lin_bp(hint_of(po_dlin), wplanes(L.lin.comp));  // (the backward-data GEMM of the linear below)
    // deriv w.r.t. the layer input = linear backprop (overwrites) + bypass_scale * d_out on the output-grid rows
    tdnnf_mat d_in = M(d_next, ni, Hd);
    tdnnf_mat d_byp = sub_grid_view(d_next, L.gin, L.gout, B, Hd);
    if (d_byp.rows == d_out.rows) {  // contiguous rows: fused into the GEMM epilogue
      const int row0 = (int)((d_byp.data - d_in.data) / d_in.stride);
      CK(tdnn_backprop_data_impl(&L.lin.ix, &d_lin, Wp(n, L.lin.comp), L.lin.K * Hd, L.bn, Hd, lin_eff, 1, &d_out, c.bypass_scale, row0,
                                 &d_in, s));
    } else {
      CK(tdnn_backprop_data_impl(&L.lin.ix, &d_lin, Wp(n, L.lin.comp), L.lin.K * Hd, L.bn, Hd, lin_eff, 1, nullptr, 0.f, 0, &d_in, s));
      tdnnf_mat d_o = tdnnf_mat{d_cur, L.gout.n, d_byp.cols, B * ldpad(Hd)};
      CK(tdnnf_add_scaled(&d_o, c.bypass_scale, &d_byp, s));
    }
    std::swap(d_cur, d_next);
  }
  {  // tdnn1: batchnorm -> relu -> affine (the lda layer is fixed: no input derivative needed)
    FroBound fb_d;
    PlanesOperand po_d;
    {
      FroBoundScope fbs(pl_on ? n->fro_buf : nullptr, &fb_d.blocks);
      CK(bn_relu_backward(n->t1_relu, d_cur, N0, n->t1_bn_memo, n->t1_relu_stats, bias_target(n->tdnn1.comp), n->tdnn1.comp, 0, mask_of(0), &po_d));
    }
    tdnnf_mat d_aff = M(d_cur, N0, Hd);
    if (!po_d.base) CK(split(d_aff, 0, kT, &po_d, s, fb_d));
    PlanesHintScope ph(hint_of(po_d), hint_of(po_lda));
    CK(param_grad(n->tdnn1.comp, ix1, 1, lda_dim, Hd, &lda_out, &d_aff, nullptr, true, nullptr, 0, false));
  }
  CK(close_bucket(-1));
  CK(phase_mark(n, 4, s));
  if (n->wg_on)  // join the weight-gradient streams
    for (unsigned b = 1; b <= 3 && b <= n->pg_count; b++) TDNNF_HIP(hipStreamWaitEvent(s, n->ev_pg[(n->pg_count - b) & 3], 0));
  n->wg_two = false;
  if (use_ng) {  // join the side stream: every bucket has been committed into grads
    TDNNF_HIP(hipEventRecord(n->ev_s3, n->s3));
    TDNNF_HIP(hipStreamWaitEvent(s, n->ev_s3, 0));
  }
  CK(phase_mark(n, 5, s));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_net_phase_times(tdnnf_net *n, double *ms_out, int capacity, int *count) {
  TDNNF_REQUIRE(n && ms_out && count && capacity >= tdnnf_net::kPhases - 1, "net_phase_times: bad arguments (capacity >= 7)");
  *count = 0;
  for (int k = 0; k < tdnnf_net::kPhases; k++)
    if (!n->phase_rec[k]) return TDNNF_OK;  // (option phase_events was off for the last step)
  TDNNF_HIP(hipEventSynchronize(n->ev_phase[tdnnf_net::kPhases - 1]));
  for (int k = 0; k + 1 < tdnnf_net::kPhases; k++) {
    float ms = 0.f;
    TDNNF_HIP(hipEventElapsedTime(&ms, n->ev_phase[k], n->ev_phase[k + 1]));
    ms_out[k] = ms;
  }
  *count = tdnnf_net::kPhases - 1;
  return TDNNF_OK;
}

int tdnnf_net_update(tdnnf_net *n, float lr, float l2_scale, long long step, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && n->params && n->grads, "net_update: call net_set_buffers first");
  TDNNF_REQUIRE(lr >= 0.f && l2_scale >= 0.f, "net_update: learning rate and l2 scale must be >= 0 (nnet-utils.cc:2240)");
  TraceRange trace_update("tdnnf_net_update");
  hipStream_t s = (hipStream_t)stream;
  CK(phase_mark(n, 6, s));
  const int nc = (int)n->comps.size();
  UpdTable tb;
  memset(&tb, 0, sizeof(tb));
  std::vector<long long> begin(nc + 1);
  std::vector<float> mc(nc);
  for (int i = 0; i < nc; i++) {
    const CompDesc &c = n->comps[i];
    begin[i] = tb.begin[i] = c.begin;
    const float lrc = lr * c.lr_factor;
    tb.lr[i] = lrc;
    tb.l2coef[i] = -2.0f * l2_scale * lrc * c.l2;  // ApplyL2Regularization, nnet-utils.cc:2241
    mc[i] = c.max_change;
  }
  begin[nc] = tb.begin[nc] = n->num_params;
  // component i owns [begin[i], begin[i+1]) including alignment padding (padding stays zero).  delta = lr g + l2 theta, max-change and the
  // update as three launches over all components (optim_group.hip)
  if (!n->upd || upd_group_params(n->upd) != n->params) {
    upd_group_destroy(n->upd);
    n->upd = nullptr;
    std::vector<UpdComp> uc(nc);
    for (int i = 0; i < nc; i++) {
      const CompDesc &c = n->comps[i];
      uc[i] = UpdComp{begin[i], begin[i + 1], c.rows, c.cols, c.orthonormal};
    }
    CK(upd_group_create(uc, n->params, &n->upd));
  }
  CK(upd_group_step(n->upd, n->params, n->grads, tb.lr, tb.l2coef, mc.data(), n->cfg.max_param_change, s));
  // ScaleBatchnormStats
  if (n->cfg.batchnorm_stats_scale != 1.0f && !n->cfg.cv_update) {  // (BatchNormTestComponents are not scaled)
    const int Hd = n->cfg.hidden_dim, S = n->cfg.prefinal_small_dim;
    ScaleTable tb;
    memset(&tb, 0, sizeof(tb));
    int nb = 0, maxn = 0;
    auto sc = [&](double *st, int D) {
      if (nb < 48) {
        tb.p[nb] = st;
        tb.n[nb] = 1 + 2 * D;
        maxn = std::max(maxn, 1 + 2 * D);
        nb++;
      }
    };
    sc(n->t1_bn_stats, Hd);
    for (auto &L : n->layers) sc(L.bn_stats, Hd);
    for (int h = 0; h < 2; h++) {
      sc(n->head[h].bn1_stats, Hd);
      sc(n->head[h].bn2_stats, S);
    }
    TDNNF_REQUIRE(nb == (int)n->layers.size() + 5, "net_update: too many BatchNorm components for one launch");
    hipLaunchKernelGGL(scale_doubles_kernel, dim3((maxn + 255) / 256, nb), dim3(256), 0, s, tb, (double)n->cfg.batchnorm_stats_scale);
  }
  // ConstrainOrthonormal: each constrained component with probability 1/4 (nnet-utils.cc:1062); the ones chosen this minibatch run
  // together as grouped launches (tall matrices -- none in the recipes' graphs -- keep the per-component path on the transpose)
  std::vector<int> chosen;
  for (int i = 0; i < nc; i++) {
    const CompDesc &c = n->comps[i];
    if (c.orthonormal == 0.f) continue;
    if (::tdnnf::tdnnf_decision((unsigned long long)step, 2 * (unsigned long long)i + 1) % 4 != 0) continue;  // RandInt(0,3) != 0
    if (upd_group_can_ortho(n->upd, i)) {
      chosen.push_back(i);
    } else if (c.rows <= c.cols) {
      CK(tdnnf_constrain_orthonormal(c.orthonormal, n->params + c.begin, c.rows, c.cols, c.cols, n->ws, n->ws_bytes, s));
    } else {  // tall matrix: constrain the transpose (nnet-utils.cc:1068-1075)
      TDNNF_REQUIRE(n->orthoT, "net_update: no transpose buffer for %s", c.name.c_str());
      const long long total = (long long)c.rows * c.cols;
      hipLaunchKernelGGL(transpose_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, n->params + c.begin, c.rows, c.cols, n->orthoT);
      CK(tdnnf_constrain_orthonormal(c.orthonormal, n->orthoT, c.cols, c.rows, c.rows, n->ws, n->ws_bytes, s));
      hipLaunchKernelGGL(transpose_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, n->orthoT, c.cols, c.rows, n->params + c.begin);
    }
  }
  if (!chosen.empty()) CK(upd_group_ortho(n->upd, chosen, s));
  CK(phase_mark(n, 7, s));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

}  // extern "C"
