// net.h -- data structures of the chain trainer (net.hip), shared with the model reader / writer (model_io.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include <map>

#include "common.h"
#include "planes_gemm.h"
#include "tdnnf_hip.h"

namespace tdnnf {
struct RowsGemmGroup;  // gemm_f32.h
struct PlanesSplitGroup;  // planes_gemm.h

struct NgGroup;
struct NgFin;
struct UpdGroup;

struct Grid {
  int t0, step, n;
  int last() const { return t0 + step * (n - 1); }
};

struct CompDesc {
  std::string name;
  long long begin;
  int rows, cols, has_bias;
  float lr_factor, l2, max_change, orthonormal;
  int num_alpha;  // DARTS: K architecture logits between the weights and the bias
  bool plain = false;  // updated without natural gradient (OnehotFunction / ConstantFunction output_ vectors)
  bool updatable = true;  // false: the fixed lda layer (FixedAffineComponent: no learning-rate factor, never a gradient)
  long long size() const { return (long long)rows * cols + num_alpha + (has_bias ? rows : 0); }
};

struct Tdnn {  // one TdnnComponent instance inside the net
  int comp;    // index into comps
  int Di, Do, K;
  int offsets[TDNNF_MAX_OFFSETS];
  Grid in, out;
  // DARTS (TdnnDARTSV3Component): share index, first random draw, coefficient memo [coef(K) | eff(K)] on device
  bool darts;
  int share, draw0;
  float *memo;
  int *active;  // device: [count, tap ids...] of the non-zero taps (uniform-sample mode)
  tdnnf_tdnn_indexes ix;
  int rows_in, rows_out;
};

struct TdnnfLayer {
  int stride, bn;
  int left, right;  // X.linear taps {-left, 0}, X.affine taps {0, right} (= stride unless the config gives layer offsets)
  Tdnn lin, aff;
  Grid gin, gout, glin;
  bool perm;  // affine input needs the rho row order
  int bypass_row0, bypass_rowstep;  // rows of the layer input that line up with the output grid
  // activations (arena)
  float *lin_out, *lin_perm, *relu_out, *noop_out;
  float *bn_memo;
  double *bn_stats, *relu_stats;
  // bottleneck-dimension supernet: component of the C-vector (X.softmax / X.alpha), first random draw, choice
  // probabilities p (C) and the column mask (bn), masked linear output
  int c_arch, arch_draw0;
  float *arch_p, *arch_mask, *lin_masked;
};

}  // namespace tdnnf

struct tdnnf_net {
  tdnnf_net_config cfg;
  std::vector<tdnnf::CompDesc> comps;
  long long num_params;
  float *params, *grads;
  float *paramsT;      // gemm_precision 1 / 2: per component, the transpose of its weight matrix at the same offset (backward-data GEMMs)
  // Pre-split plane operands (planes_gemm.h; gemm_precision 3 "f16x3", or 2 "bf16x6" with option planes): 0 = off, else planes per element.
  // Every matrix that is a GEMM operand has a slot (storage for its row-major and transposed planes and its scale), keyed by the
  // matrix's base pointer; a slot's CONTENT is only what the last split put there -- forward_backward() splits an operand where it is
  // produced and hands the description to the GEMM as a hint.  pw: the weight matrices' planes, re-split at the start of every step.
  int planes_np = 0;
  struct PlaneSlot {
    void *P = nullptr, *PT = nullptr;
    size_t bytesP = 0, bytesPT = 0;
    float *scale = nullptr;
    long long last[5] = {-1, -1, -1, -1, -1};  // (rows, cols, lead, R, layouts) of the last split: the same again leaves the zero rows as they are
  };
  std::map<const float *, PlaneSlot> plane_slots;
  std::vector<tdnnf::PlanesOperand> pw;  // by component
  std::vector<float *> pw_scale;         // their scale records
  void *planes_ws = nullptr;
  double *fro_buf = nullptr;  // FroBoundScope target: per-block norm bounds a BatchNorm finalize launch leaves for the next split (common.h)
  int B, T, Tout;
  // graph
  tdnnf::Grid g_lda, g_feat;
  tdnnf::Tdnn tdnn1;  // affine 220 -> hidden on g_lda
  std::vector<tdnnf::TdnnfLayer> layers;
  int c_lda, c_prefinal_l;
  struct Head {
    int c_affine, c_linear, c_output;
    float *aff_relu, *bn1_out, *lin_out, *bn2_out, *y;
    float *bn1_memo, *bn2_memo;
    double *bn1_stats, *bn2_stats, *relu_stats;
  } head[2];  // 0 chain, 1 xent
  float *xent_logsoftmax;
  // arena
  char *arena;
  size_t arena_bytes;
  float *lda_in, *lda_out, *t1_relu, *t1_bn;
  float *t1_bn_memo;
  double *t1_bn_stats, *t1_relu_stats;
  float *prefinal_l_out;
  float *dA, *dB, *dC, *d_small, *d_small2;  // derivative scratch
  float *d_y, *d_xent;
  float *tapgrad;      // DARTS: unscaled per-tap weight gradients (Do x K*Di) of the component being processed
  double *tapdots;     // DARTS: s_i = <dW_i, W_i>
  const float *draws;  // DARTS: uniform draws of this step (caller-owned device buffer)
  std::vector<tdnnf_ng *> ng_in, ng_out;  // per component (natural gradient)
  float *orthoT;       // transpose of a constrained matrix with more rows than columns (null when there is none)
  // Natural gradient.  Per component the N-sized passes (raw gradient into T, H = X W^T either side, J on a refresh) run on the
  // caller's stream (or the weight-gradient stream) into buffers the component keeps for the step; everything latency-bound that
  // follows (L, traces, the rank-R projections of T, commit; K and the host hand-off on a refresh) runs ONCE PER GRADIENT BUCKET
  // as grouped launches on stream s3 (ng.h, ng_group.hip).  The per-object form of that chain remains for the first minibatch
  // (the preconditioners initialise themselves from it) and for TDNNF_NG_GROUPED=0.
  struct NgComp {
    float *H_in = nullptr, *H_out = nullptr, *T = nullptr, *bsum = nullptr;
    double *part_in = nullptr, *part_out = nullptr;
    int N = 0;
  };
  std::vector<NgComp> ngc;         // by component
  float *ng_bsum_all = nullptr;    // the components' raw bias gradients (NgComp::bsum), one block: zeroed once per step
  size_t ng_bsum_floats = 0;
  float *ngTmp = nullptr;          // per-object chain: projection scratch
  void *ng_side_ws = nullptr;      // per-object chain: workspace of L = H^T H
  size_t ngset_ws_bytes = 0;
  struct NgBucket {
    int key;
    std::vector<int> comps;
    tdnnf::NgGroup *group;
  };
  std::vector<NgBucket> ng_buckets;  // created the first time a bucket closes with every preconditioner initialised
  std::vector<int> ng_cur;           // components enqueued since the last bucket closed
  tdnnf::NgFin *ngfin = nullptr;
  bool ng_grouped = true;
  hipEvent_t ev_ngc = nullptr;       // the components' N-sized passes of a bucket are enqueued
  hipStream_t s3;
  hipEvent_t ev_s3;
  hipEvent_t ev_fin0, ev_fin;  // step start -> s3, and s3's refresh uploads -> the backward pass
  float *s3_scratch;   // split-K scratch of the GEMMs launched on s3
  size_t s3_scratch_bytes;
  // Weight-gradient stream: a component's parameter gradient (and, with natural gradient, the N-sized statistics passes) is
  // independent of the backward-data GEMM that follows it.  At the recipes' minibatch (3 200 rows) neither fills the chip, so
  // they run side by side: param_grad() goes to s4 with a workspace of its own, the caller's stream waits for the one before
  // the last (the buffers a gradient reads are rewritten two components later at the earliest).  Off for minibatches whose
  // GEMMs fill the chip by themselves (TDNNF_WGRAD_STREAM=0|1 forces it).
  bool wg_on;
  hipStream_t s4;
  hipEvent_t ev_pg[4], ev_pg_in;
  int wg_lag = 3;        // the caller's stream waits for the weight gradient of the component wg_lag back (option wgrad_lag: 1 or 3)
  float *dC2 = nullptr, *dS[2] = {nullptr, nullptr};  // wg_lag 3: second buffers for the derivative matrices weight gradients read (net.hip)
  unsigned pg_count;
  void *ws4;
  void *ws2;           // the same for components whose weight gradients go to the denominator's stream (wg_two)
  float *s2_scratch;
  float *s4_scratch;   // split-K scratch of the GEMMs launched on s4
  hipStream_t s5 = nullptr;  // option wgrad_stream 3: a third weight-gradient stream with its workspace and split-K scratch
  void *ws5 = nullptr;
  float *s5_scratch = nullptr;
  // Input-side natural-gradient statistics ahead of the backward pass: H_in = X~ W_x^T (and J = H^T X on refresh steps) of a component
  // needs its forward input and the preconditioner state only, so from the second grouped minibatch on they are launched on s4 as soon as
  // the forward pass is enqueued -- beside the denominator and the xent head, HBM-bound work beside MFMA-bound work -- with the
  // arguments the component's backward call recorded one minibatch earlier (early_rec), checked again when that call comes.
  struct EarlyIn {
    unsigned char xin[256];  // a tdnnf::NgInput (net.h does not see ng.h)
    long long recorded = -1, done = -1;  // fb_count values
  };
  std::vector<EarlyIn> early;
  long long fb_count = 0;
  bool early_on = false, early_any = false;
  bool early_group = false;                    // the early passes as ONE grouped launch on s4 (minibatches with the weight-gradient streams)
  tdnnf::RowsGemmGroup *early_launch = nullptr;  // its device-side task table
  tdnnf::PlanesSplitGroup *wsplit_group = nullptr;  // f16x3: the task table of the grouped split of a step's weight matrices
  hipEvent_t ev_early_in = nullptr, ev_early = nullptr;
  size_t s4_scratch_bytes;
  bool wg_two;  // this step, from the denominator's join on: weight-gradient components alternate between s4 and s2
  float *gtmp;         // this minibatch's gradient; committed into `grads` only when the objective was finite
  // NonlinearComponent::StoreBackpropStats skips a minibatch w.p. 1/4 only "&& oderiv_count_ != 0" (nnet-component-itf.cc:466): whether a
  // ReLU's oderiv_count is non-zero, in the order of stat_blocks() (host copy: set by set_stats / read_model and by every store);
  // a net made by tdnnf_net_create_shared looks at the primary's
  std::vector<char> oderiv_nonzero_own;
  std::vector<char> *oderiv_nonzero = &oderiv_nonzero_own;
  tdnnf::BnSync bn_sync{nullptr, nullptr, nullptr, 1};  // tdnnf_net_set_batchnorm_sync; buf lives in the arena
  // diagnostics (option phase_events): events on the caller's stream at the phase boundaries of the last step
  enum { kPhases = 8 };  // 0 start, 1 forward trunk done, 2 heads forward + xent objective issued, 3 heads' backward done, 4 trunk backward done, 5 side streams joined, 6 update start, 7 update done
  hipEvent_t ev_phase[kPhases] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool phase_rec[kPhases] = {false, false, false, false, false, false, false, false};
  tdnnf::UpdGroup *upd = nullptr;  // the optimizer step's grouped launches (optim_group.h): built at the first update for the parameter buffer in use
  bool den_split = true;  // the denominator's two recursions side by side (option den_split, read when the net's first step sizes the workspace)
  hipStream_t s2;      // the denominator runs here, beside the xent head on the caller's stream
  hipEvent_t ev_fork, ev_den, ev_num;  // ev_num: the numerator recursion (side stream) is done
  hipEvent_t ev_den_rec = nullptr;     // option xent_behind_den: the denominator's two recursions are done
  hipEvent_t ev_comm = nullptr;        // tdnnf_net_allreduce_grads_rccl: the last bucket's collective
  int num_draws;
  bool owns_ng = true;    // false: created by tdnnf_net_create_shared, the preconditioners belong to the primary net
  int dropout_draw0;      // first of the (num_layers + 1) * B * hidden_dim dropout draws (cfg.use_dropout)
  float *dropout_masks;   // (num_layers + 1) x B x hidden_dim: tdnn1, then the tdnnf layers
  float dropout_proportion;
  void *ws;
  size_t ws_bytes;
  void *chain_ws;
  size_t chain_ws_bytes;
  std::vector<std::pair<std::string, tdnnf_mat>> named;
  // Gradient buckets for a data-parallel caller (tdnnf_net_grad_bucket): contiguous ranges of the flat gradient buffer in the
  // order the backward pass finishes them; `ready` is recorded once the range is final in `grads`
  struct GradBucket {
    long long begin, end;
    int close_key;  // closed after: -2 prefinal-l (heads + prefinal-l), l >= 0 tdnnf layer l, -1 tdnn1 (the rest)
    hipEvent_t ready, handoff;
  };
  std::vector<GradBucket> buckets;
  // debugging / parity (tdnnf_net_set_capture): copies of the backward pass's derivative matrices, which live in recycled
  // scratch buffers, kept under names ("tdnnf5.affine.deriv", ...) that tdnnf_net_get_activation serves
  bool capture_on = false;
  std::vector<float *> captured;
};

namespace tdnnf {
// parameter / gradient views of component `comp` inside the flat buffers
inline float *net_W(const tdnnf_net *n, int comp) { return n->params + n->comps[comp].begin; }
inline float *net_alpha(const tdnnf_net *n, int comp) {
  const CompDesc &c = n->comps[comp];
  return c.num_alpha ? n->params + c.begin + (long long)c.rows * c.cols : nullptr;
}
inline float *net_bias(const tdnnf_net *n, int comp) {
  const CompDesc &c = n->comps[comp];
  return c.has_bias ? n->params + c.begin + (long long)c.rows * c.cols + c.num_alpha : nullptr;
}
}  // namespace tdnnf
