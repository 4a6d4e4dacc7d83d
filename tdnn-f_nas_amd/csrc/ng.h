// ng.h -- trainer-internal interface of the OnlineNaturalGradient implementation (ng.hip).
//
// The reference preconditions copies of the spliced input X~ (N x Dx) and of the output derivative dY (N x Do)
// and multiplies them (nnet-tdnn-component.cc:592-624, nnet-simple-component.cc:2984-3023).  Both preconditioners
// are projections X^ = X~ (I - Wx^T Wx), dY' = dY (I - Wy^T Wy), so
//     dY'^T X^ = (I - Wy^T Wy) (dY^T X~) (I - Wx^T Wx):
// the preconditioned gradient is the raw gradient with two rank-R corrections of parameter size.  The N-sized
// work that remains is one pass per side for H = X W^T (with ||X||^2 as a by-product) and, on the steps that
// refresh W, a second pass for J = H^T X.  Nothing N x D is copied or written.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"
#include "gemm_f32.h"
#include "tdnnf_hip.h"

// State of one OnlineNaturalGradient object (ng.hip owns it; ng_group.hip reads the device pointers for its grouped launches).
struct tdnnf_ng {
  int rank, Rp, update_period, t, D, Dp, frozen;
  float num_samples_history, alpha, epsilon, delta, rho;
  std::vector<float> d;
  // device state (one allocation)
  float *dev;
  float *W, *WT, *WWT, *wlast, *J, *W1, *Kd, *Ld, *Ad, *coeff, *tmpR, *neg_one, *scale_f;
  double *scal;  // [0] tr(X X^T)  [1] tr(X^ X^^T)
  // pinned host staging
  float *pin;
  float *h_K, *h_L, *h_At, *h_coeff, *h_scale;
  double *h_tr0;
  // scratch of the component-level entry point
  float *scratch;
  size_t scratch_floats;
  // deferred refresh
  int pending, job_done, job_N;
  hipEvent_t ev_job;   // the refresh's K, L and tr(XX^T) have reached the pinned buffers: the pool thread waits for it
  hipEvent_t ev_wait;  // what the pool thread waits for: ev_job, or the event of the grouped launch that staged this object's copies
  int device;
  bool cur_upd;  // the call in flight between ng_stats_main and ng_stats_side
  int cur_N, cur_ones;
  std::vector<float> d_next;
  float rho_next;
  bool must_reorth;
  std::vector<double> sqrt_e1, inv_sqrt_e1;
};

namespace tdnnf {

// One side's data: K row-shifted taps of x scaled by eff[] (null = ones) [+ a column of ones]: D = K*Di + ones.
struct NgInput {
  MatView x;
  tdnnf_tdnn_indexes ix;
  int Di, ones, N;        // N = rows of the spliced matrix
  const float *eff;       // device, K floats or null
  const int *active;      // optional compaction of the non-zero taps (gemm_f32.h)
  int max_active;
};

// H = X~ W^T on the vector ALUs (ng_valu.hip): thread per row, W^T from SGPRs.  Row m of X~ is the concatenation over the nseg taps of
// eff[i] * X[m * row_stride * ldx + seg_off[i] + (0 .. Di)]; WT is D x Rp (k-major), D = nseg * Di [+ 1: the column of ones, whose
// row of W^T comes in as `bias`]; kWtPadRows finite rows must follow its D.  part (optional): part_cap doubles, their sum = ||X~||_F^2 without the ones.
constexpr int kWtPadRows = 64;  // zero rows every W^T buffer holds behind its D rows
struct NgRowdotArgs {
  const float *X;
  long long ldx;        // floats between consecutive rows of X
  int row_stride;       // X rows between consecutive rows of X~
  int nseg, Di;
  long long seg_off[16];  // floats from X to tap i's view (row offset * ldx)
  const float *eff;     // device, nseg floats, or null
  const float *WT;
  int Rp;               // 20, 40 or 80
  const float *bias;    // Rp floats or null
  float *H;
  int ldh, N;
  double *part;
  int part_cap;
};
bool ng_rowdot_ok(const NgRowdotArgs &a);  // shapes / alignments the kernel takes (otherwise: the MFMA rows GEMM)
hipError_t ng_rowdot(const NgRowdotArgs &a, hipStream_t s);

size_t ng_stats_workspace_bytes(int rank, int D, int K, int N);
// Statistics of one PreconditionDirections call: H = X W_t^T into H (N x ld, ld = ng_h_ld()), tr(X X^T),
// tr(X^ X^^T) and the scale on the device; on refresh steps also J, K, L and the (asynchronous) host update that
// produces W_{t+1}.  Increments t.
int ng_stats_step(tdnnf_ng *ng, const NgInput &in, float *H, void *ws, size_t ws_bytes, hipStream_t s);
// The same in two halves, so that the latency-bound R x R half can run on a side stream: ng_stats_main (the caller's
// main stream) installs a pending refresh, forms H and the per-block ||X||^2 partials (`part`:
// rows_gemm_sumsq_blocks(N) doubles) and, on refresh steps, J; ng_stats_side (any stream ordered after it) forms L, the
// traces and the scale, on refresh steps K and the hand-off to the host, and increments t.  ws of ng_stats_main must hold
// ng_stats_workspace_bytes(); ws of ng_stats_side wgrad_workspace_bytes(rank_padded, rank_padded, 1, N).
int ng_stats_main(tdnnf_ng *ng, const NgInput &in, float *H, double *part, void *ws, size_t ws_bytes, hipStream_t s);
int ng_stats_side(tdnnf_ng *ng, const float *H, const double *part, void *ws, size_t ws_bytes, hipStream_t s);
// ng_stats_main in two steps for a caller that launches the H passes of several objects as ONE grouped launch (rows_gemm_group,
// gemm_f32.h): prepare completes a pending refresh on s and hands out the arguments of the pass (the object must be initialised, rank > 0);
// after the caller's launch has been enqueued, finish (on a stream ordered behind it) does the bookkeeping and, on a refresh, J = H^T X.
int ng_stats_main_prepare(tdnnf_ng *ng, const NgInput &in, float *H, double *part, hipStream_t s, RowsGemmArgs *out);
int ng_stats_main_finish(tdnnf_ng *ng, const NgInput &in, const float *H, void *ws, size_t ws_bytes, hipStream_t s);
// The first half with H formed by the caller's own kernel (fused.hip: the BatchNorm/ReLU backward pass produces dY and
// H = dY W^T in one sweep): ng_external_begin completes a pending refresh and hands out W_t (rank_padded x ldw, rows
// >= rank are zero); *W == nullptr means "use ng_stats_main" (first minibatch, W_0 is initialised from the data).  After
// the producer is enqueued on s (H with leading dimension rank_padded, `part` as for ng_stats_main), ng_external_end does
// the bookkeeping and, on refresh steps, J.
// The second half of a refresh (W_{t+1} on the device) is otherwise enqueued by the next call on the object, on that call's stream.
// If the host part has finished already, enqueue it on `s` now (*did = 1); the caller orders `s` before the object's next use.
int ng_finalize_if_ready(tdnnf_ng *ng, hipStream_t s, int *did);
int ng_external_begin(tdnnf_ng *ng, int D, const float **W, int *Rp, int *ldw, hipStream_t s);
int ng_external_end(tdnnf_ng *ng, const NgInput &in, const float *H, void *ws, size_t ws_bytes, hipStream_t s);
int ng_h_ld(const tdnnf_ng *ng);        // leading dimension (padded rank) of H, W W^T, ...
int ng_dim(const tdnnf_ng *ng);         // D (0 before the first call)
const float *ng_scale_dev(const tdnnf_ng *ng);  // device float: sqrt(tr(XX^T)/tr(X^X^^T)) of the last call
const float *ng_w_dev(const tdnnf_ng *ng);      // W_t used by the last call (rank_padded x ldw)
int ng_w_ld(const tdnnf_ng *ng);

// T (Do x Dx, ld = ldT, ldT % 4 == 0, pad columns zero) <- (I - Wy^T Wy) T (I - Wx^T Wx) with the W_t of the last
// ng_stats_step on each side.  tmp: ng_project_tmp_floats(...) floats.
size_t ng_project_tmp_floats(const tdnnf_ng *in, const tdnnf_ng *out, int Do, int ldT);
int ng_project(tdnnf_ng *in, tdnnf_ng *out, float *T, int Do, int Dx, int ldT, float *tmp, hipStream_t s);


// ---------------------------------------------------------------------------------------------------------------------------
// Grouped side chain (ng_group.hip).  What follows the N-sized passes of a component -- L = H^T H and the traces per side, the
// two rank-R projections of the raw gradient, the commit, on a refresh K = J J^T and the hand-off to the host -- is
// latency-bound and independent between components: ~17 small launches each, 36 components per step.  A group runs those
// stages ONCE for all of its components: every stage is one launch over a task table in device memory.
struct NgGroupComp {
  tdnnf_ng *in, *out;       // both initialised (D != 0) and of rank > 0
  float *T;                 // Do x ldT raw gradient of the spliced input [+ bias column at ldw], projected in place
  int Do, Dx, ldT, ldw;     // Dx = ldw + (bsum ? 1 : 0)
  const float *bsum;        // raw bias gradient (Do floats), copied into column ldw of T first; null: no bias
  const float *H_in, *H_out;
  const double *part_in, *part_out;
  int N;                    // rows of H_in / H_out
  float *W_acc, *bias_acc;  // this minibatch's gradient: W_acc (Do x ldw) += a b T[:, :ldw], bias_acc += a b T[:, ldw]
};
struct NgGroup;
int ng_group_create(const std::vector<NgGroupComp> &comps, NgGroup **out);
void ng_group_destroy(NgGroup *g);
// the whole chain for the group's components on stream s (ordered after their ng_stats_main / ng_external_end calls)
int ng_group_run(NgGroup *g, hipStream_t s);
// Second half of the refreshes pending on a fixed set of (initialised) objects -- W_{t+1} = A_t (J + diag(c) W_t), W^T, the last
// column and W W^T on the device -- as five grouped launches on s.  wait = false: only if every host part has finished (*did = 0
// otherwise); wait = true: blocks the host until they have.
struct NgFin;
int ng_fin_create(const std::vector<tdnnf_ng *> &objs, NgFin **out);
void ng_fin_destroy(NgFin *f);
int ng_fin_run(NgFin *f, hipStream_t s, bool wait, int *did);
// hooks into ng.hip for the grouped path
bool ng_updating(const tdnnf_ng *ng);
void ng_pool_push(tdnnf_ng *ng);
void ng_pool_wait(tdnnf_ng *ng);
bool ng_pool_done(tdnnf_ng *ng);
int ng_finalize_one(tdnnf_ng *ng, hipStream_t s);  // the per-object form (also does a rare re-orthogonalisation)

}  // namespace tdnnf
