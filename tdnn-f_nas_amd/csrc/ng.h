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

#include "common.h"
#include "tdnnf_hip.h"

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

}  // namespace tdnnf
