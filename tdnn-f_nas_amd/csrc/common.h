// common.h -- shared helpers of libtdnnf_hip (error reporting, matrix views).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "tdnnf_hip.h"

namespace tdnnf {

struct MatView {
  float *data;
  int rows, cols, stride;
};
inline MatView view(const tdnnf_mat *m) { return MatView{m->data, m->rows, m->cols, m->stride}; }

void set_error(const char *fmt, ...);
int hip_status(hipError_t e, const char *what);

inline bool mat_ok(const tdnnf_mat *m) {
  return m && m->rows >= 0 && m->cols >= 0 && m->stride >= m->cols && (m->data || m->rows * m->cols == 0);
}
inline bool same_dim(const tdnnf_mat *a, const tdnnf_mat *b) { return a->rows == b->rows && a->cols == b->cols; }

inline bool vec4_ok(const MatView &m) {
  return (reinterpret_cast<uintptr_t>(m.data) & 15) == 0 && m.stride % 4 == 0 && m.cols % 4 == 0;
}

#define TDNNF_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::tdnnf::set_error(__VA_ARGS__);      \
      return TDNNF_EINVAL;                  \
    }                                       \
  } while (0)

#define TDNNF_HIP(expr)                                         \
  do {                                                          \
    hipError_t e__ = (expr);                                    \
    if (e__ != hipSuccess) return ::tdnnf::hip_status(e__, #expr); \
  } while (0)

#define TDNNF_LAUNCH_CHECK(name) TDNNF_HIP(hipGetLastError())

// Tuning options (tdnnf_set_option / tdnnf_get_option, include/tdnnf_hip.h): process-wide integers that select between code paths
// which are all parity-tested; the library reads no environment variable for them.
struct Options {
  int ng_grouped = 1;     // natural gradient: 1 the side chain of a gradient bucket as grouped launches, 0 per object (read by tdnnf_net_create)
  int ng_fuse = 1;        // output-side statistic H = dY Wy^T: 0 by its own GEMM, 1 inside the BatchNorm / ReLU backward sweep when that pays, 2 always
  int ng_early_in = 1;    // input-side statistics ahead of the backward pass: 0 never, 1 for minibatches without the weight-gradient streams, 2 always (a launch per component), 3 always, with the weight-gradient streams as ONE grouped launch (read by tdnnf_net_create)
  int wgrad_stream = -1;  // parameter gradients on a stream of their own: -1 by minibatch size, 0 off, 1 on (read by tdnnf_net_create)
  int gemm_ring = 1;      // the persistent LDS-DMA-ring form of the rows GEMM where it applies
  int planes = 1;         // gemm_precision 2: the pre-split bf16-plane GEMMs where they apply (0: the in-kernel split everywhere)
  int den_mw_test_abort = 0;  // tests: raise the multi-workgroup denominator's abort word before its launch (the one-workgroup kernels must then redo the minibatch)
  int planes_group = 1;   // f16x3 trainer: the plain components' weight matrices split by ONE grouped pair of launches per step (planes_split_group)
  int planes_check_bound = 0;  // tests: after every split that took its scale from a norm bound, measure the norm and count violations (tdnnf_planes_bound_checks)
  int wgrad_lag = 3;      // trainer, weight-gradient stream on: the caller's stream runs 3 (default) or 1 component(s) ahead of the gradients (read by tdnnf_net_create)
  int wgrad_on_caller = 0;  // trainer: the xent head's weight gradients on the caller's stream when the early statistics occupy the gradient stream
  int splitk_partial_round = 1;  // rows GEMM: split K when the tiles fill only part of one round of resident blocks (gemm_f32.hip launch_rows_balanced)
  int reverse_passes = 0;  // HBM-bound passes walk their matrix from the last rows to the first: bit 0 bn_apply_bypass, bit 1 bn_relu_bwd's apply pass
  int gemm_alt_taps = 1;  // rows GEMM with two taps of one matrix: odd row tiles visit the taps in reverse order, so that both readers of a row block fetch it together
  int splitk_per_cu = 2;  // rows GEMM, few tiles and a long reduction: K slices per CU (2: fill every resident slot; 1: half the partial tiles)
  int wgrad_small = 0;    // weight gradients of launches with at most this many rows on 64 x 64 tiles (0: off)
  int ng_bk = 0;          // natural-gradient statistics passes H = X W^T, longer K steps: bit 0 = 64 instead of 32 for rank <= 32, bit 1 = 32 instead of 16 for rank <= 96
  int ng_early_fork = 1;  // trainer, minibatches without weight-gradient streams: the trunk components' early input statistics start where the trunk's forward pass ends (beside the denominator's recursions) instead of behind the xent head's backward pass
  int xent_behind_den = -1;  // trainer: the xent head's forward pass waits for the denominator's two recursions (their 1024-thread workgroups pin half the CUs): -1 minibatches without weight-gradient streams, 0 never, 1 always
  int ng_pform = 1;       // natural-gradient statistics of a component whose K taps are row shifts of one matrix (the .linear inputs): one pass over the matrix for all taps' products (ng.hip pform_pass) instead of K
  int ng_valu = 0;        // natural-gradient statistics passes H = X W^T on the vector ALUs (ng_valu.hip) where the rank is 20 / 40 / 80 (measured: no gain, docs/experiments.md r5-n); 0: the MFMA rows GEMM
  int ng_diag_skip = 0;   // diagnostics (timing only, results wrong): skip the statistics passes H = X W^T -- bit 0 two-tap inputs >= 1024 wide, bit 1 every other
  int phase_events = 0;   // diagnostics: the trainer records an event on the caller's stream at every phase boundary of a step (tdnnf_net_phase_times)
  int den_split = -1;     // trainer: the denominator's two recursions side by side (then the occupancies of all frames at once): -1 by minibatch size, 0 / 1
};
Options &options();

// Named ranges for profilers (rocprofv3 --marker-trace; the reference's NVTX_RANGE at nnet-normalize-component.cc:185,476 is
// the same idea): roctxRangePush / Pop from librocprofiler-sdk-roctx.so (or libroctx64.so), resolved with dlopen on first use so
// that the library has no link-time dependency on a profiler; without the library (or with TDNNF_ROCTX=0) a range is a no-op.
struct TraceRange {
  explicit TraceRange(const char *name);
  ~TraceRange();
  bool on;
};

// CuMatrix::ApplyFloor semantics: `if (x < floor) x = floor`, so a NaN stays a NaN (fmaxf would swallow it and a
// diverged minibatch would no longer be detected by the objective's finiteness check).
__host__ __device__ inline float floor_keep_nan(float x, float f) { return x < f ? f : x; }

inline int grid_for(long long work, int block, int cap = 2048) {
  long long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// Reproducible stand-in for the reference's RandInt()/RandUniform() control decisions (orthonormal
// schedule nnet-utils.cc:1062, ReLU stats / self-repair coin flips nnet-simple-component.cc:1017,1084):
// splitmix64 of (step, k).  Tests restate it in Python.
inline unsigned long long tdnnf_decision(unsigned long long step, unsigned long long k) {
  unsigned long long z = step * 0x9E3779B97F4A7C15ULL + k * 0xBF58476D1CE4E5B9ULL + 0x94D049BB133111EBULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (z ^ (z >> 31)) >> 8;
}

// Second stage of the column reductions: a 256-thread block owns 8 columns, its 32 lanes per column walk the partial rows
// four requests deep, then lane 0 adds the lanes' sums in a fixed order (deterministic).  The partial rows are few MB at most,
// the stage is pure latency: with 4 lanes x 24 blocks it took 45-300 us per call, ~5 ms per training step.  Rounds 2-4 ran it as
// 1024-thread blocks of 32 columns: beside another stream's kernels such a block waits until one CU has sixteen free wave slots at once
// (the BatchNorm finalize of the xent head beside the denominator and the statistics passes: 9 us alone, 234 us on average, 4.3 ms
// at worst in the round-5 trace); four-wave blocks fit wherever anything fits, and there are four times as many of them.
constexpr int kFinCols = 8, kFinLanes = 32, kFinThreads = kFinCols * kFinLanes;
inline unsigned finalize_grid(int D) { return (unsigned)((D + kFinCols - 1) / kFinCols); }
#ifdef __HIPCC__
// q[k] = sum over c < chunks of partial[((long long)k * qstride_rows + c) * D + d] for the calling thread's column d;
// valid afterwards in the threads with (threadIdx.x >> 5) == 0.  red: NQ * kFinLanes * (kFinCols + 1) elements of Acc.
template <int NQ, class Acc>
__device__ __forceinline__ void finalize_sums(const float *partial, int chunks, long long qstride_rows, int D, int nq, Acc (&q)[NQ], Acc *red) {
  const int tc = threadIdx.x & (kFinCols - 1), lane = threadIdx.x / kFinCols, d = blockIdx.x * kFinCols + tc;
#pragma unroll
  for (int k = 0; k < NQ; k++) q[k] = 0;
  if (d < D) {
    // (the NQ quantities side by side: their loads of a round are issued together -- one after the other, five quantities took 15 us
    // where two took 5)
    const float *p = partial + d;
    const long long qs = qstride_rows * D;
    int c = lane;
    for (; c + 3 * kFinLanes < chunks; c += 4 * kFinLanes) {
      float v[NQ][4];
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        if (k < nq) {
          const float *pk = p + (long long)k * qs;
          v[k][0] = pk[(long long)c * D]; v[k][1] = pk[(long long)(c + kFinLanes) * D];
          v[k][2] = pk[(long long)(c + 2 * kFinLanes) * D]; v[k][3] = pk[(long long)(c + 3 * kFinLanes) * D];
        }
      }
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        if (k < nq) { q[k] += v[k][0]; q[k] += v[k][1]; q[k] += v[k][2]; q[k] += v[k][3]; }
      }
    }
    for (; c < chunks; c += kFinLanes) {
      float v[NQ];
#pragma unroll
      for (int k = 0; k < NQ; k++)
        if (k < nq) v[k] = p[(long long)k * qs + (long long)c * D];
#pragma unroll
      for (int k = 0; k < NQ; k++)
        if (k < nq) q[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NQ; k++) red[(k * kFinLanes + lane) * (kFinCols + 1) + tc] = q[k];
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      Acc s = 0;
      for (int l = 0; l < kFinLanes; l++) s += red[(k * kFinLanes + l) * (kFinCols + 1) + tc];
      q[k] = s;
    }
  }
}
#endif

// Synchronised BatchNorm (data-parallel training): while one of these is installed, every train-mode BatchNorm of the calling thread
// all-reduces its column sums over the ranks -- forward [sum x, sum x^2], backward [sum z dz, sum dz, sum dz^2] (2 D / 3 D doubles
// in `buf`) -- through the caller's collective `fn(ctx, buf, count, stream)` before it forms mean / scale and the backward terms
// with the GLOBAL row count, so that a sharded minibatch normalises exactly as the whole one does
// (/root/reference/src/nnet3/nnet-normalize-component.cc:433-445 takes its statistics over all rows of the minibatch).
struct BnSync {
  int (*fn)(void *ctx, double *buf, long long count, tdnnf_stream stream);
  void *ctx;
  double *buf;      // device, >= 5 * max D doubles (the fused BatchNorm / ReLU backward stages five column sums, three are reduced)
  int world;
};
BnSync *bn_sync_current();
struct BnSyncScope {
  BnSync *prev;
  explicit BnSyncScope(BnSync *b);
  ~BnSyncScope();
};

// While one of these is installed, the BatchNorm finalize launches of the calling thread -- the forward statistics (memo rows 0-2) and the
// backward terms of the fused BatchNorm / ReLU sweep -- also write, per block of kFinCols columns, an UPPER BOUND of the squared Frobenius
// norm of what the pass behind them produces (forward: z = (x - mean) scale, whose column sums of squares are N scale^2 var exactly;
// backward: the ReLU's input derivative, bounded by the column sums of squares of its output derivative, which the finalize forms
// anyway, plus the self-repair term).  planes_split takes its scale from such a bound instead of a pass over the matrix.
// buf: >= finalize_grid(D) doubles; *blocks receives how many were written (0: none -- test-mode BatchNorm has no such bound).
struct FroBoundScope {
  double *prev_buf;
  int *prev_blocks;
  FroBoundScope(double *buf, int *blocks);
  ~FroBoundScope();
};
double *fro_bound_buf();
int *fro_bound_blocks();

// two-stage deterministic column reduction (colreduce.hip)
struct ColReducePlan {
  int chunks, rows_per_chunk;
};
ColReducePlan colreduce_plan(int rows, int cols);
size_t colreduce_bytes(int rows, int cols);
// partial[q][chunk][col] for q < nq; kind: 0 = (sum a), 1 = (sum a, sum a*a), 2 = (sum a*b, sum b), 3 = (sum a, sum a>0)
hipError_t colreduce_partial(int kind, MatView a, MatView b, float *partial, hipStream_t s);
// the same with the chunking given: partial[c * cols + col] and, second quantity, partial[(sq_row_offset + c) * cols + col], c < chunks
hipError_t colreduce_partial_into(int kind, MatView a, MatView b, int chunks, int rows_per_chunk, int sq_row_offset, float *partial, hipStream_t s);
// BatchNorm forward: memo rows 0-2 from column partial sums / sums of squares laid out as above with sq_row_offset == chunks
// store_stats (optional): BatchNormComponent::StoreStats ([count, sum[D], sumsq[D]] doubles += this minibatch) in the same launch
hipError_t batchnorm_stats_from_partials(const float *partial, int chunks, int rows, int cols, float epsilon, float target_rms, float *memo, hipStream_t s,
                                         double *store_stats = nullptr);
hipError_t batchnorm_stats(MatView a, float epsilon, float target_rms, float *memo, void *ws, hipStream_t s, double *store_stats = nullptr);
hipError_t colsum_add(MatView a, float scale, float *acc, void *ws, hipStream_t s);  // ws: colreduce_bytes(rows, cols)

// the three separately launchable parts of the chain objective (chain.hip)
bool log_softmax_propagate_with_aux(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_mat *aux, float aux_scale, hipStream_t s);  // elementwise.hip
float chain_supervision_weight(const tdnnf_supervision *sp);
// beside_other_work: the caller runs other kernels next to the denominator (the trainer: the xent head), so the persistent form keeps
// its one-kernel backward pass instead of running the two recursions side by side on a further stream
size_t chain_split_region_bytes(const tdnnf_den_graph *g, int B, int T);
// ev_recursions (optional): recorded on s once both recursions of the side-by-side form are done, in front of the occupancies; *ev_recorded says
// whether this call's form did
int chain_den(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float leaky, tdnnf_mat *deriv, void *ws, hipStream_t s,
              bool beside_other_work = false, hipStream_t caller_aux = nullptr, hipEvent_t ev_recursions = nullptr, bool *ev_recorded = nullptr);
int chain_num_recursion(const tdnnf_supervision *sp, const tdnnf_den_graph *g, const tdnnf_mat *y, void *ws, hipStream_t s);
int chain_num_xent(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, const tdnnf_mat *xent_output, float xent_regularize,
                   tdnnf_mat *xent_deriv, void *ws, hipStream_t s, bool xent_deriv_initialised = false);
int chain_num(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, const tdnnf_mat *xent_output,
              float xent_regularize, tdnnf_mat *xent_deriv, void *ws, hipStream_t s, bool xent_deriv_initialised = false);
int chain_finish(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float l2_regularize, double *results,
                 tdnnf_mat *deriv, tdnnf_mat *xent_deriv, void *ws, hipStream_t s);

// trainer-internal variants of the TDNN entry points (abi_tdnn.hip)
int tdnn_propagate_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in, const float *W, int ldw, int Do, int Di,
                        const float *bias, const float *eff_coef, int init_mode, int relu, tdnnf_mat *out, tdnnf_stream stream,
                        float *colstats = nullptr, int *colstats_rows = nullptr);
int tdnn_update_simple_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do, int Di,
                            const float *eff_coef, float lr, float *W_acc, int ldw, float *bias_acc, void *ws, size_t ws_bytes,
                            const int *active_dev, int max_active, tdnnf_stream stream, bool overwrite = false);
int tdnn_backprop_data_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *out_deriv, const float *W, int ldw, int Do, int Di,
                            const float *eff_coef, int overwrite, const tdnnf_mat *add, float add_scale, int add_lo,
                            tdnnf_mat *in_deriv, tdnnf_stream stream);

}  // namespace tdnnf
