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
hipError_t batchnorm_stats_from_partials(const float *partial, int chunks, int rows, int cols, float epsilon, float target_rms, float *memo, hipStream_t s);
hipError_t batchnorm_stats(MatView a, float epsilon, float target_rms, float *memo, void *ws, hipStream_t s);
hipError_t colsum_add(MatView a, float scale, float *acc, void *ws, hipStream_t s);  // ws: colreduce_bytes(rows, cols)

// the three separately launchable parts of the chain objective (chain.hip)
int chain_den(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float leaky, tdnnf_mat *deriv, void *ws, hipStream_t s);
int chain_num(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, const tdnnf_mat *xent_output,
              float xent_regularize, tdnnf_mat *xent_deriv, void *ws, hipStream_t s);
int chain_finish(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float l2_regularize, double *results,
                 tdnnf_mat *deriv, tdnnf_mat *xent_deriv, void *ws, hipStream_t s);

// trainer-internal variants of the TDNN entry points (abi_tdnn.hip)
int tdnn_propagate_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in, const float *W, int ldw, int Do, int Di,
                        const float *bias, const float *eff_coef, int init_mode, int relu, tdnnf_mat *out, tdnnf_stream stream,
                        float *colstats = nullptr, int *colstats_rows = nullptr);
int tdnn_update_simple_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int Do, int Di,
                            const float *eff_coef, float lr, float *W_acc, int ldw, float *bias_acc, void *ws, size_t ws_bytes,
                            const int *active_dev, int max_active, tdnnf_stream stream);
int tdnn_backprop_data_impl(const tdnnf_tdnn_indexes *ix, const tdnnf_mat *out_deriv, const float *W, int ldw, int Do, int Di,
                            const float *eff_coef, int overwrite, const tdnnf_mat *add, float add_scale, int add_lo,
                            tdnnf_mat *in_deriv, tdnnf_stream stream);

}  // namespace tdnnf
