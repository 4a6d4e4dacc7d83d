// gemm_f32.h -- host-side interface of the exact-f32 MFMA GEMM kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <vector>

namespace tdnnf {

constexpr int kMaxSeg = 16;

// One K-segment of a "rows" GEMM: C[m][n] (+)= sum_s coef[s] * sum_k A_s[m][k] * B_s[k][n].
struct GemmSeg {
  long long a_off;  // element offset added to A for this segment
  long long b_off;  // element offset added to B for this segment
  int klen;         // reduction length of this segment
  int m_lo, m_hi;   // output rows for which A_s is defined (others contribute 0)
};

struct RowsGemmArgs {
  const float *A;
  long long lda;  // elements between consecutive A rows (already multiplied by any row step)
  const float *B;
  long long ldb;  // B_KC: B[n*ldb + k]; else B[k*ldb + n]
  float *C;
  long long ldc;  // elements between consecutive C rows (already multiplied by any row step)
  int M, N;
  const float *bias;  // N floats (init_mode 1)
  const float *coef;  // nseg floats or nullptr (= ones); a segment with coef == 0 is skipped
  int init_mode;      // 0: C += acc, 1: C = bias + acc, 2: C = acc
  int relu;           // 1: C = max(C, 0) after everything else
  int c_vec;          // set by rows_gemm(): C rows (and bias) allow 16-byte accesses
  // optional addend fused into the epilogue: C[m][n] += add_scale * add[(m - add_lo) * ldadd + n] for add_lo <= m < add_hi
  const float *add;
  long long ldadd;
  float add_scale;
  int add_lo, add_hi;
  // split-K over the concatenated reduction range (set by rows_gemm() for the tail tiles of a launch whose
  // tile count does not fill whole rounds of resident blocks): block (tile, sp) reduces K-elements
  // [sp*kchunk, (sp+1)*kchunk) and stores its raw partial tile to partial[sp][m][n] (ld = N rounded up to 4).
  int ksplit;       // 0/1: no split
  int kchunk;       // multiple of the kernel's K step
  float *partial;
  // optional: sumsq[block] = sum over this block's rows of sum_s coef[s]^2 * |A_s row|^2 (the Frobenius norm of the
  // scaled, spliced A operand, a by-product of staging it).  Needs N to fit one column tile (every A element is
  // then staged exactly once): rows_gemm() fails otherwise.  One double per 128-row block: rows_gemm_sumsq_blocks(M).
  double *sumsq;
  int sumsq_cap;  // set by rows_gemm(): rows_gemm_sumsq_blocks(M)
  // optional: column sums and sums of squares of the STORED C values (after bias / addend / ReLU) -- the BatchNorm statistics
  // of the layer output as a by-product of writing it.  One partial row per 128-row tile (and per chunk of the rows a split-K
  // tail launch finishes): colstats[c * N + n] sums, colstats[(rows + c) * N + n] sums of squares, rows = *colstats_rows as
  // set by rows_gemm() (0: this launch cannot do it -- not the exact-f32 128-wide tile -- and nothing was written).
  // Capacity needed: 2 * N * rows_gemm_colstats_cap(M) floats.
  float *colstats;
  int *colstats_rows;  // host, out
  int colstats_stride;  // set by rows_gemm(): the `rows` above
  // 0: exact f32 MFMA (v_mfma_f32_32x32x2_f32).  1: split-bf16 (three v_mfma_f32_32x32x16_bf16 per 16 k, products accurate
  // to ~2^-16 relative, f32 accumulation); needs a k-contiguous B and 16-byte alignment, otherwise the f32 kernel runs.
  int prec;
  int serial_epilogue;  // 1: the one-segment-at-a-time epilogue for every tile (rows_gemm() sets 0)
  int alt_seg_order;    // set by rows_gemm(): odd row tiles visit the K segments in reverse order (taps = row shifts of one matrix, gemm_f32.hip)
  int nseg;
  GemmSeg seg[kMaxSeg];
};

// Event-timing class of the launches made while one of these is alive (tdnnf_profile_*: 0 rows_gemm 128x128,
// 1 rows_gemm 128x160, 2 wgrad, 3 natural-gradient skinny GEMMs).
struct ProfClassOverride {
  int prev;
  explicit ProfClassOverride(int cls);
  ~ProfClassOverride();
};

// Event timing of an HBM-bound pass while tdnnf_profile_enable is on: class 4 bn_apply_bypass, 5 bn_relu_bwd (both stages),
// 6 denominator (forward + backward recursions), 7 planes_split; `bytes` = the pass's algorithmic HBM bytes (every element once).
struct ProfHbmRange {
  void *c;
  hipStream_t s;
  ProfHbmRange(int cls, double bytes, hipStream_t stream);
  ~ProfHbmRange();
};

// Event pair + algorithmic work of ONE launch made outside gemm_f32.hip that belongs to a GEMM class (0 .. 3; ng_valu.hip: class 3).
struct ProfGemmRange {
  void *c;
  hipStream_t s;
  ProfGemmRange(int cls, double flops, double bytes, hipStream_t stream);
  ~ProfGemmRange();
};

// Scales the algorithmic FLOPs recorded for the launches made while alive: the host cannot see device-side tap
// coefficients, so a caller that knows only `active` of K taps are non-zero (DARTS uniform-sample mode) says so.
struct ProfFlopsScale {
  double prev;
  explicit ProfFlopsScale(double f);
  ~ProfFlopsScale();
};

// rows_gemm() keeps one process-wide scratch buffer for its split-K partial tiles, which is only safe for launches that
// are ordered on one stream.  Launches made on another stream while one of these is alive use `buf` instead.
struct SplitKScratchOverride {
  float *prev_buf;
  size_t prev_bytes;
  SplitKScratchOverride(float *buf, size_t bytes);
  ~SplitKScratchOverride();
};

// Default arithmetic of the GEMMs launched while one of these is alive (RowsGemmArgs::prec / WgradArgs::prec == 0 means
// "the default"): 0 exact f32 MFMA, 1 split-bf16, 2 exact f32 even inside a split-bf16 scope (the natural-gradient
// statistics: their eigen-decomposition amplifies operand errors).  The chain trainer wraps its forward / backward pass in one.
struct GemmPrecisionScope {
  int prev;
  explicit GemmPrecisionScope(int prec);
  ~GemmPrecisionScope();
};
// Transposed copies of weight matrices for the split-bf16 backward-data GEMMs (their B operand must be k-contiguous):
// while alive, a weight pointer W inside [w_base, w_base + n) has its transpose (cols x rows, dense) at wt_base + (W - w_base).
struct TransposedWeightsScope {
  const float *prev_w, *prev_wt;
  long long prev_n;
  TransposedWeightsScope(const float *w_base, const float *wt_base, long long n);
  ~TransposedWeightsScope();
};
const float *transposed_weights(const float *W);  // null when there is none (or split-bf16 is not the default)

// capacity of the p.sumsq array for M rows: one entry per 128-row block, or per (block, K-slice) when rows_gemm() splits K
// over idle CUs for a launch of few blocks (then blocks x slices <= the chip's resident blocks <= 1024); the kernel zeroes
// the entries it does not write, consumers sum all of them
inline int rows_gemm_colstats_cap(int M) { return (M + 63) / 64 + 64; }  // (row tiles of 64 for launches that do not fill the chip with 128)
inline int rows_gemm_sumsq_blocks(int M) { return (M + 127) / 128 > 1024 ? (M + 127) / 128 : 1024; }

// b_kcontig: B element (k, n) at B[n*ldb + k] (true) or B[k*ldb + n] (false).
hipError_t rows_gemm(const RowsGemmArgs &args, bool b_kcontig, hipStream_t stream);

// Several statistics passes (sumsq set, N <= 32, k-contiguous B, 16-byte aligned operands: rows_gemm_group_ok) as ONE launch: task i runs
// exactly the blocks a launch of its own without K split would.  *cache keeps the device-side task table between calls (null at first;
// rows_gemm_group_destroy at the end); the table is uploaded again only when the calls differ from the last ones.
struct RowsGemmGroup;
bool rows_gemm_group_ok(const RowsGemmArgs &a);
hipError_t rows_gemm_group(const std::vector<RowsGemmArgs> &calls, RowsGemmGroup **cache, hipStream_t stream);
void rows_gemm_group_destroy(RowsGemmGroup *g);

// Weight gradient: G[o][i*Di + d] (+)= scale * coef[i] * sum_r dY[r][o] * X[(row_off[i] + r*row_stride)][d]
struct WgradArgs {
  const float *dY;
  long long lddy;
  const float *X;
  long long ldx;
  int Do, Di, K, N;
  int row_stride;
  int row_offsets[kMaxSeg];
  const float *coef;  // K floats or nullptr
  float scale;        // learning rate
  float *G;           // Do x (K*Di), ld = ldg
  long long ldg;
  int accumulate;     // 1: G += ..., 0: G = ...
  float *bias_acc;    // optional: bias_acc[o] += scale * sum_r dY[r][o]
  // optional compaction of the taps with a non-zero coefficient (DARTS uniform-sample mode: <= 2 of K taps):
  // active[0] = count n, active[1..n] = tap ids (device memory); max_active bounds n on the host.
  const int *active;
  int max_active;
  int prec;  // 0: exact f32 MFMA, 1: split-bf16 (as RowsGemmArgs::prec)
  int xcd_order;  // set by wgrad(): XCD-aware block order (the taps of a tile side by side on one XCD)
};
size_t wgrad_workspace_bytes(int Do, int Di, int K, int N);  // valid for any max_active <= K
hipError_t wgrad(const WgradArgs &args, void *workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace tdnnf
