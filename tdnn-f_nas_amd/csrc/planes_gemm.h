// planes_gemm.h -- host interface of the pre-split plane GEMMs (planes_gemm.hip): f32-equivalent products on the 16-bit matrix cores
// from operands split ONCE into 16-bit planes in HBM.
//
//   np = 3  "bf16x6": x = p0 + p1 + p2 (three bf16 planes, 24 mantissa bits), the six products p_i q_j with i + j <= 2.
//   np = 2  "f16x3" : x s = h + l (two f16 planes of the operand scaled by a power of two s chosen from its Frobenius norm so that
//                     no element can overflow), the three products h h', h l', l h'; the result is multiplied by 1 / (s s').
// Both accumulate in f32, smallest products first.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <vector>

#include "common.h"

namespace tdnnf {

// One K-segment (a tap of a TdnnComponent: a row-shifted view of the A matrix against a column block of the weights).
struct PlanesSeg {
  long long a_row;  // first row of the segment's view in the A plane buffer (lead rows included): tile row m reads a_row + m
  long long b_row;  // first row of the segment's view in the B plane buffer: output column n reads b_row + n
  int a_kb0;        // first K block (of 16) of A
  int b_kb0;        // first K block of B
  int nkb;          // K blocks in the segment
};

struct PlanesGemmArgs {
  int np;         // 3: bf16 triples, 2: scaled f16 pairs
  const void *A;  // P16 planes of the A operand (planes_split)
  long long RA;   // rows per (K block, plane) chunk of A: lead + rows + tail
  const void *B;  // P16 planes of the B operand, row n = output column n, k contiguous
  long long RB;
  const float *scale_a, *scale_b;  // np == 2: device pointers to the operands' [s, 1 / s] pairs (planes_split); null = 1
  float *C;
  long long ldc;
  int M, N;
  const float *bias;  // init_mode 1
  int init_mode;      // 0: C += acc, 1: C = bias + acc, 2: C = acc
  int relu;
  const float *add;   // optional addend: C[m][n] += add_scale * add[(m - add_lo) * ldadd + n] for add_lo <= m < add_hi
  long long ldadd;
  float add_scale;
  int add_lo, add_hi;
  int nseg;
  PlanesSeg seg[32];
  // optional (device, nseg floats; in tap mode ntap floats): a segment / tap whose coefficient is zero is skipped (the coefficients
  // themselves are folded into the B planes or applied by the caller's reduce: this is only the skip)
  const float *skip_coef;
  // optional (plain launches: ntap <= 1, ksplit <= 1): column sums and sums of squares of the STORED C values, one partial row per row tile --
  // colstats[tile_m * N + n] and colstats[(colstats_stride + tile_m) * N + n] (the BatchNorm statistics of a layer output as a by-product
  // of writing it, as RowsGemmArgs::colstats)
  float *colstats;
  int colstats_stride;
  // != 0: A holds the ROW-MAJOR planes of a matrix whose COLUMNS are the tile rows and whose ROWS are the reduction index (a weight
  // gradient's big operand, read with transposing LDS loads): RA = that buffer's rows, seg.a_row = the first matrix row of the K range
  // (a multiple of 16, lead rows included), seg.a_kb0 / tap_a_kb count K steps of 16 rows; the buffer must hold whole 256-column tiles
  // (K blocks padded to a multiple of 16).
  int a_rows_as_k;
  // Tap mode (weight gradients, nseg == 1): `ntap` products per output tile position; product t runs with the A operand advanced by
  // tap_a_kb[t] K blocks and the B operand by tap_b_kb[t], and lands tap_off_c * t floats further in C (tap_off_p * t in a partial slab).
  // ntap <= 1: off.
  int ntap;
  int tap_a_kb[16], tap_b_kb[16];
  long long tap_off_c, tap_off_p;
  // Split-K (ksplit > 1): block (tile, sp) multiplies K blocks [sp * kb_per_split, (sp + 1) * kb_per_split) of the concatenated range and
  // stores its raw accumulators (no scale, bias, ...) to partial[sp * partial_stride + m * ldp_m + n * ldp_n]; the caller reduces.
  int ksplit, kb_per_split;
  float *partial;
  long long partial_stride, ldp_m, ldp_n;
  int alt_seg_order;  // odd row tiles visit the K segments in reverse order (two taps = row shifts of one matrix; gemm_f32.hip rows_gemm_kernel)
};

// P16 layout of an R x C matrix:  e16 P[kb][plane][row][16],  kb = c / 16, plane 0..np-1, row 0..R-1, R = lead + rows + tail
// (lead / tail rows are zeros); the two 16-byte halves of a 32-byte row record are swapped when bit 3 of the row is set.
// The TRANSPOSED planes of the same matrix (the operand of a product that reduces over the matrix's rows: weight gradients) are the
// P16 planes of its transpose: "row" = column c of the matrix (Rt = cols + tail_t of them), k = row r.
size_t planes_bytes(int np, long long rows_total, long long k_blocks);
inline long long planes_kblocks(int cols) { return (cols + 15) / 16; }
inline long long planes_t_kblocks(int rows) { return ((rows + 63) / 64) * 4; }  // the split kernel writes whole 64-row tiles
// a chunk stride (rows x 32 bytes) that is a multiple of 8 KB puts every plane and K block on the same memory channels (2.7 x slower):
// the row count to allocate for `rows` data rows + `pad` zero rows
inline long long planes_rows_padded(long long rows) { return rows % 256 == 0 ? rows + 8 : rows; }

struct PlanesSplitArgs {
  int np;
  MatView x;
  // row-major planes (k = column), null to skip
  void *P;
  int lead;
  long long R;
  // transposed planes (k = row), null to skip
  void *PT;
  long long Rt;
  // np == 2: [s, 1 / s, ||X||_F or its bound] (device, 3 floats) receives the scale record; `sumsq_ws` (device, planes_sumsq_ws_bytes()) is scratch of the norm pass
  float *scale;
  void *sumsq_ws;
  // optional: element (r, c) is multiplied by col_coef[c / col_coef_period] (device) before it is split -- the tap coefficients of a
  // TdnnDARTSV3Component folded into its weight planes (|coef| <= 1: the unscaled norm stays a valid bound for the scale)
  const float *col_coef = nullptr;
  int col_coef_period = 0;
  // np == 2, optional: the scale from an UPPER BOUND of the matrix's Frobenius norm instead of a pass over it (common.h FroBoundScope):
  //   ||X||_F <= fro_mul * sqrt(sum of fro2_bound[0 .. fro2_blocks)) + add_coef * add_rec[2]
  // (add_rec: the scale record [s, 1 / s, norm bound] of a matrix added into this one with coefficient add_coef: the bypass sum).
  const double *fro2_bound = nullptr;
  int fro2_blocks = 0;
  float fro_mul = 1.0f, add_coef = 0.0f;
  const float *add_rec = nullptr;
  // the zero rows around the matrix are in place already (the buffers were last split with exactly these dimensions, or were zeroed and
  // never held another shape): skip the pad launches
  bool pads_done = false;
};
size_t planes_sumsq_ws_bytes();
// the scale record [s, 1 / s, bound] of a matrix of `numel` elements from a norm bound alone (PlanesSplitArgs::fro2_bound): for a producer that
// writes the planes itself (fused.h PlanesSink)
// tests (option planes_check_bound): measure ||x||_F and compare it with the bound in rec[2]; counts into tdnnf_planes_bound_checks
hipError_t planes_check_bound(MatView x, const float *rec, void *sumsq_ws, hipStream_t s);
hipError_t planes_scale_bound(const double *fro2_bound, int blocks, double numel, float mul, float add_coef, const float *add_rec, float *rec, hipStream_t s);
hipError_t planes_split(const PlanesSplitArgs &a, hipStream_t s);
// The splits of several small f16x3 matrices whose zero rows are in place (planes_split_group_ok: np 2, <= 4 M elements, 16-byte aligned rows,
// pads_done, no norm bound) as TWO launches -- the norm passes, then the splits; each matrix gets exactly the blocks, partial sums and scale
// planes_split() would give it.  *cache keeps the device-side table (null at first; planes_split_group_destroy at the end).
struct PlanesSplitGroup;
bool planes_split_group_ok(const PlanesSplitArgs &a);
hipError_t planes_split_group(const std::vector<PlanesSplitArgs> &v, PlanesSplitGroup **cache, hipStream_t s);
void planes_split_group_destroy(PlanesSplitGroup *g);
// zero the rows [0, lead) and [lead + rows, R) of every (K block, plane) chunk of a row-major plane buffer (a producer that writes the data rows itself)
hipError_t planes_pad(int np, void *P, long long k_blocks, long long R, int lead, long long rows, hipStream_t s);
// tile shape the GEMM uses for an N-column output: the A buffer needs tail >= tile rows beyond the last row read, the B buffer rows padded to the tile's columns
// ---- routing of the f32 GEMM entry points (rows_gemm / wgrad, gemm_f32.h) onto the plane kernels.
// The caller that owns the matrices (net.hip) splits an operand, describes the result in a PlanesOperand and installs it as a hint
// around the call; rows_gemm() / wgrad() check that the hinted planes really are those of their operands (base pointer, leading
// dimension, row range, zero rows around the views) and run the plane kernels, else their own.
struct PlanesOperand {
  const float *base = nullptr;  // the f32 matrix the planes were split from: rows x cols, leading dimension ld
  int rows = 0, cols = 0;
  long long ld = 0;
  int np = 0;
  const void *P = nullptr;  // row-major planes (k = column), R rows of which `lead` zero rows in front
  long long R = 0;
  int lead = 0;
  long long kb_alloc = 0;   // K blocks the P buffer has room for (>= ceil(cols / 16); whole 256-column tiles for the rows-as-K reads)
  const void *PT = nullptr;  // planes of the transpose (k = row), Rt rows
  long long Rt = 0;
  const float *scale = nullptr;  // np == 2: [s, 1 / s] on the device
  // a weight matrix split WITH tap coefficients (PlanesSplitArgs::col_coef): the planes hold coef[c / coef_period] * W[r][c]; a GEMM
  // that passes exactly this coefficient vector (TdnnDARTSV3Component's effective coefficients) may use them, one that passes none or another may not
  const float *coef = nullptr;
  int coef_period = 0;
};
struct PlanesHintScope {
  const PlanesOperand *prev_a, *prev_b;
  PlanesHintScope(const PlanesOperand *a, const PlanesOperand *b);
  ~PlanesHintScope();
};
// how many rows GEMMs / weight gradients ran on the plane kernels so far in this process (tests: the routing did route)
extern long long g_planes_routed_rows, g_planes_routed_wgrad;
const PlanesOperand *planes_hint_a();
const PlanesOperand *planes_hint_b();

int planes_gemm_tile_rows(int N);
// the tile rows planes_gemm() will use for THIS launch (256, or 128 for short reductions into 256-column tiles): what a caller needs to count its row tiles
int planes_gemm_launch_tile_rows(const PlanesGemmArgs &a);
int planes_gemm_tile_cols(int N);
hipError_t planes_gemm(const PlanesGemmArgs &a, hipStream_t s);
// epilogue of a split-K launch whose slabs hold whole output rows (ksplit > 1, ntap <= 1, ldp_n == 1): C[m][n] = f(scale * sum_sp partial[sp][m][n])
// with the launch's init / bias / addend / ReLU rules
hipError_t planes_splitk_finish(const PlanesGemmArgs &a, hipStream_t s);

}  // namespace tdnnf
