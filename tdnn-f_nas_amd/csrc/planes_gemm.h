// planes_gemm.h -- host interface of the pre-split bf16-plane GEMMs (planes_gemm.hip): f32-equivalent products on the bf16
// matrix cores (gemm_precision 2, "bf16x6").
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "common.h"

namespace tdnnf {

// One K-segment (a tap of a TdnnComponent: a row-shifted view of the A matrix against a column block of the weights).
struct PlanesSeg {
  long long a_row;  // first row of the segment's view in the A plane buffer (lead rows included): tile row m reads a_row + m
  int a_kb0;        // first K block (of 16) of A
  int b_kb0;        // first K block of B
  int nkb;          // K blocks in the segment
};

struct PlanesGemmArgs {
  const void *A;  // P16 planes of the A operand (planes_split)
  long long RA;   // rows per (K block, plane) chunk of A: lead + rows + tail
  const void *B;  // P16 planes of the B operand, row n = output column n, k contiguous (weights: one row per output)
  long long RB;
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
  PlanesSeg seg[16];
};

// bytes of the P16 plane buffer of a rows x cols matrix with `lead` zero rows in front and `tail` behind
size_t planes_bytes(int rows, int cols, int lead, int tail);
// x (f32) -> planes; the lead / tail rows are zeroed
hipError_t planes_split(MatView x, int lead, int tail, void *planes, hipStream_t s);
// tile shape the GEMM will use for an N-column output: the A buffer needs tail >= tile rows, the B buffer rows padded to the tile's columns
int planes_gemm_tile_rows(int N);
int planes_gemm_tile_cols(int N);
hipError_t planes_gemm(const PlanesGemmArgs &a, hipStream_t s);

}  // namespace tdnnf
