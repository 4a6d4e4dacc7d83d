// gemm_ring.h -- persistent LDS-DMA-ring form of the exact-f32 rows GEMM (gemm_ring.hip); called by rows_gemm() only.
#pragma once
#include "gemm_f32.h"

namespace tdnnf {

bool rows_gemm_ring_enabled();  // TDNNF_GEMM_RING != 0
int rows_gemm_ring_mode();      // 1: the 128 x 128 tile only (default); 2: the 128 x 160 tile as well
// k-contiguous B, whole-K-step segments, exact f32, 16-byte aligned operands (vec), no tap coefficients / sumsq / split-K
bool rows_gemm_ring_ok(const RowsGemmArgs &a, bool b_kcontig, bool vec);
// resident blocks of the chip for the 128 x 128 (tile_cols 128) or 128 x 160 tile: the grid of a launch, and what rows_gemm() balances with
int rows_gemm_ring_slots(int tile_cols);
// a.c_vec / a.colstats_stride as rows_gemm() sets them; tile_cols 128 or 160
hipError_t rows_gemm_ring(const RowsGemmArgs &a, bool b_kcontig, int tile_cols, hipStream_t s);

}  // namespace tdnnf
