// planes_gemm.hip -- f32-equivalent GEMMs on the bf16 matrix cores from operands PRE-SPLIT into bf16 planes in HBM.
//
// gemm_precision 2 ("bf16x6"): a = a0 + a1 + a2 (three bf16 planes, 24 mantissa bits), a.b ~ the six products
// a_i b_j with i + j <= 2, accumulated in f32, smallest first -- results inside every exact-f32 tolerance of the parity tests
// at 6/16 of the f32 MFMA's cycles.  The round-2 kernels (gemm_f32.hip, rows_gemm_x3_kernel) split the f32 operands when a
// staged tile goes to LDS; what they wait for is that staging path (global load -> split in registers -> LDS write -> barrier,
// one 128 x 128 tile, K step 16), not the matrix cores.  Here the split happens ONCE per operand, in a pass of its own
// (planes_split_kernel: every activation / derivative matrix is the A operand of one GEMM and of its weight-gradient twin),
// into a layout made for the consumer:
//
//   P16 planes of an R x C matrix:  bf16 P[kb][plane][row][16],  kb = c / 16 (K blocks of 16), plane 0..2, row 0..R-1
//   (R = lead + rows + tail: zero rows in front and behind, so that row-shifted tap views and tile overhang read zeros);
//   a row record is 32 bytes, its two 16-byte halves (k 0..7 | k 8..15) swapped when bit 3 of the row index is set, which makes
//   the 16-byte fragment reads of 16 consecutive rows fall on 16 different 16-byte columns of the 256-byte LDS bank row.
//
// A K step of a (BM x BN) tile is then 3 CONTIGUOUS chunks of BM x 32 bytes of A and 3 of BN x 32 bytes of B: the kernel moves
// them with LDS-DMA (global_load_lds_dwordx4: no registers, no conversion, no LDS write instructions) into a ring of three
// stages, two K steps ahead of the one being multiplied, with counted vmcnt waits and one raw barrier per step
// (MI355X guide, "Pipelining across barriers").  One block per CU; a wave owns a 64 x 160 (or 64 x 128) block of the tile:
// 10 (8) accumulator tiles of v_mfma_f32_32x32x16_bf16, 60 (48) MFMAs per K step between two barriers.
//
// Reference semantics: the GEMMs of TdnnComponent::Propagate / Backprop (/root/reference/src/nnet3/nnet-tdnn-component.cc:302-324,
// :378-411) with K-segments = taps (row-shifted views of one matrix), as rows_gemm().
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "planes_gemm.h"

namespace tdnnf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kStages = 3;

__device__ __forceinline__ void split3(float x, __bf16 &p0, __bf16 &p1, __bf16 &p2) {
  p0 = (__bf16)x;
  float r = x - (float)p0;
  p1 = (__bf16)r;
  r -= (float)p1;
  p2 = (__bf16)r;
}

// X (rows x cols, ld) -> P16 planes with `lead` zero rows in front and `tail` behind.  A block: 32 rows x 2 K blocks per pass
// (a thread: 4 floats = 16 bytes in, 3 x 8 bytes out; a row's 128 bytes are read by 8 neighbouring threads).
__global__ __launch_bounds__(256) void planes_split_kernel(const float *X, long long ld, int rows, int cols, int lead, long long R, __bf16 *P) {
  const int t = threadIdx.x, lr = t >> 3, q = t & 7;
  const int kb = blockIdx.y * 2 + (q >> 2), c0 = kb * 16 + (q & 3) * 4;
  const int nkb = (cols + 15) / 16;
  if (kb >= nkb) return;
  for (int r = blockIdx.x * 32 + lr; r < rows; r += gridDim.x * 32) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float *src = X + (long long)r * ld + c0;
    if (c0 + 3 < cols && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
      const float4 f = *reinterpret_cast<const float4 *>(src);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (c0 + j < cols) v[j] = src[j];
    }
    bf16x4 pl[3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      __bf16 a, b, c;
      split3(v[j], a, b, c);
      pl[0][j] = a; pl[1][j] = b; pl[2][j] = c;
    }
    const long long ra = lead + r;
    const int k = (q & 3) * 4;                       // k offset inside the record: 0, 4, 8, 12
    const int half = (k >> 3) ^ (int)((ra >> 3) & 1);  // swizzled half
    const int off = half * 8 + (k & 7);
#pragma unroll
    for (int p = 0; p < 3; p++) *reinterpret_cast<bf16x4 *>(P + (((long long)kb * 3 + p) * R + ra) * 16 + off) = pl[p];
  }
}

// zero the lead / tail rows of every (kb, plane) chunk
__global__ __launch_bounds__(256) void planes_pad_kernel(__bf16 *P, int nkb, long long R, int lead, int rows) {
  const long long pad = R - rows;  // per chunk
  const long long total = (long long)nkb * 3 * pad * 2;  // 16-byte pieces
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const long long chunk = e / (pad * 2), w = e % (pad * 2);
    long long row = w / 2;
    if (row >= lead) row += rows;
    *reinterpret_cast<uint4 *>(P + (chunk * R + row) * 16 + (w & 1) * 8) = make_uint4(0, 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------------ the GEMM
// block = WM x WN waves, a wave owns TM x TN accumulator tiles of 32 x 32; BM = WM TM 32, BN = WN TN 32.
template <int WM, int WN, int TM, int TN, bool DB>
__global__ __launch_bounds__(WM *WN * 64) void planes_gemm_kernel(const PlanesGemmArgs p, int ntm, int ntn) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_BYTES = 3 * BM * 32, B_BYTES = 3 * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = STAGE / 16, PPT = (PIECES + NT - 1) / NT;  // 16-byte pieces per stage / per thread
  constexpr int STAGE_PAD = PPT * NT * 16;  // every thread copies PPT pieces per stage (the surplus ones into the pad): one vmcnt count for all waves
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile order (workgroups are dealt to the eight XCDs round-robin): each XCD a contiguous run of tiles, tile_n fastest
  const int nblk = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m = bid / ntn, tile_n = bid % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // ---- the stages in K order: (segment, K block).  Stage g -> ring slot g % 3.
  int total = 0;
  for (int s = 0; s < p.nseg; s++) total += p.seg[s].nkb;
  // what this thread copies per stage: piece q = t + NT j of the stage image [A p0 | A p1 | A p2 | B p0 | B p1 | B p2]
  // (a plane chunk is contiguous in global memory as well: BM / BN row records of 32 bytes)
  int lds_off[PPT];
  bool isA[PPT], live[PPT];
  long long rel[PPT];  // byte offset inside the (kb, plane 0) chunk group of its operand, relative to the segment's first row
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    const int q = t + NT * j;
    live[j] = q < PIECES;
    lds_off[j] = q * 16;
    isA[j] = q < A_BYTES / 16 || !live[j];
    const int w = isA[j] ? q : q - A_BYTES / 16;            // piece inside the operand's part
    const int rowsb = isA[j] ? BM * 2 : BN * 2;            // pieces per plane chunk
    const int pl = w / rowsb, inner = w % rowsb;
    rel[j] = live[j] ? ((long long)pl * (isA[j] ? p.RA : p.RB)) * 32 + (long long)inner * 16 : 0;  // (surplus pieces re-read the tile's first 16 bytes)
  }
  // Requests: every piece keeps a running source pointer, advanced by its operand's K-block stride after each stage; the segment
  // table (kernel arguments) is only read when a segment ends.
  const char *srcp[PPT];
  long long kstride[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) kstride[j] = (isA[j] ? p.RA : p.RB) * 96;  // 3 planes x 32 bytes x rows
  int ld_seg = -1, ld_left = 0;
  auto next_request_segment = [&]() {
    ld_seg++;
    if (ld_seg >= p.nseg) return;
    const PlanesSeg sg = p.seg[ld_seg];
    ld_left = sg.nkb;
    const char *ga = reinterpret_cast<const char *>(p.A) + ((long long)sg.a_kb0 * 3 * p.RA + sg.a_row + m0) * 32;
    const char *gb = reinterpret_cast<const char *>(p.B) + ((long long)sg.b_kb0 * 3 * p.RB + n0) * 32;
#pragma unroll
    for (int j = 0; j < PPT; j++) srcp[j] = (isA[j] ? ga : gb) + rel[j];
  };
  next_request_segment();
  auto request = [&](int slot) {
    char *dst = smem + slot * STAGE_PAD;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcp[j]), (__attribute__((address_space(3))) void *)(dst + lds_off[j]), 16, 0, 0);
      srcp[j] += kstride[j];
    }
    if (--ld_left == 0) next_request_segment();
  };

  // fragment addressing: lane (li, lh) of a 32 x 32 x 16 MFMA holds k = 8 lh .. 8 lh + 7 of row li; the halves of a row record are
  // swapped when bit 3 of its absolute row is set (B: n0 is a multiple of 32; A: the segment's first row decides)
  int cs_seg = 0, cs_left = p.seg[0].nkb;
  int arow0 = (int)((p.seg[0].a_row + m0) & 15);  // only bit 3 of (a_row + m0 + local) matters
  int a_off[TM], b_off[TN];  // byte offsets of this lane's fragments inside plane 0 of a stage
  auto set_a_off = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++) {
      const int row = wm * TM * 32 + i * 32 + li;
      a_off[i] = row * 32 + ((lh ^ (((row + arow0) >> 3) & 1)) << 4);
    }
  };
  set_a_off();
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int row = wn * TN * 32 + j * 32 + li;
    b_off[j] = A_BYTES + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4);
  }
  // Fragments of stage g + 1 are read into a second register set while stage g is multiplied: a wave has its SIMD to itself, so
  // LDS latency (and the LDS bandwidth of four waves reading 18 KB each) would otherwise sit in front of every K step's MFMAs.
  auto read_frags = [&](int slot, bf16x8 (&a)[3][TM], bf16x8 (&b)[3][TN]) {
    const char *st = smem + slot * STAGE_PAD;
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
      for (int j = 0; j < TN; j++) b[q][j] = *reinterpret_cast<const bf16x8 *>(st + q * BN * 32 + b_off[j]);
#pragma unroll
      for (int i = 0; i < TM; i++) a[q][i] = *reinterpret_cast<const bf16x8 *>(st + q * BM * 32 + a_off[i]);
    }
    if (--cs_left == 0 && ++cs_seg < p.nseg) {  // (the NEXT read belongs to the next segment: its rows may swap other halves)
      cs_left = p.seg[cs_seg].nkb;
      arow0 = (int)((p.seg[cs_seg].a_row + m0) & 15);
      set_a_off();
    }
  };
  auto multiply = [&](const bf16x8 (&a)[3][TM], const bf16x8 (&b)[3][TN]) {
    // six products per accumulator tile, smallest terms first, the leading term last
#pragma unroll
    for (int d = 2; d >= 0; d--)
#pragma unroll
      for (int q = 0; q <= d; q++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[d - q][j], acc[i][j], 0, 0, 0);
  };
  // One K step.  On entry the fragments of stage g are in (a0, b0) and the requests of stages g + 1, g + 2 are in flight.
  //   wait until stage g + 1 has landed (at most stage g + 2's pieces outstanding); barrier: everybody's pieces of stage g + 1 are
  //   in LDS and everybody has read stage g's fragments, so slot g % 3 is free: request stage g + 3 into it; read the fragments of
  //   stage g + 1 into (a1, b1) while stage g is multiplied.
  auto step = [&](int g, bf16x8 (&a0)[3][TM], bf16x8 (&b0)[3][TN], bf16x8 (&a1)[3][TM], bf16x8 (&b1)[3][TN]) {
    if (g + 1 < total) {
      if (g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (g + 3 < total) request(g % kStages);
      read_frags((g + 1) % kStages, a1, b1);
    }
    multiply(a0, b0);
  };

  if constexpr (DB) {
    // prologue: three stages requested, the first one's fragments read
    bf16x8 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
    if (total > 0) request(0);
    if (total > 1) request(1);
    if (total > 2) request(2);
    if (total > 0) {
      if (total > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
      else if (total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(0, fa0, fb0);
    }
    for (int g = 0; g < total; g += 2) {
      step(g, fa0, fb0, fa1, fb1);
      if (g + 1 < total) step(g + 1, fa1, fb1, fa0, fb0);
    }
  } else {
    // One register set (tiles whose accumulators leave no room for a second): stage g is read and multiplied behind the barrier
    // that follows its wait; two stages are in flight meanwhile.
    bf16x8 fa[3][TM], fb[3][TN];
    if (total > 0) request(0);
    if (total > 1) request(1);
    for (int g = 0; g < total; g++) {
      if (g + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everybody's pieces of stage g are in LDS; everybody is done reading stage g - 1
      if (g + 2 < total) request((g + 2) % kStages);  // into the slot stage g - 1 used
      read_frags(g % kStages, fa, fb);
      multiply(fa, fb);
    }
  }

  // ---- epilogue: straight from the accumulators (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
  // a store instruction writes two 128-byte row segments
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + (wn * TN + j) * 32 + li;
      if (n >= p.N) continue;
      const float bias = p.init_mode == 1 ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M) continue;
        float *c = p.C + (long long)m * p.ldc + n;
        float v = acc[i][j][r] + bias;
        if (p.init_mode == 0) v += *c;
        if (p.add && m >= p.add_lo && m < p.add_hi) v += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n];
        if (p.relu) v = floor_keep_nan(v, 0.f);
        *c = v;
      }
    }
}

template <int WM, int WN, int TM, int TN, bool DB = false>
hipError_t launch(const PlanesGemmArgs &a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NT = WM * WN * 64, PIECES = (3 * BM * 32 + 3 * BN * 32) / 16, PPT = (PIECES + NT - 1) / NT;
  constexpr size_t lds = (size_t)kStages * PPT * NT * 16;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void *)planes_gemm_kernel<WM, WN, TM, TN, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((planes_gemm_kernel<WM, WN, TM, TN, DB>), dim3(ntm * ntn), dim3(WM * WN * 64), lds, s, a, ntm, ntn);
  return hipGetLastError();
}

}  // namespace

size_t planes_bytes(int rows, int cols, int lead, int tail) {
  const long long R = (long long)lead + rows + tail, nkb = (cols + 15) / 16;
  return (size_t)(nkb * 3 * R * 32);
}

hipError_t planes_split(MatView x, int lead, int tail, void *planes, hipStream_t s) {
  if (x.rows <= 0 || x.cols <= 0) return hipSuccess;
  const long long R = (long long)lead + x.rows + tail;
  const int nkb = (x.cols + 15) / 16;
  if (lead + tail > 0) {
    const long long pieces = (long long)nkb * 3 * (lead + tail) * 2;
    hipLaunchKernelGGL(planes_pad_kernel, dim3(grid_for(pieces, 256)), dim3(256), 0, s, (__bf16 *)planes, nkb, R, lead, x.rows);
  }
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)std::min(2048, (x.rows + 31) / 32), (nkb + 1) / 2), dim3(256), 0, s, x.data, (long long)x.stride, x.rows,
                     x.cols, lead, R, (__bf16 *)planes);
  return hipGetLastError();
}

int planes_gemm_tile_rows(int N) { return 256; }
int planes_gemm_tile_cols(int N) {
  const int w128 = ((N + 127) / 128) * 128 - N, w160 = ((N + 159) / 160) * 160 - N;
  if (w160 < w128) return 160;
  return N % 256 == 0 ? 256 : 128;  // (256 x 256 tiles: 2/3 of the 256 x 128 tile's operand bytes per flop)
}

hipError_t planes_gemm(const PlanesGemmArgs &a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.nseg <= 0) return hipSuccess;
  // 160-wide tiles for the TDNN-F bottleneck, 128-wide otherwise; 4 waves, a wave = 64 rows x the tile's width
  // Measured on MI355X (tools/planes_bench.py, f32-equivalent TFLOP/s; exact-f32 kernel of gemm_f32.hip in brackets):
  //   256 x 256 tile, 8 waves of 64 x 128:  N = 1536, K = 2 x 1536: 227 [130];  K = 2 x 160 (.affine forward): 162 [116]
  //   256 x 160 tile, 8 waves of 32 x 160:  N = 160, K = 2 x 1536 (.linear forward): 151 [117] -- one column tile, so A (6 bytes per
  //   element as planes) is streamed from HBM once: 53 flop per byte, ~250 TFLOP/s at 5 TB/s; 782 tiles on 256 CUs = 4 rounds for 3.05
  //   256 x 128 tile, 4 x 2 waves of 64 x 64: 201 / 150.  Four waves of 64 rows x the tile's width: 194 / 121 (one wave per SIMD
  //   leaves every LDS / barrier wait exposed); fragments double-buffered in registers: no gain, spills on the wide tiles.
  const int bn = planes_gemm_tile_cols(a.N);
  if (bn == 160) return launch<8, 1, 1, 5>(a, s);
  if (bn == 256) return launch<4, 2, 2, 4>(a, s);
  return launch<4, 2, 2, 2>(a, s);
}

}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

size_t tdnnf_planes_bytes(int rows, int cols, int lead_rows, int tail_rows) {
  if (rows < 0 || cols <= 0 || lead_rows < 0 || tail_rows < 0) return 0;
  return planes_bytes(rows, cols, lead_rows, tail_rows);
}

int tdnnf_planes_split(const tdnnf_mat *x, int lead_rows, int tail_rows, void *planes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(x) && planes && lead_rows >= 0 && tail_rows >= 0 && x->cols > 0, "planes_split: bad arguments");
  TDNNF_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "planes_split: the plane buffer must be 16-byte aligned");
  TDNNF_HIP(planes_split(view(x), lead_rows, tail_rows, planes, (hipStream_t)stream));
  return TDNNF_OK;
}

int tdnnf_planes_gemm(const void *a_planes, long long a_rows_total, const void *b_planes, long long b_rows_total, int num_segments, const long long *a_row,
                      const int *a_first_col, const int *b_first_col, const int *seg_cols, const float *bias, int init_mode, int relu, tdnnf_mat *c,
                      tdnnf_stream stream) {
  TDNNF_REQUIRE(a_planes && b_planes && mat_ok(c) && num_segments >= 1 && num_segments <= 16 && a_row && a_first_col && b_first_col && seg_cols,
                "planes_gemm: bad arguments (1..16 segments)");
  TDNNF_REQUIRE(init_mode >= 0 && init_mode <= 2 && (init_mode != 1 || bias), "planes_gemm: init_mode 0 (+=), 1 (bias), 2 (=)");
  PlanesGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = a_planes; a.RA = a_rows_total; a.B = b_planes; a.RB = b_rows_total;
  a.C = c->data; a.ldc = c->stride; a.M = c->rows; a.N = c->cols;
  a.bias = bias; a.init_mode = init_mode; a.relu = relu; a.nseg = num_segments;
  const int BM = planes_gemm_tile_rows(c->cols), BN = planes_gemm_tile_cols(c->cols);
  TDNNF_REQUIRE(b_rows_total >= (long long)((c->cols + BN - 1) / BN) * BN, "planes_gemm: the B plane buffer needs %d rows (output columns padded to the %d-column tile)",
                ((c->cols + BN - 1) / BN) * BN, BN);
  for (int i = 0; i < num_segments; i++) {
    TDNNF_REQUIRE(a_first_col[i] % 16 == 0 && b_first_col[i] % 16 == 0 && seg_cols[i] > 0 && a_row[i] >= 0, "planes_gemm: segment %d: columns must start on a multiple of 16", i);
    TDNNF_REQUIRE(a_row[i] + (long long)((c->rows + BM - 1) / BM) * BM <= a_rows_total,
                  "planes_gemm: segment %d reads rows %lld..%lld of an A plane buffer of %lld rows (tail rows must cover the %d-row tile)", i, a_row[i],
                  a_row[i] + (long long)((c->rows + BM - 1) / BM) * BM, a_rows_total, BM);
    a.seg[i].a_row = a_row[i];
    a.seg[i].a_kb0 = a_first_col[i] / 16;
    a.seg[i].b_kb0 = b_first_col[i] / 16;
    a.seg[i].nkb = (seg_cols[i] + 15) / 16;
  }
  TDNNF_HIP(planes_gemm(a, (hipStream_t)stream));
  return TDNNF_OK;
}

}  // extern "C"
