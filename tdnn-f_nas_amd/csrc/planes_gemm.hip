// planes_gemm.hip -- f32-equivalent GEMMs on the 16-bit matrix cores from operands PRE-SPLIT into 16-bit planes in HBM.
//
// Two arithmetics (planes_gemm.h):
//   np = 3 "bf16x6": a = a0 + a1 + a2 (three bf16 planes, 24 mantissa bits), a.b ~ the six products a_i b_j with i + j <= 2: 6/16 of
//     the f32 MFMA's cycles, 6 bytes per operand element.
//   np = 2 "f16x3":  a s = h + l with s a power of two taken from the operand's Frobenius norm (|x| <= ||X||_F, so s ||X||_F <= 65504
//     means NO element can overflow, whatever the data), two f16 planes (11 + 11 bits and the sign of the remainder); a.b ~ h h' + h l' +
//     l h': 3/16 of the f32 MFMA's cycles and 4 bytes per operand element -- what an f32 operand costs.  Norm-wise the error sits below
//     the exact-f32 kernel's own summation error (tests/test_gpu_planes_gemm.py, against float64): an element that is small against
//     its matrix's rms loses relative precision (its low plane becomes subnormal: absolute error <= 2^-25 of the scaled value 1), which a
//     product that sums thousands of typical elements does not see.
// The round-2 kernels (gemm_f32.hip, rows_gemm_x3_kernel) split f32 operands when a staged tile goes to LDS and wait for exactly that
// path.  Here the split happens ONCE per operand, in a pass of its own (planes_split_kernel), into a layout made for the consumer:
//
//   P16 planes of an R x C matrix:  e16 P[kb][plane][row][16],  kb = c / 16 (K blocks of 16), row 0..R-1
//   (R = lead + rows + tail: zero rows in front and behind, so that row-shifted tap views and tile overhang read zeros);
//   a row record is 32 bytes, its two 16-byte halves (k 0..7 | k 8..15) swapped when bit 3 of the row index is set, which makes
//   the 16-byte fragment reads of 16 consecutive rows fall on 16 different 16-byte columns of the 256-byte LDS bank row.
//   The same pass writes the planes of the TRANSPOSE (k = row index) for the products that reduce over rows (weight gradients).
//
// A K step of a (BM x BN) tile is then np CONTIGUOUS chunks of BM x 32 bytes of A and np of BN x 32 bytes of B: the kernel moves
// them with LDS-DMA (global_load_lds_dwordx4: no registers, no conversion, no LDS write instructions) into a ring of three
// stages, two K steps ahead of the one being multiplied, with counted vmcnt waits and one raw barrier per step
// (MI355X guide, "Pipelining across barriers").  One block of 8 waves per CU.
//
// Reference semantics: the GEMMs of TdnnComponent::Propagate / Backprop / UpdateSimple (/root/reference/src/nnet3/nnet-tdnn-component.cc:302-324,
// :378-411, :452) with K-segments = taps (row-shifted views of one matrix), as rows_gemm() / wgrad().
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "gemm_f32.h"
#include "planes_gemm.h"

namespace tdnnf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// LDS ring depth per tile.  Three everywhere: the 256-row tiles of two planes would have room for four (128 KB, still one block per CU), measured
// on the trainer's shapes and in the step: no difference (the K loop is not bound by the latency of its loads) -- and 96 KB leave room beside it.
constexpr int kBigStages = 3;
template <int NP, int BM>
constexpr int stages_of() { return (NP == 2 && BM >= 256) ? kBigStages : 3; }

template <int NP>
struct Plane;
template <>
struct Plane<3> {
  typedef __bf16 E;
  typedef bf16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(float x, E (&p)[3]) {
    p[0] = (__bf16)x;
    float r = x - (float)p[0];
    p[1] = (__bf16)r;
    r -= (float)p[1];
    p[2] = (__bf16)r;
  }
};
template <>
struct Plane<2> {
  typedef _Float16 E;
  typedef f16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(float x, E (&p)[2]) {  // x already scaled
    p[0] = (_Float16)x;
    p[1] = (_Float16)(x - (float)p[0]);
  }
};

// ------------------------------------------------------------------------------------------------------ the split pass
constexpr int kSumsqBlocks = 1024;
// partial[b] = sum of squares of the elements block b walks (fixed assignment: deterministic)
__global__ __launch_bounds__(256) void planes_sumsq_kernel(MatView x, double *partial) {
  __shared__ double red[4];
  const long long total = (long long)x.rows * x.cols;
  double acc = 0;
  float run = 0.f;
  int cnt = 0;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / x.cols), c = (int)(e % x.cols);
    const float v = x.data[(long long)r * x.stride + c];
    run += v * v;
    if (++cnt == 64) {  // short float runs, double across them
      acc += run;
      run = 0.f;
      cnt = 0;
    }
  }
  acc += run;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// the same with 16-byte reads (rows and base 16-byte aligned): a thread owns float4 columns, a row's last few columns one by one
__global__ __launch_bounds__(256) void planes_sumsq4_kernel(MatView x, double *partial) {
  __shared__ double red[4];
  const int c4 = (x.cols + 3) >> 2;
  const long long total = (long long)x.rows * c4;
  double acc = 0;
  float run = 0.f;
  int cnt = 0;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / c4), c = (int)(e % c4);
    const float *src = x.data + (long long)r * x.stride + 4 * c;
    if (4 * c + 3 < x.cols) {
      const float4 v = *reinterpret_cast<const float4 *>(src);
      run += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      for (int j = 0; 4 * c + j < x.cols; j++) run += src[j] * src[j];
    }
    if (++cnt == 16) {
      acc += run;
      run = 0.f;
      cnt = 0;
    }
  }
  acc += run;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// scale[0] = s = 2^e, the largest power of two with s ||X||_F <= 65504 (every |x| <= ||X||_F: nothing overflows) and s rms(X) <= 64;
// scale[1] = 1 / s; scale[2] = the norm (or the upper bound it was taken from).  An all-zero matrix gets s = 1; a NaN / Inf norm gives a
// NaN scale (the product is then NaN, as in f32).  With `mul` / `add_rec`: sqrt(sum) is only part of a bound, mul sqrt(sum) + add_coef add_rec[2].
// (all 256 threads of a block; every block that calls it with the same partials gets the same s: fixed summation order)
__device__ __forceinline__ float planes_scale_of(const double *partial, int nb, double numel, float mul, float add_coef, const float *add_rec, double *red,
                                                 double *fro_out) {
  double a = 0;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double sum = red[0];
  const double fro = (double)mul * sqrt(sum) + (add_rec ? (double)add_coef * (double)add_rec[2] : 0.0);
  float s = 1.0f;
  if (fro != fro || fro > 1.0e150) {
    s = __int_as_float(0x7fc00000);
  } else if (fro > 0) {
    const double rms = fro / sqrt(numel);
    int e = (int)floor(log2(65504.0 / fro));
    const int e2 = (int)floor(log2(64.0 / rms));
    if (e2 < e) e = e2;
    if (e > 120) e = 120;
    if (e < -120) e = -120;
    s = ldexpf(1.0f, e);
  }
  *fro_out = fro;
  __syncthreads();  // (red is reused by the caller)
  return s;
}
__global__ void planes_scale_kernel(const double *partial, int nb, double numel, float *scale, float mul, float add_coef, const float *add_rec) {
  __shared__ double red[256];
  double fro;
  const float s = planes_scale_of(partial, nb, numel, mul, add_coef, add_rec, red, &fro);
  if (threadIdx.x != 0) return;
  scale[0] = s;
  scale[1] = 1.0f / s;
  scale[2] = (float)(fro * 1.000001);  // (rounded up: the record may feed the bound of a matrix this one is added into)
}

// X (rows x cols, ld) -> P16 planes (k = column; `lead` zero rows in front) and / or the planes of the transpose (k = row).
// A block: a 64 x 64 tile; thread t reads 16 consecutive floats of row t / 4 (one row record of P), the transposed records go
// through LDS (thread t then owns column t % 64, rows 16 (t / 64) ..+15).
// sq_partial != null (small matrices): the scale is formed here from the norm pass's partials -- by every block, identically -- and
// block (0, 0) writes the record; saves the launch of planes_scale_kernel in front of every small split
template <int NP>
__device__ __forceinline__ void planes_split_block(const float *X, long long ld, int rows, int cols, const float *scale, int lead, long long R, void *Pv,
                                                   long long Rt, void *PTv, int vec_ok, const double *sq_partial, int sq_nb, float *scale_out,
                                                   const float *col_coef, int col_coef_period, int bx, int by) {
  typedef typename Plane<NP>::E E;
  __shared__ __attribute__((aligned(16))) E tile[NP][64][64 + 2];
  E *P = reinterpret_cast<E *>(Pv), *PT = reinterpret_cast<E *>(PTv);
  const int t = threadIdx.x, lr = t >> 2, cq = t & 3;
  const int r = bx * 64 + lr, c0 = by * 64 + cq * 16;
  float s = scale ? scale[0] : 1.0f;
  if (sq_partial) {
    double fro;
    s = planes_scale_of(sq_partial, sq_nb, (double)rows * cols, 1.0f, 0.0f, nullptr, reinterpret_cast<double *>(&tile[0][0][0]), &fro);
    if (bx == 0 && by == 0 && t == 0) {
      scale_out[0] = s;
      scale_out[1] = 1.0f / s;
      scale_out[2] = (float)(fro * 1.000001);
    }
  }
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = 0.f;
  if (r < rows && c0 < cols) {
    const float *src = X + (long long)r * ld + c0;
    if (vec_ok && c0 + 15 < cols) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 f = *reinterpret_cast<const float4 *>(src + 4 * q);
        v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (c0 + j < cols) v[j] = src[j];
    }
  }
  if (col_coef) {  // tap coefficients folded into the planes
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (c0 + j < cols) v[j] *= col_coef[(c0 + j) / col_coef_period];
  }
  E pl[NP][16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    E e[NP];
    Plane<NP>::split(v[j] * s, e);
#pragma unroll
    for (int p = 0; p < NP; p++) pl[p][j] = e[p];
  }
  const int nkb = (cols + 15) / 16, kb = c0 >> 4;
  if (P && r < rows && kb < nkb) {
    const long long ra = (long long)lead + r;
    const int sw = (int)((ra >> 3) & 1);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      E *dst = P + (((long long)kb * NP + p) * R + ra) * 16;
      *reinterpret_cast<uint4 *>(dst + (0 ^ sw) * 8) = *reinterpret_cast<const uint4 *>(&pl[p][0]);
      *reinterpret_cast<uint4 *>(dst + (1 ^ sw) * 8) = *reinterpret_cast<const uint4 *>(&pl[p][8]);
    }
  }
  if (!PT) return;
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int j = 0; j < 16; j++) tile[p][lr][cq * 16 + j] = pl[p][j];
  __syncthreads();
  const int c = by * 64 + (t & 63), kq = t >> 6;
  if (c >= cols) return;  // (rows of PT beyond `cols` are zeroed by the pad kernel)
  const long long kbt = (long long)bx * 4 + kq;
  const int sw = (c >> 3) & 1;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    E rec[16];
#pragma unroll
    for (int j = 0; j < 16; j++) rec[j] = tile[p][kq * 16 + j][t & 63];
    E *dst = PT + ((kbt * NP + p) * Rt + c) * 16;
    *reinterpret_cast<uint4 *>(dst + (0 ^ sw) * 8) = *reinterpret_cast<const uint4 *>(&rec[0]);
    *reinterpret_cast<uint4 *>(dst + (1 ^ sw) * 8) = *reinterpret_cast<const uint4 *>(&rec[8]);
  }
}

template <int NP>
__global__ __launch_bounds__(256) void planes_split_kernel(const float *X, long long ld, int rows, int cols, const float *scale, int lead, long long R, void *Pv,
                                                           long long Rt, void *PTv, int vec_ok, const double *sq_partial, int sq_nb, float *scale_out,
                                                           const float *col_coef, int col_coef_period) {
  planes_split_block<NP>(X, ld, rows, cols, scale, lead, R, Pv, Rt, PTv, vec_ok, sq_partial, sq_nb, scale_out, col_coef, col_coef_period, (int)blockIdx.x,
                         (int)blockIdx.y);
}
// Grouped form for many SMALL matrices (a net's weight matrices at the start of a step: 36 x (norm pass + split) launches, strictly serial on
// the caller's stream, nothing else in flight): matrix i owns the blocks [first[i], first[i + 1]) of each of the two launches.
struct PlanesSplitItem {
  const float *X;
  long long ld, R, Rt;
  int rows, cols, lead, vec_ok;
  void *P, *PT;
  float *scale;
  const float *col_coef;
  int col_coef_period;
  int sq_first, sq_nb;  // its norm-pass blocks / partials: [sq_first, sq_first + sq_nb)
  int sp_first, sp_gx;  // its split blocks: sp_first + by * sp_gx + bx
};
__device__ __forceinline__ int planes_item_of(const int *first, int n, int b) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b >= first[mid]) lo = mid;
    else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void planes_sumsq4_group_kernel(const PlanesSplitItem *items, const int *first, int n, double *partial) {
  __shared__ double red[4];
  const PlanesSplitItem it = items[planes_item_of(first, n, (int)blockIdx.x)];
  const int b = (int)blockIdx.x - it.sq_first;
  const int c4 = (it.cols + 3) >> 2;
  const long long total = (long long)it.rows * c4;
  double acc = 0;
  float run = 0.f;
  int cnt = 0;
  for (long long e = b * 256LL + threadIdx.x; e < total; e += (long long)it.sq_nb * 256) {  // (exactly planes_sumsq4_kernel's walk with sq_nb blocks)
    const int r = (int)(e / c4), c = (int)(e % c4);
    const float *src = it.X + (long long)r * it.ld + 4 * c;
    if (4 * c + 3 < it.cols) {
      const float4 v = *reinterpret_cast<const float4 *>(src);
      run += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      for (int j = 0; 4 * c + j < it.cols; j++) run += src[j] * src[j];
    }
    if (++cnt == 16) {
      acc += run;
      run = 0.f;
      cnt = 0;
    }
  }
  acc += run;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void planes_split_group_kernel(const PlanesSplitItem *items, const int *first, int n, const double *partial) {
  const PlanesSplitItem it = items[planes_item_of(first, n, (int)blockIdx.x)];
  const int b = (int)blockIdx.x - it.sp_first;
  planes_split_block<2>(it.X, it.ld, it.rows, it.cols, it.scale, it.lead, it.R, it.P, it.Rt, it.PT, it.vec_ok, partial + it.sq_first, it.sq_nb, it.scale, it.col_coef,
                        it.col_coef_period, b % it.sp_gx, b / it.sp_gx);
}

// zero the rows [0, lead) and [lead + rows, R) of every (kb, plane) chunk (32-byte records of 2-byte elements, whatever the type)
__global__ __launch_bounds__(256) void planes_pad_kernel(void *Pv, long long nchunks, long long R, int lead, long long rows) {
  unsigned short *P = reinterpret_cast<unsigned short *>(Pv);
  const long long pad = R - rows;  // per chunk
  const long long total = nchunks * pad * 2;  // 16-byte pieces
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const long long chunk = e / (pad * 2), w = e % (pad * 2);
    long long row = w / 2;
    if (row >= lead) row += rows;
    *reinterpret_cast<uint4 *>(P + (chunk * R + row) * 16 + (w & 1) * 8) = make_uint4(0, 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------------ the GEMM
// block = WM x WN waves, a wave owns TM x TN accumulator tiles of 32 x 32; BM = WM TM 32, BN = WN TN 32.
// ATR: the A operand is given by ROW-MAJOR planes of the matrix whose COLUMNS are the tile rows (a product that reduces over the
// matrix's rows, i.e. a weight gradient, without planes of the transpose): a K step is 16 consecutive matrix rows, the tile's 256
// columns are 16 K-block chunks of 16 x 32-byte row records, staged as [chunk pair][row 0..15][chunk parity][32 bytes] and read
// with ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group fetches 4 rows x 16 columns and every lane receives one
// column), two reads per operand register pair; rows of a 32-lane half cover 256 contiguous bytes: conflict-free.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NP, int WM, int WN, int TM, int TN, bool DB, bool ATR = false>
__global__ __launch_bounds__(WM *WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void planes_gemm_kernel(const PlanesGemmArgs p, int ntm, int ntn) {
  typedef typename Plane<NP>::V8 V8;
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_BYTES = NP * BM * 32, B_BYTES = NP * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = STAGE / 16, PPT = (PIECES + NT - 1) / NT;  // 16-byte pieces per stage / per thread
  constexpr int STAGE_PAD = PPT * NT * 16;  // every thread copies PPT pieces per stage (the surplus ones into the pad): one vmcnt count for all waves
  constexpr int NS = stages_of<NP, BM>();
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile order (workgroups are dealt to the eight XCDs round-robin): each XCD a contiguous run of tiles, column tiles
  // (and taps) fastest so that neighbours share the A chunk, K splits slowest
  const int ntaps = p.ntap > 1 ? p.ntap : 1, ncol = ntn * ntaps, nsplit = p.ksplit > 1 ? p.ksplit : 1;
  const int nblk = ntm * ncol * nsplit;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int sp = bid / (ntm * ncol), tile_m = (bid / ncol) % ntm, tcol = bid % ncol, tap = tcol / ntn, tile_n = tcol % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // ---- the stages in K order: (segment, K block).  This block multiplies stages [g_begin, g_begin + total).
  if (ntaps > 1 && p.skip_coef && p.skip_coef[tap] == 0.f) return;  // a tap with a zero coefficient (uniform-sample mode): its slab is not read either
  int all = 0;
  for (int s = 0; s < p.nseg; s++)
    if (!(ntaps <= 1 && p.skip_coef && p.skip_coef[s] == 0.f)) all += p.seg[s].nkb;
  // (alt_seg_order, as rows_gemm_kernel: odd row tiles visit the two taps in reverse order, so that the row block two neighbouring tiles
  // share is fetched by both in the same phase of the launch)
  const bool rev_seg = p.alt_seg_order && (tile_m & 1);
  const int g_begin = nsplit > 1 ? sp * p.kb_per_split : 0;
  const int total = nsplit > 1 ? max(0, min(all - g_begin, p.kb_per_split)) : all;
  const int tap_akb = ntaps > 1 ? p.tap_a_kb[tap] : 0, tap_bkb = ntaps > 1 ? p.tap_b_kb[tap] : 0;
  // what this thread copies per stage: piece q = t + NT j of the stage image [A planes | B planes]
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
    if (ATR && isA[j] && live[j]) {  // piece `inner` of the image [chunk pair][row][parity][half]: K-block chunk 2 cp + parity, row record q
      const int cp = inner >> 6, rem = inner & 63, q = rem >> 2, par = (rem >> 1) & 1, h16 = rem & 1;
      rel[j] = ((long long)pl * p.RA + (long long)(2 * cp + par) * NP * p.RA + q) * 32 + h16 * 16;
    }
  }
  // Requests: every piece keeps a running source pointer, advanced by its operand's K-block stride after each stage; the segment
  // table (kernel arguments) is only read when a segment ends.
  const char *srcp[PPT];
  long long kstride[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) kstride[j] = (ATR && isA[j]) ? 512 : (isA[j] ? p.RA : p.RB) * (32 * NP);  // (ATR: a K step is 16 rows of 32 bytes)
  int ld_seg = -1, ld_left = 0, ld_skip = g_begin;
  auto next_request_segment = [&]() {
    for (;;) {
      ld_seg++;
      if (ld_seg >= p.nseg) return;
      const int si = rev_seg ? p.nseg - 1 - ld_seg : ld_seg;
      if (ntaps <= 1 && p.skip_coef && p.skip_coef[si] == 0.f) continue;
      const PlanesSeg sg = p.seg[si];
      if (ld_skip >= sg.nkb) {  // (a split that starts behind this segment)
        ld_skip -= sg.nkb;
        continue;
      }
      ld_left = sg.nkb - ld_skip;
      const char *ga = ATR ? reinterpret_cast<const char *>(p.A) + ((long long)(m0 >> 4) * NP * p.RA + sg.a_row + 16LL * (sg.a_kb0 + tap_akb + ld_skip)) * 32
                           : reinterpret_cast<const char *>(p.A) + ((long long)(sg.a_kb0 + tap_akb + ld_skip) * NP * p.RA + sg.a_row + m0) * 32;
      const char *gb = reinterpret_cast<const char *>(p.B) + ((long long)(sg.b_kb0 + tap_bkb + ld_skip) * NP * p.RB + sg.b_row + n0) * 32;
      ld_skip = 0;
#pragma unroll
      for (int j = 0; j < PPT; j++) srcp[j] = (isA[j] ? ga : gb) + rel[j];
      return;
    }
  };
  next_request_segment();
  auto request_piece = [&](int slot, int j) {
    // (as an instruction the compiler does not see: it books the builtin as a flat access pending on BOTH counters and, knowing nothing of the
    // counted vmcnt waits below, turns every later lgkmcnt wait into lgkmcnt(0) -- the LDS reads could not be counted past each other)
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(smem + slot * STAGE_PAD + lds_off[j]));
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(srcp[j]), "s"(m0v) : "memory");  // (m0 is a reserved register nothing else in these kernels uses)
    srcp[j] += kstride[j];
  };
  auto request_done = [&]() {
    if (--ld_left == 0) next_request_segment();
  };
  auto request = [&](int slot) {
#pragma unroll
    for (int j = 0; j < PPT; j++) request_piece(slot, j);
    request_done();
  };

  // fragment addressing: lane (li, lh) of a 32 x 32 x 16 MFMA holds k = 8 lh .. 8 lh + 7 of row li; the halves of a row record are
  // swapped when bit 3 of its absolute row is set (the segment's first row decides)
  int cs_seg = -1, cs_left = 0, cs_skip = g_begin;
  int a_off[TM], b_off[TN];  // byte offsets of this lane's fragments inside plane 0 of a stage
  auto next_compute_segment = [&]() {
    for (;;) {
      cs_seg++;
      if (cs_seg >= p.nseg) return;
      const int si = rev_seg ? p.nseg - 1 - cs_seg : cs_seg;
      if (ntaps <= 1 && p.skip_coef && p.skip_coef[si] == 0.f) continue;
      const PlanesSeg sg = p.seg[si];
      if (cs_skip >= sg.nkb) {
        cs_skip -= sg.nkb;
        continue;
      }
      cs_left = sg.nkb - cs_skip;
      cs_skip = 0;
      const int arow0 = (int)((sg.a_row + m0) & 15), brow0 = (int)((sg.b_row + n0) & 15);
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int row = wm * TM * 32 + i * 32 + li;
        a_off[i] = row * 32 + ((lh ^ (((row + arow0) >> 3) & 1)) << 4);
        if (ATR) {  // lane = 16 G + 4 q + pp: group G reads chunk parity G & 1, rows 8 (G >> 1) + q (then + 4), columns 4 pp ..
          const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, kg = G >> 1;
          a_off[i] = (wm * TM + i) * 1024 + (8 * kg + q) * 64 + (G & 1) * 32 + ((((pp >> 1) ^ kg) & 1) << 4) + ((pp & 1) << 3);
        }
      }
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int row = wn * TN * 32 + j * 32 + li;
        b_off[j] = A_BYTES + row * 32 + ((lh ^ (((row + brow0) >> 3) & 1)) << 4);
      }
      return;
    }
  };
  next_compute_segment();
  // Fragments of stage g + 1 are read into a second register set while stage g is multiplied: a wave has its SIMD to itself, so
  // LDS latency (and the LDS bandwidth of the waves reading their fragments) would otherwise sit in front of every K step's MFMAs.
  auto read_a = [&](const char *st, int q, V8 (&a)[NP][TM]) {
#pragma unroll
    for (int i = 0; i < TM; i++) {
      if constexpr (ATR) {
        typedef __attribute__((address_space(3))) s16x4 *lds4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(st + q * BM * 32 + a_off[i]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(st + q * BM * 32 + a_off[i] + 256));
        union { s16x4 h[2]; V8 v; } u;
        u.h[0] = lo;
        u.h[1] = hi;
        a[q][i] = u.v;
      } else {
        a[q][i] = *reinterpret_cast<const V8 *>(st + q * BM * 32 + a_off[i]);
      }
    }
  };
  auto read_b = [&](const char *st, int q, V8 (&b)[NP][TN]) {
#pragma unroll
    for (int j = 0; j < TN; j++) b[q][j] = *reinterpret_cast<const V8 *>(st + q * BN * 32 + b_off[j]);
  };
  auto read_frags = [&](int slot, V8 (&a)[NP][TM], V8 (&b)[NP][TN]) {
    const char *st = smem + slot * STAGE_PAD;
#pragma unroll
    for (int q = 0; q < NP; q++) {
      read_b(st, q, b);
      read_a(st, q, a);
    }
    if (--cs_left == 0) next_compute_segment();  // (the NEXT read belongs to the next segment: its rows may swap other halves)
  };
  auto product = [&](const V8 (&a)[NP][TM], int qa, const V8 (&b)[NP][TN], int qb) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = Plane<NP>::mfma(a[qa][i], b[qb][j], acc[i][j]);
  };
  auto multiply = [&](const V8 (&a)[NP][TM], const V8 (&b)[NP][TN]) {
    // the products a_q b_(d - q), d = np - 1 .. 0: smallest terms first, the leading term last
#pragma unroll
    for (int d = NP - 1; d >= 0; d--)
#pragma unroll
      for (int q = 0; q <= d; q++) product(a, q, b, d - q);
  };
  // the same product with the request pieces [jlo, jhi) of `slot` issued between its MFMAs, evenly spaced (a piece issued alone costs the
  // wave ~60 cycles, several in a row behind a barrier 100-185 each -- with both waves of a SIMD there together the matrix pipe idles)
  auto product_req = [&](const V8 (&a)[NP][TM], int qa, const V8 (&b)[NP][TN], int qb, bool req, int slot, int jlo, int jhi) {
    constexpr int total_m = TM * TN;
    const int np = jhi - jlo;
#pragma unroll
    for (int k = 0; k < total_m; k++) {
      acc[k / TN][k % TN] = Plane<NP>::mfma(a[qa][k / TN], b[qb][k % TN], acc[k / TN][k % TN]);
#pragma unroll
      for (int n = 0; n < PPT; n++)
        if (n < np && k == ((n + 1) * total_m) / (np + 1) - 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (req) request_piece(slot, jlo + n);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  // One K step.  On entry the fragments of stage g are in (a0, b0) and the requests of stages g + 1, g + 2 are in flight.
  //   wait until stage g + 1 has landed (at most stage g + 2's pieces outstanding); barrier: everybody's pieces of stage g + 1 are
  //   in LDS and everybody has read stage g's fragments, so slot g % 3 is free: request stage g + 3 into it; read the fragments of
  //   stage g + 1 into (a1, b1) while stage g is multiplied.
  // wait until at most `k` stages' requests of this thread are outstanding (k <= NS - 1; uniform)
  auto wait_stages = [&](int k) {
    static_assert((NS - 1) * PPT <= 63, "vmcnt is a 6-bit count");
    if (k >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPT > 63 ? 63 : 4 * PPT) : "memory");
    else if (k == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPT > 63 ? 63 : 3 * PPT) : "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto step = [&](int g, V8 (&a0)[NP][TM], V8 (&b0)[NP][TN], V8 (&a1)[NP][TM], V8 (&b1)[NP][TN]) {
    if (g + 1 < total) {
      wait_stages(min(NS - 2, total - g - 2));  // stage g + 1 has landed; stages g + 2 .. may still be in flight
      __builtin_amdgcn_s_barrier();
      if (g + NS < total) request(g % NS);
      read_frags((g + 1) % NS, a1, b1);
    }
    multiply(a0, b0);
  };

  if constexpr (NP == 2 && !DB) {
    // Two planes, one register set, no exposed LDS phase: a K step's three products are ordered l h', h h', h l' and the fragments of the NEXT
    // stage are read into each operand's registers as soon as its last product of THIS stage has been issued -- l after the first product,
    // h' after the second, h and l' after the third -- so every LDS read has eight MFMAs (256 cycles of the matrix pipe) or more between
    // its issue and its first use.  (Read all at once behind the barrier, the twelve reads of a step sat in front of its MFMAs in both
    // waves of a SIMD together: the compute phase alone ran at 1.26 us per K step of a 256 x 256 tile where the MFMAs take 0.65-0.8.)
    // The barrier sits behind the first product: stage g + 1 has landed for everybody and everybody's reads of stage g are complete
    // (lgkmcnt(0): they were issued a product earlier), so slot g % NS takes stage g + NS.
    V8 fa[NP][TM], fb[NP][TN];
#pragma unroll
    for (int i = 0; i < NS; i++)
      if (total > i) request(i);
    if (total > 0) {
      wait_stages(min(NS - 1, total - 1));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), as an instruction the compiler's counter model sees: kernel-argument loads still pending here
                                           // (scalar loads share the counter and return out of order) would make it wait for ALL LDS reads at the loop head
      read_a(smem, 1, fa);  // (in the loop's order: the first product's operands are the oldest reads on either way into the loop)
      read_b(smem, 0, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_a(smem, 0, fa);
      read_b(smem, 1, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (total > 0) {
      product(fa, 1, fb, 0);
      __builtin_amdgcn_sched_barrier(0);
      // (the loop begins behind step g's first product, at its barrier: the one place of a step where no LDS read is outstanding, so the
      // compiler's wait-count model, which gives up precision across a loop's back edge, has nothing to be conservative about)
      for (int g = 0; g + 1 < total; g++) {
        const char *st = smem + ((g + 1) % NS) * STAGE_PAD;
        wait_stages(min(NS - 2, total - g - 2));
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (--cs_left == 0) next_compute_segment();  // the fragment offsets of stage g + 1 (the segment table is read with scalar loads, which share the LDS reads' counter)
        const bool req = g + NS < total;
        read_a(st, 1, fa);
        __builtin_amdgcn_sched_barrier(0);
        product_req(fa, 0, fb, 0, req, g % NS, 0, PPT / 2);
        __builtin_amdgcn_sched_barrier(0);
        read_b(st, 0, fb);
        __builtin_amdgcn_sched_barrier(0);
        product_req(fa, 0, fb, 1, req, g % NS, PPT / 2, PPT);
        if (req) request_done();
        __builtin_amdgcn_sched_barrier(0);
        read_a(st, 0, fa);
        read_b(st, 1, fb);
        __builtin_amdgcn_sched_barrier(0);
        product(fa, 1, fb, 0);  // (of step g + 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      product(fa, 0, fb, 0);
      product(fa, 0, fb, 1);
    }
  } else if constexpr (DB) {
    // prologue: NS stages requested, the first one's fragments read
    V8 fa0[NP][TM], fb0[NP][TN], fa1[NP][TM], fb1[NP][TN];
#pragma unroll
    for (int i = 0; i < NS; i++)
      if (total > i) request(i);
    if (total > 0) {
      wait_stages(min(NS - 1, total - 1));
      __builtin_amdgcn_s_barrier();
      read_frags(0, fa0, fb0);
    }
    for (int g = 0; g < total; g += 2) {
      step(g, fa0, fb0, fa1, fb1);
      if (g + 1 < total) step(g + 1, fa1, fb1, fa0, fb0);
    }
  } else {
    // One register set (tiles whose accumulators leave no room for a second): stage g is read and multiplied behind the barrier
    // that follows its wait; NS - 1 stages are in flight meanwhile.
    V8 fa[NP][TM], fb[NP][TN];
#pragma unroll
    for (int i = 0; i < NS - 1; i++)
      if (total > i) request(i);
    for (int g = 0; g < total; g++) {
      wait_stages(min(NS - 2, total - g - 1));
      __builtin_amdgcn_s_barrier();  // everybody's pieces of stage g are in LDS; everybody is done reading stage g - 1
      if (g + NS - 1 < total) request((g + NS - 1) % NS);  // into the slot stage g - 1 used
      read_frags(g % NS, fa, fb);
      multiply(fa, fb);
    }
  }

  // ---- epilogue: straight from the accumulators (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
  // a store instruction writes two 128-byte row segments
  if (nsplit > 1) {  // raw partial tile into this split's slab (the caller reduces, scales, accumulates)
    float *P = p.partial + (long long)sp * p.partial_stride + (long long)tap * p.tap_off_p;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + (wn * TN + j) * 32 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M) P[(long long)m * p.ldp_m + (long long)n * p.ldp_n] = acc[i][j][r];
        }
      }
    return;
  }
  float sc = 1.0f;
  if (NP == 2) sc = (p.scale_a ? p.scale_a[1] : 1.0f) * (p.scale_b ? p.scale_b[1] : 1.0f);
  // Row-contiguous epilogue: a wave passes its 32 x 64 accumulator chunks through a private LDS slab ([32][64 + 4] floats, in the ring's
  // memory) and works on them as rows -- 16 lanes x 16 bytes per row, four rows per instruction -- so that the output, the `+=` operand and
  // the bypass addend move as 256-byte row segments in dwordx4 accesses, the loads of a chunk's eight row groups issued together.  (Straight
  // from the accumulators a lane owns one column: 4-byte accesses, 128 of them per thread and operand, each behind its own bounds test --
  // the .linear backward-data GEMM, which adds the bypass derivative, took 675 us where the same product without an addend took 280.)
  // Chunks at the ragged right edge, and operands that are not 16-byte aligned, take the element-wise path.
  constexpr int CW = 64, LDW = CW + 4, NCH = (TN + 1) / 2;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && p.ldc % 4 == 0 && p.tap_off_c % 4 == 0 &&
                      (!p.add || ((reinterpret_cast<uintptr_t>(p.add) & 15) == 0 && p.ldadd % 4 == 0)) &&
                      (p.init_mode != 1 || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
  float cs1[TN], cs2[TN];      // column sums / sums of squares of what this lane stores, element-wise chunks (p.colstats)
  float vs1[NCH][4], vs2[NCH][4];  // the same, row-contiguous chunks: columns 4 (lane & 15) .. + 3 of the chunk
#pragma unroll
  for (int j = 0; j < TN; j++) cs1[j] = cs2[j] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int e = 0; e < 4; e++) vs1[c][e] = vs2[c][e] = 0.f;
  __builtin_amdgcn_s_barrier();  // every wave has read its last fragments: the ring's memory is free
  float *scr = reinterpret_cast<float *>(smem) + wave * (32 * LDW);
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      const int j0 = ch * 2, nj = (TN - j0) < 2 ? (TN - j0) : 2, ncol = nj * 32;
      const int nw = n0 + (wn * TN + j0) * 32;  // first column of the chunk
      if (nw >= p.N) continue;
      if (vec_ok && nw + ncol <= p.N) {
#pragma unroll
        for (int jj = 0; jj < nj; jj++)
#pragma unroll
          for (int r = 0; r < 16; r++) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDW + jj * 32 + li] = acc[i][j0 + jj][r];
        const bool col_on = c4 < ncol;
        const int n = nw + c4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.init_mode == 1 && col_on) bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
        const int mrow0 = m0 + (wm * TM + i) * 32 + rr;
        float4 addv[8], cold[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int m = mrow0 + 4 * q;
          addv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          cold[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col_on && m < p.M) {
            if (p.add && m >= p.add_lo && m < p.add_hi) addv[q] = *reinterpret_cast<const float4 *>(p.add + (long long)(m - p.add_lo) * p.ldadd + n);
            if (p.init_mode == 0) cold[q] = *reinterpret_cast<const float4 *>(p.C + (long long)m * p.ldc + (long long)tap * p.tap_off_c + n);
          }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int m = mrow0 + 4 * q;
          if (!(col_on && m < p.M)) continue;
          const float4 a4 = *reinterpret_cast<const float4 *>(scr + (rr + 4 * q) * LDW + c4);
          float v[4] = {a4.x * sc + bias4.x + cold[q].x + p.add_scale * addv[q].x, a4.y * sc + bias4.y + cold[q].y + p.add_scale * addv[q].y,
                        a4.z * sc + bias4.z + cold[q].z + p.add_scale * addv[q].z, a4.w * sc + bias4.w + cold[q].w + p.add_scale * addv[q].w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (p.relu) v[e] = floor_keep_nan(v[e], 0.f);
            vs1[ch][e] += v[e];
            vs2[ch][e] += v[e] * v[e];
          }
          *reinterpret_cast<float4 *>(p.C + (long long)m * p.ldc + (long long)tap * p.tap_off_c + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < nj; jj++) {
          const int j = j0 + jj;
          const int n = n0 + (wn * TN + j) * 32 + li;
          if (n >= p.N) continue;
          const float bias = p.init_mode == 1 ? p.bias[n] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            float *c = p.C + (long long)m * p.ldc + (long long)tap * p.tap_off_c + n;
            float v = acc[i][j][r] * sc + bias;
            if (p.init_mode == 0) v += *c;
            if (p.add && m >= p.add_lo && m < p.add_hi) v += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n];
            if (p.relu) v = floor_keep_nan(v, 0.f);
            *c = v;
            cs1[j] += v;
            cs2[j] += v * v;
          }
        }
      }
    }
  if (p.colstats) {  // one partial row per row tile: the lanes that hold the same columns first, the WM wave rows through LDS (fixed order)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      cs1[j] += __shfl_xor(cs1[j], 32, 64);
      cs2[j] += __shfl_xor(cs2[j], 32, 64);
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        vs1[c][e] += __shfl_xor(vs1[c][e], 16, 64);
        vs1[c][e] += __shfl_xor(vs1[c][e], 32, 64);
        vs2[c][e] += __shfl_xor(vs2[c][e], 16, 64);
        vs2[c][e] += __shfl_xor(vs2[c][e], 32, 64);
      }
    __syncthreads();  // (every wave is done with its slab)
    float *red = reinterpret_cast<float *>(smem);  // [wm][2][BN]
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      const int j0 = ch * 2, nj = (TN - j0) < 2 ? (TN - j0) : 2, ncol = nj * 32;
      const int nw = n0 + (wn * TN + j0) * 32;
      if (nw >= p.N) continue;
      if (vec_ok && nw + ncol <= p.N) {
        if (rr == 0 && c4 < ncol) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            red[(wm * 2 + 0) * BN + (wn * TN + j0) * 32 + c4 + e] = vs1[ch][e];
            red[(wm * 2 + 1) * BN + (wn * TN + j0) * 32 + c4 + e] = vs2[ch][e];
          }
        }
      } else if (lh == 0) {
#pragma unroll
        for (int jj = 0; jj < nj; jj++) {
          const int nl = (wn * TN + j0 + jj) * 32 + li;
          red[(wm * 2 + 0) * BN + nl] = cs1[j0 + jj];
          red[(wm * 2 + 1) * BN + nl] = cs2[j0 + jj];
        }
      }
    }
    __syncthreads();
    for (int nl = t; nl < BN; nl += NT) {
      const int n = n0 + nl;
      if (n >= p.N) continue;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; w++) {
        s1 += red[(w * 2 + 0) * BN + nl];
        s2 += red[(w * 2 + 1) * BN + nl];
      }
      p.colstats[(long long)tile_m * p.N + n] = s1;
      p.colstats[(long long)(p.colstats_stride + tile_m) * p.N + n] = s2;
    }
  }
}

__global__ __launch_bounds__(256) void planes_splitk_finish_kernel(const PlanesGemmArgs p) {
  const long long total = (long long)p.M * p.N;
  float sc = 1.0f;
  if (p.np == 2) sc = (p.scale_a ? p.scale_a[1] : 1.0f) * (p.scale_b ? p.scale_b[1] : 1.0f);
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int m = (int)(e / p.N), n = (int)(e % p.N);
    float v = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < p.ksplit; sp++) v += p.partial[(long long)sp * p.partial_stride + (long long)m * p.ldp_m + n];
    v *= sc;
    float *c = p.C + (long long)m * p.ldc + n;
    if (p.init_mode == 1) v += p.bias[n];
    else if (p.init_mode == 0) v += *c;
    if (p.add && m >= p.add_lo && m < p.add_hi) v += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n];
    if (p.relu) v = floor_keep_nan(v, 0.f);
    *c = v;
  }
}

template <int NP, int WM, int WN, int TM, int TN, bool DB = false, bool ATR = false>
hipError_t launch(const PlanesGemmArgs &a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NT = WM * WN * 64, PIECES = (NP * BM * 32 + NP * BN * 32) / 16, PPT = (PIECES + NT - 1) / NT;
  constexpr size_t lds = (size_t)stages_of<NP, BM>() * PPT * NT * 16;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void *)planes_gemm_kernel<NP, WM, WN, TM, TN, DB, ATR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  const int nblk = ntm * ntn * (a.ntap > 1 ? a.ntap : 1) * (a.ksplit > 1 ? a.ksplit : 1);
  hipLaunchKernelGGL((planes_gemm_kernel<NP, WM, WN, TM, TN, DB, ATR>), dim3(nblk), dim3(WM * WN * 64), lds, s, a, ntm, ntn);
  return hipGetLastError();
}

}  // namespace

long long g_planes_routed_rows = 0, g_planes_routed_wgrad = 0;
static thread_local const PlanesOperand *g_hint_a = nullptr, *g_hint_b = nullptr;
PlanesHintScope::PlanesHintScope(const PlanesOperand *a, const PlanesOperand *b) : prev_a(g_hint_a), prev_b(g_hint_b) {
  g_hint_a = a;
  g_hint_b = b;
}
PlanesHintScope::~PlanesHintScope() {
  g_hint_a = prev_a;
  g_hint_b = prev_b;
}
const PlanesOperand *planes_hint_a() { return g_hint_a; }
const PlanesOperand *planes_hint_b() { return g_hint_b; }

size_t planes_bytes(int np, long long rows_total, long long k_blocks) { return (size_t)(k_blocks * np * rows_total * 32); }
size_t planes_sumsq_ws_bytes() { return sizeof(double) * kSumsqBlocks; }

__device__ unsigned g_bound_checks = 0, g_bound_violations = 0;
__global__ void planes_check_bound_kernel(const double *partial, int nb, const float *rec) {
  __shared__ double red[256];
  double a = 0;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  atomicAdd(&g_bound_checks, 1u);
  if (!(sqrt(red[0]) <= (double)rec[2])) atomicAdd(&g_bound_violations, 1u);
}
hipError_t planes_check_bound(MatView x, const float *rec, void *sumsq_ws, hipStream_t s) {
  const bool vec4 = (reinterpret_cast<uintptr_t>(x.data) & 15) == 0 && x.stride % 4 == 0;
  if (vec4) hipLaunchKernelGGL(planes_sumsq4_kernel, dim3(kSumsqBlocks), dim3(256), 0, s, x, (double *)sumsq_ws);
  else hipLaunchKernelGGL(planes_sumsq_kernel, dim3(kSumsqBlocks), dim3(256), 0, s, x, (double *)sumsq_ws);
  hipLaunchKernelGGL(planes_check_bound_kernel, dim3(1), dim3(256), 0, s, (const double *)sumsq_ws, kSumsqBlocks, rec);
  return hipGetLastError();
}

hipError_t planes_pad(int np, void *P, long long k_blocks, long long R, int lead, long long rows, hipStream_t s) {
  if (!P || R <= rows) return hipSuccess;
  hipLaunchKernelGGL(planes_pad_kernel, dim3(grid_for(k_blocks * np * (R - rows) * 2, 256)), dim3(256), 0, s, P, k_blocks * np, R, lead, rows);
  return hipGetLastError();
}

hipError_t planes_scale_bound(const double *fro2_bound, int blocks, double numel, float mul, float add_coef, const float *add_rec, float *rec, hipStream_t s) {
  hipLaunchKernelGGL(planes_scale_kernel, dim3(1), dim3(256), 0, s, fro2_bound, blocks, numel, rec, mul, add_coef, add_rec);
  return hipGetLastError();
}

hipError_t planes_split(const PlanesSplitArgs &a, hipStream_t s) {
  const MatView &x = a.x;
  if (x.rows <= 0 || x.cols <= 0 || (!a.P && !a.PT)) return hipSuccess;
  if (a.np != 2 && a.np != 3) return hipErrorInvalidValue;
  const long long nkb = planes_kblocks(x.cols), nkbt = planes_t_kblocks(x.rows);
  ProfHbmRange prof(7, (double)x.rows * x.cols * (4.0 + 2.0 * a.np * ((a.P ? 1 : 0) + (a.PT ? 1 : 0))), s);  // the matrix read once, each layout written once
  const bool vec4 = (reinterpret_cast<uintptr_t>(x.data) & 15) == 0 && x.stride % 4 == 0;
  const bool small = (long long)x.rows * x.cols <= (4LL << 20);
  const double *sq_partial = nullptr;
  int sq_nb = 0;
  if (a.np == 2) {
    if (!a.scale || !a.sumsq_ws) return hipErrorInvalidValue;
    if (a.fro2_bound && a.fro2_blocks > 0) {  // the producer's finalize launch left a bound: no pass over the matrix
      hipLaunchKernelGGL(planes_scale_kernel, dim3(1), dim3(256), 0, s, a.fro2_bound, a.fro2_blocks, (double)x.rows * x.cols, a.scale, a.fro_mul, a.add_coef, a.add_rec);
      if (options().planes_check_bound) {
        hipError_t ce = planes_check_bound(x, a.scale, a.sumsq_ws, s);
        if (ce != hipSuccess) return ce;
      }
    } else {
      // (a small matrix gets as many norm-pass blocks as it has 16 K-element pieces, and its split forms the scale itself: two launches, not three)
      sq_nb = small ? (int)std::max<long long>(1, std::min<long long>(kSumsqBlocks, ((long long)x.rows * x.cols + 16383) / 16384)) : kSumsqBlocks;
      if (vec4) hipLaunchKernelGGL(planes_sumsq4_kernel, dim3(sq_nb), dim3(256), 0, s, x, (double *)a.sumsq_ws);
      else hipLaunchKernelGGL(planes_sumsq_kernel, dim3(sq_nb), dim3(256), 0, s, x, (double *)a.sumsq_ws);
      if (small) sq_partial = (const double *)a.sumsq_ws;
      else hipLaunchKernelGGL(planes_scale_kernel, dim3(1), dim3(256), 0, s, (const double *)a.sumsq_ws, sq_nb, (double)x.rows * x.cols, a.scale, 1.0f, 0.0f, (const float *)nullptr);
    }
  }
  if (a.P && a.R > x.rows && !a.pads_done)
    hipLaunchKernelGGL(planes_pad_kernel, dim3(grid_for(nkb * a.np * (a.R - x.rows) * 2, 256)), dim3(256), 0, s, a.P, nkb * a.np, a.R, a.lead, (long long)x.rows);
  if (a.PT && a.Rt > x.cols && !a.pads_done)
    hipLaunchKernelGGL(planes_pad_kernel, dim3(grid_for(nkbt * a.np * (a.Rt - x.cols) * 2, 256)), dim3(256), 0, s, a.PT, nkbt * a.np, a.Rt, 0, (long long)x.cols);
  const dim3 grid((unsigned)((x.rows + 63) / 64), (unsigned)((x.cols + 63) / 64));
  if (a.np == 2)
    hipLaunchKernelGGL(planes_split_kernel<2>, grid, dim3(256), 0, s, x.data, (long long)x.stride, x.rows, x.cols, (const float *)a.scale, a.lead, a.R, a.P, a.Rt, a.PT, vec4 ? 1 : 0,
                       sq_partial, sq_nb, a.scale, a.col_coef, a.col_coef_period);
  else
    hipLaunchKernelGGL(planes_split_kernel<3>, grid, dim3(256), 0, s, x.data, (long long)x.stride, x.rows, x.cols, (const float *)nullptr, a.lead, a.R, a.P, a.Rt, a.PT, vec4 ? 1 : 0,
                       (const double *)nullptr, 0, (float *)nullptr, a.col_coef, a.col_coef_period);
  return hipGetLastError();
}

struct PlanesSplitGroup {
  std::vector<PlanesSplitItem> items;
  std::vector<int> sq_first, sp_first;
  PlanesSplitItem *d_items = nullptr;
  int *d_sq_first = nullptr, *d_sp_first = nullptr;
  double *d_partial = nullptr;
  int cap = 0, cap_partial = 0;
};
void planes_split_group_destroy(PlanesSplitGroup *g) {
  if (!g) return;
  for (void *p : {(void *)g->d_items, (void *)g->d_sq_first, (void *)g->d_sp_first, (void *)g->d_partial})
    if (p) hipFree(p);
  delete g;
}
bool planes_split_group_ok(const PlanesSplitArgs &a) {
  const MatView &x = a.x;
  return a.np == 2 && x.rows > 0 && x.cols > 0 && (a.P || a.PT) && a.scale && !(a.fro2_bound && a.fro2_blocks > 0) && a.pads_done &&
         (long long)x.rows * x.cols <= (4LL << 20) && (reinterpret_cast<uintptr_t>(x.data) & 15) == 0 && x.stride % 4 == 0;
}
hipError_t planes_split_group(const std::vector<PlanesSplitArgs> &v, PlanesSplitGroup **cache, hipStream_t s) {
  if (v.empty()) return hipSuccess;
  if (!*cache) *cache = new PlanesSplitGroup();
  PlanesSplitGroup &g = **cache;
  std::vector<PlanesSplitItem> items(v.size());
  std::vector<int> sqf(v.size() + 1, 0), spf(v.size() + 1, 0);
  double bytes = 0;
  for (size_t i = 0; i < v.size(); i++) {
    const PlanesSplitArgs &a = v[i];
    if (!planes_split_group_ok(a)) return hipErrorInvalidValue;
    PlanesSplitItem it;
    memset(&it, 0, sizeof(it));
    it.X = a.x.data; it.ld = a.x.stride; it.rows = a.x.rows; it.cols = a.x.cols; it.lead = a.lead; it.R = a.R; it.Rt = a.Rt; it.P = a.P; it.PT = a.PT;
    it.scale = a.scale; it.col_coef = a.col_coef; it.col_coef_period = a.col_coef_period; it.vec_ok = 1;
    it.sq_nb = (int)std::max<long long>(1, std::min<long long>(kSumsqBlocks, ((long long)a.x.rows * a.x.cols + 16383) / 16384));  // (as planes_split for a small matrix)
    it.sq_first = sqf[i];
    it.sp_gx = (a.x.rows + 63) / 64;
    it.sp_first = spf[i];
    sqf[i + 1] = sqf[i] + it.sq_nb;
    spf[i + 1] = spf[i] + it.sp_gx * ((a.x.cols + 63) / 64);
    items[i] = it;
    bytes += (double)a.x.rows * a.x.cols * (4.0 + 4.0 * ((a.P ? 1 : 0) + (a.PT ? 1 : 0)));
  }
  const bool same = g.items.size() == items.size() && memcmp(g.items.data(), items.data(), sizeof(PlanesSplitItem) * items.size()) == 0;
  if (!same) {
    if (g.cap < (int)items.size()) {
      for (void *p : {(void *)g.d_items, (void *)g.d_sq_first, (void *)g.d_sp_first})
        if (p) hipFree(p);
      hipError_t e = hipMalloc((void **)&g.d_items, sizeof(PlanesSplitItem) * items.size());
      if (e == hipSuccess) e = hipMalloc((void **)&g.d_sq_first, sizeof(int) * (items.size() + 1));
      if (e == hipSuccess) e = hipMalloc((void **)&g.d_sp_first, sizeof(int) * (items.size() + 1));
      if (e != hipSuccess) return e;
      g.cap = (int)items.size();
    }
    if (g.cap_partial < sqf.back()) {
      if (g.d_partial) hipFree(g.d_partial);
      hipError_t e = hipMalloc((void **)&g.d_partial, sizeof(double) * sqf.back());
      if (e != hipSuccess) return e;
      g.cap_partial = sqf.back();
    }
    g.items = items; g.sq_first = sqf; g.sp_first = spf;
    hipError_t e = hipMemcpyAsync(g.d_items, g.items.data(), sizeof(PlanesSplitItem) * items.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(g.d_sq_first, g.sq_first.data(), sizeof(int) * sqf.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(g.d_sp_first, g.sp_first.data(), sizeof(int) * spf.size(), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
  }
  ProfHbmRange prof(7, bytes, s);
  hipLaunchKernelGGL(planes_sumsq4_group_kernel, dim3(sqf.back()), dim3(256), 0, s, g.d_items, g.d_sq_first, (int)items.size(), g.d_partial);
  hipLaunchKernelGGL(planes_split_group_kernel, dim3(spf.back()), dim3(256), 0, s, g.d_items, g.d_sp_first, (int)items.size(), (const double *)g.d_partial);
  return hipGetLastError();
}

hipError_t planes_splitk_finish(const PlanesGemmArgs &a, hipStream_t s) {
  if (a.ksplit < 2 || a.ntap > 1 || a.ldp_n != 1 || !a.partial) return hipErrorInvalidValue;
  const long long total = (long long)a.M * a.N;
  hipLaunchKernelGGL(planes_splitk_finish_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, a);
  return hipGetLastError();
}

int planes_gemm_tile_rows(int N) { return 256; }
int planes_gemm_tile_cols(int N) {
  const int w128 = ((N + 127) / 128) * 128 - N, w160 = ((N + 159) / 160) * 160 - N;
  if (w160 < w128) return 160;
  return N % 256 == 0 ? 256 : 128;  // (256 x 256 tiles: 2/3 of the 256 x 128 tile's operand bytes per flop)
}

// Short reductions (K = 320: the .affine forward and .linear backward-data GEMMs, 1536 columns out): a tile's 256 KB of output and
// its 20 K steps of loads both run at what ONE CU can move (~20-50 GB/s), one after the other when the CU holds a single block.
// 128 x 256 tiles of four waves take 72 KB of LDS: two blocks per CU, one storing while the other multiplies.
static bool small_row_tile(const PlanesGemmArgs &a) {
  if (a.np != 2 || a.a_rows_as_k || a.ntap > 1 || a.ksplit > 1 || a.M <= 4096 || planes_gemm_tile_cols(a.N) != 256) return false;
  int nkb = 0;
  for (int i = 0; i < a.nseg; i++) nkb += a.seg[i].nkb;
  return nkb <= 40;
}
int planes_gemm_launch_tile_rows(const PlanesGemmArgs &a) { return small_row_tile(a) ? 128 : 256; }

hipError_t planes_gemm(const PlanesGemmArgs &a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.nseg <= 0) return hipSuccess;
  if (a.np != 2 && a.np != 3) return hipErrorInvalidValue;
  if (a.ntap > 1 && a.nseg != 1) return hipErrorInvalidValue;
  // 160-wide tiles for the TDNN-F bottleneck, 256- / 128-wide otherwise; 8 waves (two per SIMD)
  // Measured on MI355X for np = 3 (tools/planes_bench.py, f32-equivalent TFLOP/s; exact-f32 kernel of gemm_f32.hip in brackets):
  //   256 x 256 tile, 8 waves of 64 x 128:  N = 1536, K = 2 x 1536: 227 [130];  K = 2 x 160 (.affine forward): 162 [116]
  //   256 x 160 tile, 8 waves of 32 x 160:  N = 160, K = 2 x 1536 (.linear forward): 151 [117]
  //   256 x 128 tile, 4 x 2 waves of 64 x 64: 201 / 150.  Four waves of 64 rows x the tile's width: 194 / 121 (one wave per SIMD
  //   leaves every LDS / barrier wait exposed); fragments double-buffered in registers: no gain, spills on the wide tiles.
  const int bn = planes_gemm_tile_cols(a.N);
  if (a.a_rows_as_k) {  // the A operand through transposing LDS reads (weight gradients from row-major planes)
    if (a.np == 3) {
      if (bn == 160) return launch<3, 8, 1, 1, 5, false, true>(a, s);
      if (bn == 256) return launch<3, 4, 2, 2, 4, false, true>(a, s);
      return launch<3, 4, 2, 2, 2, false, true>(a, s);
    }
    if (bn == 160) return launch<2, 8, 1, 1, 5, false, true>(a, s);
    if (bn == 256) return launch<2, 4, 2, 2, 4, false, true>(a, s);
    return launch<2, 4, 2, 2, 2, false, true>(a, s);
  }
  if (a.np == 3) {
    if (bn == 160) return launch<3, 8, 1, 1, 5>(a, s);
    if (bn == 256) return launch<3, 4, 2, 2, 4>(a, s);
    return launch<3, 4, 2, 2, 2>(a, s);
  }
  if (bn == 160) return launch<2, 8, 1, 1, 5>(a, s);
  if (bn == 256) return small_row_tile(a) ? launch<2, 2, 2, 2, 4>(a, s) : launch<2, 4, 2, 2, 4>(a, s);
  return launch<2, 4, 2, 2, 2>(a, s);
}

}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

size_t tdnnf_planes_bytes(int num_planes, long long rows_total, long long k_blocks) {
  if ((num_planes != 2 && num_planes != 3) || rows_total <= 0 || k_blocks <= 0) return 0;
  return planes_bytes(num_planes, rows_total, k_blocks);
}
size_t tdnnf_planes_split_workspace_bytes(void) { return planes_sumsq_ws_bytes(); }
// option planes_check_bound: how many bound-derived scales were checked against the measured norm, and how many bounds were too small
// (synchronises the device)
void tdnnf_planes_bound_checks(long long *checks, long long *violations) {
  unsigned c = 0, v = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_bound_checks), sizeof(c));
  (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bound_violations), sizeof(v));
  if (checks) *checks = c;
  if (violations) *violations = v;
}
void tdnnf_planes_routed(long long *rows_gemms, long long *weight_gradients) {
  if (rows_gemms) *rows_gemms = g_planes_routed_rows;
  if (weight_gradients) *weight_gradients = g_planes_routed_wgrad;
}

int tdnnf_planes_split(int num_planes, const tdnnf_mat *x, int lead_rows, long long rows_total, void *planes, long long t_rows_total, void *planes_t,
                       float *scale_dev, void *workspace_dev, tdnnf_stream stream) {
  TDNNF_REQUIRE((num_planes == 2 || num_planes == 3) && mat_ok(x) && x->cols > 0 && lead_rows >= 0 && (planes || planes_t), "planes_split: bad arguments (2 or 3 planes)");
  TDNNF_REQUIRE(!planes || rows_total >= (long long)lead_rows + x->rows, "planes_split: rows_total must cover lead + rows");
  TDNNF_REQUIRE(!planes_t || t_rows_total >= x->cols, "planes_split: t_rows_total must cover the matrix's columns");
  TDNNF_REQUIRE(((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(planes_t)) & 15) == 0, "planes_split: the plane buffers must be 16-byte aligned");
  TDNNF_REQUIRE(num_planes == 3 || (scale_dev && workspace_dev), "planes_split: two f16 planes need the scale output and the workspace");
  PlanesSplitArgs a;
  a.np = num_planes; a.x = view(x); a.P = planes; a.lead = lead_rows; a.R = rows_total; a.PT = planes_t; a.Rt = t_rows_total; a.scale = scale_dev; a.sumsq_ws = workspace_dev;
  TDNNF_HIP(planes_split(a, (hipStream_t)stream));
  return TDNNF_OK;
}

static int planes_gemm_abi(int num_planes, const void *a_planes, long long a_rows_total, const float *a_scale_dev, const void *b_planes, long long b_rows_total,
                           const float *b_scale_dev, int num_segments, const long long *a_row, const long long *b_row, const int *a_first_col, const int *b_first_col,
                           const int *seg_cols, const float *bias, int init_mode, int relu, const tdnnf_mat *add, float add_scale, int add_first_row, float *colstats,
                           int *colstats_rows, tdnnf_mat *c, tdnnf_stream stream) {
  TDNNF_REQUIRE((num_planes == 2 || num_planes == 3) && a_planes && b_planes && mat_ok(c) && num_segments >= 1 && num_segments <= 16 && a_row && a_first_col &&
                    b_first_col && seg_cols,
                "planes_gemm: bad arguments (2 or 3 planes, 1..16 segments)");
  TDNNF_REQUIRE(init_mode >= 0 && init_mode <= 2 && (init_mode != 1 || bias), "planes_gemm: init_mode 0 (+=), 1 (bias), 2 (=)");
  TDNNF_REQUIRE(!add || (mat_ok(add) && add->cols == c->cols && add_first_row >= 0), "planes_gemm: the addend must have the output's columns");
  PlanesGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.np = num_planes;
  a.A = a_planes; a.RA = a_rows_total; a.B = b_planes; a.RB = b_rows_total; a.scale_a = a_scale_dev; a.scale_b = b_scale_dev;
  a.C = c->data; a.ldc = c->stride; a.M = c->rows; a.N = c->cols;
  a.bias = bias; a.init_mode = init_mode; a.relu = relu; a.nseg = num_segments;
  if (add) {
    a.add = add->data; a.ldadd = add->stride; a.add_scale = add_scale; a.add_lo = add_first_row; a.add_hi = add_first_row + add->rows;
  }
  const int BM = planes_gemm_tile_rows(c->cols), BN = planes_gemm_tile_cols(c->cols);
  for (int i = 0; i < num_segments; i++) {
    const long long br = b_row ? b_row[i] : 0;
    TDNNF_REQUIRE(a_first_col[i] % 16 == 0 && b_first_col[i] % 16 == 0 && seg_cols[i] > 0 && a_row[i] >= 0 && br >= 0, "planes_gemm: segment %d: columns must start on a multiple of 16", i);
    TDNNF_REQUIRE(a_row[i] + (long long)((c->rows + BM - 1) / BM) * BM <= a_rows_total,
                  "planes_gemm: segment %d reads rows %lld..%lld of an A plane buffer of %lld rows (tail rows must cover the %d-row tile)", i, a_row[i],
                  a_row[i] + (long long)((c->rows + BM - 1) / BM) * BM, a_rows_total, BM);
    TDNNF_REQUIRE(br + (long long)((c->cols + BN - 1) / BN) * BN <= b_rows_total, "planes_gemm: segment %d: the B plane buffer needs %lld rows (output columns padded to the %d-column tile)", i,
                  br + (long long)((c->cols + BN - 1) / BN) * BN, BN);
    a.seg[i].a_row = a_row[i];
    a.seg[i].b_row = br;
    a.seg[i].a_kb0 = a_first_col[i] / 16;
    a.seg[i].b_kb0 = b_first_col[i] / 16;
    a.seg[i].nkb = (seg_cols[i] + 15) / 16;
  }
  if (colstats) {  // one partial row per row tile of THIS launch (the tile height depends on the shape), sums first, sums of squares behind them
    TDNNF_REQUIRE(colstats_rows, "planes_gemm: colstats_rows must be given with colstats");
    const int tile_rows = planes_gemm_launch_tile_rows(a);
    a.colstats = colstats;
    a.colstats_stride = (c->rows + tile_rows - 1) / tile_rows;
    *colstats_rows = (int)a.colstats_stride;
  }
  TDNNF_HIP(planes_gemm(a, (hipStream_t)stream));
  return TDNNF_OK;
}

int tdnnf_planes_gemm(int num_planes, const void *a_planes, long long a_rows_total, const float *a_scale_dev, const void *b_planes, long long b_rows_total,
                      const float *b_scale_dev, int num_segments, const long long *a_row, const long long *b_row, const int *a_first_col, const int *b_first_col,
                      const int *seg_cols, const float *bias, int init_mode, int relu, tdnnf_mat *c, tdnnf_stream stream) {
  return planes_gemm_abi(num_planes, a_planes, a_rows_total, a_scale_dev, b_planes, b_rows_total, b_scale_dev, num_segments, a_row, b_row, a_first_col, b_first_col, seg_cols,
                         bias, init_mode, relu, nullptr, 0.f, 0, nullptr, nullptr, c, stream);
}

int tdnnf_planes_gemm_epilogue(int num_planes, const void *a_planes, long long a_rows_total, const float *a_scale_dev, const void *b_planes, long long b_rows_total,
                               const float *b_scale_dev, int num_segments, const long long *a_row, const long long *b_row, const int *a_first_col,
                               const int *b_first_col, const int *seg_cols, const float *bias, int init_mode, int relu, const tdnnf_mat *add, float add_scale,
                               int add_first_row, float *colstats, int *colstats_rows, tdnnf_mat *c, tdnnf_stream stream) {
  return planes_gemm_abi(num_planes, a_planes, a_rows_total, a_scale_dev, b_planes, b_rows_total, b_scale_dev, num_segments, a_row, b_row, a_first_col, b_first_col, seg_cols,
                         bias, init_mode, relu, add, add_scale, add_first_row, colstats, colstats_rows, c, stream);
}

}  // extern "C"
