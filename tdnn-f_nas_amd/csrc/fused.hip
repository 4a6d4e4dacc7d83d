// fused.hip -- fused HBM-bound passes used by the chain trainer (net.hip) between the MFMA GEMMs.
//
// The reference runs ReLU, BatchNorm, (dropout) and the bypass Sum() of a tdnnf-layer
// (/root/reference/steps/libs/nnet3/xconfig/composite_layers.py:177-213) as separate components, each a
// chain of CuMatrix passes (~12 full passes over N x 1536 per layer and direction).  Here:
//   forward : GEMM epilogue does bias + ReLU; one reduction pass for the BatchNorm statistics; ONE pass
//             computes z = (x - mean) * scale and out = z + bypass * prev (z is not stored);
//   backward: one reduction pass (sum z*dz, sum dz [, ReLU statistics]), ONE pass producing the derivative
//             w.r.t. the affine output (BatchNorm backward, ReLU mask, self-repair) together with the
//             per-column partial sums of the bias gradient.
// Arithmetic is the components' (nnet-normalize-component.cc:421-452,505-542; nnet-simple-component.cc:
// 958-1074), only the number of trips through HBM changes.  All column reductions are two-stage and
// deterministic.
#include <algorithm>

#include "common.h"
#include "fused.h"
#include "gemm_f32.h"
#include "planes_gemm.h"

namespace tdnnf {
namespace {

__device__ __forceinline__ void ld(const float *p, float (&v)[4], bool vec) {
  if (vec) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}
__device__ __forceinline__ void st(float *p, const float (&v)[4], bool vec) {
  if (vec) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else p[0] = v[0];
}

// four consecutive columns c .. c + 3 (c % 4 == 0) of row `ra` of a matrix as two f16 planes of v * s in the P16 layout (planes_gemm.h):
// K block c / 16, k = c % 16 of the row's 32-byte record, halves swapped when bit 3 of the row is set; `R` rows per chunk
__device__ __forceinline__ void store_planes4(const PlanesSink &pk, float ps, long long ra, int c, const float (&v)[4]) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float w = v[j] * ps;
    hi[j] = (_Float16)w;
    lo[j] = (_Float16)(w - (float)hi[j]);
  }
  const int kb = c >> 4, k = c & 15, half = (k >> 3) ^ (int)((ra >> 3) & 1);
  _Float16 *dst = reinterpret_cast<_Float16 *>(pk.P) + (((long long)kb * 2) * pk.R + ra) * 16 + half * 8 + (k & 7);
  *reinterpret_cast<h4 *>(dst) = hi;
  *reinterpret_cast<h4 *>(dst + pk.R * 16) = lo;
}

// out = (x - mean) * scale + bypass * prev   (prev.data may be null).  Views may be "super rows"
// (cols = k * D): the column parameters repeat with period D.
template <int VEC, bool PLANES>
__global__ __launch_bounds__(256) void bn_apply_bypass_kernel(MatView x, const float *mean, const float *scale, int D, int period,
                                                              MatView prev, float bypass, MatView out, const float *mask, int B, PlanesSink pk, int rev) {
  const int cv = x.cols / VEC;
  const float ps = PLANES ? pk.rec[0] : 1.0f;
  // PLANES: a wave covers 8 rows x 32 columns (not 1 row x 256): its plane stores are then two runs of 8 consecutive 32-byte row records per
  // plane (256 contiguous bytes) instead of 32 scattered records, and the f32 accesses are still whole 128-byte lines
  const int ncb = PLANES ? cv / 8 : 1;
  const long long total = PLANES ? (long long)((x.rows + 7) / 8) * 8 * cv : (long long)x.rows * cv;
  for (long long e0 = blockIdx.x * 256LL + threadIdx.x; e0 < total; e0 += gridDim.x * 256LL) {
    // (rev: from the matrix's last rows to its first -- what the GEMM that produced x wrote last is still in L2 / the Infinity Cache when this
    // pass starts, and what this pass writes last, the first rows, is what the next GEMM reads first)
    const long long e = rev ? total - 1 - e0 : e0;
    int r, c;
    if (PLANES) {
      const long long blk = e >> 6;
      r = (int)(blk / ncb) * 8 + (int)((e >> 3) & 7);
      c = ((int)(blk % ncb) * 8 + (int)(e & 7)) * VEC;
      if (r >= x.rows) continue;
    } else {
      r = (int)(e / cv);
      c = (int)(e % cv) * VEC;
    }
    const int cd = c % period;
    if (cd >= D) continue;  // row padding inside a super row
    float xv[4], pv[4] = {0, 0, 0, 0}, o[4];
    ld(x.data + (long long)r * x.stride + c, xv, VEC == 4);
    if (prev.data) ld(prev.data + (long long)r * prev.stride + c, pv, VEC == 4);
    // GeneralDropoutComponent between the BatchNorm and the bypass sum: one mask row per sequence, shared over time
    const float *mk = mask ? mask + (long long)(x.cols > D ? c / period : r % B) * D + cd : nullptr;
    // the column parameters as whole float4s (cd is a multiple of 4 here): 5 memory instructions per output float4 instead of 11
    float mu[4], sc[4], mv[4] = {1.f, 1.f, 1.f, 1.f};
    ld(mean + cd, mu, VEC == 4);
    ld(scale + cd, sc, VEC == 4);
    if (mk) ld(mk, mv, VEC == 4);
#pragma unroll
    for (int j = 0; j < VEC; j++) o[j] = (xv[j] - mu[j]) * sc[j] * mv[j] + bypass * pv[j];
    st(out.data + (long long)r * out.stride + c, o, VEC == 4);
    if constexpr (PLANES && VEC == 4) store_planes4(pk, ps, r, c, o);  // the same four values as two f16 planes of o * s (no lead rows)
  }
}

// Column-strip kernels: block = 64 (x VEC) columns x 4 row lanes, one chunk of rows per blockIdx.y.
// partial layout: [quantity][chunk][col].
template <int VEC, bool RELU_STATS>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(MatView x, MatView dz, const float *mean, const float *scale,
                                                                 int rows_per_chunk, int chunks, float *partial, const float *mask, int B) {
  __shared__ float red[5][4][64 * 4 + 4];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tc) * VEC;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(x.rows, r0 + rows_per_chunk);
  float s[5][4] = {};  // sum z dz, sum dz, sum dz^2 (StoreBackpropStats), [sum x, count x > 0]
  if (col < x.cols) {
    float mu[4], sc[4];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      mu[j] = mean[col + j];
      sc[j] = scale[col + j];
    }
    for (int r = r0 + tr; r < r1; r += 4) {
      float xv[4], dv[4];
      ld(x.data + (long long)r * x.stride + col, xv, VEC == 4);
      ld(dz.data + (long long)r * dz.stride + col, dv, VEC == 4);
      if (mask) {  // derivative through the dropout mask first (row r belongs to sequence r % B)
        const float *mk = mask + (long long)(r % B) * x.cols + col;
#pragma unroll
        for (int j = 0; j < VEC; j++) dv[j] *= mk[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const float z = (xv[j] - mu[j]) * sc[j];
        s[0][j] += z * dv[j];
        s[1][j] += dv[j];
        s[2][j] += dv[j] * dv[j];
        if (RELU_STATS) {
          s[3][j] += xv[j];
          s[4][j] += xv[j] > 0.f ? 1.f : 0.f;
        }
      }
    }
  }
  constexpr int NQ = RELU_STATS ? 5 : 3;
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int j = 0; j < VEC; j++) red[q][tr][tc * VEC + j] = s[q][j];
  __syncthreads();
  if (tr == 0 && col < x.cols) {
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const int k = tc * VEC + j;
        partial[((long long)q * chunks + blockIdx.y) * x.cols + col + j] = (red[q][0][k] + red[q][1][k]) + (red[q][2][k] + red[q][3][k]);
      }
  }
}

// memo rows 3 (var_deriv_mod) and 4 (temp) from the partials (nnet-normalize-component.cc:520-526);
// optionally the ReLU statistics [count, value_sum[D], deriv_sum[D]] (StoreStatsInternal) and, into oderiv =
// [oderiv_count, oderiv_sumsq[D]], NonlinearComponent::StoreBackpropStats (nnet-component-itf.cc:461-480): the column sums of
// squares of the ReLU's out_deriv dr = (dz + temp) scale + z vdm.  They need no pass of their own:
//   sum_r dr^2 = scale^2 (sum dz^2 - (sum dz)^2 / N) + 2 vdm scale sum z dz + vdm^2 sum z^2,   sum z^2 = N var scale^2
// (train mode: sum z = 0; test mode: temp = vdm = 0) from the quantities this reduction already forms, plus sum dz^2.
// Synchronised BatchNorm (common.h BnSync): sums_out != null -- only the five column sums, doubles [5][D]; sums_in != null -- the sums
// come from there (the first three all-reduced over the ranks), N is the GLOBAL row count and N_local this rank's (ReLU statistics).
__global__ __launch_bounds__(kFinThreads) void bn_relu_bwd_finalize_kernel(const float *partial, int chunks, int D, int N, float target_rms,
                                                                           float *memo, double *relu_stats, int test_mode, double *oderiv,
                                                                           double *sums_out = nullptr, const double *sums_in = nullptr, int N_local = 0,
                                                                           double *fro2 = nullptr, float repair_abs = 0.f) {
  __shared__ double red[5 * kFinLanes * (kFinCols + 1)];
  const int d = blockIdx.x * kFinCols + (threadIdx.x & (kFinCols - 1));
  double q[5];
  const int nq = relu_stats ? 5 : 3;
  if (!sums_in) finalize_sums<5, double>(partial, chunks, chunks, D, nq, q, red);
  if (sums_out) {
    if (threadIdx.x < kFinCols && d < D)
      for (int k = 0; k < nq; k++) sums_out[(size_t)k * D + d] = q[k];
    return;
  }
  if (relu_stats && blockIdx.x == 0 && threadIdx.x == 0) relu_stats[0] += (double)(sums_in ? N_local : N);
  if (oderiv && blockIdx.x == 0 && threadIdx.x == 0) oderiv[0] += (double)N;
  double bound = 0;  // upper bound of sum_r d_aff^2 for this column: d_aff = relu'(x) dr + repair, |relu'| <= 1
  if (threadIdx.x < kFinCols && d < D) {
    if (sums_in)
      for (int k = 0; k < nq; k++) q[k] = sums_in[(size_t)k * D + d];
    const float coeff = -1.0f / (target_rms * target_rms * N);
    // test mode (BatchNormTestComponent::Backprop, nnet-normalize-component.cc:879-922): in_deriv = out_deriv * scale
    const float sc = memo[2 * D + d];
    const float vdm = test_mode ? 0.f : (float)(coeff * q[0]) * sc;
    memo[3 * D + d] = vdm;
    memo[4 * D + d] = test_mode ? 0.f : (float)(-q[1] / N);
    if (relu_stats) {
      relu_stats[1 + d] += q[3];
      relu_stats[1 + D + d] += q[4];
    }
    if (oderiv || fro2) {
      const double sc2 = (double)sc * sc;
      double v = sc2 * (test_mode ? q[2] : q[2] - q[1] * q[1] / N);
      if (!test_mode) {
        const double var = (double)memo[D + d] - (double)memo[d] * memo[d];  // uvar - mean^2 (memo rows 1, 0 of the forward pass)
        v += 2.0 * vdm * sc * q[0] + (double)vdm * vdm * (double)N * (var > 0 ? var : 0.0) * sc2;
      }
      if (oderiv) oderiv[1 + d] += v > 0 ? v : 0.0;
      const double nr = sqrt(v > 0 ? v : 0.0) * 1.0001 + sqrt((double)N) * repair_abs;  // (a hair of slack for the rounding of v itself)
      bound = nr * nr;
    }
  }
  if (fro2 && threadIdx.x < 64) {
    for (int o = 16; o > 0; o >>= 1) bound += __shfl_xor(bound, o, 32);
    if (threadIdx.x == 0) fro2[blockIdx.x] = bound;
  }
}

// d_aff = relu'(x) * [ (dz + temp) * scale + z * vdm ] + repair[col];  partial[chunk][col] = column sums of d_aff.
// repair_stats (may be null): ReLU statistics deciding the self-repair term (nnet-simple-component.cc:1028-1073).
template <int VEC, bool PLANES>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(MatView x, MatView dz, const float *memo, int D,
                                                                const double *repair_stats, float self_repair_scale,
                                                                int rows_per_chunk, int chunks, MatView d_aff, float *partial, const float *mask, int B,
                                                                PlanesSink pk, int pk_lead, int rev) {
  const float pks = PLANES ? pk.rec[0] : 1.0f;
  __shared__ float red[4][64 * 4 + 4];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tc) * VEC;
  const int chunk = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;  // (rev: the chunks the reduce pass read last, first)
  const int r0 = chunk * rows_per_chunk, r1 = min(x.rows, r0 + rows_per_chunk);
  float s[4] = {0, 0, 0, 0};
  if (col < x.cols) {
    float mu[4], sc[4], vdm[4], tmp[4], rep[4];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      mu[j] = memo[col + j];
      sc[j] = memo[2 * D + col + j];
      vdm[j] = memo[3 * D + col + j];
      tmp[j] = memo[4 * D + col + j];
      rep[j] = 0.f;
      if (repair_stats) {
        const float count = (float)repair_stats[0];
        if (self_repair_scale != 0.f && count != 0.f) {
          const float stv = (float)repair_stats[1 + D + col + j];
          float v = (stv - 0.05f * count > 0.f ? 1.f : 0.f) + (stv - 0.95f * count > 0.f ? 1.f : 0.f) - 1.f;
          rep[j] = v * (-self_repair_scale / 0.5f);
        }
      }
    }
    auto apply = [&](const float (&xv)[4], float (&dv)[4], int r) {
      float o[4];
      if (mask) {
        const float *mk = mask + (long long)(r % B) * x.cols + col;
#pragma unroll
        for (int j = 0; j < VEC; j++) dv[j] *= mk[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const float z = (xv[j] - mu[j]) * sc[j];
        const float dr = (dv[j] + tmp[j]) * sc[j] + z * vdm[j];
        float v = (xv[j] > 0.f ? 1.f : 0.f) * dr;
        if (rep[j] != 0.f) v += rep[j];
        o[j] = v;
        s[j] += v;
      }
      st(d_aff.data + (long long)r * d_aff.stride + col, o, VEC == 4);
      if constexpr (PLANES && VEC == 4) store_planes4(pk, pks, (long long)pk_lead + r, col, o);
    };
    int r = r0 + tr;
    for (; r + 12 < r1; r += 16) {  // four rows requested together (d_aff may be dz itself: a row is read before it is written)
      float xv[4][4], dv[4][4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        ld(x.data + (long long)(r + 4 * u) * x.stride + col, xv[u], VEC == 4);
        ld(dz.data + (long long)(r + 4 * u) * dz.stride + col, dv[u], VEC == 4);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) apply(xv[u], dv[u], r + 4 * u);
    }
    for (; r < r1; r += 4) {
      float xv[4], dv[4];
      ld(x.data + (long long)r * x.stride + col, xv, VEC == 4);
      ld(dz.data + (long long)r * dz.stride + col, dv, VEC == 4);
      apply(xv, dv, r);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) red[tr][tc * VEC + j] = s[j];
  __syncthreads();
  if (tr == 0 && col < x.cols) {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const int k = tc * VEC + j;
      partial[(long long)chunk * x.cols + col + j] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
  }
}

// The apply pass above with the natural-gradient statistic of the output side riding on it: a block owns 128 whole rows,
// walks them 32 columns at a time, writes d_aff and stages the same values in LDS as the A operand of
// H[128 x Rp] += d_aff[128 x 32] W[Rp x 32]^T on the f32 matrix cores (v_mfma_f32_32x32x2_f32, wave w owns rows 32w..32w+31).
// The sweep is HBM-bound (x, dz in, d_aff out), the 2 N D Rp flops of H hide behind it: separately the two passes cost
// an elementwise pass plus a GEMM that re-reads d_aff.  partial[block][col] = column sums of d_aff, part[block] = sum of squares.
// MASK (a compile-time switch, not `if (mask)`): behind a run-time branch around the mask load the compiler has to wait for
// "every outstanding memory operation" where the branches join -- which includes the d_aff store of the previous row, so every
// step made four trips to HBM in a row (11 us per step of a kernel whose step has 1.3 us of MFMAs; found in the ISA).
// FULL: every row of the block exists (the per-row `if (r < rows)` around the loads made the compiler wait for ALL outstanding memory
// operations -- the previous row's store among them -- at each of them: four trips to HBM in a row per step)
template <int NT, bool MASK, bool FULL, bool PLANES>
__device__ __forceinline__ void bn_relu_bwd_apply_ng_body(MatView x, MatView dz, const float *memo, int D, const double *repair_stats,
                                                                   float self_repair_scale, MatView d_aff, float *partial, const float *mask, int B,
                                                                   NgFuse ng, PlanesSink pk, int pk_lead, int blk) {
  const float pks = PLANES ? pk.rec[0] : 1.0f;
  constexpr int BM = 128, BK = 32, LD = BK + 4;
  __shared__ __attribute__((aligned(16))) float As[BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[NT * 32 * LD];
  __shared__ float colred[4][BK];
  __shared__ double red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int c4 = t & 7, rr = t >> 3;  // this thread's float4 of a 32-column row segment, rows rr + 32 i
  const int r0 = blk * BM;
  typedef float acc_t __attribute__((ext_vector_type(16)));
  acc_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
  float ssq = 0.f;
  const float count = repair_stats ? (float)repair_stats[0] : 0.f;
  // register stage of the NEXT 32-column tile: requested before the MFMAs of the current one, so the trips to HBM (x, dz) and
  // L2 (W) run under them
  float xv[4][4], dv[4][4];
  float4 wv[NT];
  float4 mu_n, sc_n, vdm_n, tmp_n;  // the per-column constants of the requested tile
  double stv_n[4] = {0.0, 0.0, 0.0, 0.0};
  const bool repairing = repair_stats && self_repair_scale != 0.f && count != 0.f;
  auto request = [&](int k0) {
    const int col = k0 + c4 * 4;
    // (the constants first: they are what the next step touches first, and a wait for them is a wait for everything requested before)
    mu_n = *reinterpret_cast<const float4 *>(memo + col);
    sc_n = *reinterpret_cast<const float4 *>(memo + 2 * D + col);
    vdm_n = *reinterpret_cast<const float4 *>(memo + 3 * D + col);
    tmp_n = *reinterpret_cast<const float4 *>(memo + 4 * D + col);
    if (repairing) {
#pragma unroll
      for (int j = 0; j < 4; j++) stv_n[j] = repair_stats[1 + D + col + j];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + rr + 32 * i;
      if (FULL || r < x.rows) {
        ld(x.data + (long long)r * x.stride + col, xv[i], true);
        ld(dz.data + (long long)r * dz.stride + col, dv[i], true);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {  // W tile: NT * 32 rows x 32 k, rows >= Rp are zero
      const int idx = t + 256 * j, n = idx >> 3, kk = (idx & 7) * 4;
      wv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < ng.Rp) wv[j] = *reinterpret_cast<const float4 *>(ng.W + (long long)n * ng.ldw + k0 + kk);
    }
  };
  request(0);
  for (int k0 = 0; k0 < D; k0 += BK) {
    const int col = k0 + c4 * 4;
    const float4 mu = mu_n, sc = sc_n, vdm = vdm_n, tmp = tmp_n;
    float rep[4] = {0.f, 0.f, 0.f, 0.f};
    if (repairing) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float stv = (float)stv_n[j];
        const float v = (stv - 0.05f * count > 0.f ? 1.f : 0.f) + (stv - 0.95f * count > 0.f ? 1.f : 0.f) - 1.f;
        rep[j] = v * (-self_repair_scale / 0.5f);
      }
    }
    const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, sc_[4] = {sc.x, sc.y, sc.z, sc.w}, vdm_[4] = {vdm.x, vdm.y, vdm.z, vdm.w},
                tmp_[4] = {tmp.x, tmp.y, tmp.z, tmp.w};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    float o[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + rr + 32 * i;
#pragma unroll
      for (int j = 0; j < 4; j++) o[i][j] = 0.f;
      if (FULL || r < x.rows) {
        if (MASK) {
          const float *mk = mask + (long long)(r % B) * x.cols + col;
#pragma unroll
          for (int j = 0; j < 4; j++) dv[i][j] *= mk[j];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float z = (xv[i][j] - mu_[j]) * sc_[j];
          const float dr = (dv[i][j] + tmp_[j]) * sc_[j] + z * vdm_[j];
          float v = (xv[i][j] > 0.f ? 1.f : 0.f) * dr;
          if (rep[j] != 0.f) v += rep[j];
          o[i][j] = v;
          cs[j] += v;
          ssq += v * v;
        }
      }
      *reinterpret_cast<float4 *>(As + (rr + 32 * i) * LD + c4 * 4) = make_float4(o[i][0], o[i][1], o[i][2], o[i][3]);
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int idx = t + 256 * j, n = idx >> 3, kk = (idx & 7) * 4;
      *reinterpret_cast<float4 *>(Bs + n * LD + kk) = wv[j];
    }
    // the next tile is requested BEFORE this tile's d_aff goes out: memory operations retire in order, so a store issued ahead
    // of the loads would have to reach HBM before the next step could touch what it loaded
    if (k0 + BK < D) request(k0 + BK);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + rr + 32 * i;
      if (FULL || r < x.rows) {
        st(d_aff.data + (long long)r * d_aff.stride + col, o[i], true);
        if constexpr (PLANES) store_planes4(pk, pks, (long long)pk_lead + r, col, o[i]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {  // column sums over the wave's 8 row groups (lanes 8 apart hold the same columns)
      float v = cs[j];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 8) colred[wave][lane * 4 + j] = v;
    }
    __syncthreads();
    {
      const float *as = As + (wave * 32 + li) * LD + lh * 4;
      const float *bs = Bs + li * LD + lh * 4;
#pragma unroll
      for (int kg = 0; kg < BK / 8; kg++) {
        const float4 a = *reinterpret_cast<const float4 *>(as + kg * 8);
#pragma unroll
        for (int j = 0; j < NT; j++) {
          const float4 b = *reinterpret_cast<const float4 *>(bs + j * 32 * LD + kg * 8);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
        }
      }
    }
    if (t < BK) partial[(long long)blk * D + k0 + t] = (colred[0][t] + colred[1][t]) + (colred[2][t] + colred[3][t]);
    __syncthreads();  // the tile buffers are rewritten next
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int n = j * 32 + li;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int r = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (n < ng.Rp && r < x.rows) ng.H[(long long)r * ng.Rp + n] = acc[j][e];
    }
  }
  double v = ssq;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (t == 0) ng.part[blk] = (red[0] + red[1]) + (red[2] + red[3]);
  if (blk == 0)
    for (int i = gridDim.x + t; i < ng.part_cap; i += 256) ng.part[i] = 0.0;
}

template <int NT, bool MASK, bool PLANES>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_ng_kernel(MatView x, MatView dz, const float *memo, int D, const double *repair_stats,
                                                                   float self_repair_scale, MatView d_aff, float *partial, const float *mask, int B,
                                                                   NgFuse ng, PlanesSink pk, int pk_lead, int rev) {
  const int blk = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;  // (rev: the row blocks the reduce pass read last, first)
  if (blk * 128 + 128 <= x.rows)
    bn_relu_bwd_apply_ng_body<NT, MASK, true, PLANES>(x, dz, memo, D, repair_stats, self_repair_scale, d_aff, partial, mask, B, ng, pk, pk_lead, blk);
  else bn_relu_bwd_apply_ng_body<NT, MASK, false, PLANES>(x, dz, memo, D, repair_stats, self_repair_scale, d_aff, partial, mask, B, ng, pk, pk_lead, blk);
}

__global__ __launch_bounds__(kFinThreads) void colsum_add_kernel(const float *partial, int chunks, int D, float scale, float *acc) {
  __shared__ float red[kFinLanes * (kFinCols + 1)];
  const int d = blockIdx.x * kFinCols + (threadIdx.x & (kFinCols - 1));
  float q[1];
  finalize_sums<1, float>(partial, chunks, chunks, D, 1, q, red);
  if (threadIdx.x < kFinCols && d < D) acc[d] += scale * q[0];
}

}  // namespace

hipError_t bn_apply_bypass(MatView x, const float *memo, int D, int period, MatView prev, float bypass, MatView out, hipStream_t s, const float *mask, int B,
                           const PlanesSink *planes) {
  if (x.rows == 0) return hipSuccess;
  // period: a super row (cols > D) is a run of rows of D values, each padded to `period` (the plain rows' stride)
  const bool vec = vec4_ok(x) && vec4_ok(out) && (!prev.data || vec4_ok(prev)) && D % 4 == 0 && (reinterpret_cast<uintptr_t>(memo) & 15) == 0 &&
                   (!mask || (reinterpret_cast<uintptr_t>(mask) & 15) == 0);
  const long long work = (long long)x.rows * (vec ? x.cols / 4 : x.cols);
  if (planes && !(vec && x.cols == D && D % 32 == 0 && planes->P && planes->rec)) return hipErrorInvalidValue;
  ProfHbmRange prof(4, 4.0 * x.rows * x.cols * ((prev.data ? 3.0 : 2.0) + (planes ? 1.0 : 0.0)), s);  // reads x [and the bypass rows], writes out [and its planes]
  const PlanesSink none{nullptr, 0, nullptr};
  if (planes) hipLaunchKernelGGL((bn_apply_bypass_kernel<4, true>), dim3(grid_for((long long)((x.rows + 7) / 8) * 8 * (x.cols / 4), 256)), dim3(256), 0, s, x, memo, memo + 2 * D, D, period, prev, bypass, out, mask, B, *planes, 0);
  else if (vec) hipLaunchKernelGGL((bn_apply_bypass_kernel<4, false>), dim3(grid_for(work, 256)), dim3(256), 0, s, x, memo, memo + 2 * D, D, period, prev, bypass, out, mask, B, none, options().reverse_passes & 1);
  else hipLaunchKernelGGL((bn_apply_bypass_kernel<1, false>), dim3(grid_for(work, 256)), dim3(256), 0, s, x, memo, memo + 2 * D, D, period, prev, bypass, out, mask, B, none, 0);
  return hipGetLastError();
}

bool bn_relu_bwd_ng_ok(MatView x, MatView dz, MatView d_aff, int Rp) {
  return vec4_ok(x) && vec4_ok(dz) && vec4_ok(d_aff) && x.cols % 32 == 0 && Rp >= 1 && Rp <= 96 && x.rows >= 1;
}

// Two 128-row blocks are resident per CU (registers): a launch of a few blocks more than one round of them runs as two
// (514 blocks on 256 CUs: 0.50 ms against 0.32 ms for 512), and a launch of few blocks (a 9 600-row minibatch: 75) leaves most
// CUs idle while each block walks all its columns; in both cases the separate passes are faster.
bool bn_relu_bwd_ng_pays(int rows) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    slots = 2 * cus;
  }
  const int blocks = (rows + 127) / 128, rounds = (blocks + slots - 1) / slots;
  return 4 * blocks >= 3 * rounds * slots;  // the rounds at least 3/4 used: fewer blocks walk their 1536 columns on idle CUs' time
}

size_t bn_relu_bwd_workspace_bytes(int rows, int cols) {
  ColReducePlan p = colreduce_plan(rows, cols);
  const size_t blocks128 = ((size_t)rows + 127) / 128;  // column-sum partials of the natural-gradient form: one row per 128-row block
  return sizeof(float) * (5 * (size_t)p.chunks + std::max((size_t)p.chunks, blocks128)) * cols + 64;
}

// x: ReLU output (= BatchNorm input), dz: derivative w.r.t. the BatchNorm output, memo: forward memo (rows 0-2
// valid).  Writes d_aff (may alias dz) and adds lr_scale * colsum(d_aff) into bias_acc (may be null).
hipError_t bn_relu_bwd(MatView x, MatView dz, float *memo, float target_rms, bool bn_test_mode, double *relu_stats, bool store_relu_stats,
                       bool self_repair, float self_repair_scale, MatView d_aff, float *bias_acc, float bias_scale,
                       void *ws, size_t ws_bytes, hipStream_t s, const float *mask, int B, const NgFuse *ng, double *oderiv_stats, const BwdPlanes *planes) {
  if (x.rows == 0) return hipSuccess;
  if (ws_bytes < bn_relu_bwd_workspace_bytes(x.rows, x.cols)) return hipErrorInvalidValue;
  if (ng && (!bn_relu_bwd_ng_ok(x, dz, d_aff, ng->Rp) || (reinterpret_cast<uintptr_t>(memo) & 15) || (reinterpret_cast<uintptr_t>(ng->W) & 15) || ng->ldw % 4))
    return hipErrorInvalidValue;
  const int D = x.cols;
  ProfHbmRange prof(5, 4.0 * x.rows * x.cols * 5.0, s);  // two stages: (x, dz) read twice, d_aff written
  ColReducePlan pl = colreduce_plan(x.rows, D);
  const bool vec = vec4_ok(x) && vec4_ok(dz) && vec4_ok(d_aff);
  const int per = vec ? 256 : 64;
  dim3 grid((D + per - 1) / per, pl.chunks), block(256);
  float *partial = (float *)ws;                                  // quantities 0..4 of the reduction
  float *bias_partial = partial + 5 * (size_t)pl.chunks * D;     // column sums of d_aff
  // Order as in the reference: StoreStats runs with the forward pass, RepairGradients in Backprop sees the
  // statistics including this minibatch (when it was stored).
  if (store_relu_stats) {
    if (vec) hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<4, true>), grid, block, 0, s, x, dz, memo, memo + 2 * D, pl.rows_per_chunk, pl.chunks, partial, mask, B);
    else hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<1, true>), grid, block, 0, s, x, dz, memo, memo + 2 * D, pl.rows_per_chunk, pl.chunks, partial, mask, B);
  } else {
    if (vec) hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<4, false>), grid, block, 0, s, x, dz, memo, memo + 2 * D, pl.rows_per_chunk, pl.chunks, partial, mask, B);
    else hipLaunchKernelGGL((bn_relu_bwd_reduce_kernel<1, false>), grid, block, 0, s, x, dz, memo, memo + 2 * D, pl.rows_per_chunk, pl.chunks, partial, mask, B);
  }
  double *fro2 = fro_bound_buf();
  if (fro2 && fro_bound_blocks()) *fro_bound_blocks() = (int)finalize_grid(D);
  const float repair_abs = self_repair ? 2.0f * fabsf(self_repair_scale) : 0.f;  // |repair term| <= self_repair_scale / 0.5
  if (BnSync *sy = bn_test_mode ? nullptr : bn_sync_current()) {
    // [sum z dz, sum dz, sum dz^2] over all ranks' rows; the ReLU's value / derivative sums (rows 3, 4) stay this rank's
    hipLaunchKernelGGL(bn_relu_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, pl.chunks, D, x.rows, target_rms, memo,
                       store_relu_stats ? relu_stats : (double *)nullptr, 0, (double *)nullptr, sy->buf, (const double *)nullptr, 0);
    if (sy->fn(sy->ctx, sy->buf, 3LL * D, (tdnnf_stream)s)) return hipErrorUnknown;
    hipLaunchKernelGGL(bn_relu_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, pl.chunks, D, x.rows * sy->world, target_rms, memo,
                       store_relu_stats ? relu_stats : (double *)nullptr, 0, oderiv_stats, (double *)nullptr, (const double *)sy->buf, x.rows, fro2, repair_abs);
  } else {
    hipLaunchKernelGGL(bn_relu_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, pl.chunks, D, x.rows, target_rms, memo,
                       store_relu_stats ? relu_stats : (double *)nullptr, bn_test_mode ? 1 : 0, oderiv_stats, (double *)nullptr, (const double *)nullptr, 0, fro2,
                       repair_abs);
  }
  const double *rep = self_repair ? relu_stats : nullptr;
  // planes of d_aff written by the apply pass itself: the scale record first, from the bound the finalize launch just left (common.h FroBoundScope)
  PlanesSink pk{nullptr, 0, nullptr};
  int pk_lead = 0;
  if (planes && planes->P) {
    if (!(fro2 && vec && x.cols % 16 == 0)) return hipErrorInvalidValue;  // (the caller already counts on the planes)
    hipError_t e = planes_scale_bound(fro2, (int)finalize_grid(D), (double)x.rows * D, 1.0f, 0.0f, nullptr, planes->rec, s);
    if (e != hipSuccess) return e;
    pk = PlanesSink{planes->P, planes->R, planes->rec};
    pk_lead = planes->lead;
  }
  if (ng) {
    const int blocks = (x.rows + 127) / 128;
#define APPLY_NG(NT)                                                                                                                                     \
  do {                                                                                                                                               \
    if (pk.P && mask) hipLaunchKernelGGL((bn_relu_bwd_apply_ng_kernel<NT, true, true>), dim3(blocks), block, 0, s, x, dz, memo, D, rep, self_repair_scale, d_aff, bias_partial, mask, B, *ng, pk, pk_lead, (options().reverse_passes >> 1) & 1); \
    else if (pk.P) hipLaunchKernelGGL((bn_relu_bwd_apply_ng_kernel<NT, false, true>), dim3(blocks), block, 0, s, x, dz, memo, D, rep, self_repair_scale, d_aff, bias_partial, mask, B, *ng, pk, pk_lead, (options().reverse_passes >> 1) & 1); \
    else if (mask) hipLaunchKernelGGL((bn_relu_bwd_apply_ng_kernel<NT, true, false>), dim3(blocks), block, 0, s, x, dz, memo, D, rep, self_repair_scale, d_aff, bias_partial, mask, B, *ng, pk, pk_lead, (options().reverse_passes >> 1) & 1); \
    else hipLaunchKernelGGL((bn_relu_bwd_apply_ng_kernel<NT, false, false>), dim3(blocks), block, 0, s, x, dz, memo, D, rep, self_repair_scale, d_aff, bias_partial, mask, B, *ng, pk, pk_lead, (options().reverse_passes >> 1) & 1);    \
  } while (0)
    if (ng->Rp <= 32) APPLY_NG(1);
    else if (ng->Rp <= 64) APPLY_NG(2);
    else APPLY_NG(3);
#undef APPLY_NG
    if (bias_acc) hipLaunchKernelGGL(colsum_add_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, bias_partial, blocks, D, bias_scale, bias_acc);
    return hipGetLastError();
  }
  if (pk.P) hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<4, true>), grid, block, 0, s, x, dz, memo, D, rep, self_repair_scale, pl.rows_per_chunk, pl.chunks, d_aff, bias_partial, mask, B, pk, pk_lead, (options().reverse_passes >> 1) & 1);
  else if (vec) hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<4, false>), grid, block, 0, s, x, dz, memo, D, rep, self_repair_scale, pl.rows_per_chunk, pl.chunks, d_aff, bias_partial, mask, B, pk, pk_lead, (options().reverse_passes >> 1) & 1);
  else hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<1, false>), grid, block, 0, s, x, dz, memo, D, rep, self_repair_scale, pl.rows_per_chunk, pl.chunks, d_aff, bias_partial, mask, B, pk, pk_lead, (options().reverse_passes >> 1) & 1);
  if (bias_acc) hipLaunchKernelGGL(colsum_add_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, bias_partial, pl.chunks, D, bias_scale, bias_acc);
  return hipGetLastError();
}

}  // namespace tdnnf
