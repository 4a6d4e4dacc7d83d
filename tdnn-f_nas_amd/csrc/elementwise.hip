// elementwise.hip -- HBM-bound component kernels for gfx950: BatchNorm(+Test), ReLU, the DARTS
// mixing ops, CopyN, ElementwiseProduct, LogSoftmax, scaled sums.  One kernel replaces each
// chain of CuMatrix calls of the reference; all loads/stores are 16 B per lane
// when the views allow it, grids are capped at 2048 blocks and grid-stride.
//
// Reference: /root/reference/src/nnet3/nnet-normalize-component.cc, nnet-simple-component.cc
// (exact line ranges are next to each C-ABI entry in include/tdnnf_hip.h).
#include "common.h"

namespace tdnnf {
namespace {

// ----------------------------------------------------------------------------- colreduce
// block = 64 float4-columns x 4 row lanes; each block reduces `rows_per_chunk` rows of 256 columns.
template <int KIND, int VEC>
__global__ __launch_bounds__(256) void colreduce_kernel(MatView a, MatView b, int rows_per_chunk, int chunks,
                                                        float *partial) {
  __shared__ float red[2][4][64 * 4 + 4];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tc) * VEC;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(a.rows, r0 + rows_per_chunk);
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (col < a.cols) {
    auto fetch = [&](int r, float (&va)[4], float (&vb)[4]) {
      if (VEC == 4) {
        float4 x = *reinterpret_cast<const float4 *>(a.data + (long long)r * a.stride + col);
        va[0] = x.x; va[1] = x.y; va[2] = x.z; va[3] = x.w;
        if (KIND == 2) {
          float4 y = *reinterpret_cast<const float4 *>(b.data + (long long)r * b.stride + col);
          vb[0] = y.x; vb[1] = y.y; vb[2] = y.z; vb[3] = y.w;
        }
      } else if (VEC == 2) {
        float2 x = *reinterpret_cast<const float2 *>(a.data + (long long)r * a.stride + col);
        va[0] = x.x; va[1] = x.y;
        if (KIND == 2) {
          float2 y = *reinterpret_cast<const float2 *>(b.data + (long long)r * b.stride + col);
          vb[0] = y.x; vb[1] = y.y;
        }
      } else {
        va[0] = a.data[(long long)r * a.stride + col];
        if (KIND == 2) vb[0] = b.data[(long long)r * b.stride + col];
      }
    };
    auto add = [&](const float (&va)[4], const float (&vb)[4]) {
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        if (KIND == 0) s0[j] += va[j];
        if (KIND == 1) { s0[j] += va[j]; s1[j] += va[j] * va[j]; }
        if (KIND == 2) { s0[j] += va[j] * vb[j]; s1[j] += vb[j]; }
        if (KIND == 3) { s0[j] += va[j]; s1[j] += va[j] > 0.f ? 1.f : 0.f; }
      }
    };
    int r = r0 + tr;
    for (; r + 12 < r1; r += 16) {  // four rows requested before the first is added: the pass is bound by requests in flight
      float va[4][4], vb[4][4];
#pragma unroll
      for (int u = 0; u < 4; u++) fetch(r + 4 * u, va[u], vb[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) add(va[u], vb[u]);
    }
    for (; r < r1; r += 4) {
      float va[4], vb[4];
      fetch(r, va, vb);
      add(va, vb);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    red[0][tr][tc * VEC + j] = s0[j];
    red[1][tr][tc * VEC + j] = s1[j];
  }
  __syncthreads();
  if (tr == 0 && col < a.cols) {
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      const float t0 = (red[0][0][tc * VEC + j] + red[0][1][tc * VEC + j]) + (red[0][2][tc * VEC + j] + red[0][3][tc * VEC + j]);
      const float t1 = (red[1][0][tc * VEC + j] + red[1][1][tc * VEC + j]) + (red[1][2][tc * VEC + j] + red[1][3][tc * VEC + j]);
      partial[(long long)blockIdx.y * a.cols + col + j] = t0;
      if (KIND != 0) partial[((long long)chunks + blockIdx.y) * a.cols + col + j] = t1;
    }
  }
}

}  // namespace

ColReducePlan colreduce_plan(int rows, int cols) {
  ColReducePlan p;
  const int colblocks = (cols + 255) / 256;
  int chunks = (1024 + colblocks - 1) / colblocks;  // ~4 blocks per CU
  const int maxc = (rows + 31) / 32;
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  p.rows_per_chunk = (rows + chunks - 1) / chunks;
  if (p.rows_per_chunk < 1) p.rows_per_chunk = 1;
  p.chunks = (rows + p.rows_per_chunk - 1) / p.rows_per_chunk;
  if (p.chunks < 1) p.chunks = 1;
  return p;
}
size_t colreduce_bytes(int rows, int cols) {
  ColReducePlan p = colreduce_plan(rows, cols);
  return sizeof(float) * 2 * (size_t)p.chunks * cols + 64;
}
hipError_t colreduce_partial(int kind, MatView a, MatView b, float *partial, hipStream_t s) {
  ColReducePlan p = colreduce_plan(a.rows, a.cols);
  return colreduce_partial_into(kind, a, b, p.chunks, p.rows_per_chunk, p.chunks, partial, s);
}
hipError_t colreduce_partial_into(int kind, MatView a, MatView b, int chunks, int rows_per_chunk, int sq_row_offset, float *partial, hipStream_t s) {
  const bool vec = vec4_ok(a) && (kind != 2 || vec4_ok(b));
  auto vec2_ok = [](const MatView &m) { return m.cols % 2 == 0 && m.stride % 2 == 0 && (reinterpret_cast<uintptr_t>(m.data) & 7) == 0; };
  const bool vec2 = !vec && vec2_ok(a) && (kind != 2 || vec2_ok(b));  // e.g. the 6034-wide output layer
  const int per = vec ? 256 : (vec2 ? 128 : 64);
  dim3 grid((a.cols + per - 1) / per, chunks), block(256);
#define CR(K)                                                                                              \
  if (vec) hipLaunchKernelGGL((colreduce_kernel<K, 4>), grid, block, 0, s, a, b, rows_per_chunk, sq_row_offset, partial); \
  else if (vec2) hipLaunchKernelGGL((colreduce_kernel<K, 2>), grid, block, 0, s, a, b, rows_per_chunk, sq_row_offset, partial); \
  else hipLaunchKernelGGL((colreduce_kernel<K, 1>), grid, block, 0, s, a, b, rows_per_chunk, sq_row_offset, partial);
  switch (kind) {
    case 0: CR(0) break;
    case 1: CR(1) break;
    case 2: CR(2) break;
    default: CR(3) break;
  }
#undef CR
  return hipGetLastError();
}

namespace {

// ---------------------------------------------------------------------------- batchnorm
// finalize forward stats: memo rows 0 mean, 1 uvar, 2 scale  (nnet-normalize-component.cc:433-445)
// sums_out (synchronised BatchNorm, first half): only the column sums, as doubles [2][D].  sums_in (second half): the sums come
// from there (all-reduced over the ranks) instead of from the partial rows, and N is the global row count.
__global__ __launch_bounds__(kFinThreads) void bn_fwd_finalize_kernel(const float *partial, int chunks, int D, int N, float epsilon,
                                                                      float target_rms, float *memo, double *sums_out = nullptr,
                                                                      const double *sums_in = nullptr, double *store = nullptr, int store_frames = 0,
                                                                      double *fro2 = nullptr) {
  __shared__ double red[2 * kFinLanes * (kFinCols + 1)];
  const int d = blockIdx.x * kFinCols + (threadIdx.x & (kFinCols - 1));
  double q[2];
  if (!sums_in) finalize_sums<2, double>(partial, chunks, chunks, D, 2, q, red);
  const bool own = threadIdx.x < kFinCols && d < D;
  if (sums_out) {
    if (own) {
      sums_out[d] = q[0];
      sums_out[D + d] = q[1];
    }
    return;
  }
  double bound = 0;  // sum over rows of z^2 for this column: N scale^2 var (store_frames = this rank's rows)
  if (own) {
    if (sums_in) {
      q[0] = sums_in[d];
      q[1] = sums_in[D + d];
    }
    const float mean = (float)(q[0] / N), uvar = (float)(q[1] / N);
    const float var_scale = 1.0f / (target_rms * target_rms);
    float v = var_scale * uvar - var_scale * mean * mean;
    v = floor_keep_nan(v, 0.f) + var_scale * epsilon;
    memo[d] = mean;
    memo[D + d] = uvar;
    const float sc = 1.0f / sqrtf(v);
    memo[2 * D + d] = sc;
    const double var = (double)uvar - (double)mean * mean;
    bound = (double)N * sc * sc * (var > 0 ? var : 0.0);
    if (store) {  // BatchNormComponent::StoreStats (nnet-normalize-component.cc:551-589) in the same launch: count += I, sum += I mean, sumsq += I uvar
      if (d == 0) store[0] += (double)store_frames;
      store[1 + d] += (double)store_frames * mean;
      store[1 + D + d] += (double)store_frames * uvar;
    }
  }
  if (fro2 && threadIdx.x < 64) {  // (the kFinCols <= 32 owning threads are the first lanes of the first wave; the other lanes carry zeros)
    for (int o = 16; o > 0; o >>= 1) bound += __shfl_xor(bound, o, 32);
    if (threadIdx.x == 0) fro2[blockIdx.x] = bound;
  }
}
// memo rows 3 var_deriv_mod, 4 temp (:520-526)
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(const float *partial, int chunks, int D, int N, float target_rms, float *memo,
                                                                      double *sums_out = nullptr, const double *sums_in = nullptr) {
  __shared__ double red[2 * kFinLanes * (kFinCols + 1)];
  const int d = blockIdx.x * kFinCols + (threadIdx.x & (kFinCols - 1));
  double q[2];
  if (!sums_in) finalize_sums<2, double>(partial, chunks, chunks, D, 2, q, red);
  if (threadIdx.x >= kFinCols || d >= D) return;
  if (sums_out) {
    sums_out[d] = q[0];
    sums_out[D + d] = q[1];
    return;
  }
  if (sums_in) {
    q[0] = sums_in[d];
    q[1] = sums_in[D + d];
  }
  const float coeff = -1.0f / (target_rms * target_rms * N);
  memo[3 * D + d] = (float)(coeff * q[0]) * memo[2 * D + d];
  memo[4 * D + d] = (float)(-q[1] / N);
}

// generic per-column affine maps.  MODE 0: out = (in + add[c]) * mul[c]      (bn train fwd, add=-mean)
//                                  MODE 1: out = in * mul[c] + add[c]        (bn test fwd)
//                                  MODE 2: out = in * mul[c]                 (bn test bwd)
//                                  MODE 3: out = (in + add[c]) * mul[c] + z * vdm[c]   (bn train bwd)
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void colmap_kernel(MatView in, MatView z, const float *mul, const float *add,
                                                     const float *vdm, MatView out) {
  const int cv = in.cols / VEC;
  const long long total = (long long)in.rows * cv;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / cv), c = (int)(e % cv) * VEC;
    float x[4], zz[4], o[4];
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4 *>(in.data + (long long)r * in.stride + c);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      if (MODE == 3) {
        float4 w = *reinterpret_cast<const float4 *>(z.data + (long long)r * z.stride + c);
        zz[0] = w.x; zz[1] = w.y; zz[2] = w.z; zz[3] = w.w;
      }
    } else {
      x[0] = in.data[(long long)r * in.stride + c];
      if (MODE == 3) zz[0] = z.data[(long long)r * z.stride + c];
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      if (MODE == 0) o[j] = (x[j] + add[c + j]) * mul[c + j];
      if (MODE == 1) o[j] = x[j] * mul[c + j] + add[c + j];
      if (MODE == 2) o[j] = x[j] * mul[c + j];
      if (MODE == 3) o[j] = (x[j] + add[c + j]) * mul[c + j] + zz[j] * vdm[c + j];
    }
    if (VEC == 4)
      *reinterpret_cast<float4 *>(out.data + (long long)r * out.stride + c) = make_float4(o[0], o[1], o[2], o[3]);
    else
      out.data[(long long)r * out.stride + c] = o[0];
  }
}
// bn train fwd needs add = -mean: small helper writing negated means into memo row 4 is avoided by
// passing mean and using MODE 4: out = (in - sub[c]) * mul[c]
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(MatView in, const float *mean, const float *scale, MatView out) {
  const int cv = in.cols / VEC;
  const long long total = (long long)in.rows * cv;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / cv), c = (int)(e % cv) * VEC;
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4 *>(in.data + (long long)r * in.stride + c);
      const float4 m = *reinterpret_cast<const float4 *>(mean + c), s = *reinterpret_cast<const float4 *>(scale + c);
      v.x = (v.x - m.x) * s.x; v.y = (v.y - m.y) * s.y; v.z = (v.z - m.z) * s.z; v.w = (v.w - m.w) * s.w;
      *reinterpret_cast<float4 *>(out.data + (long long)r * out.stride + c) = v;
    } else {
      out.data[(long long)r * out.stride + c] = (in.data[(long long)r * in.stride + c] - mean[c]) * scale[c];
    }
  }
}

__global__ void bn_store_stats_kernel(const float *memo, int D, int num_frames, double *stats) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d == 0) stats[0] += (double)num_frames;
  if (d < D) {
    stats[1 + d] += (double)num_frames * memo[d];
    stats[1 + D + d] += (double)num_frames * memo[D + d];
  }
}
__global__ void bn_derived_kernel(const double *stats, int D, float epsilon, float target_rms, float *scale,
                                  float *offset) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const double count = stats[0];
  float off = (float)(stats[1 + d] * (-1.0 / count));
  float sc = (float)(stats[1 + D + d] * (1.0 / count));
  sc += -1.0f * off * off;
  sc = floor_keep_nan(sc, 0.f) + epsilon;
  sc = 1.0f / sqrtf(sc);
  sc *= target_rms;
  scale[d] = sc;
  offset[d] = off * sc;
}

// ------------------------------------------------------------------- generic elementwise
// OP 0: out = max(a,0)              (relu fwd)
// OP 1: out = (a>0) * b             (relu bwd: a = out_value, b = out_deriv)
// OP 2: out = sa*a + sb*b           (sum of scaled; b may alias out)
// OP 3: out = sa*a                  (scaled copy)
// OP 4: out += sa*a
template <int OP, int VEC>
__global__ __launch_bounds__(256) void ew_kernel(MatView a, MatView b, float sa, float sb, MatView out) {
  const int cv = out.cols / VEC;
  const long long total = (long long)out.rows * cv;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / cv), c = (int)(e % cv) * VEC;
    float x[4], y[4], o[4];
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4 *>(a.data + (long long)r * a.stride + c);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      if (OP == 1 || OP == 2) {
        float4 w = *reinterpret_cast<const float4 *>(b.data + (long long)r * b.stride + c);
        y[0] = w.x; y[1] = w.y; y[2] = w.z; y[3] = w.w;
      }
      if (OP == 4) {
        float4 w = *reinterpret_cast<const float4 *>(out.data + (long long)r * out.stride + c);
        y[0] = w.x; y[1] = w.y; y[2] = w.z; y[3] = w.w;
      }
    } else {
      x[0] = a.data[(long long)r * a.stride + c];
      if (OP == 1 || OP == 2) y[0] = b.data[(long long)r * b.stride + c];
      if (OP == 4) y[0] = out.data[(long long)r * out.stride + c];
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      if (OP == 0) o[j] = x[j] < 0.f ? 0.f : x[j];
      if (OP == 1) o[j] = (x[j] > 0.f ? 1.f : 0.f) * y[j];
      if (OP == 2) o[j] = sa * x[j] + sb * y[j];
      if (OP == 3) o[j] = sa * x[j];
      if (OP == 4) o[j] = y[j] + sa * x[j];
    }
    if (VEC == 4)
      *reinterpret_cast<float4 *>(out.data + (long long)r * out.stride + c) = make_float4(o[0], o[1], o[2], o[3]);
    else
      out.data[(long long)r * out.stride + c] = o[0];
  }
}

template <int OP>
hipError_t launch_ew(MatView a, MatView b, float sa, float sb, MatView out, hipStream_t s) {
  if (out.rows == 0 || out.cols == 0) return hipSuccess;
  const bool needb = (OP == 1 || OP == 2);
  const bool vec = vec4_ok(a) && vec4_ok(out) && (!needb || vec4_ok(b));
  const long long work = (long long)out.rows * (vec ? out.cols / 4 : out.cols);
  if (vec) hipLaunchKernelGGL((ew_kernel<OP, 4>), dim3(grid_for(work, 256)), dim3(256), 0, s, a, b, sa, sb, out);
  else hipLaunchKernelGGL((ew_kernel<OP, 1>), dim3(grid_for(work, 256)), dim3(256), 0, s, a, b, sa, sb, out);
  return hipGetLastError();
}

// relu stats finalize: stats = [count, value_sum[D], deriv_sum[D]]
__global__ void relu_stats_finalize_kernel(const float *partial, int chunks, int D, int rows, double *stats) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d == 0) stats[0] += (double)rows;
  if (d >= D) return;
  double vs = 0, ds = 0;
  for (int c = 0; c < chunks; c++) {
    vs += partial[(long long)c * D + d];
    ds += partial[((long long)chunks + c) * D + d];
  }
  stats[1 + d] += vs;
  stats[1 + D + d] += ds;
}
// RepairGradients (nnet-simple-component.cc:1028-1073): add +-scale/0.5 to the columns outside the thresholds
__global__ void relu_repair_kernel(const double *stats, int D, float self_repair_scale, float lower, float upper,
                                   MatView in_deriv) {
  const float count = (float)stats[0];
  if (self_repair_scale == 0.f || count == 0.f) return;
  const float lo = lower * count, hi = upper * count;
  const long long total = (long long)in_deriv.rows * D;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / D), c = (int)(e % D);
    const float st = (float)stats[1 + D + c];
    float v = (st - lo > 0.f ? 1.f : 0.f) + (st - hi > 0.f ? 1.f : 0.f) - 1.f;
    v *= -self_repair_scale / 0.5f;
    if (v != 0.f) in_deriv.data[(long long)r * in_deriv.stride + c] += v;
  }
}

// colsum finalize: acc[c] += scale * sum_chunks partial
__global__ __launch_bounds__(kFinThreads) void colsum_finalize_kernel(const float *partial, int chunks, int D, float scale, float *acc) {
  __shared__ float red[kFinLanes * (kFinCols + 1)];
  const int d = blockIdx.x * kFinCols + (threadIdx.x & (kFinCols - 1));
  float q[1];
  finalize_sums<1, float>(partial, chunks, chunks, D, 1, q, red);
  if (threadIdx.x < kFinCols && d < D) acc[d] += scale * q[0];
}

// ----------------------------------------------------------------- softmax-style row ops (cols <= 64)
__device__ __forceinline__ float gumbel(float u) { return -logf(-logf(u)); }

// one thread per row; C <= 64 columns kept in registers is overkill -- C is 8 in every recipe, loop instead
__global__ void softmax_rows_kernel(MatView in, const float *gumbel_u, float inv_temp, MatView out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= in.rows) return;
  const float *x = in.data + (long long)r * in.stride;
  float *o = out.data + (long long)r * out.stride;
  float mx = -INFINITY;
  for (int c = 0; c < in.cols; c++) {
    const float v = gumbel_u ? (x[c] + gumbel(gumbel_u[c])) * inv_temp : x[c];
    o[c] = v;
    mx = fmaxf(mx, v);
  }
  float sum = 0.f;
  for (int c = 0; c < in.cols; c++) sum += expf(o[c] - mx);
  for (int c = 0; c < in.cols; c++) o[c] = floor_keep_nan(expf(o[c] - mx) / sum, 1.0e-20f);
}
__global__ void softmax_flops_bwd_kernel(MatView p, MatView dp, float a, const float *flops, int dim, float inv_temp,
                                         MatView dx) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.rows) return;
  const float *pv = p.data + (long long)r * p.stride;
  float *e = dp.data + (long long)r * dp.stride;
  float *d = dx.data + (long long)r * dx.stride;
  if (flops)
    for (int c = 0; c < dim; c++) e[c] += a * flops[c];
  float pe = 0.f;
  for (int c = 0; c < p.cols; c++) pe += pv[c] * e[c];
  for (int c = 0; c < p.cols; c++) d[c] = pv[c] * (e[c] - pe) * inv_temp;
}
__global__ void onehot_kernel(const float *u, MatView out) {
  const float uu = u[0];
  const int C = out.cols;
  const long long total = (long long)out.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    out.data[(long long)r * out.stride + c] = (uu >= (float)c / C && uu < (float)(c + 1) / C) ? 1.f : 0.f;
  }
}
__global__ void copyn_fwd_kernel(MatView in, float scale, MatView out) {
  const int C = out.cols, d = in.cols;
  const long long total = (long long)out.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    out.data[(long long)r * out.stride + c] += scale * in.data[(long long)r * in.stride + c % d];
  }
}
__global__ void copyn_bwd_kernel(MatView dout, float scale, MatView din) {
  const int d = din.cols, nb = dout.cols / d;
  const long long total = (long long)din.rows * d;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / d), c = (int)(e % d);
    float s = 0.f;
    for (int b = 0; b < nb; b++) s += dout.data[(long long)r * dout.stride + b * d + c];
    din.data[(long long)r * din.stride + c] += scale * s;
  }
}
__global__ void rows_from_vec_kernel(const float *v, float scale, MatView out) {
  const int C = out.cols;
  const long long total = (long long)out.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL)
    out.data[(e / C) * out.stride + e % C] = scale * v[e % C];
}
__global__ void ewprod_fwd_kernel(MatView in, int od, MatView out) {
  const int n = in.cols / od;
  const long long total = (long long)in.rows * od;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / od), c = (int)(e % od);
    const float *x = in.data + (long long)r * in.stride;
    float p = x[c];
    for (int i = 1; i < n; i++) p *= x[i * od + c];
    out.data[(long long)r * out.stride + c] = p;
  }
}
__global__ void ewprod_bwd_kernel(MatView in, MatView dout, int od, MatView din) {
  const int n = in.cols / od;
  const long long total = (long long)in.rows * in.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / in.cols), cc = (int)(e % in.cols), i = cc / od, c = cc % od;
    const float *x = in.data + (long long)r * in.stride;
    float p = dout.data[(long long)r * dout.stride + c];
    for (int j = 0; j < n; j++)
      if (j != i) p *= x[j * od + c];
    din.data[(long long)r * din.stride + cc] = p;
  }
}
// GeneralDropoutComponent::GetMemo (UPSTREAM): continuous: 1 - 2p + 4p U; else (U - p > 0 ? 1 : 0) / (1 - p)
__global__ void general_dropout_mask_kernel(const float *u, long long n, float p, int continuous, float *mask) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const float v = u[i];
    mask[i] = continuous ? v * (p * 4.0f) + (1.0f - 2.0f * p) : ((v + -p > 0.0f) ? 1.0f : 0.0f) * (1.0f / (1.0f - p));
  }
}
__global__ void dropout_kernel(MatView in, const float *mask, int num_seq, MatView out) {
  const int C = in.cols;
  const long long total = (long long)in.rows * C;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / C), c = (int)(e % C);
    out.data[(long long)r * out.stride + c] = in.data[(long long)r * in.stride + c] * mask[(long long)(r % num_seq) * C + c];
  }
}

// ---------------------------------------------------------------- log-softmax: one block per row
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(MatView in, MatView out) {
  __shared__ float red[4];
  for (int r = blockIdx.x; r < in.rows; r += gridDim.x) {
    const float *x = in.data + (long long)r * in.stride;
    float *o = out.data + (long long)r * out.stride;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < in.cols; c += 256) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < in.cols; c += 256) s += expf(x[c] - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    __syncthreads();
    for (int c = threadIdx.x; c < in.cols; c += 256) o[c] = x[c] - lse;
  }
}
// One pass per row with the row held in registers (up to 256 * 4 * NV columns, 16-byte aligned rows): the 6034-wide output
// rows are read once and written once instead of three reads through L2.
template <int NV>
__global__ __launch_bounds__(256) void log_softmax_fwd_regs_kernel(MatView in, MatView out, MatView aux, float aux_scale) {
  __shared__ float red[2][4];
  const int t = threadIdx.x, nc4 = (in.cols + 3) / 4;
  for (int r = blockIdx.x; r < in.rows; r += gridDim.x) {
    const float *x = in.data + (long long)r * in.stride;
    float *o = out.data + (long long)r * out.stride;
    float4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int c4 = t + 256 * k, c = c4 * 4;
      v[k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (c4 < nc4) {
        if (c + 3 < in.cols) v[k] = *reinterpret_cast<const float4 *>(x + c);
        else {
          v[k].x = x[c];
          if (c + 1 < in.cols) v[k].y = x[c + 1];
          if (c + 2 < in.cols) v[k].z = x[c + 2];
        }
      }
      mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[0][t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) s += expf(v[k].x - mx) + expf(v[k].y - mx) + expf(v[k].z - mx) + expf(v[k].w - mx);  // exp(-inf) = 0 for the padding
    s = wave_sum(s);
    if ((t & 63) == 0) red[1][t >> 6] = s;
    __syncthreads();
    const float lse = mx + logf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int c4 = t + 256 * k, c = c4 * 4;
      if (c4 >= nc4) continue;
      if (c + 3 < in.cols) *reinterpret_cast<float4 *>(o + c) = make_float4(v[k].x - lse, v[k].y - lse, v[k].z - lse, v[k].w - lse);
      else {
        o[c] = v[k].x - lse;
        if (c + 1 < in.cols) o[c + 1] = v[k].y - lse;
        if (c + 2 < in.cols) o[c + 2] = v[k].z - lse;
      }
      if (aux.data) {  // aux = aux_scale * softmax(in): the dense part of the backward pass for a derivative with a known row sum
        float *a = aux.data + (long long)r * aux.stride;
        if (c + 3 < in.cols) *reinterpret_cast<float4 *>(a + c) = make_float4(aux_scale * expf(v[k].x - lse), aux_scale * expf(v[k].y - lse),
                                                                                aux_scale * expf(v[k].z - lse), aux_scale * expf(v[k].w - lse));
        else {
          a[c] = aux_scale * expf(v[k].x - lse);
          if (c + 1 < in.cols) a[c + 1] = aux_scale * expf(v[k].y - lse);
          if (c + 2 < in.cols) a[c + 2] = aux_scale * expf(v[k].z - lse);
        }
      }
    }
    __syncthreads();  // red[] is reused by the next row
  }
}

template <int NV>
__global__ __launch_bounds__(256) void log_softmax_bwd_regs_kernel(MatView y, MatView dy, MatView dx) {
  __shared__ float red[4];
  const int t = threadIdx.x, nc4 = (y.cols + 3) / 4;
  for (int r = blockIdx.x; r < y.rows; r += gridDim.x) {
    const float *yv = y.data + (long long)r * y.stride, *e = dy.data + (long long)r * dy.stride;
    float *d = dx.data + (long long)r * dx.stride;
    float4 ev[NV], pv[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int c4 = t + 256 * k, c = c4 * 4;
      ev[k] = pv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < nc4) {
        if (c + 3 < y.cols) {
          ev[k] = *reinterpret_cast<const float4 *>(e + c);
          const float4 q = *reinterpret_cast<const float4 *>(yv + c);
          pv[k] = make_float4(expf(q.x), expf(q.y), expf(q.z), expf(q.w));
        } else {
          ev[k].x = e[c];
          pv[k].x = expf(yv[c]);
          if (c + 1 < y.cols) { ev[k].y = e[c + 1]; pv[k].y = expf(yv[c + 1]); }
          if (c + 2 < y.cols) { ev[k].z = e[c + 2]; pv[k].z = expf(yv[c + 2]); }
        }
      }
      s += (ev[k].x + ev[k].y) + (ev[k].z + ev[k].w);
    }
    s = wave_sum(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int c4 = t + 256 * k, c = c4 * 4;
      if (c4 >= nc4) continue;
      const float4 o = make_float4(ev[k].x - pv[k].x * s, ev[k].y - pv[k].y * s, ev[k].z - pv[k].z * s, ev[k].w - pv[k].w * s);
      if (c + 3 < y.cols) *reinterpret_cast<float4 *>(d + c) = o;
      else {
        d[c] = o.x;
        if (c + 1 < y.cols) d[c + 1] = o.y;
        if (c + 2 < y.cols) d[c + 2] = o.z;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(MatView y, MatView dy, MatView dx) {
  __shared__ float red[4];
  for (int r = blockIdx.x; r < y.rows; r += gridDim.x) {
    const float *yv = y.data + (long long)r * y.stride, *e = dy.data + (long long)r * dy.stride;
    float *d = dx.data + (long long)r * dx.stride;
    float s = 0.f;
    for (int c = threadIdx.x; c < y.cols; c += 256) s += e[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    for (int c = threadIdx.x; c < y.cols; c += 256) d[c] = e[c] - expf(yv[c]) * s;
  }
}

// ----------------------------------------------------------------------- DARTS coefficient kernels
// One wave; K <= 16.  Restates nnet-tdnn-component.cc:250-289 and the effective weights of :292-328.
__global__ void darts_coef_kernel(const float *log_alpha, int K, int flags, float temp, const float *gu,
                                  const float *su, int share, float *coef, float *eff) {
  if (threadIdx.x != 0) return;
  float c[TDNNF_MAX_OFFSETS];
  for (int i = 0; i < K; i++) c[i] = log_alpha[i];
  if (flags & TDNNF_DARTS_USE_GUMBEL) {
    for (int i = 0; i < K; i++) c[i] = (c[i] + gumbel(gu[i])) * (1.0f / temp);
  }
  if ((flags & TDNNF_DARTS_USE_GUMBEL) || !(flags & TDNNF_DARTS_FREE_SELECT)) {
    float mx = c[0];
    for (int i = 1; i < K; i++) mx = fmaxf(mx, c[i]);
    float s = 0.f;
    for (int i = 0; i < K; i++) s += expf(c[i] - mx);
    for (int i = 0; i < K; i++) c[i] = floor_keep_nan(expf(c[i] - mx) / s, 1.0e-20f);
  } else {
    for (int i = 0; i < K; i++) c[i] = 1.0f / (1.0f + expf(-c[i]));
  }
  if (flags & TDNNF_DARTS_UNIFORM_SAMPLE) {
    const float u = su[0];
    for (int i = 0; i < K; i++) c[i] = (u >= (float)i / K && u < (float)(i + 1) / K) ? 1.f : 0.f;
  }
  for (int i = 0; i < K; i++) {
    coef[i] = c[i];
    float e;
    if (flags & TDNNF_DARTS_UNIFORM_SAMPLE) e = (i == share || c[i] == 1.f) ? 1.f : 0.f;
    else if (flags & TDNNF_DARTS_FREE_SELECT) e = c[i];
    else e = (i == share) ? 1.f : c[i];
    eff[i] = e;
  }
}

// s_i = <dW_i, W_i> per tap, two-stage and ordered: block (tap, slab) reduces a slab of output rows into partial[tap][slab]
// (float4 reads, no index arithmetic per element), alpha_update_kernel adds a tap's slabs in slab order.
// dots: [K | K * TDNNF_TAP_DOTS_SLABS] doubles (s_i, then the partials).
__global__ __launch_bounds__(256) void tap_dots_kernel(const float *G, int ldg, const float *W, int ldw, int Do, int Di, int K, int vec,
                                                       double *dots) {
  __shared__ double red[4];
  const int tap = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
  const int rows = (Do + nslab - 1) / nslab, o0 = slab * rows, o1 = min(Do, o0 + rows);
  double s = 0;
  if (vec) {  // Di, both leading dimensions and both pointers allow 16-byte reads
    const int q = Di >> 2;
    for (int e = threadIdx.x; e < (o1 - o0) * q; e += 256) {
      const int o = o0 + e / q, d = (e % q) << 2;
      const float4 g = *reinterpret_cast<const float4 *>(G + (long long)o * ldg + tap * Di + d);
      const float4 w = *reinterpret_cast<const float4 *>(W + (long long)o * ldw + tap * Di + d);
      s += ((double)g.x * (double)w.x + (double)g.y * (double)w.y) + ((double)g.z * (double)w.z + (double)g.w * (double)w.w);
    }
  } else {
    for (int e = threadIdx.x; e < (o1 - o0) * Di; e += 256) {
      const int o = o0 + e / Di, d = e % Di;
      s += (double)G[(long long)o * ldg + tap * Di + d] * (double)W[(long long)o * ldw + tap * Di + d];
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dots[K + tap * nslab + slab] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void alpha_update_kernel(double *dots, int nslab, const float *coef, int K, int flags, int share, float temp,
                                    float lr, float *acc) {
  // the slabs of a tap: one per lane, then a fixed shuffle tree (one wave; nslab <= 64) -- a serial chain of K x 64 dependent loads took 65 us
  if (nslab > 0) {
    for (int i = 0; i < K; i++) {
      double s = (int)threadIdx.x < nslab ? dots[K + i * nslab + threadIdx.x] : 0.0;
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (threadIdx.x == 0) dots[i] = s;
    }
  }
  if (threadIdx.x != 0) return;
  if (!(flags & TDNNF_DARTS_UNIFORM_SAMPLE)) {
    for (int i = 0; i < K; i++) {
      const float si = (float)dots[i];
      if (flags & TDNNF_DARTS_FREE_SELECT) {
        acc[i] += si * coef[i];
        acc[i] += -1.0f * si * coef[i] * coef[i];
      } else if (i != share) {
        const float tau = (flags & TDNNF_DARTS_USE_GUMBEL) ? temp : 1.0f;
        for (int j = 0; j < K; j++) acc[j] += (-1.0f * si / tau) * coef[i] * coef[j];
        acc[i] += (si / tau) * coef[i];
      }
    }
  }
  float mul = 1.0f;
  if (flags & TDNNF_DARTS_USE_ENTROPY) mul *= 5.0f;
  if (flags & TDNNF_DARTS_FREE_SELECT) mul *= 5.0f * lr;
  else if (flags & TDNNF_DARTS_USE_GUMBEL) mul *= lr;
  else mul *= 5.0f * lr;
  if (flags & TDNNF_DARTS_UPDATE_ALPHA) mul *= 10000.0f;
  for (int i = 0; i < K; i++) acc[i] *= mul;
}

__global__ __launch_bounds__(256) void dot_partial_kernel(const float *x, const float *y, size_t n, double *partial) {
  __shared__ double red[4];
  double s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) s += (double)x[i] * (double)y[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void axpy_kernel(const float *x, float a, float *y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += a * x[i];
}

}  // namespace
}  // namespace tdnnf

namespace tdnnf {
static thread_local double *g_fro_buf = nullptr;
static thread_local int *g_fro_blocks = nullptr;
FroBoundScope::FroBoundScope(double *buf, int *blocks) : prev_buf(g_fro_buf), prev_blocks(g_fro_blocks) {
  g_fro_buf = buf;
  g_fro_blocks = blocks;
  if (blocks) *blocks = 0;
}
FroBoundScope::~FroBoundScope() {
  g_fro_buf = prev_buf;
  g_fro_blocks = prev_blocks;
}
double *fro_bound_buf() { return g_fro_buf; }
int *fro_bound_blocks() { return g_fro_blocks; }
static thread_local BnSync *g_bn_sync = nullptr;
BnSync *bn_sync_current() { return g_bn_sync; }
BnSyncScope::BnSyncScope(BnSync *b) : prev(g_bn_sync) { g_bn_sync = b; }
BnSyncScope::~BnSyncScope() { g_bn_sync = prev; }

// memo rows 0-2 from partial column sums; with a BnSync installed the sums are all-reduced over the ranks first
static hipError_t bn_fwd_finalize(const float *partial, int chunks, int rows, int cols, float epsilon, float target_rms, float *memo, hipStream_t s,
                                  double *store = nullptr) {
  BnSync *sy = bn_sync_current();
  double *fro2 = fro_bound_buf();
  if (fro2 && fro_bound_blocks()) *fro_bound_blocks() = (int)finalize_grid(cols);
  if (!sy) {
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(finalize_grid(cols)), dim3(kFinThreads), 0, s, partial, chunks, cols, rows, epsilon, target_rms, memo,
                       (double *)nullptr, (const double *)nullptr, store, rows, fro2);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(finalize_grid(cols)), dim3(kFinThreads), 0, s, partial, chunks, cols, rows, epsilon, target_rms, memo, sy->buf,
                     (const double *)nullptr);
  if (sy->fn(sy->ctx, sy->buf, 2LL * cols, (tdnnf_stream)s)) return hipErrorUnknown;
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(finalize_grid(cols)), dim3(kFinThreads), 0, s, partial, chunks, cols, rows * sy->world, epsilon, target_rms, memo,
                     (double *)nullptr, (const double *)sy->buf, store, rows, fro2);  // (the bound then covers all ranks' rows: still an upper bound of this rank's)
  return hipGetLastError();
}
static hipError_t bn_bwd_finalize(const float *partial, int chunks, int rows, int D, float target_rms, float *memo, hipStream_t s) {
  BnSync *sy = bn_sync_current();
  if (!sy) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, chunks, D, rows, target_rms, memo);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, chunks, D, rows, target_rms, memo, sy->buf, (const double *)nullptr);
  if (sy->fn(sy->ctx, sy->buf, 2LL * D, (tdnnf_stream)s)) return hipErrorUnknown;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(finalize_grid(D)), dim3(kFinThreads), 0, s, partial, chunks, D, rows * sy->world, target_rms, memo, (double *)nullptr,
                     (const double *)sy->buf);
  return hipGetLastError();
}
// BatchNorm forward statistics only (memo rows 0-2); the trainer applies them in a fused pass (fused.hip)
hipError_t batchnorm_stats(MatView a, float epsilon, float target_rms, float *memo, void *ws, hipStream_t s, double *store_stats) {
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  hipError_t e = colreduce_partial(1, a, a, (float *)ws, s);
  if (e != hipSuccess) return e;
  return bn_fwd_finalize((const float *)ws, pl.chunks, a.rows, a.cols, epsilon, target_rms, memo, s, store_stats);
}
hipError_t batchnorm_stats_from_partials(const float *partial, int chunks, int rows, int cols, float epsilon, float target_rms, float *memo, hipStream_t s,
                                         double *store_stats) {
  return bn_fwd_finalize(partial, chunks, rows, cols, epsilon, target_rms, memo, s, store_stats);
}
// acc[c] += scale * colsum(a)[c]   (two-stage, float4 loads)
hipError_t colsum_add(MatView a, float scale, float *acc, void *ws, hipStream_t s) {
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  hipError_t e = colreduce_partial(0, a, a, (float *)ws, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(finalize_grid(a.cols)), dim3(kFinThreads), 0, s, (const float *)ws, pl.chunks, a.cols, scale, acc);
  return hipGetLastError();
}
}  // namespace tdnnf

using namespace tdnnf;

// =================================================================================== C-ABI
extern "C" {

size_t tdnnf_colreduce_workspace_bytes(int rows, int cols) { return colreduce_bytes(rows, cols); }

int tdnnf_batchnorm_propagate(const tdnnf_mat *in, float epsilon, float target_rms, tdnnf_mat *out, float *memo,
                              void *ws, size_t ws_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && memo, "batchnorm_propagate: bad matrices");
  TDNNF_REQUIRE(in->rows > 0 && epsilon > 0 && target_rms > 0, "batchnorm_propagate: empty input or bad epsilon/target-rms");
  TDNNF_REQUIRE(ws && ws_bytes >= colreduce_bytes(in->rows, in->cols), "batchnorm_propagate: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  MatView a = view(in), o = view(out);
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  TDNNF_HIP(colreduce_partial(1, a, a, (float *)ws, s));
  TDNNF_HIP(bn_fwd_finalize((const float *)ws, pl.chunks, a.rows, a.cols, epsilon, target_rms, memo, s));
  const bool vec = vec4_ok(a) && vec4_ok(o) && (reinterpret_cast<uintptr_t>(memo) & 15) == 0;
  const long long work = (long long)a.rows * (vec ? a.cols / 4 : a.cols);
  if (vec) hipLaunchKernelGGL((bn_apply_kernel<4>), dim3(grid_for(work, 256)), dim3(256), 0, s, a, memo, memo + 2 * a.cols, o);
  else hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(grid_for(work, 256)), dim3(256), 0, s, a, memo, memo + 2 * a.cols, o);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_batchnorm_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, float target_rms, float *memo,
                             tdnnf_mat *in_deriv, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_value) && mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(out_value, out_deriv) &&
                    same_dim(out_value, in_deriv) && memo,
                "batchnorm_backprop: bad matrices");
  TDNNF_REQUIRE(out_value->rows > 0, "batchnorm_backprop: empty input");
  TDNNF_REQUIRE(ws && ws_bytes >= colreduce_bytes(out_value->rows, out_value->cols), "batchnorm_backprop: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  MatView z = view(out_value), dz = view(out_deriv), dx = view(in_deriv);
  const int D = z.cols;
  ColReducePlan pl = colreduce_plan(z.rows, D);
  TDNNF_HIP(colreduce_partial(2, z, dz, (float *)ws, s));
  TDNNF_HIP(bn_bwd_finalize((const float *)ws, pl.chunks, z.rows, D, target_rms, memo, s));
  const bool vec = vec4_ok(z) && vec4_ok(dz) && vec4_ok(dx);
  const long long work = (long long)z.rows * (vec ? D / 4 : D);
  // dx = (dz + temp) * scale + z * vdm
  if (vec) hipLaunchKernelGGL((colmap_kernel<3, 4>), dim3(grid_for(work, 256)), dim3(256), 0, s, dz, z, memo + 2 * D, memo + 4 * D, memo + 3 * D, dx);
  else hipLaunchKernelGGL((colmap_kernel<3, 1>), dim3(grid_for(work, 256)), dim3(256), 0, s, dz, z, memo + 2 * D, memo + 4 * D, memo + 3 * D, dx);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_batchnorm_store_stats(const float *memo, int D, int num_frames, double *stats, tdnnf_stream stream) {
  TDNNF_REQUIRE(memo && stats && D > 0 && num_frames > 0, "batchnorm_store_stats: bad arguments");
  hipLaunchKernelGGL(bn_store_stats_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, memo, D, num_frames, stats);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_batchnorm_compute_derived(const double *stats, int D, float epsilon, float target_rms, float *scale,
                                    float *offset, tdnnf_stream stream) {
  TDNNF_REQUIRE(stats && scale && offset && D > 0, "batchnorm_compute_derived: bad arguments");
  hipLaunchKernelGGL(bn_derived_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, D, epsilon, target_rms, scale, offset);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_batchnorm_test_propagate(const tdnnf_mat *in, const float *scale, const float *offset, tdnnf_mat *out,
                                   tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && scale && offset, "batchnorm_test_propagate: bad arguments");
  MatView a = view(in), o = view(out);
  if (a.rows == 0) return TDNNF_OK;
  const bool vec = vec4_ok(a) && vec4_ok(o);
  const long long work = (long long)a.rows * (vec ? a.cols / 4 : a.cols);
  if (vec) hipLaunchKernelGGL((colmap_kernel<1, 4>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, a, a, scale, offset, scale, o);
  else hipLaunchKernelGGL((colmap_kernel<1, 1>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, a, a, scale, offset, scale, o);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_batchnorm_test_backprop(const tdnnf_mat *out_deriv, const float *scale, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(out_deriv, in_deriv) && scale, "batchnorm_test_backprop: bad arguments");
  MatView a = view(out_deriv), o = view(in_deriv);
  if (a.rows == 0) return TDNNF_OK;
  const bool vec = vec4_ok(a) && vec4_ok(o);
  const long long work = (long long)a.rows * (vec ? a.cols / 4 : a.cols);
  if (vec) hipLaunchKernelGGL((colmap_kernel<2, 4>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, a, a, scale, scale, scale, o);
  else hipLaunchKernelGGL((colmap_kernel<2, 1>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, a, a, scale, scale, scale, o);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_softmax_flops_propagate(const tdnnf_mat *in, const float *gumbel_u, float temp, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && in->cols > 0, "softmax_flops_propagate: bad matrices");
  TDNNF_REQUIRE(!gumbel_u || temp > 0, "softmax_flops_propagate: temp-proportion must be > 0");
  if (in->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((in->rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(in), gumbel_u,
                     gumbel_u ? 1.0f / temp : 1.0f, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_softmax_flops_backprop(const tdnnf_mat *out_value, tdnnf_mat *out_deriv, float scale, const float *flops, int dim,
                                 float temp, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_value) && mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(out_value, out_deriv) &&
                    same_dim(out_value, in_deriv),
                "softmax_flops_backprop: bad matrices");
  TDNNF_REQUIRE(temp > 0 && (!flops || (dim > 0 && dim <= out_value->cols)), "softmax_flops_backprop: bad temp/dim");
  if (out_value->rows == 0) return TDNNF_OK;
  const float a = scale / out_deriv->rows / out_deriv->cols;
  hipLaunchKernelGGL(softmax_flops_bwd_kernel, dim3((out_value->rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     view(out_value), view(out_deriv), a, flops, dim, 1.0f / temp, view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_onehot_propagate(const float *u, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(u && mat_ok(out), "onehot_propagate: bad arguments");
  if (out->rows * out->cols == 0) return TDNNF_OK;
  hipLaunchKernelGGL(onehot_kernel, dim3(grid_for((long long)out->rows * out->cols, 256)), dim3(256), 0, (hipStream_t)stream, u, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_copyn_propagate(const tdnnf_mat *in, float scale, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && in->rows == out->rows && in->cols > 0 && out->cols % in->cols == 0,
                "copyn_propagate: output-dim must be a multiple of input-dim");
  if (out->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(copyn_fwd_kernel, dim3(grid_for((long long)out->rows * out->cols, 256)), dim3(256), 0, (hipStream_t)stream, view(in), scale, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int tdnnf_copyn_backprop(const tdnnf_mat *out_deriv, float scale, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_deriv) && mat_ok(in_deriv) && in_deriv->rows == out_deriv->rows && in_deriv->cols > 0 &&
                    out_deriv->cols % in_deriv->cols == 0,
                "copyn_backprop: output-dim must be a multiple of input-dim");
  if (in_deriv->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(copyn_bwd_kernel, dim3(grid_for((long long)in_deriv->rows * in_deriv->cols, 256)), dim3(256), 0, (hipStream_t)stream, view(out_deriv), scale, view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_constant_function_propagate(const float *output, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(output && mat_ok(out), "constant_function_propagate: bad arguments");
  if (out->rows * out->cols == 0) return TDNNF_OK;
  hipLaunchKernelGGL(rows_from_vec_kernel, dim3(grid_for((long long)out->rows * out->cols, 256)), dim3(256), 0, (hipStream_t)stream, output, 1.0f, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int tdnnf_constant_function_backprop(const tdnnf_mat *out_deriv, float lr, float *output_acc, void *ws, size_t ws_bytes,
                                     tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_deriv) && output_acc, "constant_function_backprop: bad arguments");
  if (out_deriv->rows == 0) return TDNNF_OK;
  TDNNF_REQUIRE(ws && ws_bytes >= colreduce_bytes(out_deriv->rows, out_deriv->cols), "constant_function_backprop: workspace too small");
  MatView a = view(out_deriv);
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  TDNNF_HIP(colreduce_partial(0, a, a, (float *)ws, (hipStream_t)stream));
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(finalize_grid(a.cols)), dim3(kFinThreads), 0, (hipStream_t)stream, (const float *)ws, pl.chunks, a.cols, 5.0f * lr, output_acc);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

// OnehotFunctionComponent::Backprop nnet-simple-component.cc:9521-9552, the branch the recipes configure
// ("is-updatable=true use-natural-gradient=false", generate_bottleneckCB8share_onehottrain_config.py:12):
// output_.AddRowSumMat(learning_rate, out_deriv); the component has no input derivative
int tdnnf_onehot_backprop(const tdnnf_mat *out_deriv, float lr, float *output_acc, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_deriv) && output_acc, "onehot_backprop: bad arguments");
  if (out_deriv->rows == 0) return TDNNF_OK;
  TDNNF_REQUIRE(ws && ws_bytes >= colreduce_bytes(out_deriv->rows, out_deriv->cols), "onehot_backprop: workspace too small");
  MatView a = view(out_deriv);
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  TDNNF_HIP(colreduce_partial(0, a, a, (float *)ws, (hipStream_t)stream));
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(finalize_grid(a.cols)), dim3(kFinThreads), 0, (hipStream_t)stream, (const float *)ws, pl.chunks, a.cols, lr, output_acc);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_flops_constraint_backprop(const float *flops, float scale, int rows_in, int cols_in, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(flops && mat_ok(in_deriv) && rows_in > 0 && cols_in > 0, "flops_constraint_backprop: bad arguments");
  if (in_deriv->rows * in_deriv->cols == 0) return TDNNF_OK;
  hipLaunchKernelGGL(rows_from_vec_kernel, dim3(grid_for((long long)in_deriv->rows * in_deriv->cols, 256)), dim3(256), 0,
                     (hipStream_t)stream, flops, scale / rows_in / cols_in, view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_elementwise_product_propagate(const tdnnf_mat *in, int output_dim, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && output_dim > 0 && in->cols > output_dim && in->cols % output_dim == 0 &&
                    out->cols == output_dim && out->rows == in->rows,
                "elementwise_product_propagate: input-dim must be a proper multiple of output-dim");
  if (in->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(ewprod_fwd_kernel, dim3(grid_for((long long)in->rows * output_dim, 256)), dim3(256), 0, (hipStream_t)stream, view(in), output_dim, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int tdnnf_elementwise_product_backprop(const tdnnf_mat *in_value, const tdnnf_mat *out_deriv, int output_dim,
                                       tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in_value) && mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(in_value, in_deriv) && output_dim > 0 &&
                    in_value->cols % output_dim == 0 && out_deriv->cols == output_dim && out_deriv->rows == in_value->rows,
                "elementwise_product_backprop: bad dimensions");
  if (in_value->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(ewprod_bwd_kernel, dim3(grid_for((long long)in_value->rows * in_value->cols, 256)), dim3(256), 0, (hipStream_t)stream, view(in_value), view(out_deriv), output_dim, view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_relu_propagate(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out), "relu_propagate: bad matrices");
  TDNNF_HIP(launch_ew<0>(view(in), view(in), 0, 0, view(out), (hipStream_t)stream));
  return TDNNF_OK;
}
int tdnnf_relu_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_value) && mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(out_value, out_deriv) && same_dim(out_value, in_deriv),
                "relu_backprop: bad matrices");
  TDNNF_HIP(launch_ew<1>(view(out_value), view(out_deriv), 0, 0, view(in_deriv), (hipStream_t)stream));
  return TDNNF_OK;
}
int tdnnf_relu_repair(const double *stats, int dim, float self_repair_scale, float lower, float upper, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(stats && mat_ok(in_deriv) && in_deriv->cols == dim, "relu_repair: bad arguments");
  TDNNF_REQUIRE(self_repair_scale >= 0.0f && self_repair_scale < 0.1f, "relu_repair: self-repair-scale out of range");
  if (in_deriv->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(relu_repair_kernel, dim3(grid_for((long long)in_deriv->rows * dim, 256)), dim3(256), 0, (hipStream_t)stream, stats, dim, self_repair_scale, lower, upper, view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int tdnnf_relu_store_stats(const tdnnf_mat *out_value, double *stats, void *ws, size_t ws_bytes, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_value) && stats, "relu_store_stats: bad arguments");
  if (out_value->rows == 0) return TDNNF_OK;
  TDNNF_REQUIRE(ws && ws_bytes >= colreduce_bytes(out_value->rows, out_value->cols), "relu_store_stats: workspace too small");
  MatView a = view(out_value);
  ColReducePlan pl = colreduce_plan(a.rows, a.cols);
  TDNNF_HIP(colreduce_partial(3, a, a, (float *)ws, (hipStream_t)stream));
  hipLaunchKernelGGL(relu_stats_finalize_kernel, dim3((a.cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)ws, pl.chunks, a.cols, a.rows, stats);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_log_softmax_propagate(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && in->cols > 0, "log_softmax_propagate: bad matrices");
  if (in->rows == 0) return TDNNF_OK;
  const dim3 grid(in->rows < 8192 ? in->rows : 8192);
  const bool al = in->stride % 4 == 0 && out->stride % 4 == 0 && ((uintptr_t)in->data & 15) == 0 && ((uintptr_t)out->data & 15) == 0;
  if (al && in->cols <= 256 * 4 * 2) hipLaunchKernelGGL((log_softmax_fwd_regs_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, view(in), view(out), MatView{nullptr, 0, 0, 0}, 0.f);
  else if (al && in->cols <= 256 * 4 * 8) hipLaunchKernelGGL((log_softmax_fwd_regs_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, view(in), view(out), MatView{nullptr, 0, 0, 0}, 0.f);
  else hipLaunchKernelGGL(log_softmax_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, view(in), view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
}  // extern "C"
namespace tdnnf {
// LogSoftmax forward that also leaves aux = aux_scale * softmax(in).  For an output derivative dy whose rows all sum to the
// same known c (xent_regularize * weight * numerator posteriors), the backward pass dy - softmax * rowsum(dy) is then
// aux (aux_scale = -c) plus dy's few non-zeros added on top: no zero fill, no second dense pass.  false: shapes the
// one-pass kernel does not take (nothing launched).
bool log_softmax_propagate_with_aux(const tdnnf_mat *in, tdnnf_mat *out, tdnnf_mat *aux, float aux_scale, hipStream_t s) {
  const bool al = in->stride % 4 == 0 && out->stride % 4 == 0 && aux->stride % 4 == 0 && ((uintptr_t)in->data & 15) == 0 &&
                  ((uintptr_t)out->data & 15) == 0 && ((uintptr_t)aux->data & 15) == 0;
  if (!al || in->cols > 256 * 4 * 8 || in->rows == 0) return false;
  const dim3 grid(in->rows < 8192 ? in->rows : 8192);
  if (in->cols <= 256 * 4 * 2) hipLaunchKernelGGL((log_softmax_fwd_regs_kernel<2>), grid, dim3(256), 0, s, view(in), view(out), view(aux), aux_scale);
  else hipLaunchKernelGGL((log_softmax_fwd_regs_kernel<8>), grid, dim3(256), 0, s, view(in), view(out), view(aux), aux_scale);
  return true;
}
}  // namespace tdnnf
extern "C" {
int tdnnf_log_softmax_backprop(const tdnnf_mat *out_value, const tdnnf_mat *out_deriv, tdnnf_mat *in_deriv, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(out_value) && mat_ok(out_deriv) && mat_ok(in_deriv) && same_dim(out_value, out_deriv) && same_dim(out_value, in_deriv),
                "log_softmax_backprop: bad matrices");
  if (out_value->rows == 0) return TDNNF_OK;
  const dim3 grid(out_value->rows < 8192 ? out_value->rows : 8192);
  const bool al = out_value->stride % 4 == 0 && out_deriv->stride % 4 == 0 && in_deriv->stride % 4 == 0 && ((uintptr_t)out_value->data & 15) == 0 &&
                  ((uintptr_t)out_deriv->data & 15) == 0 && ((uintptr_t)in_deriv->data & 15) == 0;
  if (al && out_value->cols <= 256 * 4 * 2) hipLaunchKernelGGL((log_softmax_bwd_regs_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, view(out_value), view(out_deriv), view(in_deriv));
  else if (al && out_value->cols <= 256 * 4 * 8) hipLaunchKernelGGL((log_softmax_bwd_regs_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, view(out_value), view(out_deriv), view(in_deriv));
  else hipLaunchKernelGGL(log_softmax_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, view(out_value), view(out_deriv), view(in_deriv));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_sum_scaled(const tdnnf_mat *a, float sa, const tdnnf_mat *b, float sb, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(a) && mat_ok(out) && same_dim(a, out) && (!b || (mat_ok(b) && same_dim(b, out))), "sum_scaled: bad matrices");
  if (b) TDNNF_HIP(launch_ew<2>(view(a), view(b), sa, sb, view(out), (hipStream_t)stream));
  else TDNNF_HIP(launch_ew<3>(view(a), view(a), sa, 0, view(out), (hipStream_t)stream));
  return TDNNF_OK;
}
int tdnnf_add_scaled(const tdnnf_mat *a, float sc, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(a) && mat_ok(out) && same_dim(a, out), "add_scaled: bad matrices");
  TDNNF_HIP(launch_ew<4>(view(a), view(a), sc, 0, view(out), (hipStream_t)stream));
  return TDNNF_OK;
}
int tdnnf_general_dropout(const tdnnf_mat *in, const float *mask, int num_seq, tdnnf_mat *out, tdnnf_stream stream) {
  TDNNF_REQUIRE(mat_ok(in) && mat_ok(out) && same_dim(in, out) && mask && num_seq > 0 && in->rows % num_seq == 0, "general_dropout: bad arguments");
  if (in->rows == 0) return TDNNF_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((long long)in->rows * in->cols, 256)), dim3(256), 0, (hipStream_t)stream, view(in), mask, num_seq, view(out));
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_general_dropout_mask(const float *uniform, long long n, float proportion, int continuous, float *mask, tdnnf_stream stream) {
  TDNNF_REQUIRE(n >= 0 && (n == 0 || (uniform && mask)) && proportion >= 0.f && proportion < 1.f, "general_dropout_mask: bad arguments (proportion in [0, 1))");
  if (n == 0) return TDNNF_OK;
  hipLaunchKernelGGL(general_dropout_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, uniform, n, proportion, continuous, mask);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_tdnn_darts_coef(const float *log_alpha, int K, int flags, float temp, const float *gumbel_u, const float *sample_u,
                          int share_index, float *coef_memo, float *eff_coef, tdnnf_stream stream) {
  TDNNF_REQUIRE(log_alpha && coef_memo && eff_coef && K >= 1 && K <= TDNNF_MAX_OFFSETS, "tdnn_darts_coef: K out of range");
  TDNNF_REQUIRE(!(flags & TDNNF_DARTS_USE_GUMBEL) || (gumbel_u && temp > 0), "tdnn_darts_coef: gumbel mode needs draws and temp > 0");
  TDNNF_REQUIRE(!(flags & TDNNF_DARTS_UNIFORM_SAMPLE) || sample_u, "tdnn_darts_coef: uniform-sample mode needs a draw");
  TDNNF_REQUIRE(share_index >= 0 && share_index < K, "tdnn_darts_coef: bad share index");
  hipLaunchKernelGGL(darts_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, log_alpha, K, flags, temp, gumbel_u, sample_u, share_index, coef_memo, eff_coef);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

int tdnnf_tdnn_darts_alpha_update(const float *tap_grad, int ldg, const float *W, int ldw, int Do, int Di, int K,
                                  const float *coef_memo, int flags, int share_index, float temp, float lr,
                                  float *alpha_acc, double *tap_dots, tdnnf_stream stream) {
  // uniform-sample mode adds no gradient (the reference computes and discards it, :502-507): only the scalings run and
  // tap_grad may be null
  const bool uniform = (flags & TDNNF_DARTS_UNIFORM_SAMPLE) != 0;
  TDNNF_REQUIRE((tap_grad || uniform) && W && coef_memo && alpha_acc && tap_dots, "tdnn_darts_alpha_update: null pointer (tap_dots_dev is required scratch of TDNNF_TAP_DOTS_DOUBLES(K) doubles)");
  TDNNF_REQUIRE(K >= 1 && K <= TDNNF_MAX_OFFSETS && Do > 0 && Di > 0 && ldg >= K * Di && ldw >= K * Di, "tdnn_darts_alpha_update: bad dimensions");
  const int nslab = std::min(TDNNF_TAP_DOTS_SLABS, Do);
  if (tap_grad) {
    const int vec = Di % 4 == 0 && ldg % 4 == 0 && ldw % 4 == 0 && ((reinterpret_cast<uintptr_t>(tap_grad) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
    hipLaunchKernelGGL(tap_dots_kernel, dim3(K, nslab), dim3(256), 0, (hipStream_t)stream, tap_grad, ldg, W, ldw, Do, Di, K, vec, tap_dots);
  }
  hipLaunchKernelGGL(alpha_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tap_dots, tap_grad ? nslab : 0, coef_memo, K, flags, share_index, temp, lr, alpha_acc);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

// UpdatableComponent::DotProduct (e.g. /root/reference/src/nnet3/nnet-tdnn-component.cc:949-958: TraceMatMat + VecVec): <x, y> in double
// over two device vectors; the result goes to the host, so the call synchronises the stream (model combination / diagnostics, not the
// training step).  Two stages, fixed order: 256 block partials, then the host adds them.
int tdnnf_dot(const float *x, const float *y, size_t n, double *result_host, tdnnf_stream stream) {
  TDNNF_REQUIRE(result_host && (n == 0 || (x && y)), "dot: null pointer");
  *result_host = 0.0;
  if (n == 0) return TDNNF_OK;
  constexpr int kBlocks = 256;
  double *partial = nullptr;
  TDNNF_HIP(hipMalloc((void **)&partial, sizeof(double) * kBlocks));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(kBlocks), dim3(256), 0, (hipStream_t)stream, x, y, n, partial);
  double host[kBlocks];
  hipError_t e = hipMemcpyAsync(host, partial, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(partial);
  TDNNF_HIP(e);
  double acc = 0;
  for (int i = 0; i < kBlocks; i++) acc += host[i];
  *result_host = acc;
  return TDNNF_OK;
}

int tdnnf_axpy(const float *x, float a, float *y, size_t n, tdnnf_stream stream) {
  TDNNF_REQUIRE(x && y, "axpy: null pointer");
  if (n == 0 || a == 0.0f) return TDNNF_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, x, a, y, n);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

}  // extern "C"
