// optim.hip -- per-minibatch optimizer step on gfx950 without host round trips:
// ConstrainOrthonormalInternal, UpdateNnetWithMaxChange (/root/reference/src/nnet3/nnet-utils.cc:914-1032,
// :2085-2175).  The reference reads Trace / DotProduct results back to the host for every component
// (~60 D2H syncs per minibatch); here the scalars stay in device memory and the
// scale factors are applied by the next kernel.
#include <string.h>

#include "common.h"
#include "gemm_f32.h"

namespace tdnnf {
namespace {

// scal[0] = -4*alpha (coefficient of the update GEMM), scal[1] = scale, scal[2] = ratio.  Also P -= scale^2 I.
__global__ __launch_bounds__(256) void ortho_scalars_kernel(float *P, int rows, float scale_in, float *scal) {
  __shared__ double red[2][4];
  double tr = 0, trpp = 0;
  for (int e = threadIdx.x; e < rows * rows; e += 256) {
    const double v = P[e];
    trpp += v * v;
    if (e / rows == e % rows) tr += v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    tr += __shfl_xor(tr, o, 64);
    trpp += __shfl_xor(trpp, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = tr;
    red[1][threadIdx.x >> 6] = trpp;
  }
  __syncthreads();
  tr = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  trpp = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  float update_speed = 0.125f, scale = scale_in, ratio = 1.f;
  if (scale_in < 0.f) {  // floating scale, nnet-utils.cc:938-986
    scale = sqrtf((float)(trpp / tr));
    ratio = (float)(trpp * rows / (tr * tr));
    if (ratio > 1.02f) {
      update_speed *= 0.5f;
      if (ratio > 1.1f) update_speed *= 0.5f;
    }
  }
  for (int i = threadIdx.x; i < rows; i += 256) P[i * rows + i] -= scale * scale;  // :987
  if (threadIdx.x == 0) {
    scal[0] = -4.0f * (update_speed / (scale * scale));  // :1019, :1030
    scal[1] = scale;
    scal[2] = ratio;
  }
}

struct CompTable {
  long long begin[129];
  float max_change[128];
};
constexpr int kDotBlocks = 32;

__global__ __launch_bounds__(256) void comp_dot_kernel(const float *delta, CompTable tb, double *partial) {
  __shared__ double red[4];
  const int c = blockIdx.y;
  const long long b = tb.begin[c], e = tb.begin[c + 1];
  double s = 0;
  for (long long i = b + blockIdx.x * 256LL + threadIdx.x; i < e; i += kDotBlocks * 256LL) {
    const double v = delta[i];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[c * kDotBlocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// UpdateNnetWithMaxChange :2095-2172 (single thread; num_comp <= 128)
__global__ void max_change_factors_kernel(const double *partial, CompTable tb, int n, float max_param_change,
                                          float max_change_scale, float scale, float *factors, float *info) {
  if (threadIdx.x != 0) return;
  float param_delta_squared = 0.f;
  for (int i = 0; i < n; i++) {
    double d = 0;
    for (int b = 0; b < kDotBlocks; b++) d += partial[i * kDotBlocks + b];
    const float dot = (float)d, mc = tb.max_change[i];
    float f = 1.0f;
    if (mc != 0.f && sqrtf(dot) * fabsf(scale) > mc * max_change_scale) f = mc * max_change_scale / (sqrtf(dot) * fabsf(scale));
    factors[i] = f;
    param_delta_squared += f * f * dot;
  }
  float param_delta = sqrtf(param_delta_squared) * fabsf(scale);
  float ok = 1.f;
  if (max_param_change != 0.f && param_delta > max_param_change * max_change_scale) {
    if (param_delta - param_delta != 0.f) ok = 0.f;  // infinite change: do not apply (:2144-2147)
    else scale *= max_param_change * max_change_scale / param_delta;
  }
  for (int i = 0; i < n; i++) factors[i] = ok != 0.f ? factors[i] * scale : 0.f;
  factors[n] = ok;
  if (info) {
    for (int i = 0; i <= n; i++) info[i] = factors[i];
  }
}

__global__ __launch_bounds__(256) void comp_apply_kernel(float *params, float *delta, CompTable tb, const float *factors,
                                                         int zero_delta) {
  const int c = blockIdx.y;
  const long long b = tb.begin[c], e = tb.begin[c + 1];
  const float f = factors[c];
  for (long long i = b + blockIdx.x * 256LL + threadIdx.x; i < e; i += gridDim.x * 256LL) {
    if (f != 0.f) params[i] += f * delta[i];
    if (zero_delta) delta[i] = 0.f;
  }
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

size_t tdnnf_constrain_orthonormal_workspace_bytes(int rows, int cols) {
  return sizeof(float) * ((size_t)rows * rows + (size_t)rows * cols + 8) + 64;
}

int tdnnf_constrain_orthonormal(float scale, float *M, int rows, int cols, int ld, void *ws, size_t ws_bytes,
                                tdnnf_stream stream) {
  TDNNF_REQUIRE(M && rows > 0 && cols > 0 && ld >= cols, "constrain_orthonormal: bad matrix");
  TDNNF_REQUIRE(scale != 0.0f, "constrain_orthonormal: scale must be nonzero (nnet-utils.cc:915)");
  TDNNF_REQUIRE(rows <= cols, "constrain_orthonormal: pass the transpose when rows > cols (nnet-utils.cc:1068-1075)");
  TDNNF_REQUIRE(ws && ws_bytes >= tdnnf_constrain_orthonormal_workspace_bytes(rows, cols), "constrain_orthonormal: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float *scal = (float *)ws;  // 8 floats, 16-byte aligned block first
  float *P = scal + 8;
  float *U = P + (size_t)rows * rows;
  RowsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = M; a.lda = ld; a.B = M; a.ldb = ld; a.C = P; a.ldc = rows; a.M = rows; a.N = rows; a.init_mode = 2; a.nseg = 1;
  a.seg[0].klen = cols; a.seg[0].m_lo = 0; a.seg[0].m_hi = rows;
  TDNNF_HIP(rows_gemm(a, true, s));  // P = M M^T  (SymAddMat2 + CopyLowerToUpper)
  hipLaunchKernelGGL(ortho_scalars_kernel, dim3(1), dim3(256), 0, s, P, rows, scale, scal);
  RowsGemmArgs b;
  memset(&b, 0, sizeof(b));
  b.A = P; b.lda = rows; b.B = M; b.ldb = ld; b.C = U; b.ldc = cols; b.M = rows; b.N = cols; b.init_mode = 2; b.nseg = 1;
  b.coef = scal;  // -4*alpha, read on device
  b.seg[0].klen = rows; b.seg[0].m_lo = 0; b.seg[0].m_hi = rows;
  TDNNF_HIP(rows_gemm(b, false, s));  // U = -4 alpha (P - s^2 I) M
  tdnnf_mat um{U, rows, cols, cols}, mm{M, rows, cols, ld};
  return tdnnf_add_scaled(&um, 1.0f, &mm, stream);  // M += U
}

size_t tdnnf_max_change_workspace_bytes(int num_comp) {
  return sizeof(double) * (size_t)num_comp * kDotBlocks + sizeof(float) * (num_comp + 2) + 64;
}

int tdnnf_update_with_max_change(float *params, float *delta, int num_comp, const long long *comp_begin,
                                 const float *max_change, float max_param_change, float max_change_scale, float scale,
                                 int zero_delta, void *ws, size_t ws_bytes, float *info, tdnnf_stream stream) {
  TDNNF_REQUIRE(params && delta && comp_begin && max_change && num_comp > 0 && num_comp <= 128,
                "update_with_max_change: need 1..128 components");
  TDNNF_REQUIRE(ws && ws_bytes >= tdnnf_max_change_workspace_bytes(num_comp), "update_with_max_change: workspace too small");
  CompTable tb;
  memset(&tb, 0, sizeof(tb));
  for (int i = 0; i <= num_comp; i++) {
    tb.begin[i] = comp_begin[i];
    TDNNF_REQUIRE(i == 0 || comp_begin[i] >= comp_begin[i - 1], "update_with_max_change: component offsets must be non-decreasing");
  }
  for (int i = 0; i < num_comp; i++) {
    TDNNF_REQUIRE(max_change[i] >= 0.0f, "update_with_max_change: max-change must be >= 0 (nnet-utils.cc:2111)");
    tb.max_change[i] = max_change[i];
  }
  hipStream_t s = (hipStream_t)stream;
  double *partial = (double *)ws;
  float *factors = (float *)(partial + (size_t)num_comp * kDotBlocks);
  hipLaunchKernelGGL(comp_dot_kernel, dim3(kDotBlocks, num_comp), dim3(256), 0, s, delta, tb, partial);
  hipLaunchKernelGGL(max_change_factors_kernel, dim3(1), dim3(64), 0, s, partial, tb, num_comp, max_param_change,
                     max_change_scale, scale, factors, info);
  hipLaunchKernelGGL(comp_apply_kernel, dim3(kDotBlocks, num_comp), dim3(256), 0, s, params, delta, tb, factors, zero_delta);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

}  // extern "C"
