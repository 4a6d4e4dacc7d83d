// ggemm.h -- grouped exact-f32 GEMM over device-resident task lists (one 64 x 64 output tile x one K slice per block), shared by the
// natural-gradient side chain (ng_group.hip) and the grouped optimizer step (optim_group.hip).  Included into each translation unit's
// anonymous namespace: the kernels are per-TU copies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace tdnnf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

inline int pad4(int x) { return (x + 3) & ~3; }

// ------------------------------------------------------------------------------------------------ generic grouped GEMM
// One task = one 64 x 64 output tile (x one K slice): C[m][n] (op)= alpha * sum_{k in [k0, k1)} A(m, k) B(k, n) with
// A(m, k) = A[m sam + k sak], B(k, n) = B[k sbk + n sbn] (any of the four orientations; the contiguous one is loaded 16 bytes
// at a time).  4 waves, each a 32 x 32 block of v_mfma_f32_32x32x2_f32 (exact f32), K step 16 through LDS.
struct GTask {
  const float *A, *B;
  float *C;
  long long sam, sak, sbk, sbn, ldc;
  int M, N, k0, k1, m0, n0;
  float alpha;
  int mode;  // 0: C = alpha acc   1: C += alpha acc   2: raw partial tile (C = 64 x 64 slot, ldc = 64)
  int vecA, vecB;
};
struct RTask {  // sums the K slices of one output tile: C (op)= alpha * sum_s part[s]
  float *C;
  const float *part;
  long long ldc;
  int M, N, m0, n0, nsplit, mode;
  float alpha;
};

constexpr int GT = 64, GK = 16, GLD = GT + 4;

__device__ __forceinline__ void ggemm_body(const GTask &p, float (*As)[GLD], float (*Bs)[GLD]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.M, N = p.N, m0 = p.m0, n0 = p.n0, k1 = p.k1;
  const long long sam = p.sam, sak = p.sak, sbk = p.sbk, sbn = p.sbn;
  const float *A = p.A, *B = p.B;
  const bool a_kc = sak == 1, b_kc = sbk == 1;  // k-contiguous operands: a thread owns 4 consecutive k of one row / column
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  float ra[4], rb[4];
  auto load = [&](int kc) {
    if (a_kc) {
      const int m = m0 + (t >> 2), k = kc + (t & 3) * 4;
      const float *q = A + (long long)m * sam + k;
      if (m < M && p.vecA && k + 3 < k1) {
        const float4 v = *reinterpret_cast<const float4 *>(q);
        ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) ra[j] = (m < M && k + j < k1) ? q[j] : 0.f;
      }
    } else {
      const int k = kc + (t >> 4), m = m0 + (t & 15) * 4;
      const float *q = A + (long long)m * sam + (long long)k * sak;
      if (k < k1 && p.vecA && sam == 1 && m + 3 < M) {
        const float4 v = *reinterpret_cast<const float4 *>(q);
        ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) ra[j] = (k < k1 && m + j < M) ? q[(long long)j * sam] : 0.f;
      }
    }
    if (b_kc) {
      const int n = n0 + (t >> 2), k = kc + (t & 3) * 4;
      const float *q = B + (long long)n * sbn + k;
      if (n < N && p.vecB && k + 3 < k1) {
        const float4 v = *reinterpret_cast<const float4 *>(q);
        rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) rb[j] = (n < N && k + j < k1) ? q[j] : 0.f;
      }
    } else {
      const int k = kc + (t >> 4), n = n0 + (t & 15) * 4;
      const float *q = B + (long long)k * sbk + (long long)n * sbn;
      if (k < k1 && p.vecB && sbn == 1 && n + 3 < N) {
        const float4 v = *reinterpret_cast<const float4 *>(q);
        rb[0] = v.x; rb[1] = v.y; rb[2] = v.z; rb[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) rb[j] = (k < k1 && n + j < N) ? q[(long long)j * sbn] : 0.f;
      }
    }
  };
  auto store = [&]() {
    if (a_kc) {
#pragma unroll
      for (int j = 0; j < 4; j++) As[(t & 3) * 4 + j][t >> 2] = ra[j];
    } else {
      *reinterpret_cast<float4 *>(&As[t >> 4][(t & 15) * 4]) = make_float4(ra[0], ra[1], ra[2], ra[3]);
    }
    if (b_kc) {
#pragma unroll
      for (int j = 0; j < 4; j++) Bs[(t & 3) * 4 + j][t >> 2] = rb[j];
    } else {
      *reinterpret_cast<float4 *>(&Bs[t >> 4][(t & 15) * 4]) = make_float4(rb[0], rb[1], rb[2], rb[3]);
    }
  };
  int kc = p.k0;
  if (kc < k1) {
    load(kc);
    for (;;) {
      __syncthreads();  // the previous tile's fragment reads are done
      store();
      __syncthreads();
      kc += GK;
      const bool more = kc < k1;
      if (more) load(kc);  // in flight under the MFMAs
#pragma unroll
      for (int kk = 0; kk < GK; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + lh][wm * 32 + li], Bs[kk + lh][wn * 32 + li], acc, 0, 0, 0);
      if (!more) break;
    }
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int n = n0 + wn * 32 + li;
  const float alpha = p.alpha;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, m = m0 + row;
    if (p.mode == 2) {
      p.C[row * GT + wn * 32 + li] = acc[r];
    } else if (m < M && n < N) {
      float *c = p.C + (long long)m * p.ldc + n;
      *c = p.mode == 1 ? *c + alpha * acc[r] : alpha * acc[r];
    }
  }
}

__global__ __launch_bounds__(256) void ggemm_kernel(const GTask *tasks) {
  __shared__ float As[GK][GLD], Bs[GK][GLD];
  ggemm_body(tasks[blockIdx.x], As, Bs);
}

// (four blocks per tile, 16 rows each: the K slices are added serially per element, so the chain is nsplit loads long)
__device__ __forceinline__ void ggemm_reduce_body(const RTask &p, int quarter) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = quarter * 1024 + j * 256 + t;
    const int row = e / GT, col = e % GT, m = p.m0 + row, n = p.n0 + col;
    if (m >= p.M || n >= p.N) continue;
    const float *src = p.part + e;
    float v = 0.f;
    int s = 0;
    for (; s + 3 < p.nsplit; s += 4) {
      const float v0 = src[(size_t)s * GT * GT], v1 = src[(size_t)(s + 1) * GT * GT], v2 = src[(size_t)(s + 2) * GT * GT], v3 = src[(size_t)(s + 3) * GT * GT];
      v += v0; v += v1; v += v2; v += v3;
    }
    for (; s < p.nsplit; s++) v += src[(size_t)s * GT * GT];
    float *c = p.C + (long long)m * p.ldc + n;
    *c = p.mode == 1 ? *c + p.alpha * v : p.alpha * v;
  }
}
__global__ __launch_bounds__(256) void ggemm_reduce_kernel(const RTask *tasks) { ggemm_reduce_body(tasks[blockIdx.x >> 2], blockIdx.x & 3); }

// The same two kernels over a SELECTION of precomputed task ranges (a launch's blocks = the concatenation of up to 32 ranges of the
// device-resident lists): block b belongs to range j with start[j] <= b < start[j + 1] and runs list entry base[j] + (b - start[j]).
struct SelRanges {
  int n;
  int start[33];
  int base[32];
  __host__ __device__ int find(int b, int *local) const {
    int j = 0;
    while (j + 1 < n && start[j + 1] <= b) j++;
    *local = b - start[j];
    return j;
  }
};
__global__ __launch_bounds__(256) void ggemm_sel_kernel(const GTask *tasks, SelRanges sel) {
  __shared__ float As[GK][GLD], Bs[GK][GLD];
  int local;
  const int j = sel.find((int)blockIdx.x, &local);
  ggemm_body(tasks[sel.base[j] + local], As, Bs);
}
__global__ __launch_bounds__(256) void ggemm_reduce_sel_kernel(const RTask *tasks, SelRanges sel) {  // (ranges in units of blocks: 4 per task)
  int local;
  const int j = sel.find((int)blockIdx.x, &local);
  ggemm_reduce_body(tasks[sel.base[j] + (local >> 2)], local & 3);
}


inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// host-side builder of ggemm task lists
struct GemmList {
  std::vector<GTask> tasks;
  std::vector<RTask> rtasks;
  size_t slots = 0;  // 64 x 64 partial tiles needed
  // C (M x N, ldc) (op)= alpha * A (M x K) * B (K x N); partial slot offsets are relative (fixed up against the buffer later)
  // (whole_k: one K slice per tile whatever K is -- no reduction stage)
  void add(const float *A, long long sam, long long sak, const float *B, long long sbk, long long sbn, float *C, long long ldc, int M, int N, int K,
           float alpha, int mode, bool whole_k = false) {
    // a task is latency-bound (one K step of look-ahead, 8 MFMAs per step): short K slices on many CUs, not long ones on few
    int nsplit = (K + 127) / 128;
    if (nsplit > 64) nsplit = 64;
    if (whole_k) nsplit = 1;
    int kchunk = (((K + nsplit - 1) / nsplit) + GK - 1) / GK * GK;
    nsplit = (K + kchunk - 1) / kchunk;
    const int vecA = al16(A) && (sak == 1 ? sam % 4 == 0 : (sam == 1 && sak % 4 == 0));
    const int vecB = al16(B) && (sbk == 1 ? sbn % 4 == 0 : (sbn == 1 && sbk % 4 == 0));
    for (int m0 = 0; m0 < M; m0 += GT)
      for (int n0 = 0; n0 < N; n0 += GT) {
        if (nsplit == 1) {
          tasks.push_back(GTask{A, B, C, sam, sak, sbk, sbn, ldc, M, N, 0, K, m0, n0, alpha, mode, vecA, vecB});
          continue;
        }
        rtasks.push_back(RTask{C, (const float *)(slots * GT * GT * sizeof(float)), ldc, M, N, m0, n0, nsplit, mode, alpha});
        for (int s = 0; s < nsplit; s++) {
          tasks.push_back(GTask{A, B, (float *)((slots + s) * GT * GT * sizeof(float)), sam, sak, sbk, sbn, GT, M, N, s * kchunk,
                                std::min(K, (s + 1) * kchunk), m0, n0, 1.0f, 2, vecA, vecB});
        }
        slots += nsplit;
      }
  }
  void fixup(float *part) {
    for (auto &t : tasks)
      if (t.mode == 2) t.C = part + (size_t)t.C / sizeof(float);
    for (auto &r : rtasks) r.part = part + (size_t)r.part / sizeof(float);
  }
};

struct DevList {  // a GemmList on the device
  GTask *tasks = nullptr;
  RTask *rtasks = nullptr;
  int nt = 0, nr = 0;
};

inline int launch_list(const DevList &l, hipStream_t s) {
  if (l.nt) hipLaunchKernelGGL(ggemm_kernel, dim3(l.nt), dim3(256), 0, s, l.tasks);
  if (l.nr) hipLaunchKernelGGL(ggemm_reduce_kernel, dim3(4 * l.nr), dim3(256), 0, s, l.rtasks);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

template <class T>
T *carve(char *&p, size_t n) {
  p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  T *r = (T *)p;
  p += sizeof(T) * n;
  return r;
}

}  // namespace
}  // namespace tdnnf
