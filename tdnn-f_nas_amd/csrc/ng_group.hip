// ng_group.hip -- the latency-bound half of OnlineNaturalGradient for MANY components at once (ng.h, "Grouped side chain").
//
// Per component and side the reference's PreconditionDirections (UPSTREAM Kaldi nnet3/natural-gradient-online.cc; call sites
// /root/reference/src/nnet3/nnet-tdnn-component.cc:598-599, nnet-simple-component.cc:3001-3002) needs, after the N-sized pass
// H = X W^T:  L = H^T H, tr(X^ X^^T) = tr(XX^T) - 2 tr(L) + <L, W W^T> and the scale; on a refresh K = J J^T and the R x R
// eigen-problem (host, host_linalg.h).  The trainer then projects the raw gradient, T <- (I - Wy^T Wy) T (I - Wx^T Wx), and adds
// a b T to the minibatch's gradient (nnet-tdnn-component.cc:604-624).  Run per component that is ~17 launches of a few
// microseconds of work each, 36 components per step: round 2's step at the recipes' minibatch was a chain of ~1 290 dependent
// launches.  Here every stage is ONE launch over all components of a group (a gradient bucket of the trainer):
//   1  bias columns into T                                   ng_set_columns_kernel
//   2  L partials, slab-parallel over the rows of H          ng_l_partial_kernel      (f32 MFMA, symmetric tiles only)
//   3  L, traces, scale per (component, side)                ng_l_finish_kernel
//   4  Q = T Wx^T      5  T -= Q Wx      6  P = Wy T      7  T -= Wy^T P           ggemm_kernel (+ ggemm_reduce_kernel when K is split)
//   8  gradient += a b T                                     ng_commit_kernel
//   refresh steps:  K = J J^T (ggemm), K / L / tr(XX^T) to pinned host memory (ng_stage_kernel), one event for the pool threads;
//   next step:      W_{t+1} = A_t (J + diag(c) W_t), W^T, last column, W W^T for ALL refreshed objects (ng_group_finalize).
// Results are deterministic: every reduction has a fixed order (no atomics).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gemm_f32.h"
#include "ggemm.h"
#include "ng.h"

namespace tdnnf {
namespace {

// ------------------------------------------------------------------------------------------------ L = H^T H, traces, scale
struct PairDesc {  // one (component, side)
  const float *H;
  const double *part;  // ||X||^2 partials of the pass that formed H
  float *Ld;
  const float *WWT;
  double *scal;
  float *scale_f;
  // refresh
  const float *Kd;
  float *hK, *hL;  // pinned host memory (device-visible)
  double *h_tr0;
  double ones_term;
  int N, Rp, nt, npart;
  int slab0, nslab, rows_per_slab;
};
__device__ __forceinline__ int ntile_pairs(int nt) { return nt * (nt + 1) / 2; }

// block (pair, slab): the slab's contribution to the upper-triangular 32 x 32 tiles of H^T H, as raw accumulator images
// [tile pair][register][lane].  A wave takes every fourth pair of rows; lane (li, lh) holds H[row + lh][32 c + li] for the
// column tiles c, which is both the A fragment (A[i = li][k = lh]) and the B fragment (B[k = lh][j = li]) of the MFMA.
template <int NT>
__device__ __forceinline__ void l_partial_body(const PairDesc &p, int slab, float *partial, float *lds) {
  constexpr int NP = NT * (NT + 1) / 2;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int Rp = p.Rp, r0 = slab * p.rows_per_slab, r1 = min(p.N, r0 + p.rows_per_slab);
  f32x16 acc[NP];
#pragma unroll
  for (int q = 0; q < NP; q++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
  bool cv[NT];
#pragma unroll
  for (int c = 0; c < NT; c++) cv[c] = c * 32 + li < Rp;
  constexpr int U = 4;  // pairs of rows requested together
  for (int base = r0 + 2 * wave; base < r1; base += 8 * U) {
    float a[U][NT];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int row = base + 8 * u + lh;
      const float *h = p.H + (size_t)row * Rp + li;
#pragma unroll
      for (int c = 0; c < NT; c++) a[u][c] = (row < r1 && cv[c]) ? h[c * 32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = i; j < NT; j++, q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], a[u][j], acc[q], 0, 0, 0);
    }
  }
  // waves 2, 3 -> LDS, waves 0, 1 add; wave 1 -> LDS, wave 0 adds and stores (fixed order)
  float *mine = lds + (size_t)(wave & 1) * NP * 1024;
  if (wave >= 2) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) mine[(q * 16 + r) * 64 + lane] = acc[q][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[q][r] += mine[(q * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) lds[(q * 16 + r) * 64 + lane] = acc[q][r];
  }
  __syncthreads();
  if (wave == 0) {
    float *out = partial + (size_t)(p.slab0 + slab) * 6 * 1024;
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) out[(q * 16 + r) * 64 + lane] = acc[q][r] + lds[(q * 16 + r) * 64 + lane];
  }
}

__global__ __launch_bounds__(256) void ng_l_partial_kernel(const PairDesc *pairs, int npairs, float *partial) {
  extern __shared__ float lds[];
  int pi = 0;
  {  // block -> (pair, slab): the pairs' slab ranges are consecutive
    int lo = 0, hi = npairs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pairs[mid].slab0 <= (int)blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    pi = lo;
  }
  const PairDesc &p = pairs[pi];
  const int slab = blockIdx.x - p.slab0;
  if (p.nt == 1) l_partial_body<1>(p, slab, partial, lds);
  else if (p.nt == 2) l_partial_body<2>(p, slab, partial, lds);
  else l_partial_body<3>(p, slab, partial, lds);
}

// one block (1024 threads: an accumulator image per pass) per (component, side): L from the slab partials (slabs added in order), then
//   tr0 = sum ||X||^2 partials + ones_term,  tr1 = tr0 - 2 tr(L) + <L, W W^T>,  scale = sqrt(tr0 / tr1)
__global__ __launch_bounds__(1024) void ng_l_finish_kernel(const PairDesc *pairs, const float *partial) {
  __shared__ double red[3][16];
  const PairDesc &p = pairs[blockIdx.x];
  const int t = threadIdx.x, Rp = p.Rp, nt = p.nt;
  double a = 0, b = 0, c = 0;
  for (int i = t; i < p.npart; i += 1024) a += p.part[i];
  int q = 0;
  for (int ti = 0; ti < nt; ti++)
    for (int tj = ti; tj < nt; tj++, q++) {
      const int e = t, r = e >> 6, lane = e & 63;
      const int m = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = tj * 32 + (lane & 31);
      if (m >= Rp || n >= Rp) continue;
      const float *src = partial + (size_t)p.slab0 * 6 * 1024 + (size_t)q * 1024 + e;
      float v = 0.f;
      int s = 0;
      for (; s + 7 < p.nslab; s += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = src[(size_t)(s + u) * 6144];
#pragma unroll
        for (int u = 0; u < 8; u++) v += w[u];
      }
      for (; s < p.nslab; s++) v += src[(size_t)s * 6144];
      p.Ld[m * Rp + n] = v;
      const double w = (double)v * (double)p.WWT[m * Rp + n];
      if (ti != tj) {
        p.Ld[n * Rp + m] = v;
        c += 2.0 * w;
      } else {
        c += w;
        if (m == n) b += v;
      }
    }
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  if ((t & 63) == 0) {
    red[0][t >> 6] = a;
    red[1][t >> 6] = b;
    red[2][t >> 6] = c;
  }
  __syncthreads();
  if (t == 0) {
    double tr0 = p.ones_term, trL = 0, trLW = 0;
    for (int w = 0; w < 16; w++) {
      tr0 += red[0][w];
      trL += red[1][w];
      trLW += red[2][w];
    }
    const double tr1 = tr0 - 2.0 * trL + trLW;
    p.scal[0] = tr0;
    p.scal[1] = tr1;
    *p.scale_f = (tr0 <= 0.0 || !(tr1 > 0.0)) ? 1.0f : (float)sqrt(tr0 / tr1);
  }
}

// refresh: K, L and tr(XX^T) of every pair to the pinned buffers the pool threads read
__global__ __launch_bounds__(256) void ng_stage_kernel(const PairDesc *pairs) {
  const PairDesc &p = pairs[blockIdx.x];
  const int n = p.Rp * p.Rp;
  for (int i = threadIdx.x; i < n; i += 256) {
    p.hK[i] = p.Kd[i];
    p.hL[i] = p.Ld[i];
  }
  if (threadIdx.x == 0) *p.h_tr0 = p.scal[0];
}

// ------------------------------------------------------------------------------------------------ per-component stages
struct CompDesc {
  float *T;
  const float *bsum, *sa, *sb;
  float *W_acc, *bias_acc;
  int Do, ldT, ldw, Dx;
  int blk0;  // first block of this component in the commit launch
};
__device__ __forceinline__ int find_comp(const CompDesc *c, int n, int blk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (c[mid].blk0 <= blk) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}
// T[o][ldw] = bsum[o], zeros in the row padding (one block per component)
__global__ __launch_bounds__(256) void ng_set_columns_kernel(const CompDesc *comps) {
  const CompDesc &c = comps[blockIdx.x];
  if (!c.bsum && c.ldT == c.ldw) return;
  for (int o = threadIdx.x; o < c.Do; o += 256) {
    float *row = c.T + (size_t)o * c.ldT;
    int col = c.ldw;
    if (c.bsum) row[col++] = c.bsum[o];
    for (; col < c.ldT; col++) row[col] = 0.f;
  }
}
// W_acc[o][c] += a b T[o][c] (c < ldw), bias_acc[o] += a b T[o][ldw]: "local_lrate = scale * learning_rate_"
// (nnet-tdnn-component.cc:604-624); a, b: the two preconditioners' scales, on the device.  1024 elements per block.
__global__ __launch_bounds__(256) void ng_commit_group_kernel(const CompDesc *comps, int ncomps) {
  const int ci = find_comp(comps, ncomps, blockIdx.x);
  const CompDesc &c = comps[ci];
  const float sc = c.sa[0] * c.sb[0];
  const int C = c.Dx;
  const long long total = (long long)c.Do * C, e0 = (long long)(blockIdx.x - c.blk0) * 1024;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const long long e = e0 + j * 256 + threadIdx.x;
    if (e >= total) break;
    const int o = (int)(e / C), col = (int)(e % C);
    const float v = sc * c.T[(size_t)o * c.ldT + col];
    if (col < c.ldw) c.W_acc[(size_t)o * c.ldw + col] += v;
    else c.bias_acc[o] += v;
  }
}

// ------------------------------------------------------------------------------------------------ grouped finalize
struct FinDesc {  // one refreshed object
  float *J, *W, *W1, *WT, *wlast;
  const float *h_coeff;  // pinned
  int Rp, D, Dp;
  int blk0;
};
__device__ __forceinline__ int find_fin(const FinDesc *c, int n, int blk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (c[mid].blk0 <= blk) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}
// J[r][d] += coeff[r] W[r][d]   (B_t = J_t + (1 - eta) / (eta / N) (D_t + rho_t I) W_t)
__global__ __launch_bounds__(256) void ng_fin_adddiag_kernel(const FinDesc *f, int nf) {
  const FinDesc &p = f[find_fin(f, nf, blockIdx.x)];
  const long long total = (long long)p.Rp * p.Dp, e0 = (long long)(blockIdx.x - p.blk0) * 1024;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const long long e = e0 + j * 256 + threadIdx.x;
    if (e >= total) break;
    p.J[e] += p.h_coeff[e / p.Dp] * p.W[e];
  }
}
// W = W1, W^T, last column
__global__ __launch_bounds__(256) void ng_fin_derive_kernel(const FinDesc *f, int nf) {
  const FinDesc &p = f[find_fin(f, nf, blockIdx.x)];
  const long long total = (long long)p.Rp * p.Dp, e0 = (long long)(blockIdx.x - p.blk0) * 1024;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const long long e = e0 + j * 256 + threadIdx.x;
    if (e >= total) break;
    const int r = (int)(e / p.Dp), d = (int)(e % p.Dp);
    const float v = p.W1[e];
    p.W[e] = v;
    if (d < p.D) {
      p.WT[(size_t)d * p.Rp + r] = v;
      if (d == p.D - 1) p.wlast[r] = v;
    }
  }
}


}  // namespace

struct NgGroup {
  std::vector<NgGroupComp> comps;
  std::vector<tdnnf_ng *> objs;  // 2 per component: in, out
  char *dev = nullptr;           // one allocation: tables, partial tiles, Q / P
  PairDesc *pairs = nullptr;
  CompDesc *cdesc = nullptr;
  float *lpart = nullptr;
  int npairs = 0, nslabs = 0, commit_blocks = 0;
  size_t l_lds = 0;
  DevList stage[4], kstage;
  hipEvent_t ev_staged = nullptr;
};

int ng_group_create(const std::vector<NgGroupComp> &comps, NgGroup **out) {
  TDNNF_REQUIRE(out && !comps.empty(), "ng_group_create: no components");
  NgGroup *g = new NgGroup();
  g->comps = comps;
  std::vector<PairDesc> pairs;
  std::vector<CompDesc> cdesc;
  GemmList st[4], kst;
  size_t qp_floats = 0;
  std::vector<size_t> q_off, p_off;
  int slab_total = 0, blk = 0;
  for (const NgGroupComp &c : comps) {
    TDNNF_REQUIRE(c.in && c.out && c.in->D == c.Dx && c.in->Dp == c.ldT && c.out->D == c.Do && c.in->rank > 0 && c.out->rank > 0 && c.in->Rp <= 96 &&
                      c.out->Rp <= 96 && c.N > 0,
                  "ng_group_create: a preconditioner is not initialised for its component (or its rank is outside 1..96)");
    q_off.push_back(qp_floats);
    qp_floats += ((size_t)c.Do * c.in->Rp + 63) & ~(size_t)63;
    p_off.push_back(qp_floats);
    qp_floats += ((size_t)c.out->Rp * c.ldT + 63) & ~(size_t)63;
    for (int side = 0; side < 2; side++) {
      tdnnf_ng *ng = side == 0 ? c.in : c.out;
      g->objs.push_back(ng);
      PairDesc p;
      memset(&p, 0, sizeof(p));
      p.H = side == 0 ? c.H_in : c.H_out;
      p.part = side == 0 ? c.part_in : c.part_out;
      p.Ld = ng->Ld;
      p.WWT = ng->WWT;
      p.scal = ng->scal;
      p.scale_f = ng->scale_f;
      p.Kd = ng->Kd;
      TDNNF_HIP(hipHostGetDevicePointer((void **)&p.hK, ng->h_K, 0));
      TDNNF_HIP(hipHostGetDevicePointer((void **)&p.hL, ng->h_L, 0));
      TDNNF_HIP(hipHostGetDevicePointer((void **)&p.h_tr0, ng->h_tr0, 0));
      p.ones_term = (side == 0 && c.bsum) ? (double)c.N : 0.0;
      p.N = c.N;
      p.Rp = ng->Rp;
      p.nt = (ng->Rp + 31) / 32;
      p.npart = rows_gemm_sumsq_blocks(c.N);
      // slabs: at least 256 rows, at most 32 per pair (the finish kernel adds them serially)
      int rps = std::max(256, (c.N + 31) / 32);
      rps = (rps + 7) & ~7;
      p.rows_per_slab = rps;
      p.nslab = (c.N + rps - 1) / rps;
      p.slab0 = slab_total;
      slab_total += p.nslab;
      pairs.push_back(p);
      // refresh: K = J J^T
      kst.add(ng->J, ng->Dp, 1, ng->J, 1, ng->Dp, ng->Kd, ng->Rp, ng->Rp, ng->Rp, ng->Dp, 1.0f, 0);
    }
    CompDesc d;
    memset(&d, 0, sizeof(d));
    d.T = c.T;
    d.bsum = c.bsum;
    d.sa = c.in->scale_f;
    d.sb = c.out->scale_f;
    d.W_acc = c.W_acc;
    d.bias_acc = c.bias_acc;
    d.Do = c.Do;
    d.ldT = c.ldT;
    d.ldw = c.ldw;
    d.Dx = c.Dx;
    d.blk0 = blk;
    blk += (int)(((long long)c.Do * c.Dx + 1023) / 1024);
    cdesc.push_back(d);
  }
  g->npairs = (int)pairs.size();
  g->nslabs = slab_total;
  g->commit_blocks = blk;
  // Q and P live behind the tables; the projection stages (ng.h: T <- (I - Wy^T Wy) T (I - Wx^T Wx))
  // are built once their addresses are known
  size_t bytes = 0;
  auto room = [&](size_t b) { bytes = ((bytes + 255) & ~(size_t)255) + b; };
  room(sizeof(PairDesc) * pairs.size());
  room(sizeof(CompDesc) * cdesc.size());
  room(sizeof(float) * (size_t)slab_total * 6 * 1024);
  room(sizeof(float) * qp_floats);
  // (task lists are sized after a dry build with null Q / P: their counts do not depend on the addresses)
  auto build = [&](float *qp) {
    for (auto &l : st) l = GemmList();
    for (size_t i = 0; i < comps.size(); i++) {
      const NgGroupComp &c = comps[i];
      float *Q = qp + q_off[i], *P = qp + p_off[i];
      const int Rx = c.in->Rp, Ry = c.out->Rp;
      st[0].add(c.T, c.ldT, 1, c.in->WT, Rx, 1, Q, Rx, c.Do, Rx, c.Dx, 1.0f, 0);            // Q = T Wx^T   (WT: D x Rp)
      st[1].add(Q, Rx, 1, c.in->W, c.in->Dp, 1, c.T, c.ldT, c.Do, c.ldT, Rx, -1.0f, 1);     // T -= Q Wx
      st[2].add(c.out->W, c.out->Dp, 1, c.T, c.ldT, 1, P, c.ldT, Ry, c.ldT, c.Do, 1.0f, 0);  // P = Wy T
      st[3].add(c.out->WT, Ry, 1, P, c.ldT, 1, c.T, c.ldT, c.Do, c.ldT, Ry, -1.0f, 1);      // T -= Wy^T P
    }
  };
  build(nullptr);
  size_t max_slots = kst.slots;
  for (auto &l : st) max_slots = std::max(max_slots, l.slots);
  for (auto *l : {&st[0], &st[1], &st[2], &st[3], &kst}) {
    room(sizeof(GTask) * l->tasks.size());
    room(sizeof(RTask) * l->rtasks.size());
  }
  room(sizeof(float) * max_slots * GT * GT);
  bytes += 1024;
  if (hipMalloc((void **)&g->dev, bytes) != hipSuccess) {
    set_error("ng_group_create: cannot allocate %zu bytes", bytes);
    delete g;
    return TDNNF_EHIP;
  }
  char *cur = g->dev;
  g->pairs = carve<PairDesc>(cur, pairs.size());
  g->cdesc = carve<CompDesc>(cur, cdesc.size());
  g->lpart = carve<float>(cur, (size_t)slab_total * 6 * 1024);
  float *qp = carve<float>(cur, qp_floats);
  build(qp);
  GemmList *lists[5] = {&st[0], &st[1], &st[2], &st[3], &kst};
  DevList *dls[5] = {&g->stage[0], &g->stage[1], &g->stage[2], &g->stage[3], &g->kstage};
  for (int i = 0; i < 5; i++) {
    dls[i]->nt = (int)lists[i]->tasks.size();
    dls[i]->nr = (int)lists[i]->rtasks.size();
    dls[i]->tasks = carve<GTask>(cur, lists[i]->tasks.size());
    dls[i]->rtasks = carve<RTask>(cur, lists[i]->rtasks.size());
  }
  float *part = carve<float>(cur, max_slots * GT * GT);  // the stages run one after the other: one partial buffer
  for (int i = 0; i < 5; i++) {
    lists[i]->fixup(part);
    if (dls[i]->nt) TDNNF_HIP(hipMemcpy(dls[i]->tasks, lists[i]->tasks.data(), sizeof(GTask) * dls[i]->nt, hipMemcpyHostToDevice));
    if (dls[i]->nr) TDNNF_HIP(hipMemcpy(dls[i]->rtasks, lists[i]->rtasks.data(), sizeof(RTask) * dls[i]->nr, hipMemcpyHostToDevice));
  }
  TDNNF_HIP(hipMemcpy(g->pairs, pairs.data(), sizeof(PairDesc) * pairs.size(), hipMemcpyHostToDevice));
  TDNNF_HIP(hipMemcpy(g->cdesc, cdesc.data(), sizeof(CompDesc) * cdesc.size(), hipMemcpyHostToDevice));
  TDNNF_HIP(hipEventCreateWithFlags(&g->ev_staged, hipEventDisableTiming | hipEventBlockingSync));
  g->l_lds = sizeof(float) * 2 * 6 * 1024;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)ng_l_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->l_lds);
    attr_done = true;
  }
  *out = g;
  return TDNNF_OK;
}

void ng_group_destroy(NgGroup *g) {
  if (!g) return;
  for (tdnnf_ng *ng : g->objs)
    if (ng->pending && ng->ev_wait == g->ev_staged) ng_pool_wait(ng);  // a pool thread may still be waiting on this group's event
  if (g->ev_staged) (void)hipEventDestroy(g->ev_staged);
  (void)hipFree(g->dev);
  delete g;
}

int ng_group_run(NgGroup *g, hipStream_t s) {
  TDNNF_REQUIRE(g, "ng_group_run: null group");
  const int nc = (int)g->comps.size();
  // every object of the group is in the same place of its refresh schedule (they are used once per minibatch, from the same start)
  const bool upd = g->objs[0]->cur_upd;
  for (tdnnf_ng *ng : g->objs)
    TDNNF_REQUIRE(ng->cur_upd == upd && ng->cur_N > 0 && !ng->pending, "ng_group_run: the group's preconditioners are out of step");
  hipLaunchKernelGGL(ng_set_columns_kernel, dim3(nc), dim3(256), 0, s, g->cdesc);
  hipLaunchKernelGGL(ng_l_partial_kernel, dim3(g->nslabs), dim3(256), g->l_lds, s, g->pairs, g->npairs, g->lpart);
  hipLaunchKernelGGL(ng_l_finish_kernel, dim3(g->npairs), dim3(1024), 0, s, g->pairs, g->lpart);
  if (upd) {
    int rc = launch_list(g->kstage, s);
    if (rc) return rc;
    hipLaunchKernelGGL(ng_stage_kernel, dim3(g->npairs), dim3(256), 0, s, g->pairs);
    TDNNF_HIP(hipEventRecord(g->ev_staged, s));
  }
  for (int i = 0; i < 4; i++) {
    int rc = launch_list(g->stage[i], s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ng_commit_group_kernel, dim3(g->commit_blocks), dim3(256), 0, s, g->cdesc, nc);
  TDNNF_LAUNCH_CHECK();
  for (tdnnf_ng *ng : g->objs) {
    if (upd) {
      ng->job_N = ng->cur_N;
      ng->job_done = 0;
      ng->pending = 1;
      ng->ev_wait = g->ev_staged;
      ng_pool_push(ng);
    }
    ng->cur_N = 0;
    ng->t += 1;
  }
  return TDNNF_OK;
}

struct NgFin {
  std::vector<tdnnf_ng *> objs;
  char *dev = nullptr;
  FinDesc *fd = nullptr;
  int nf = 0, blocks = 0;
  DevList w1, wwt;
};

int ng_fin_create(const std::vector<tdnnf_ng *> &objs_in, NgFin **out) {
  TDNNF_REQUIRE(out, "ng_fin_create: null argument");
  NgFin *f = new NgFin();
  for (tdnnf_ng *ng : objs_in)
    if (ng && ng->rank > 0 && ng->D != 0) f->objs.push_back(ng);
  std::vector<FinDesc> fd;
  GemmList w1, wwt;
  int blk = 0;
  for (tdnnf_ng *ng : f->objs) {
    FinDesc d;
    d.J = ng->J; d.W = ng->W; d.W1 = ng->W1; d.WT = ng->WT; d.wlast = ng->wlast;
    TDNNF_HIP(hipHostGetDevicePointer((void **)&d.h_coeff, ng->h_coeff, 0));
    d.Rp = ng->Rp; d.D = ng->D; d.Dp = ng->Dp;
    d.blk0 = blk;
    blk += (int)(((long long)ng->Rp * ng->Dp + 1023) / 1024);
    fd.push_back(d);
    float *hAt = nullptr;
    TDNNF_HIP(hipHostGetDevicePointer((void **)&hAt, ng->h_At, 0));
    w1.add(hAt, ng->Rp, 1, ng->J, ng->Dp, 1, ng->W1, ng->Dp, ng->Rp, ng->Dp, ng->Rp, 1.0f, 0);      // W1 = A_t (J + diag(c) W)
    wwt.add(ng->W, ng->Dp, 1, ng->W, 1, ng->Dp, ng->WWT, ng->Rp, ng->Rp, ng->Rp, ng->Dp, 1.0f, 0);  // W W^T
  }
  f->nf = (int)fd.size();
  f->blocks = blk;
  if (f->nf == 0) {
    *out = f;
    return TDNNF_OK;
  }
  size_t bytes = 0;
  auto room = [&](size_t b) { bytes = ((bytes + 255) & ~(size_t)255) + b; };
  room(sizeof(FinDesc) * fd.size());
  room(sizeof(GTask) * w1.tasks.size());
  room(sizeof(RTask) * w1.rtasks.size());
  room(sizeof(GTask) * wwt.tasks.size());
  room(sizeof(RTask) * wwt.rtasks.size());
  room(sizeof(float) * std::max(w1.slots, wwt.slots) * GT * GT);
  bytes += 1024;
  if (hipMalloc((void **)&f->dev, bytes) != hipSuccess) {
    set_error("ng_fin_create: cannot allocate %zu bytes", bytes);
    delete f;
    return TDNNF_EHIP;
  }
  char *cur = f->dev;
  f->fd = carve<FinDesc>(cur, fd.size());
  f->w1.nt = (int)w1.tasks.size(); f->w1.nr = (int)w1.rtasks.size();
  f->wwt.nt = (int)wwt.tasks.size(); f->wwt.nr = (int)wwt.rtasks.size();
  f->w1.tasks = carve<GTask>(cur, w1.tasks.size());
  f->w1.rtasks = carve<RTask>(cur, w1.rtasks.size());
  f->wwt.tasks = carve<GTask>(cur, wwt.tasks.size());
  f->wwt.rtasks = carve<RTask>(cur, wwt.rtasks.size());
  float *part = carve<float>(cur, std::max(w1.slots, wwt.slots) * GT * GT);
  w1.fixup(part);
  wwt.fixup(part);
  TDNNF_HIP(hipMemcpy(f->fd, fd.data(), sizeof(FinDesc) * fd.size(), hipMemcpyHostToDevice));
  if (f->w1.nt) TDNNF_HIP(hipMemcpy(f->w1.tasks, w1.tasks.data(), sizeof(GTask) * f->w1.nt, hipMemcpyHostToDevice));
  if (f->w1.nr) TDNNF_HIP(hipMemcpy(f->w1.rtasks, w1.rtasks.data(), sizeof(RTask) * f->w1.nr, hipMemcpyHostToDevice));
  if (f->wwt.nt) TDNNF_HIP(hipMemcpy(f->wwt.tasks, wwt.tasks.data(), sizeof(GTask) * f->wwt.nt, hipMemcpyHostToDevice));
  if (f->wwt.nr) TDNNF_HIP(hipMemcpy(f->wwt.rtasks, wwt.rtasks.data(), sizeof(RTask) * f->wwt.nr, hipMemcpyHostToDevice));
  *out = f;
  return TDNNF_OK;
}

void ng_fin_destroy(NgFin *f) {
  if (!f) return;
  (void)hipFree(f->dev);
  delete f;
}

int ng_fin_run(NgFin *f, hipStream_t s, bool wait, int *did) {
  *did = 0;
  TDNNF_REQUIRE(f, "ng_fin_run: null argument");
  size_t pending = 0;
  for (tdnnf_ng *ng : f->objs) pending += ng->pending ? 1 : 0;
  if (pending == 0) return TDNNF_OK;
  if (!wait)
    for (tdnnf_ng *ng : f->objs)
      if (ng->pending && !ng_pool_done(ng)) return TDNNF_OK;
  bool reorth = false;
  for (tdnnf_ng *ng : f->objs)
    if (ng->pending) {
      ng_pool_wait(ng);
      reorth = reorth || ng->must_reorth;
    }
  *did = 1;
  if (reorth || pending != f->objs.size()) {  // ReorthogonalizeRt1 (rare, synchronous) / objects out of step: one by one
    for (tdnnf_ng *ng : f->objs) {
      int rc = ng_finalize_one(ng, s);
      if (rc) return rc;
    }
    return TDNNF_OK;
  }
  hipLaunchKernelGGL(ng_fin_adddiag_kernel, dim3(f->blocks), dim3(256), 0, s, f->fd, f->nf);
  int rc = launch_list(f->w1, s);
  if (rc) return rc;
  hipLaunchKernelGGL(ng_fin_derive_kernel, dim3(f->blocks), dim3(256), 0, s, f->fd, f->nf);
  rc = launch_list(f->wwt, s);
  if (rc) return rc;
  TDNNF_LAUNCH_CHECK();
  for (tdnnf_ng *ng : f->objs) {
    ng->pending = 0;
    ng->d = ng->d_next;
    ng->rho = ng->rho_next;
  }
  return TDNNF_OK;
}

}  // namespace tdnnf
