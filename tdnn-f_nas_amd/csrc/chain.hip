// chain.hip -- LF-MMI objective and derivative on gfx950: chain::ComputeChainObjfAndDeriv,
// DenominatorComputation and NumeratorComputation (UPSTREAM Kaldi, not shipped in the reference;
// reached via /root/reference/steps/nnet3/chain/train.py:515; options pinned by
// local/chain_NAS/run_TDNN_DARTSV3_fbk_stride_pretrain.sh:185-195).  SURVEY.md 8(a) row A7.
//
// MI355X design (vs. upstream's 2*T kernel launches with atomics):
//  * ONE persistent workgroup per sequence walks all T frames inside a single launch; the HMM state
//    vectors (alpha/beta) and the exponentiated output row live in LDS, the per-frame renormaliser
//    is a workgroup reduction.  Sequences are independent, so there is no inter-workgroup traffic.
//  * the denominator graph is stored three times in sliced-ELL (SELL-64) form -- by destination
//    (forward), by source (beta) and by pdf (occupancies) -- so every arc gather is a coalesced
//    stream and every sum is a fixed-order per-thread loop: no float atomics, results are bitwise
//    reproducible.  An arc is 8 bytes: (state | pdf << 16, prob).
//  * the numerator (tiny time-synchronous graphs) runs one wave per sequence in the log domain.
#include <math.h>
#include <string.h>

#include <algorithm>

#include <vector>

#include "common.h"
#include "gemm_f32.h"

struct tdnnf_den_graph {
  int H, A, P;
  // SELL-64 over rows sorted by descending degree (so a slice's rows have near-equal degree and padding is
  // negligible): slot s = 64*k + lane holds original row row[s] (0xffffffff = padding slot); its j-th arc is
  // arc[base[k] + j*64 + lane] = (packed key, prob bits), j < (base[k+1]-base[k])/64.
  struct Sell {
    int nrows, nslices;
    int *base;       // nslices + 1 (device)
    unsigned *row;   // nslices * 64
    uint2 *arc;      // key: (other-state-or-src | pdf << 16), or (src | dst << 16) for the by-pdf table
    uint4 *arc4;     // the same arcs as (key, prob, prob * init[key & 0xffff] or 0, 0): what the wide form loads
    long long entries;
    int mw_max_arcs[9];  // [G]: most arc entries any of G workgroups owns when slice k belongs to workgroup k % G (den_*_mw_kernel); G = 2, 4, 8
  } by_dst, by_src, by_pdf;
  float *init;  // H
  float init_sum;
};

struct tdnnf_supervision {
  int B, T;
  int num_states, num_arcs;
  float weight;
  int *seq_state_begin;  // B+1
  int *state_time;
  float *final_logprob;
  // arcs grouped by destination state (forward) and by source state (backward), CSR over global state ids
  int *in_begin, *in_src, *in_pdf;
  float *in_lp;
  int *out_begin, *out_dst, *out_pdf;
  float *out_lp;
  // states of a sequence are sorted by time; frame_state_begin[s*(T+2) + t] = first state with time t
  int *frame_state_begin;
  int max_states_per_seq;
};

namespace tdnnf {
namespace {

constexpr int kDenThreads = 1024;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum(float v, float *red, int nwaves) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();  // protect red from the previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nwaves; w++) s += red[w];
  return s;
}

// ApplyExpLimited(-30, 30): comparisons (not fmin/fmax) so that a NaN stays a NaN and trips the
// objf-not-finite failure path, as in the reference stack.
__device__ __forceinline__ float exp_limited(float v) {
  v = v < -30.f ? -30.f : (v > 30.f ? 30.f : v);
  return expf(v);
}

struct DenDev {
  int H, P;
  tdnnf_den_graph::Sell by_dst, by_src, by_pdf;
  const float *init;
  float init_sum;
};

// The frame's output row (P floats, from HBM) one frame ahead in registers: a frame used to begin with the dependent load of its own row --
// 2-3 us of HBM latency in front of a 17-19 us frame, 500 times per recursion.  (More than kDenRowRegs * kDenThreads pdfs: the kernels load the rest
// the old way.)
constexpr int kDenRowRegs = 8;
struct RowAhead {
  float v[kDenRowRegs];
  __device__ __forceinline__ void load(const float *yr, int P, int tid) {
#pragma unroll
    for (int i = 0; i < kDenRowRegs; i++) {
      const int p = tid + i * kDenThreads;
      v[i] = p < P ? yr[p] : 0.f;
    }
  }
};

// The sum over one row of a sliced-ELL table (ap: the lane's first arc, w: the slice's width, uniform over the wave), arcs in order.  A row is a chain
// of dependent-latency loads from L2: batches of eight, then four, then ONE batch for the last one to three arcs (indices clamped, the surplus terms
// replaced by exact zeros) -- "#pragma unroll 4" left up to three single loads behind every row, "#pragma unroll 8" up to seven (measured: slower).
template <class Term>
__device__ __forceinline__ float sell_row_sum(const uint2 *ap, int w, Term term) {
  float acc = 0.f;
  int j = 0;
  for (; j + 8 <= w; j += 8) {
    uint2 a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = ap[(j + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += term(a[u]);
  }
  if (j + 4 <= w) {
    uint2 a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = ap[(j + u) * 64];
#pragma unroll
    for (int u = 0; u < 4; u++) acc += term(a[u]);
    j += 4;
  }
  if (j < w) {
    uint2 a[3];
#pragma unroll
    for (int u = 0; u < 3; u++) a[u] = ap[min(j + u, w - 1) * 64];
#pragma unroll
    for (int u = 0; u < 3; u++) acc += (j + u < w) ? term(a[u]) : 0.f;
  }
  return acc;
}

// Forward: alpha_dash(t, .) for t = 0..T stored to `alpha` [(T+1) x Hs] per sequence, alpha sums to
// `asum` [T+1], per-sequence log-prob to logprob[s].
// FAST (the host checks: state vectors in LDS, at most kDenFastSlots rows and kDenFastStates states per thread): everything that does not change from
// frame to frame -- a thread's slices, rows and initial probabilities -- stays in registers, and a frame's new vector is formed in a second LDS
// buffer: no load inside a frame depends on another load or store of the same frame except the arcs themselves (the plain loop re-reads the slice
// table, the row ids, init[] and -- after a global store -- its own alpha row, each a trip to L2 in front of the next barrier).
// res_cap > 0: the first res_cap arcs of the table (its widest slices) stay in LDS for all frames -- used when the launch holds every CU anyway.
constexpr int kDenFastSlots = 4, kDenFastStates = 4;  // (both tables of these kernels have one row per state)
template <bool LDS_STATE, bool FAST = false>
__global__ __launch_bounds__(kDenThreads) void den_forward_kernel(DenDev g, MatView y, int B, int T, float leaky,
                                                                  float *alpha_all, float *asum_all, int Hs,
                                                                  double *logprob, float *gstate, const unsigned *only_if, int res_cap) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[kDenThreads / 64];
  if (only_if && *only_if == 0) return;  // fallback launch behind the multi-workgroup recursion: runs only when that one gave up
  const int s = blockIdx.x, tid = threadIdx.x;
  const int H = g.H, P = g.P;
  float *x = smem;  // P
  float *prev = LDS_STATE ? smem + ((P + 3) & ~3) : gstate + (size_t)s * 2 * Hs;
  float *nxt = smem + ((P + 3) & ~3) + Hs;  // FAST: the frame's new vector
  uint2 *lres = reinterpret_cast<uint2 *>(smem + ((P + 3) & ~3) + (LDS_STATE ? Hs : 0) + (FAST ? Hs : 0));
  for (int i = tid; i < res_cap; i += kDenThreads) lres[i] = g.by_dst.arc[i];
  int sb0[kDenFastSlots], sw[kDenFastSlots];
  unsigned srow[kDenFastSlots];
  float hinit[kDenFastStates];
  if constexpr (FAST) {
#pragma unroll
    for (int k = 0; k < kDenFastSlots; k++) {
      const int slot = tid + k * kDenThreads;
      sb0[k] = 0;
      sw[k] = 0;
      srow[k] = 0xffffffffu;
      if (slot < g.by_dst.nslices * 64) {
        sb0[k] = g.by_dst.base[slot >> 6];
        sw[k] = (g.by_dst.base[(slot >> 6) + 1] - sb0[k]) >> 6;
        srow[k] = g.by_dst.row[slot];
      }
    }
#pragma unroll
    for (int i = 0; i < kDenFastStates; i++) hinit[i] = tid + i * kDenThreads < H ? g.init[tid + i * kDenThreads] : 0.f;
  }
  float *alpha = alpha_all + (size_t)s * (T + 1) * Hs;
  float *asum = asum_all + (size_t)s * (T + 1);

  // AlphaFirstFrame + AlphaDash(0)
  for (int h = tid; h < H; h += kDenThreads) {
    const float a = g.init[h] + leaky * g.init_sum * g.init[h];
    prev[h] = a;
    alpha[h] = a;
  }
  if (tid == 0) asum[0] = g.init_sum;
  float prev_sum = g.init_sum;
  double logcorr = 0.0;
  RowAhead ra;
  ra.load(y.data + (size_t)s * y.stride, P, tid);
  __syncthreads();
  for (int t = 1; t <= T; t++) {
#pragma unroll
    for (int i = 0; i < kDenRowRegs; i++)
      if (tid + i * kDenThreads < P) x[tid + i * kDenThreads] = exp_limited(ra.v[i]);
    for (int p = tid + kDenRowRegs * kDenThreads; p < P; p += kDenThreads) x[p] = exp_limited(y.data[(size_t)((t - 1) * B + s) * y.stride + p]);  // (more than 8 192 pdfs: the rest as before)
    if (t < T) ra.load(y.data + (size_t)(t * B + s) * y.stride, P, tid);  // the next frame's row: lands while this frame's arcs are walked
    __syncthreads();
    const float inv = 1.0f / prev_sum;
    logcorr += (double)logf(prev_sum);
    float *cur = alpha + (size_t)t * Hs;
    float local = 0.f;
    auto term = [&](const uint2 a) { return prev[a.x & 0xffffu] * __uint_as_float(a.y) * x[a.x >> 16]; };
    if constexpr (FAST) {
      const int ln = tid & 63;
#pragma unroll
      for (int k = 0; k < kDenFastSlots; k++) {
        if (srow[k] == 0xffffffffu && sw[k] == 0) continue;
        float acc = sb0[k] + sw[k] * 64 <= res_cap ? sell_row_sum(lres + sb0[k] + ln, sw[k], term) : sell_row_sum(g.by_dst.arc + sb0[k] + ln, sw[k], term);
        if (srow[k] != 0xffffffffu) {
          acc *= inv;
          nxt[srow[k]] = acc;  // alpha(t,h) before the leaky term
          local += acc;
        }
      }
      const float sum = block_sum(local, red, kDenThreads / 64);  // (its barriers: every row of nxt is written)
      if (tid == 0) asum[t] = sum;
#pragma unroll
      for (int i = 0; i < kDenFastStates; i++) {  // AlphaDash(t)
        const int h = tid + i * kDenThreads;
        if (h < H) {
          const float a = nxt[h] + leaky * sum * hinit[i];
          nxt[h] = a;
          cur[h] = a;
        }
      }
      float *other = prev;
      prev = nxt;
      nxt = other;
      prev_sum = sum;
      __syncthreads();
      continue;
    }
    for (int slot = tid; slot < g.by_dst.nslices * 64; slot += kDenThreads) {
      const int sl = slot >> 6, ln = slot & 63;
      const int b0 = g.by_dst.base[sl], w = (g.by_dst.base[sl + 1] - b0) >> 6;
      float acc = b0 + w * 64 <= res_cap ? sell_row_sum(lres + b0 + ln, w, term) : sell_row_sum(g.by_dst.arc + b0 + ln, w, term);
      const unsigned h = g.by_dst.row[slot];
      if (h != 0xffffffffu) {
        acc *= inv;
        cur[h] = acc;  // alpha(t,h) before the leaky term
        local += acc;
      }
    }
    const float sum = block_sum(local, red, kDenThreads / 64);
    if (tid == 0) asum[t] = sum;
    for (int h = tid; h < H; h += kDenThreads) {  // AlphaDash(t)
      const float a = cur[h] + leaky * sum * g.init[h];
      cur[h] = a;
      prev[h] = a;
    }
    prev_sum = sum;
    __syncthreads();
  }
  float local = 0.f;
  for (int h = tid; h < H; h += kDenThreads) local += prev[h];
  const float tot = block_sum(local, red, kDenThreads / 64);
  if (tid == 0) {
    logprob[s] = (double)logf(tot) + logcorr;
    asum[T] = tot;  // reuse: total of alpha_dash(T) (asum[T] itself is not needed by the backward pass)
  }
}

// Backward: deriv[t*B+s][p] = deriv_weight * gamma_den(t, p)   (overwrites the whole row)
template <bool LDS_STATE>
__global__ __launch_bounds__(kDenThreads) void den_backward_kernel(DenDev g, MatView y, int B, int T, float leaky,
                                                                   const float *alpha_all, const float *asum_all,
                                                                   int Hs, float deriv_weight, MatView deriv,
                                                                   float *gstate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[kDenThreads / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int H = g.H, P = g.P, P4 = (P + 3) & ~3, H4 = (H + 3) & ~3;
  float *x = smem;
  float *bnext = LDS_STATE ? smem + P4 : gstate + (size_t)s * 3 * Hs;
  float *bcur = LDS_STATE ? bnext + H4 : bnext + Hs;
  float *ad = LDS_STATE ? bcur + H4 : bcur + Hs;  // alpha_dash(t) * inv_asum(t)
  const float *alpha = alpha_all + (size_t)s * (T + 1) * Hs;
  const float *asum = asum_all + (size_t)s * (T + 1);

  {  // BetaDashLastFrame + Beta(T)
    const float bd = 1.0f / asum[T];
    const float lsum = g.init_sum * bd;
    for (int h = tid; h < H; h += kDenThreads) bnext[h] = bd + leaky * lsum;
  }
  RowAhead ra;
  ra.load(y.data + (size_t)((T - 1) * B + s) * y.stride, P, tid);
  __syncthreads();
  for (int t = T - 1; t >= 0; t--) {
    const float inv = 1.0f / asum[t];
#pragma unroll
    for (int i = 0; i < kDenRowRegs; i++)
      if (tid + i * kDenThreads < P) x[tid + i * kDenThreads] = exp_limited(ra.v[i]);
    for (int p = tid + kDenRowRegs * kDenThreads; p < P; p += kDenThreads) x[p] = exp_limited(y.data[(size_t)(t * B + s) * y.stride + p]);
    if (t > 0) ra.load(y.data + (size_t)((t - 1) * B + s) * y.stride, P, tid);
    for (int h = tid; h < H; h += kDenThreads) ad[h] = alpha[(size_t)t * Hs + h] * inv;
    __syncthreads();
    // beta_dash(t, i) = sum over out-arcs
    float local = 0.f;
    for (int slot = tid; slot < g.by_src.nslices * 64; slot += kDenThreads) {
      const int sl = slot >> 6, ln = slot & 63;
      const int b0 = g.by_src.base[sl], w = (g.by_src.base[sl + 1] - b0) >> 6;
      const uint2 *ap = g.by_src.arc + b0 + ln;
      const float acc0 = sell_row_sum(ap, w, [&](const uint2 a) { return __uint_as_float(a.y) * x[a.x >> 16] * bnext[a.x & 0xffffu]; });
      float acc = acc0;
      const unsigned h = g.by_src.row[slot];
      if (h != 0xffffffffu) {
        acc *= inv;
        bcur[h] = acc;
        local += g.init[h] * acc;
      }
    }
    // occupancies by pdf: gamma(t,p) = x[p] * sum_arcs prob * alpha_dash(t,src)/A(t) * beta(t+1,dst)
    float *dr = deriv.data + (size_t)(t * B + s) * deriv.stride;
    for (int slot = tid; slot < g.by_pdf.nslices * 64; slot += kDenThreads) {
      const int sl = slot >> 6, ln = slot & 63;
      const int b0 = g.by_pdf.base[sl], w = (g.by_pdf.base[sl + 1] - b0) >> 6;
      const uint2 *ap = g.by_pdf.arc + b0 + ln;
      const float acc0 = sell_row_sum(ap, w, [&](const uint2 a) { return __uint_as_float(a.y) * ad[a.x & 0xffffu] * bnext[a.x >> 16]; });
      float acc = acc0;
      const unsigned p = g.by_pdf.row[slot];
      if (p != 0xffffffffu) dr[p] = deriv_weight * acc * x[p];
    }
    const float ls = block_sum(local, red, kDenThreads / 64);  // also orders the reads of bnext above
    for (int h = tid; h < H; h += kDenThreads) bcur[h] += leaky * ls;  // Beta(t)
    float *tmp = bnext;
    bnext = bcur;
    bcur = tmp;
    __syncthreads();
  }
}


// The backward half in two kernels that do not wait for the forward one (persistent form, state vectors in LDS).  As in the wide
// form below, the backward recursion is linear and homogeneous in its last frame, so it can run SELF-NORMALISED beside the
// forward recursion:  b(T, h) = 1,  S(t) = sum_h init_h b(t, h),
//   b(t, h) = sum_arcs p x(t, pdf) (b(t+1, dst) / S(t+1) + leaky),
// every b(t) and S(t) kept (another (T+1) x Hs floats per sequence); the occupancies then need no dependence between frames:
//   gamma(t, p) = x(t, p) sum_arcs p alpha_dash(t, src) (b(t+1, dst) / S(t+1) + leaky) / Zd(t),   Zd(t) = sum_h alpha_dash(t, h) b(t, h)
// -- one workgroup per (frame, sequence).  128 x 500 frames, 4 000 states: 17.8 -> 11.6 ms stand-alone (10 000 states: 34.5 -> 27.9),
// which is what the component-level entry point gets.  The trainer, which runs the denominator beside the xent head on a stream
// of its own, keeps the one-kernel backward pass: there the further stream bought nothing at 1500 x 128 (133.8 -> 134.6 ms) and
// cost 7 ms at 150 x 64 (18.5 -> 25.7: with the weight-gradient stream that is a fifth stream in flight, and beyond four they
// share hardware queues -- the same cliff as one side stream per natural-gradient buffer set).
template <bool FAST = false>
__global__ __launch_bounds__(kDenThreads) void den_beta_kernel(DenDev g, MatView y, int B, int T, float leaky, float *b_all, float *S_all, int Hs,
                                                               const unsigned *only_if, int res_cap) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[kDenThreads / 64];
  if (only_if && *only_if == 0) return;
  const int s = blockIdx.x, tid = threadIdx.x;
  const int H = g.H, P = g.P, P4 = (P + 3) & ~3;
  float *x = smem;       // P: exp of the frame's output row
  float *bn = smem + P4;  // H: b(t+1, .) / S(t+1) + leaky
  float *nb = bn + ((H + 3) & ~3);  // FAST: the frame's raw sums (den_forward_kernel)
  uint2 *lres = reinterpret_cast<uint2 *>(bn + ((H + 3) & ~3) * (FAST ? 2 : 1));  // (den_forward_kernel: the table's first res_cap arcs)
  for (int i = tid; i < res_cap; i += kDenThreads) lres[i] = g.by_src.arc[i];
  int sb0[kDenFastSlots], sw[kDenFastSlots];
  unsigned srow[kDenFastSlots];
  float sinit[kDenFastSlots];
  if constexpr (FAST) {
#pragma unroll
    for (int k = 0; k < kDenFastSlots; k++) {
      const int slot = tid + k * kDenThreads;
      sb0[k] = 0;
      sw[k] = 0;
      srow[k] = 0xffffffffu;
      sinit[k] = 0.f;
      if (slot < g.by_src.nslices * 64) {
        sb0[k] = g.by_src.base[slot >> 6];
        sw[k] = (g.by_src.base[(slot >> 6) + 1] - sb0[k]) >> 6;
        srow[k] = g.by_src.row[slot];
        if (srow[k] != 0xffffffffu) sinit[k] = g.init[srow[k]];
      }
    }
  }
  float *brow = b_all + (size_t)s * (T + 1) * Hs;
  float *S = S_all + (size_t)s * (T + 1);
  for (int h = tid; h < H; h += kDenThreads) {
    brow[(size_t)T * Hs + h] = 1.0f;
    bn[h] = 1.0f / g.init_sum + leaky;
  }
  if (tid == 0) S[T] = g.init_sum;
  RowAhead ra;
  ra.load(y.data + (size_t)((T - 1) * B + s) * y.stride, P, tid);
  __syncthreads();
  for (int t = T - 1; t >= 0; t--) {
#pragma unroll
    for (int i = 0; i < kDenRowRegs; i++)
      if (tid + i * kDenThreads < P) x[tid + i * kDenThreads] = exp_limited(ra.v[i]);
    for (int p = tid + kDenRowRegs * kDenThreads; p < P; p += kDenThreads) x[p] = exp_limited(y.data[(size_t)(t * B + s) * y.stride + p]);
    if (t > 0) ra.load(y.data + (size_t)((t - 1) * B + s) * y.stride, P, tid);
    __syncthreads();
    float *bcur = brow + (size_t)t * Hs;
    float local = 0.f;
    auto term = [&](const uint2 a) { return __uint_as_float(a.y) * x[a.x >> 16] * bn[a.x & 0xffffu]; };
    if constexpr (FAST) {
      const int ln = tid & 63;
#pragma unroll
      for (int k = 0; k < kDenFastSlots; k++) {
        if (srow[k] == 0xffffffffu && sw[k] == 0) continue;
        const float acc = sb0[k] + sw[k] * 64 <= res_cap ? sell_row_sum(lres + sb0[k] + ln, sw[k], term) : sell_row_sum(g.by_src.arc + sb0[k] + ln, sw[k], term);
        if (srow[k] != 0xffffffffu) {
          nb[srow[k]] = acc;
          local += sinit[k] * acc;
        }
      }
      const float St = block_sum(local, red, kDenThreads / 64);  // (its barriers: bn is no longer read, every row of nb is written)
      if (tid == 0) S[t] = St;
      const float inv = 1.0f / St;
#pragma unroll
      for (int i = 0; i < kDenFastStates; i++) {
        const int h = tid + i * kDenThreads;
        if (h < H) {
          const float v = nb[h];
          bcur[h] = v;
          bn[h] = v * inv + leaky;
        }
      }
      __syncthreads();
      continue;
    }
    for (int slot = tid; slot < g.by_src.nslices * 64; slot += kDenThreads) {
      const int sl = slot >> 6, ln = slot & 63;
      const int b0 = g.by_src.base[sl], w = (g.by_src.base[sl + 1] - b0) >> 6;
      float acc = b0 + w * 64 <= res_cap ? sell_row_sum(lres + b0 + ln, w, term) : sell_row_sum(g.by_src.arc + b0 + ln, w, term);
      const unsigned h = g.by_src.row[slot];
      if (h != 0xffffffffu) {
        bcur[h] = acc;
        local += g.init[h] * acc;
      }
    }
    const float St = block_sum(local, red, kDenThreads / 64);  // (its barriers also order the reads of bn above)
    if (tid == 0) S[t] = St;
    const float inv = 1.0f / St;
    for (int h = tid; h < H; h += kDenThreads) bn[h] = bcur[h] * inv + leaky;  // (other threads' rows: visible after block_sum's barriers)
    __syncthreads();
  }
}

// ---- The two recursions with SEVERAL workgroups per sequence (few sequences: the 8-GPU shard of a minibatch, the recipes' own egs).
// One workgroup per sequence walks T dependent frames at ~15 us each -- 384 KB of arcs from L2 per frame on one CU, and the fixed work of
// a frame -- while the other CUs have nothing to do.  Here G workgroups share a sequence: slice k of the SELL table belongs to workgroup
// k % G (every workgroup gets the same mix of row degrees), its arcs stay in LDS for the whole kernel, and per frame a workgroup
//   publishes the new values of its rows (exchange buffer in slot order, double-buffered by frame parity: agent-scope stores, every
//   wave's vmcnt(0), workgroup barrier, one agent-scope add to the sequence's counter),
//   waits until the counter shows all G slices of the frame (one lane polls; bounded: on a time-out it raises the abort word, which
//   every poll also reads, and the host reports an error instead of hanging), and
//   reads the whole vector back (agent-scope loads) into LDS, where the frame's normaliser is summed.
// MI355X_MICROARCH.md, inter-workgroup visibility: stores and loads of the handed-off bytes all sc1, the signal behind every storing
// wave's wait and a barrier, the loads behind the poll and a barrier.  A buffer of parity q is rewritten for frame t + 2 only after the
// counter has reached G (t + 1), i.e. after every workgroup has published frame t + 1, which it does after reading frame t.
// The workgroups of a sequence sit on one XCD when the sequence count is a multiple of 8 (blocks b and b + 8 share an XCD).
struct MwCtl {
  unsigned long long *ctr;  // [2 * B]: forward counters, then backward counters
  unsigned *abort_flag;
  float *xf, *xb;           // [B][2][NSp] exchange buffers of the forward / backward recursion
  int NSp;
};
constexpr unsigned kMwSpinLimit = 1u << 22;  // polls of ~0.5 us: seconds -- a launch that cannot make progress ends, it does not hang

__device__ __forceinline__ void mw_block_of(int b, int B, int G, int *s, int *gi) {
  if (B % 8 == 0) {
    const int x = b & 7, j = b >> 3;
    *s = (j / G) * 8 + x;
    *gi = j % G;
  } else {
    *s = b / G;
    *gi = b % G;
  }
}
// publish: all of this workgroup's stores are issued; wait, barrier, one add.
__device__ __forceinline__ void mw_publish(unsigned long long *ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consume: wait for `target` adds (one lane polls, the others wait at the barrier).  Returns false on abort.
__device__ __forceinline__ bool mw_wait(unsigned long long *ctr, unsigned *abort_flag, unsigned target, unsigned *lds_flag) {
  if (threadIdx.x == 0) {
    unsigned spins = 0, ab = 0;
    while (true) {
      const unsigned long long v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)v >= target || ab) break;
      if (++spins > kMwSpinLimit) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *lds_flag = ab;
  }
  __syncthreads();
  return *lds_flag == 0;
}

// dir 0: alpha recursion over by_dst (as den_forward_kernel<true>); dir 1: the self-normalised beta recursion over by_src (den_beta_kernel)
template <int DIR>
__global__ __launch_bounds__(kDenThreads) void den_mw_kernel(DenDev g, MwCtl ctl, int G, MatView y, int B, int T, float leaky, float *vec_all, float *sum_all,
                                                             int Hs, double *logprob) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[kDenThreads / 64];
  __shared__ unsigned flag;
  __shared__ int loff[64 + 1];  // LDS offsets (arc entries) of the slices this workgroup owns
  const tdnnf_den_graph::Sell &tab = DIR == 0 ? g.by_dst : g.by_src;
  int s, gi;
  mw_block_of(blockIdx.x, B, G, &s, &gi);
  const int tid = threadIdx.x;
  const int H = g.H, P = g.P, P4 = (P + 3) & ~3, H4 = (H + 3) & ~3, ns = tab.nslices, NSp = ctl.NSp;
  float *x = smem;            // P: exp of the frame's output row
  float *cur = smem + P4;     // H: the previous frame's vector, by state (forward: alpha_dash(t-1); backward: b(t+1)/S(t+1) + leaky)
  uint2 *arcs = reinterpret_cast<uint2 *>(cur + H4);
  unsigned long long *ctr = ctl.ctr + (DIR == 0 ? s : B + s);
  float *xch = (DIR == 0 ? ctl.xf : ctl.xb) + (size_t)s * 2 * NSp;
  float *vec = vec_all + (size_t)s * (T + 1) * Hs;
  float *sums = sum_all + (size_t)s * (T + 1);
  const int nown = ns > gi ? (ns - gi + G - 1) / G : 0;  // slices gi, gi + G, ...
  if (tid == 0) {
    int o = 0;
    for (int i = 0; i < nown; i++) {
      loff[i] = o;
      const int k = gi + i * G;
      o += tab.base[k + 1] - tab.base[k];
    }
    loff[nown] = o;
  }
  __syncthreads();
  for (int i = 0; i < nown; i++) {
    const int k = gi + i * G, b0 = tab.base[k], n = tab.base[k + 1] - b0;
    for (int e = tid; e < n; e += kDenThreads) arcs[loff[i] + e] = tab.arc[b0 + e];
  }
  // the share of the final arrays this workgroup writes: states [h0, h1)
  const int h0 = (int)((long long)H * gi / G), h1 = (int)((long long)H * (gi + 1) / G);
  float prev_sum = g.init_sum;
  double logcorr = 0.0;
  if (DIR == 0) {  // AlphaFirstFrame + AlphaDash(0)
    for (int h = tid; h < H; h += kDenThreads) {
      const float a = g.init[h] + leaky * g.init_sum * g.init[h];
      cur[h] = a;
      if (h >= h0 && h < h1) vec[h] = a;
    }
    if (gi == 0 && tid == 0) sums[0] = g.init_sum;
  } else {  // b(T, .) = 1, S(T) = sum init
    for (int h = tid; h < H; h += kDenThreads) {
      cur[h] = 1.0f / g.init_sum + leaky;
      if (h >= h0 && h < h1) vec[(size_t)T * Hs + h] = 1.0f;
    }
    if (gi == 0 && tid == 0) sums[T] = g.init_sum;
  }
  __syncthreads();
  // the output row of a frame is requested (into registers) before the exchange of the frame before it and turned into x behind it:
  // its trip to memory runs under the wait for the other workgroups
  constexpr int kRowRegs = 8, kRowThreads = kDenThreads - 64;  // waves 1-15: P <= 8 * 960 (checked on the host)
  float yv[kRowRegs];
  const int rt = tid - 64;
  auto request_row = [&](int step) {
    const int yrow = DIR == 0 ? step - 1 : T - step;
    const float *yr = y.data + (size_t)(yrow * B + s) * y.stride;
    if (rt >= 0) {
#pragma unroll
      for (int i = 0; i < kRowRegs; i++) yv[i] = rt + i * kRowThreads < P ? yr[rt + i * kRowThreads] : 0.f;
    }
  };
  auto row_to_x = [&]() {
    if (rt >= 0) {
#pragma unroll
      for (int i = 0; i < kRowRegs; i++)
        if (rt + i * kRowThreads < P) x[rt + i * kRowThreads] = exp_limited(yv[i]);
    }
  };
  float *stage = reinterpret_cast<float *>(arcs + loff[nown]);  // nown * 64: this workgroup's new values, for 16-byte stores
  request_row(1);
  row_to_x();
  __syncthreads();
  for (int step = 1; step <= T; step++) {
    const int t = DIR == 0 ? step : T - step;  // the frame whose vector is formed
    const float inv = 1.0f / prev_sum;
    if (DIR == 0) logcorr += (double)logf(prev_sum);
    float *xw = xch + (size_t)(step & 1) * NSp;
    for (int q = tid; q < nown * 64; q += kDenThreads) {
      const int i = q >> 6, ln = q & 63;
      const int w = (loff[i + 1] - loff[i]) >> 6;
      const uint2 *ap = arcs + loff[i] + ln;
      const float acc0 = sell_row_sum(ap, w, [&](const uint2 a) { return cur[a.x & 0xffffu] * __uint_as_float(a.y) * x[a.x >> 16]; });
      float acc = acc0;
      if (DIR == 0) acc *= inv;
      stage[q] = acc;  // (a padding slot: 0)
    }
    __syncthreads();
    for (int q = tid; q < nown * 16; q += kDenThreads) {  // 16 bytes per store, agent scope
      const int i = q >> 4, c = q & 15, k = gi + i * G;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + i * 64 + c * 4);
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(xw + k * 64 + c * 4), "v"(v) : "memory");
    }
    mw_publish(ctr);
    if (step < T) request_row(step + 1);  // (behind the signal; not by the polling wave, whose polls would queue behind these loads)
    if (!mw_wait(ctr, ctl.abort_flag, (unsigned)(G * step), &flag)) return;
    // the whole vector of this frame, slot order -> by state; its normaliser
    float local = 0.f;
    for (int q4 = tid; q4 < ns * 16; q4 += kDenThreads) {
      f32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(xw + q4 * 4) : "memory");
      const uint4 hh = *reinterpret_cast<const uint4 *>(tab.row + q4 * 4);
      const unsigned hs[4] = {hh.x, hh.y, hh.z, hh.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (hs[e] != 0xffffffffu) {
          cur[hs[e]] = v[e];
          local += DIR == 0 ? v[e] : g.init[hs[e]] * v[e];
        }
      }
    }
    const float sum = block_sum(local, red, kDenThreads / 64);  // (its barriers order the writes of cur above before the reads below)
    if (gi == 0 && tid == 0) sums[t] = sum;
    if (DIR == 0) {
      for (int h = tid; h < H; h += kDenThreads) {  // AlphaDash(t)
        const float a = cur[h] + leaky * sum * g.init[h];
        cur[h] = a;
        if (h >= h0 && h < h1) vec[(size_t)t * Hs + h] = a;
      }
    } else {
      const float is = 1.0f / sum;
      for (int h = tid; h < H; h += kDenThreads) {
        const float v = cur[h];
        if (h >= h0 && h < h1) vec[(size_t)t * Hs + h] = v;  // b(t, h), as den_beta_kernel keeps it
        cur[h] = v * is + leaky;
      }
    }
    if (step < T) row_to_x();
    prev_sum = sum;
    __syncthreads();
  }
  if (DIR == 0) {
    float local = 0.f;
    for (int h = tid; h < H; h += kDenThreads) local += cur[h];
    const float tot = block_sum(local, red, kDenThreads / 64);
    if (gi == 0 && tid == 0) {
      logprob[s] = (double)logf(tot) + logcorr;
      sums[T] = tot;  // (as den_forward_kernel: the total of alpha_dash(T))
    }
  }
}

// a multi-workgroup launch that gave up (mw_wait's time-out: its workgroups were not co-resident, e.g. under a CU mask or beside another
// process): the one-workgroup kernels launched behind it redo both recursions (they look at the same word), and the host learns of it
// through a counter in pinned memory -- chain_den() reports it once and stops using the multi-workgroup form in this process
__global__ void den_mw_check_kernel(const unsigned *abort_flag, unsigned *host_fallbacks) {
  if (*abort_flag != 0 && threadIdx.x == 0) __hip_atomic_fetch_add(host_fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Occupancies of one (frame, sequence): deriv[t*B+s][p] = deriv_weight * gamma_den(t, p) (overwrites the whole row)
constexpr int kGammaThreads = 512;
__global__ __launch_bounds__(kGammaThreads) void den_gamma_kernel(DenDev g, MatView y, int B, int T, float leaky, const float *alpha_all, const float *b_all,
                                                                  const float *S_all, int Hs, float deriv_weight, MatView deriv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[kGammaThreads / 64];
  const int t = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int H = g.H, P = g.P, P4 = (P + 3) & ~3, H4 = (H + 3) & ~3;
  float *x = smem, *ad = smem + P4, *bn = ad + H4;
  const float *alpha = alpha_all + ((size_t)s * (T + 1) + t) * Hs;     // alpha_dash(t, .)
  const float *bt = b_all + ((size_t)s * (T + 1) + t) * Hs, *bt1 = bt + Hs;
  const float inv = 1.0f / S_all[(size_t)s * (T + 1) + t + 1];
  const float *yr = y.data + (size_t)(t * B + s) * y.stride;
  for (int p = tid; p < P; p += kGammaThreads) x[p] = exp_limited(yr[p]);
  float local = 0.f;
  for (int h = tid; h < H; h += kGammaThreads) {
    const float a = alpha[h];
    ad[h] = a;
    bn[h] = bt1[h] * inv + leaky;
    local += a * bt[h];
  }
  const float Zd = block_sum(local, red, kGammaThreads / 64);  // (also the barrier before ad / bn / x are read)
  const float scale = deriv_weight / Zd;
  float *dr = deriv.data + (size_t)(t * B + s) * deriv.stride;
  // (the slice table and row ids of a thread's rows first, all at once: fetched row by row they sat, one trip to L2 each, in front of every row's arcs
  // -- twelve rows per thread at 6 034 pdfs, 33 us per block of which the arcs themselves are a third)
  constexpr int kRows = 12;
  const int nslot = g.by_pdf.nslices * 64, ln = tid & 63;
  auto term = [&](const uint2 a) { return __uint_as_float(a.y) * ad[a.x & 0xffffu] * bn[a.x >> 16]; };
  for (int s0 = tid; s0 < nslot; s0 += kRows * kGammaThreads) {
    int b0[kRows], w[kRows];
    unsigned row[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      const int slot = s0 + k * kGammaThreads;
      b0[k] = 0;
      w[k] = 0;
      row[k] = 0xffffffffu;
      if (slot < nslot) {
        b0[k] = g.by_pdf.base[slot >> 6];
        w[k] = (g.by_pdf.base[(slot >> 6) + 1] - b0[k]) >> 6;
        row[k] = g.by_pdf.row[slot];
      }
    }
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      if (row[k] == 0xffffffffu) continue;
      const float acc = sell_row_sum(g.by_pdf.arc + b0[k] + ln, w[k], term);
      dr[row[k]] = scale * acc * x[row[k]];
    }
  }
}
// ---------------------------------------------------------------------------------------------- denominator, wide form
// The persistent kernels above give a sequence one workgroup and keep its state vectors in LDS: right while they fit (up to
// ~10 000 states), a crawl beyond (30 000 states / 360 000 arcs: every arc is a 4-byte gather from L2, 580 ms per
// minibatch).  The wide form runs the recursion one frame per launch over ALL sequences with the SEQUENCE as the fastest
// index of every array (Kaldi's own choice, for the same reason), in GROUPS of SG sequences (32; 16 for minibatches of <= 16):
//   alpha[t][group][state][SG],  x[t][group][pdf][SG]          (SG = 32: one whole 128-byte line per state, line-aligned)
//  * a wave is SG sequences x 64/SG rows (states / pdfs).  The arcs of its rows are not fetched arc by arc: one coalesced
//    16-byte load per lane brings 64 arcs (4 arc positions of the wave's 16 rows) as (state | pdf << 16, p, p init_src), and
//    each lane group picks the arc of its row out of the holder's registers with ds_bpermute (the LDS crossbar, no memory).
//    Per arc step that leaves the two gathers the recursion cannot do without (alpha_src, x_pdf): the leaky-HMM term
//    alpha_dash = alpha + leaky A init_src costs no third one because p init_src travels in the arc.
//  * a workgroup serves ONE group: group = blockIdx.x % num_groups.  Workgroups go to the eight XCDs round-robin, so an XCD
//    only ever touches the frame slices of its own group(s) -- at 128 sequences and SG 32: 30 000 states x 128 bytes = 3.8 MB
//    of alpha + 0.8 MB of x against a 4 MB L2 (measured hit rate 77 %, 3.4 M L2 requests per launch; SG 16 fits better, 89 %
//    of 6.3 M, and is slower: L2 requests are per line, and a 64-byte run is half a line).  Runs that straddle a line cost a
//    second request each: every array of the wide form is 128-byte aligned.  The arc table streams past with non-temporal loads.
//  * the normaliser A(t-1, s) = sum_h alpha(t-1, h, s) is NOT a launch of its own between two frames: every workgroup of frame
//    t starts by summing the previous launch's partial sums for its group's sequences (fixed order, float4 loads all in
//    flight at once), so a frame is ONE launch forward and one backward.
// Measured (30 000 states / 360 000 arcs, 128 x 500 frames, tools/den_bench.py): 56 ms for the whole objective, from 105 ms with
// one 256-byte run per state and arc-by-arc loads; a recursion launch takes 48 us alone, 55 us beside the other direction's.
constexpr int kWideBatch = 8;   // arc steps whose picks and gathers are issued together (a multiple of 4)
constexpr int kWideSlices = 2;  // SELL slices (of 64 rows) per 256-thread block of the recursions: a wave takes 16 rows of each (TDNNF_WIDE_SLICES: experiments)

struct WideDims {
  int B, NG, SG;  // sequences, groups, sequences per group (16 or 32; NG * SG >= B)
};

typedef unsigned uv4 __attribute__((ext_vector_type(4)));

// xT[t][g][p][sl] = exp(clamp(y[t*B + g*SG + sl][p])): per frame a B x P -> P x B transpose through LDS
__global__ __launch_bounds__(256) void den_wide_prep_kernel(MatView y, WideDims d, int P, float *xT) {
  __shared__ float tile[64][65];
  const int t = blockIdx.z, p0 = blockIdx.x * 64, s0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int sq = s0 + i, p = p0 + tx;
    tile[i][tx] = (sq < d.B && p < P) ? exp_limited(y.data[(size_t)(t * d.B + sq) * y.stride + p]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, sq = s0 + tx;
    if (p < P && sq < d.NG * d.SG) xT[(((size_t)t * d.NG + sq / d.SG) * P + p) * d.SG + sq % d.SG] = tile[tx][i];
  }
}
// deriv[t*B + s][p] = dT[t][g][p][sl]
__global__ __launch_bounds__(256) void den_wide_unprep_kernel(const float *dT, WideDims d, int P, MatView deriv) {
  __shared__ float tile[64][65];
  const int t = blockIdx.z, p0 = blockIdx.x * 64, s0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, sq = s0 + tx;
    tile[i][tx] = (p < P && sq < d.B) ? dT[(((size_t)t * d.NG + sq / d.SG) * P + p) * d.SG + sq % d.SG] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int sq = s0 + i, p = p0 + tx;
    if (sq < d.B && p < P) deriv.data[(size_t)(t * d.B + sq) * deriv.stride + p] = tile[tx][i];
  }
}

// frame 0 of alpha (before the leaky term) / frame T of b: v[g][h][sl] = init_h or 1; norm[s] = init_sum
__global__ __launch_bounds__(256) void den_wide_init_kernel(DenDev g, WideDims d, int Hs, bool ones, float *v, float *norm) {
  const long long total = (long long)d.NG * Hs * d.SG;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {  // (grid_for caps the grid)
    const int h = (int)((e / d.SG) % Hs);
    v[e] = h < g.H ? (ones ? 1.0f : g.init[h]) : 0.f;
  }
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < d.B; e += gridDim.x * 256LL) norm[e] = g.init_sum;
}

// where a workgroup of the recursion kernels stands: its group, its block of slices, its lane's sequence and row lane
template <int SG>
struct WideLane {
  static constexpr int RL = 64 / SG;  // rows side by side in a wave
  static constexpr int IT = 16 / RL;  // passes over a wave's 16 rows
  int grp, blk, sl, rl, sq, wave, lane;
  bool on;
  __device__ WideLane(const WideDims &d, int g, int b) {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    grp = g;
    blk = b;
    sl = lane % SG;
    rl = lane / SG;
    sq = grp * SG + sl;
    on = sq < d.B;
  }
};
// Normaliser of the previous launch for this lane's sequence: the sum of the group's `nblk` partial sums, 256/SG threads per
// sequence and four of them per load, then the threads' sums in fixed order (the same number in every workgroup); stored by
// the first workgroup of a group.  part[group][sequence][npad], npad = nblk rounded up to 4 with the tail zeroed: the sums of one
// sequence are contiguous, and every thread has all its loads (<= 8 float4 for up to 1 024 partial sums at SG 32) in flight at
// once -- the partial sums were written by the previous launch, possibly through another XCD's L2, so each load is a trip to
// memory, and a chain of them was most of a frame's time.
__device__ __forceinline__ int wide_npad(int nblk) { return (nblk + 3) & ~3; }
template <int SG>
__device__ __forceinline__ float wide_norm(const float *part, int nblk, float *norm_out, float *red, const WideLane<SG> &L) {
  constexpr int NQ = 256 / SG;
  const int q = threadIdx.x / SG, nch = wide_npad(nblk) >> 2;
  const float4 *pp = (const float4 *)(part + ((size_t)L.grp * SG + L.sl) * wide_npad(nblk));
  float v = 0.f;
  for (int c0 = q; c0 < nch; c0 += 8 * NQ) {
    float4 f[8];
#pragma unroll
    for (int u = 0; u < 8; u++) f[u] = c0 + u * NQ < nch ? pp[c0 + u * NQ] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; u++) v += (f[u].x + f[u].y) + (f[u].z + f[u].w);
  }
  red[q * SG + L.sl] = v;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; i++) tot += red[i * SG + L.sl];
  __syncthreads();  // red is used again for the block's own partial
  if (L.blk == 0 && q == 0 && L.on) norm_out[L.sq] = tot;
  return L.on ? tot : 1.0f;
}
// sum of `v` over the block's rows for each sequence of the group -> part[grp][sl][blk] (padding sequences: 0)
template <int SG>
__device__ __forceinline__ void wide_store_partial(float v, float *red, float *part, int nblk, const WideLane<SG> &L) {
#pragma unroll
  for (int o = SG; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  if (L.rl == 0) red[L.wave * SG + L.sl] = v;
  __syncthreads();
  if (threadIdx.x < SG) {
    float *row = part + ((size_t)L.grp * SG + L.sl) * wide_npad(nblk);
    row[L.blk] = (red[L.sl] + red[SG + L.sl]) + (red[2 * SG + L.sl] + red[3 * SG + L.sl]);
    if (L.blk == nblk - 1)
      for (int k = nblk; k < wide_npad(nblk); k++) row[k] = 0.f;
  }
}
// One load for 4 consecutive arc positions of a wave's 16 rows: lane l holds arc (position jb + l/16, row l%16 of the 16)
__device__ __forceinline__ uv4 wide_load_arcs(const uv4 *ap, int jb, int w, int lane) {
  const int j = jb + (lane >> 4);
  uv4 a = {0u, 0u, 0u, 0u};
  if (j < w) a = __builtin_nontemporal_load(ap + (size_t)j * 64);
  return a;
}

// The arcs of one SELL slice for this wave's 16 rows.  Every arc step needs the arc's (key, p, p init_src) and two gathered
// values: t1[(key & 0xffff) * SG + sl] and t2[(key >> 16) * SG + sl].  kWideBatch steps at a time: all their picks and gathers are
// issued before the first product is formed, and the load of the next four arc positions goes out BEHIND the gathers -- loads
// complete in order, so an arc load (a trip to memory) issued ahead of them would hold every gather's data back.
// MODE 0 forward:  acc += t2 p t1,  acl += t2 p init_src           (t1 = alpha(t-1), t2 = x(t-1))
// MODE 1 backward: acc += p t2 t1,  acl += p t2                    (t1 = b(t+1),     t2 = x(t))
// MODE 2 occupancy: acc += (t1 p + c0 p init_src) (t2 c1 + c2)      (t1 = alpha(t),   t2 = b(t+1); c0 = leaky A, c1 = 1/S, c2 = leaky)
template <int SG, int MODE>
__device__ __forceinline__ void wide_slice(const tdnnf_den_graph::Sell &T, int slice, const float *t1, const float *t2, const WideLane<SG> &L, float c0, float c1,
                                           float c2, float *acc, float *acl) {
  using WL = WideLane<SG>;
  const int b0 = T.base[slice], w = (T.base[slice + 1] - b0) >> 6;
  const uv4 *ap = (const uv4 *)T.arc4 + b0 + L.wave * 16 + (L.lane & 15);
  uv4 a = wide_load_arcs(ap, 0, w, L.lane);
  constexpr int BS = kWideBatch, NB = WL::IT * 4 / BS;  // arc steps per batch, batches per four arc positions
  for (int jb = 0; jb < w; jb += 4) {
    uv4 an = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int h = 0; h < NB; h++) {
      float g1[BS], g2[BS], pr[BS], iq[BS];
#pragma unroll
      for (int k = 0; k < BS; k++) {
        const int e = h * BS + k, row4 = ((e / 4) * WL::RL + L.rl) * 4 + (e % 4) * 64;  // step e: row pass e/4, arc position e%4
        const unsigned key = (unsigned)__builtin_amdgcn_ds_bpermute(row4, (int)a.x);
        pr[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(row4, (int)a.y));
        iq[k] = MODE != 1 ? __int_as_float(__builtin_amdgcn_ds_bpermute(row4, (int)a.z)) : 0.f;
        g1[k] = t1[(key & 0xffffu) * SG + L.sl];
        g2[k] = t2[(key >> 16) * SG + L.sl];
      }
      if (h == NB - 1) an = wide_load_arcs(ap, jb + 4, w, L.lane);  // zeros past the end
#pragma unroll
      for (int k = 0; k < BS; k++) {
        const int it = (h * BS + k) / 4;
        if (MODE == 0) {
          acc[it] += g2[k] * pr[k] * g1[k];
          acl[it] += g2[k] * iq[k];
        } else if (MODE == 1) {
          const float px = pr[k] * g2[k];
          acc[it] += px * g1[k];
          acl[it] += px;
        } else {
          acc[it] += (g1[k] * pr[k] + c0 * iq[k]) * (g2[k] * c1 + c2);
        }
      }
      // keep the next batch's picks and loads behind this batch's arithmetic (4 BS live values each): the arc registers are
      // "redefined" here and the batch's sums "used"
      asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z)::"memory");
#pragma unroll
      for (int it = h * BS / 4; it < (h + 1) * BS / 4; it++) {
        asm volatile("" : "+v"(acc[it]));
        if (MODE != 2) asm volatile("" : "+v"(acl[it]));
      }
    }
    a = an;
  }
}

// alpha(t, h, s) = 1/A(t-1, s) sum_arcs alpha_dash(t-1, src, s) p x(t-1, pdf, s)
//               = (1/A) sum_arcs alpha(t-1, src, s) p x + leaky sum_arcs p init_src x        (alpha_dash = alpha + leaky A init).
// part_prev: launch t-1's partial sums of A(t-1) (null at t = 1: A(0) is in asum[0] already); nsl: slices per block
template <int SG>
__global__ __launch_bounds__(256) void den_wide_fwd_kernel(DenDev g, WideDims d, int t, float leaky, const float *xT, float *alphaT, float *asum, int Hs,
                                                           const float *part_prev, int nblk, int nsl, float *part) {
  using WL = WideLane<SG>;
  __shared__ float red[256];
  const WL L(d, blockIdx.x % d.NG, blockIdx.x / d.NG);
  const size_t frame = (size_t)d.NG * Hs * SG;
  const float *prev = alphaT + (size_t)(t - 1) * frame + (size_t)L.grp * Hs * SG;  // uniform; lanes add (state * SG + sl)
  float *cur = alphaT + (size_t)t * frame + (size_t)L.grp * Hs * SG;
  const float *x = xT + ((size_t)(t - 1) * d.NG + L.grp) * g.P * SG;
  const int s0 = L.blk * nsl, s1 = min(s0 + nsl, g.by_dst.nslices);
  const float Aprev = part_prev ? wide_norm<SG>(part_prev, nblk, asum + (size_t)(t - 1) * d.B, red, L) : (L.on ? asum[(size_t)(t - 1) * d.B + L.sq] : 1.f);
  const float inv = 1.0f / Aprev;
  float total = 0.f;
  for (int slice = s0; slice < s1; slice++) {
    float acc[WL::IT], acl[WL::IT];
#pragma unroll
    for (int it = 0; it < WL::IT; it++) acc[it] = acl[it] = 0.f;
    wide_slice<SG, 0>(g.by_dst, slice, prev, x, L, 0.f, 0.f, 0.f, acc, acl);
#pragma unroll
    for (int it = 0; it < WL::IT; it++) {
      const unsigned h = g.by_dst.row[slice * 64 + L.wave * 16 + it * WL::RL + L.rl];
      if (h != 0xffffffffu && L.on) {
        const float v = acc[it] * inv + leaky * acl[it];
        cur[h * SG + L.sl] = v;
        total += v;
      }
    }
  }
  wide_store_partial<SG>(total, red, part, nblk, L);
}

// out[s] = sum over the nblk partial rows, in the order wide_norm takes them (after the last frame of a recursion)
template <int SG>
__global__ __launch_bounds__(256) void den_wide_sum_kernel(const float *part, int nblk, WideDims d, float *out) {
  __shared__ float red[256];
  const WideLane<SG> L(d, blockIdx.x, 0);
  wide_norm<SG>(part, nblk, out, red, L);
}

// tot(s) = sum_h alpha_dash(T, h, s) = A(T, s) (1 + leaky init_sum); log-prob of the sequence
__global__ __launch_bounds__(256) void den_wide_total_kernel(DenDev g, int B, int T, float leaky, const float *asum, double *logprob) {
  const int sq = blockIdx.x * 256 + threadIdx.x;
  if (sq >= B) return;
  const float tt = asum[(size_t)T * B + sq] * (1.0f + leaky * g.init_sum);
  double lc = 0.0;
  for (int t = 0; t < T; t++) lc += (double)logf(asum[(size_t)t * B + sq]);
  logprob[sq] = (double)logf(tt) + lc;
}

// The backward recursion does not wait for the forward one: it runs SELF-NORMALISED on a stream of its own, beside it.
// beta_dash is linear and homogeneous in its last frame, so with b(T, h) = 1, S(t) = sum_h init_h b(t, h) and
//   b(t, h, s) = sum_arcs p x(t, pdf, s) (b(t+1, dst, s) / S(t+1, s) + leaky)
// the true beta_dash(t) is a per-(frame, sequence) multiple of b(t) (the leaky term of the normalised vector is the constant
// `leaky`: sum_h init_h b/S = 1).  The multiple never has to be formed: the occupancies of a frame sum to one, so
//   gamma(t, p, s) = x(t, p, s) sum_arcs p alpha_dash(t, src, s) (b(t+1, dst, s) / S(t+1, s) + leaky) / Zd(t, s),
//   Zd(t, s) = sum_h alpha_dash(t, h, s) b(t, h, s)        (= the sum over p of the numerators, by the recursion above),
// which needs alpha (kept for every frame anyway) and b for every frame (another (T+1) x H x B floats: 7.7 GB at 30 000 states,
// 128 x 500 frames -- what 288 GB are for) and leaves the occupancy pass with no dependence between frames: ONE launch over
// all of them instead of one per frame.

// b(t) from b(t+1) (bnextT) and S(t+1) = (1/S) sum_arcs p x b(t+1, dst) + leaky sum_arcs p x; partials of S(t) = sum_h init_h b(t, h, s).
// part_prev: the partial sums of launch t+1 (null at t = T-1: S(T) is in S already)
template <int SG>
__global__ __launch_bounds__(256) void den_wide_beta_kernel(DenDev g, WideDims d, int t, float leaky, const float *xT, const float *bnextT, float *S, int Hs,
                                                            float *bcurT, const float *part_prev, int nblk, int nsl, float *part) {
  using WL = WideLane<SG>;
  __shared__ float red[256];
  const WL L(d, blockIdx.x % d.NG, blockIdx.x / d.NG);
  const float *bnext = bnextT + (size_t)L.grp * Hs * SG;
  float *bcur = bcurT + (size_t)L.grp * Hs * SG;
  const float *x = xT + ((size_t)t * d.NG + L.grp) * g.P * SG;
  const int s0 = L.blk * nsl, s1 = min(s0 + nsl, g.by_src.nslices);
  const float Snext = part_prev ? wide_norm<SG>(part_prev, nblk, S + (size_t)(t + 1) * d.B, red, L) : (L.on ? S[(size_t)(t + 1) * d.B + L.sq] : 1.f);
  const float inv = 1.0f / Snext;
  float total = 0.f;
  for (int slice = s0; slice < s1; slice++) {
    float acc[WL::IT], acl[WL::IT];
#pragma unroll
    for (int it = 0; it < WL::IT; it++) acc[it] = acl[it] = 0.f;
    wide_slice<SG, 1>(g.by_src, slice, bnext, x, L, 0.f, 0.f, 0.f, acc, acl);
#pragma unroll
    for (int it = 0; it < WL::IT; it++) {
      const unsigned h = g.by_src.row[slice * 64 + L.wave * 16 + it * WL::RL + L.rl];
      if (h != 0xffffffffu && L.on) {
        const float v = acc[it] * inv + leaky * acl[it];
        bcur[h * SG + L.sl] = v;
        total += g.init[h] * v;
      }
    }
  }
  wide_store_partial<SG>(total, red, part, nblk, L);
}

// Zd(t, s) = sum_h (alpha(t, h, s) + leaky A(t, s) init_h) b(t, h, s) for every frame: block (t, group), 256/SG threads per
// sequence stride the states
template <int SG>
__global__ __launch_bounds__(256) void den_wide_dot_kernel(DenDev g, WideDims d, float leaky, const float *alphaT, const float *asum, int Hs, const float *bT,
                                                           float *Zd) {
  __shared__ double red[256];
  const int t = blockIdx.x, grp = blockIdx.y, sl = threadIdx.x % SG, hl = threadIdx.x / SG, sq = grp * SG + sl;
  const size_t off = ((size_t)t * d.NG + grp) * Hs * SG + sl;
  const float *alpha = alphaT + off, *b = bT + off;
  double acc = 0.0;
  if (sq < d.B) {
    const float lka = leaky * asum[(size_t)t * d.B + sq];
    for (int h = hl; h < g.H; h += 256 / SG) acc += (double)((alpha[(size_t)h * SG] + lka * g.init[h]) * b[(size_t)h * SG]);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (hl == 0 && sq < d.B) {
    double tot = 0.0;
    for (int q = 0; q < 256 / SG; q++) tot += red[q * SG + sl];
    Zd[(size_t)t * d.B + sq] = (float)tot;
  }
}

// x(t, p, s) <- deriv_weight gamma_den(t, p, s)  (in place), every frame in one launch: blockIdx.y = t, one SELL slice per
// block (x fastest: the resident waves stay within a frame or two, whose alpha and b slices an XCD's L2 can hold)
template <int SG>
__global__ __launch_bounds__(256) void den_wide_gamma_kernel(DenDev g, WideDims d, float leaky, float *xT, const float *alphaT, const float *asum, int Hs,
                                                             const float *bT, const float *S, const float *Zd, float deriv_weight) {
  using WL = WideLane<SG>;
  const WL L(d, blockIdx.x % d.NG, blockIdx.x / d.NG);
  const int t = blockIdx.y, slice = L.blk;
  const size_t frame = (size_t)d.NG * Hs * SG;
  const float *alpha = alphaT + (size_t)t * frame + (size_t)L.grp * Hs * SG, *bnext = bT + (size_t)(t + 1) * frame + (size_t)L.grp * Hs * SG;
  float *x = xT + ((size_t)t * d.NG + L.grp) * g.P * SG;
  float lka = 0.f, inv = 0.f, scale = 0.f;
  if (L.on) {
    lka = leaky * asum[(size_t)t * d.B + L.sq];
    inv = 1.0f / S[(size_t)(t + 1) * d.B + L.sq];
    scale = deriv_weight / Zd[(size_t)t * d.B + L.sq];
  }
  float acc[WL::IT];
#pragma unroll
  for (int it = 0; it < WL::IT; it++) acc[it] = 0.f;
  wide_slice<SG, 2>(g.by_pdf, slice, alpha, bnext, L, lka, inv, leaky, acc, nullptr);
#pragma unroll
  for (int it = 0; it < WL::IT; it++) {
    const unsigned p = g.by_pdf.row[slice * 64 + L.wave * 16 + it * WL::RL + L.rl];
    if (p != 0xffffffffu && L.on) x[p * SG + L.sl] *= scale * acc[it];
  }
}

// The numerator recursion runs in the log domain, where the values grow with the frame index (log alpha ~ -8 t): at 500
// frames a float's 24 bits leave an absolute error of ~2e-4 in the exponent of a posterior, 2e-3 in the posteriors of a
// 1500-frame chunk (measured: frame sums of gamma_num off by up to 2.2e-3, derivative 6.8e-4 from a float64 evaluation).  Doubles:
// one wave per sequence walks a few states per frame, the arithmetic is free.
__device__ __forceinline__ double log_add(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double m = fmax(a, b), d = fmin(a, b) - m;
  return m + log1p(exp(d));
}

struct SupDev {
  int B, T;
  float weight;
  const int *seq_state_begin, *frame_state_begin;
  const float *final_logprob;
  const int *in_begin, *in_src, *in_pdf;
  const float *in_lp;
  const int *out_begin, *out_dst, *out_pdf;
  const float *out_lp;
};

// one wave per sequence.  la/lb: global scratch indexed by global state id.
__global__ __launch_bounds__(64) void numerator_kernel(SupDev sp, MatView y, MatView xent_out, double *la, double *lb,
                                                       double *num_logprob, double *xent_objf, MatView deriv,
                                                       MatView xent_deriv, float xent_scale, int phases) {
  // phases bit 0: forward-backward recursion (la, lb, total -> num_logprob): needs the chain output y only;
  //        bit 2: xent posteriors / objective; bit 1: deriv += weight * gamma_num  (both need the recursion's la / lb / total,
  //        possibly from an earlier launch)
  const int s = blockIdx.x, lane = threadIdx.x, B = sp.B, T = sp.T;
  const int *fsb = sp.frame_state_begin + (size_t)s * (T + 2);
  const int s0 = sp.seq_state_begin[s], s1 = sp.seq_state_begin[s + 1];
  double tot = -INFINITY;
  if (phases & 1) {
  for (int i = s0 + lane; i < s1; i += 64) la[i] = (i == s0) ? 0.0 : -INFINITY;
  __syncthreads();
  for (int t = 1; t <= T; t++) {  // states entered at time t
    for (int st = fsb[t] + lane; st < fsb[t + 1]; st += 64) {
      double v = -INFINITY;
      for (int a = sp.in_begin[st]; a < sp.in_begin[st + 1]; a++)
        v = log_add(v, la[sp.in_src[a]] + ((double)sp.in_lp[a] + (double)y.data[(size_t)((t - 1) * B + s) * y.stride + sp.in_pdf[a]]));
      la[st] = v;
    }
    __syncthreads();
  }
  for (int st = fsb[T] + lane; st < fsb[T + 1]; st += 64) {
    const float f = sp.final_logprob[st];
    lb[st] = (double)f;
    if (f != -INFINITY) tot = log_add(tot, la[st] + (double)f);
  }
  for (int o = 32; o > 0; o >>= 1) tot = log_add(tot, __shfl_xor(tot, o, 64));
  __syncthreads();
  for (int t = T - 1; t >= 0; t--) {
    for (int st = fsb[t] + lane; st < fsb[t + 1]; st += 64) {
      double v = -INFINITY;
      for (int a = sp.out_begin[st]; a < sp.out_begin[st + 1]; a++)
        v = log_add(v, ((double)sp.out_lp[a] + (double)y.data[(size_t)(t * B + s) * y.stride + sp.out_pdf[a]]) + lb[sp.out_dst[a]]);
      lb[st] = v;
    }
    __syncthreads();
  }
  } else {
    tot = num_logprob[s];
  }
  const bool do_xent = (phases & 4) != 0, do_deriv = (phases & 2) != 0;
  if ((phases & 1) && lane == 0) num_logprob[s] = tot;
  // posteriors: lane = frame (distinct output rows per lane, fixed arc order -> deterministic)
  double xo = 0.0;
  for (int t = lane; t < T; t += 64) {
    const size_t row = (size_t)(t * B + s);
    for (int st = fsb[t]; st < fsb[t + 1]; st++)
      for (int a = sp.out_begin[st]; a < sp.out_begin[st + 1]; a++) {
        const int pdf = sp.out_pdf[a];
        const double ll = (double)sp.out_lp[a] + (double)y.data[row * y.stride + pdf];
        const float gam = sp.weight * (float)exp(la[st] + ll + lb[sp.out_dst[a]] - tot);
        if (do_deriv && deriv.data) deriv.data[row * deriv.stride + pdf] += gam;
        if (do_xent && xent_deriv.data) xent_deriv.data[row * xent_deriv.stride + pdf] += xent_scale * gam;
        if (do_xent && xent_out.data) xo += (double)gam * (double)xent_out.data[row * xent_out.stride + pdf];
      }
  }
  if (do_xent) {
    for (int o = 32; o > 0; o >>= 1) xo += __shfl_xor(xo, o, 64);
    if (lane == 0) xent_objf[s] = xo;
  }
}

// results: [0] objf [1] l2_term [2] weight [3] num [4] den [5] ok [6] xent objf
__global__ void chain_finalize_kernel(const double *num_lp, const double *den_lp, const double *xent, const double *l2sum,
                                      int B, int T, float weight, float l2_regularize, double *results) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double num = 0, den = 0, xo = 0;
  for (int s = 0; s < B; s++) {
    num += num_lp[s];
    den += den_lp[s];
    xo += xent[s];
  }
  num *= weight;
  den *= weight;
  double objf = num - den;
  const double w = (double)weight * B * T;
  const bool ok = (objf - objf == 0.0);
  if (!ok) objf = -10.0 * w;
  results[0] = objf;
  results[1] = (l2_regularize == 0.f || !l2sum) ? 0.0 : -0.5 * (double)weight * l2_regularize * l2sum[0];
  results[2] = w;
  results[3] = num;
  results[4] = den;
  results[5] = ok ? 1.0 : 0.0;
  results[6] = ok ? xo : 0.0;
}

// failure path (objf not finite): zero the derivatives; otherwise add the l2 term's derivative.
__global__ void chain_guard_kernel(const double *results, MatView y, float l2_scale, MatView d, MatView xd) {
  const bool ok = results[5] != 0.0;
  if (ok && l2_scale == 0.f) return;
  const long long total = (long long)d.rows * d.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int r = (int)(e / d.cols), c = (int)(e % d.cols);
    if (!ok) {
      d.data[(size_t)r * d.stride + c] = 0.f;
      if (xd.data) xd.data[(size_t)r * xd.stride + c] = 0.f;
    } else {
      d.data[(size_t)r * d.stride + c] += -l2_scale * y.data[(size_t)r * y.stride + c];
    }
  }
}
__global__ __launch_bounds__(256) void sumsq_kernel(MatView y, double *out) {
  __shared__ double red[4];
  double s = 0;
  const long long total = (long long)y.rows * y.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const double v = y.data[(size_t)(e / y.cols) * y.stride + e % y.cols];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void zero_rows_kernel(MatView m) {
  const long long total = (long long)m.rows * m.cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL)
    m.data[(size_t)(e / m.cols) * m.stride + e % m.cols] = 0.f;
}

// -------------------------------------------------------------------------------- host helpers
template <class T>
int to_device(const std::vector<T> &v, T **out) {
  *out = nullptr;
  if (v.empty()) {
    TDNNF_HIP(hipMalloc((void **)out, sizeof(T)));
    return TDNNF_OK;
  }
  TDNNF_HIP(hipMalloc((void **)out, sizeof(T) * v.size()));
  TDNNF_HIP(hipMemcpy(*out, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return TDNNF_OK;
}

// rows[r] = list of (key, prob); builds SELL-64 over rows sorted by descending degree (stable)
// init (may be null): initial probabilities indexed by the low 16 bits of the key (the arc's source state), for arc4.z
int build_sell(int nrows, const std::vector<std::vector<std::pair<unsigned, float>>> &rows, const std::vector<float> *init, tdnnf_den_graph::Sell *out) {
  const int ns = (nrows + 63) / 64;
  std::vector<int> order(nrows);
  for (int r = 0; r < nrows; r++) order[r] = r;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a].size() > rows[b].size(); });
  std::vector<int> base(ns + 1, 0);
  for (int k = 0; k < ns; k++) base[k + 1] = base[k] + (int)rows[order[64 * k]].size() * 64;
  std::vector<unsigned> rowid((size_t)ns * 64, 0xffffffffu);
  std::vector<uint2> arc(base[ns], make_uint2(0u, 0u));  // padding: state 0 / pdf 0 with prob 0
  std::vector<uint4> arc4(base[ns], make_uint4(0u, 0u, 0u, 0u));
  // The order of a row's arcs is free (a sum).  The j-th arcs of a slice's 64 rows are read by one wave instruction and become two LDS gathers
  // (the two 16-bit halves of the key index state / pdf vectors): 64 random addresses put 4-5 on the worst of 64 banks.  Greedy per position: every
  // row takes, among its arcs not placed yet, the one whose two banks are least used at this position so far (half a wave -- 32 lanes -- is what the
  // LDS serves at once).  The persistent recursions are bound by these gathers, not by the arc loads (docs/experiments.md r4-i).
  std::vector<std::vector<std::pair<unsigned, float>>> placed(nrows);
  for (int k = 0; k < ns; k++) {
    const int w = (base[k + 1] - base[k]) / 64;
    std::vector<std::vector<char>> used(64);
    for (int l = 0; l < 64 && 64 * k + l < nrows; l++) used[l].assign(rows[order[64 * k + l]].size(), 0);
    for (int j = 0; j < w; j++) {
      int load[2][2][64];  // [half-wave][which key half][bank]
      memset(load, 0, sizeof(load));
      for (int l = 0; l < 64 && 64 * k + l < nrows; l++) {
        const auto &src = rows[order[64 * k + l]];
        int best = -1, best_cost = 1 << 30;
        for (size_t c = 0; c < src.size(); c++) {
          if (used[l][c]) continue;
          const int cost = load[l >> 5][0][src[c].first & 63u] + load[l >> 5][1][(src[c].first >> 16) & 63u];
          if (cost < best_cost) {
            best_cost = cost;
            best = (int)c;
          }
        }
        if (best < 0) continue;  // (this row is shorter than the slice: padding from here on)
        used[l][best] = 1;
        load[l >> 5][0][src[best].first & 63u]++;
        load[l >> 5][1][(src[best].first >> 16) & 63u]++;
        placed[order[64 * k + l]].push_back(src[best]);
      }
    }
  }
  for (int s = 0; s < nrows; s++) {
    const int r = order[s];
    rowid[s] = (unsigned)r;
    for (size_t j = 0; j < placed[r].size(); j++) {
      unsigned bits;
      memcpy(&bits, &placed[r][j].second, 4);
      arc[base[s / 64] + j * 64 + s % 64] = make_uint2(placed[r][j].first, bits);
      const float ip = init ? placed[r][j].second * (*init)[placed[r][j].first & 0xffffu] : 0.f;
      unsigned ibits;
      memcpy(&ibits, &ip, 4);
      arc4[base[s / 64] + j * 64 + s % 64] = make_uint4(placed[r][j].first, bits, ibits, 0u);
    }
  }
  out->nrows = nrows;
  out->nslices = ns;
  out->entries = base[ns];
  for (int G = 0; G <= 8; G++) {
    out->mw_max_arcs[G] = 0;
    if (G != 2 && G != 4 && G != 8) continue;
    for (int g = 0; g < G; g++) {
      int own = 0;
      for (int k = g; k < ns; k += G) own += base[k + 1] - base[k];
      out->mw_max_arcs[G] = std::max(out->mw_max_arcs[G], own);
    }
  }
  int rc;
  if ((rc = to_device(base, &out->base))) return rc;
  if ((rc = to_device(rowid, &out->row))) return rc;
  if ((rc = to_device(arc4, &out->arc4))) return rc;
  return to_device(arc, &out->arc);
}

struct ChainPlan {
  int Hs;
  bool lds_state;
  bool split;         // persistent form with the backward recursion beside the forward one (den_beta_kernel + den_gamma_kernel)
  bool wide;          // den_wide_*: one launch per frame over all sequences, sequence-minor arrays
  int wide_blocks;    // partial rows of the widest launch
  int SG, NG;         // wide: sequences per group, groups
  size_t alpha_floats, asum_floats, gstate_floats, la_floats;
  size_t lds_fwd, lds_bwd;
};
int g_den_mode = 0;  // tdnnf_chain_set_denominator_mode: 0 automatic, 1 persistent, 2 wide, 3 persistent with one workgroup per sequence
// per device (a process may drive several): a time-out on one device says nothing about the others
struct MwDev {
  unsigned *fallbacks = nullptr;  // pinned host memory, written by den_mw_check_kernel
  bool off = false;               // a multi-workgroup launch gave up once on this device: not used again there
  int cus = 0;                    // compute units (0 = not asked yet, -1 = unknown)
};
constexpr int kMaxDev = 64;
MwDev g_mw[kMaxDev];
MwDev *mw_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  MwDev &d = g_mw[dev];
  if (d.cus == 0) {
    hipDeviceProp_t prop;
    d.cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : -1;
  }
  return &d;
}
// exchange buffers and counters of the multi-workgroup recursions (den_mw_kernel), behind b_all / S_all in the split region
size_t mw_slots(const tdnnf_den_graph *g) { return (size_t)std::max(g->by_dst.nslices, g->by_src.nslices) * 64; }
size_t mw_extra_floats(const tdnnf_den_graph *g, int B) { return 4 * (size_t)B * mw_slots(g) + 4 * (size_t)B + 64; }
// workgroups per sequence for B sequences (0: one workgroup per sequence, the other kernels): both recursions at once must fit the chip at one
// workgroup per CU, every workgroup's arcs and state vectors its LDS, and a workgroup owns at most 64 slices.
// Four per sequence.  Measured in the step at 1500 x 16 (ms): one 27.3, two 27.5, four 23.8, eight 25.2 -- eight are faster alone (5.0 against
// 5.6 ms for both recursions) but their 256 workgroups hold every CU while the xent head's backward pass wants them; two cost what they gain.
int mw_groups(const tdnnf_den_graph *g, int B, int T) {
  MwDev *md = mw_dev();
  if (g_den_mode == 3 || !md || md->off || T < 8) return 0;
  (void)hipGetLastError();
  const int cus = md->cus, G = 4;
  if (cus <= 0 || 2 * B * G > cus) return 0;
  const int ns = std::min(g->by_dst.nslices, g->by_src.nslices), nsmax = std::max(g->by_dst.nslices, g->by_src.nslices);
  if (ns < G || (nsmax + G - 1) / G > 64) return 0;
  const size_t arcs = (size_t)std::max(g->by_dst.mw_max_arcs[G], g->by_src.mw_max_arcs[G]);
  const size_t lds = sizeof(float) * (((g->P + 3) & ~3) + ((g->H + 3) & ~3)) + 8 * arcs + 256 * (size_t)((nsmax + G - 1) / G);
  if (lds > 150 * 1024 || g->P > 8 * 960) return 0;
  return G;
}

ChainPlan chain_plan(const tdnnf_den_graph *g, int B, int T, int num_states_sup) {
  ChainPlan p;
  p.Hs = (g->H + 3) & ~3;
  const int P4 = (g->P + 3) & ~3;
  p.lds_fwd = sizeof(float) * (P4 + p.Hs);
  p.lds_bwd = sizeof(float) * (P4 + 3 * p.Hs);
  p.lds_state = p.lds_bwd <= 150 * 1024;
  if (!p.lds_state) {
    p.lds_fwd = sizeof(float) * P4;
    p.lds_bwd = sizeof(float) * P4;
  }
  const int mode = g_den_mode;
  p.wide = mode == 2 || (mode == 0 && !p.lds_state);
  p.split = !p.wide && p.lds_state;
  const int rows = std::max(std::max(g->by_dst.nslices, g->by_src.nslices), g->by_pdf.nslices) * 64;
  p.wide_blocks = (rows / 64 + 3) & ~3;  // partial sums per sequence (at most one per slice), padded to float4
  // 32 sequences per group (a whole 128-byte line per state) unless the minibatch has no more than 16
  p.SG = B > 16 ? 32 : 16;
  p.NG = (B + p.SG - 1) / p.SG;
  const size_t Bw = p.wide ? (size_t)p.NG * p.SG : (size_t)B;  // the wide arrays hold whole groups
  p.alpha_floats = Bw * (T + 1) * p.Hs;
  p.asum_floats = ((size_t)B * (3 * T + 4) + 31) & ~(size_t)31;  // wide: A(0..T), S(0..T) of the backward recursion, Zd(0..T-1)
  // wide: the backward vectors of every frame, two double-buffered sets of partial rows (the recursions run side by side) and x = exp(clamp(y)) /
  // the derivative, sequence-minor (T x P x B)
  p.gstate_floats = p.wide ? Bw * (size_t)(T + 1) * p.Hs + 4 * Bw * p.wide_blocks + (size_t)T * g->P * Bw : (p.lds_state ? (p.split ? (size_t)B * (T + 1) * (p.Hs + 1) + 32 + mw_extra_floats(g, B) : 0) : (size_t)B * 3 * p.Hs);
  p.la_floats = 4 * (size_t)num_states_sup + 2;  // two arrays of DOUBLES (log alpha, log beta of the numerator), 8-byte aligned
  return p;
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

int tdnnf_den_graph_create(int H, int A, int P, const int *src, const int *dst, const int *pdf, const float *prob,
                           const float *initial_probs, int start_state, tdnnf_den_graph **out) {
  TDNNF_REQUIRE(out && H > 0 && A > 0 && P > 0 && src && dst && pdf && prob, "den_graph_create: bad arguments");
  TDNNF_REQUIRE(H <= 65535 && P <= 65535, "den_graph_create: num_states and num_pdfs must be <= 65535 (16-bit packed arcs)");
  for (int a = 0; a < A; a++)
    TDNNF_REQUIRE(src[a] >= 0 && src[a] < H && dst[a] >= 0 && dst[a] < H && pdf[a] >= 0 && pdf[a] < P && prob[a] >= 0,
                  "den_graph_create: arc %d out of range", a);
  std::vector<float> init(H);
  if (initial_probs) {
    init.assign(initial_probs, initial_probs + H);
  } else {  // DenominatorGraph::SetInitialProbs: 100-step average occupancy of the row-normalised graph
    TDNNF_REQUIRE(start_state >= 0 && start_state < H, "den_graph_create: bad start state");
    std::vector<double> norm(H, 0.0), cur(H, 0.0), nxt(H), avg(H, 0.0);
    for (int a = 0; a < A; a++) norm[src[a]] += prob[a];
    cur[start_state] = 1.0;
    for (int it = 0; it < 100; it++) {
      for (int h = 0; h < H; h++) avg[h] += cur[h] / 100;
      std::fill(nxt.begin(), nxt.end(), 0.0);
      for (int a = 0; a < A; a++)
        if (norm[src[a]] > 0) nxt[dst[a]] += cur[src[a]] * prob[a] / norm[src[a]];
      cur.swap(nxt);
    }
    for (int h = 0; h < H; h++) init[h] = (float)avg[h];
  }
  tdnnf_den_graph *g = new tdnnf_den_graph();
  memset(g, 0, sizeof(*g));
  g->H = H;
  g->A = A;
  g->P = P;
  std::vector<std::vector<std::pair<unsigned, float>>> bd(H), bs(H), bp(P);
  for (int a = 0; a < A; a++) {
    bd[dst[a]].push_back({(unsigned)src[a] | ((unsigned)pdf[a] << 16), prob[a]});
    bs[src[a]].push_back({(unsigned)dst[a] | ((unsigned)pdf[a] << 16), prob[a]});
    bp[pdf[a]].push_back({(unsigned)src[a] | ((unsigned)dst[a] << 16), prob[a]});
  }
  int rc;
  if ((rc = build_sell(H, bd, &init, &g->by_dst)) || (rc = build_sell(H, bs, nullptr, &g->by_src)) || (rc = build_sell(P, bp, &init, &g->by_pdf)) ||
      (rc = to_device(init, &g->init))) {
    tdnnf_den_graph_destroy(g);
    return rc;
  }
  float s = 0.f;  // float sum in index order, as the device kernels assume
  double sd = 0;
  for (int h = 0; h < H; h++) sd += init[h];
  s = (float)sd;
  g->init_sum = s;
  *out = g;
  return TDNNF_OK;
}

void tdnnf_den_graph_destroy(tdnnf_den_graph *g) {
  if (!g) return;
  tdnnf_den_graph::Sell *t[3] = {&g->by_dst, &g->by_src, &g->by_pdf};
  for (auto *x : t) {
    hipFree(x->base);
    hipFree(x->row);
    hipFree(x->arc);
    hipFree(x->arc4);
  }
  hipFree(g->init);
  delete g;
}

int tdnnf_supervision_create(int B, int T, const int *seq_state_begin, const int *seq_arc_begin, const int *state_time,
                             const float *final_logprob, const int *arc_src, const int *arc_dst, const int *arc_pdf,
                             const float *arc_logprob, float weight, tdnnf_supervision **out) {
  TDNNF_REQUIRE(out && B > 0 && T > 0 && seq_state_begin && seq_arc_begin && state_time && final_logprob && arc_src &&
                    arc_dst && arc_pdf && arc_logprob,
                "supervision_create: bad arguments");
  const int NS = seq_state_begin[B], NA = seq_arc_begin[B];
  std::vector<int> fsb((size_t)B * (T + 2), 0);
  int max_states = 0;
  for (int s = 0; s < B; s++) {
    const int s0 = seq_state_begin[s], s1 = seq_state_begin[s + 1];
    max_states = std::max(max_states, s1 - s0);
    TDNNF_REQUIRE(s1 > s0 && state_time[s0] == 0, "supervision_create: sequence %d must start with its time-0 start state", s);
    int st = s0;
    for (int t = 0; t <= T + 1; t++) {
      while (st < s1 && state_time[st] < t) st++;
      fsb[(size_t)s * (T + 2) + t] = st;
    }
    for (int i = s0 + 1; i < s1; i++)
      TDNNF_REQUIRE(state_time[i] >= state_time[i - 1] && state_time[i] <= T, "supervision_create: states must be sorted by time");
    TDNNF_REQUIRE(fsb[(size_t)s * (T + 2) + 1] == s0 + 1, "supervision_create: exactly one time-0 state per sequence");
    for (int a = seq_arc_begin[s]; a < seq_arc_begin[s + 1]; a++)
      TDNNF_REQUIRE(arc_src[a] >= s0 && arc_src[a] < s1 && arc_dst[a] >= s0 && arc_dst[a] < s1 && arc_pdf[a] >= 0 &&
                        state_time[arc_dst[a]] == state_time[arc_src[a]] + 1,
                    "supervision_create: arc %d must advance exactly one frame inside its sequence", a);
  }
  std::vector<int> in_begin(NS + 1, 0), out_begin(NS + 1, 0);
  for (int a = 0; a < NA; a++) {
    in_begin[arc_dst[a] + 1]++;
    out_begin[arc_src[a] + 1]++;
  }
  for (int i = 0; i < NS; i++) {
    in_begin[i + 1] += in_begin[i];
    out_begin[i + 1] += out_begin[i];
  }
  std::vector<int> in_src(NA), in_pdf(NA), out_dst(NA), out_pdf(NA), ipos(in_begin.begin(), in_begin.end() - 1),
      opos(out_begin.begin(), out_begin.end() - 1);
  std::vector<float> in_lp(NA), out_lp(NA);
  for (int a = 0; a < NA; a++) {  // stable: original arc order within each state
    int i = ipos[arc_dst[a]]++, o = opos[arc_src[a]]++;
    in_src[i] = arc_src[a];
    in_pdf[i] = arc_pdf[a];
    in_lp[i] = arc_logprob[a];
    out_dst[o] = arc_dst[a];
    out_pdf[o] = arc_pdf[a];
    out_lp[o] = arc_logprob[a];
  }
  tdnnf_supervision *sp = new tdnnf_supervision();
  memset(sp, 0, sizeof(*sp));
  sp->B = B;
  sp->T = T;
  sp->num_states = NS;
  sp->num_arcs = NA;
  sp->weight = weight;
  sp->max_states_per_seq = max_states;
  std::vector<int> ssb(seq_state_begin, seq_state_begin + B + 1), stime(state_time, state_time + NS);
  std::vector<float> fin(final_logprob, final_logprob + NS);
  int rc;
  if ((rc = to_device(ssb, &sp->seq_state_begin)) || (rc = to_device(stime, &sp->state_time)) ||
      (rc = to_device(fin, &sp->final_logprob)) || (rc = to_device(in_begin, &sp->in_begin)) ||
      (rc = to_device(in_src, &sp->in_src)) || (rc = to_device(in_pdf, &sp->in_pdf)) || (rc = to_device(in_lp, &sp->in_lp)) ||
      (rc = to_device(out_begin, &sp->out_begin)) || (rc = to_device(out_dst, &sp->out_dst)) ||
      (rc = to_device(out_pdf, &sp->out_pdf)) || (rc = to_device(out_lp, &sp->out_lp)) ||
      (rc = to_device(fsb, &sp->frame_state_begin))) {
    tdnnf_supervision_destroy(sp);
    return rc;
  }
  *out = sp;
  return TDNNF_OK;
}

void tdnnf_supervision_destroy(tdnnf_supervision *sp) {
  if (!sp) return;
  void *ptrs[] = {sp->seq_state_begin, sp->state_time, sp->final_logprob, sp->in_begin, sp->in_src, sp->in_pdf,
                  sp->in_lp, sp->out_begin, sp->out_dst, sp->out_pdf, sp->out_lp, sp->frame_state_begin};
  for (void *p : ptrs) hipFree(p);
  delete sp;
}

// workspace layout: [doubles: den_lp[B], num_lp[B], xent[B], l2sum[1]] [alpha] [asum] [gstate] [la, lb]
// (the numerator scratch is sized for up to 4*(T+1) states per sequence; larger graphs are rejected)
int tdnnf_chain_set_denominator_mode(int mode) {
  TDNNF_REQUIRE(mode >= 0 && mode <= 3, "chain_set_denominator_mode: 0 automatic, 1 persistent, 2 wide, 3 persistent with one workgroup per sequence");
  g_den_mode = mode;
  return TDNNF_OK;
}

// tests / diagnostics: how many minibatches the one-workgroup kernels redid behind a multi-workgroup launch that gave up, and whether that
// form is switched off for the process; reset != 0 clears both
int tdnnf_chain_den_mw_status(int *fallbacks, int *disabled, int reset) {
  (void)hipDeviceSynchronize();
  MwDev *md = mw_dev();  // (the current device's)
  if (fallbacks) *fallbacks = md && md->fallbacks ? (int)*(volatile unsigned *)md->fallbacks : 0;
  if (disabled) *disabled = md && md->off ? 1 : 0;
  if (reset && md) {
    if (md->fallbacks) *md->fallbacks = 0;
    md->off = false;
  }
  return TDNNF_OK;
}

size_t tdnnf_chain_workspace_bytes(const tdnnf_den_graph *g, int B, int T) {
  if (!g || B <= 0 || T <= 0) return 0;
  ChainPlan p = chain_plan(g, B, T, B * 4 * (T + 1));
  return sizeof(double) * (3 * (size_t)B + 2) + sizeof(float) * (p.alpha_floats + p.asum_floats + p.gstate_floats + p.la_floats) + 768;
}

}  // extern "C"

namespace tdnnf {
// Bytes at the END of the workspace that only the split persistent form (den_forward beside den_beta, then den_gamma) touches: a caller
// that always passes beside_other_work = true to chain_den may allocate that much less.
size_t chain_split_region_bytes(const tdnnf_den_graph *g, int B, int T) {
  if (!g || B <= 0 || T <= 0) return 0;
  ChainPlan p = chain_plan(g, B, T, B * 4 * (T + 1));
  return p.split ? sizeof(float) * p.gstate_floats : 0;
}
}  // namespace tdnnf

namespace tdnnf {
namespace {
struct ChainBufs {
  ChainPlan p;
  double *den_lp, *num_lp, *xent, *l2sum;
  float *alpha, *asum, *gstate;
  double *la, *lb;  // numerator log alpha / log beta
};
ChainBufs chain_bufs(const tdnnf_den_graph *g, int B, int T, void *ws) {
  ChainBufs b;
  b.p = chain_plan(g, B, T, B * 4 * (T + 1));
  b.den_lp = (double *)ws;
  b.num_lp = b.den_lp + B;
  b.xent = b.num_lp + B;
  b.l2sum = b.xent + B;
  b.alpha = (float *)(((uintptr_t)(b.l2sum + 2) + 127) & ~(uintptr_t)127);  // the wide form gathers 64- / 128-byte runs: keep them in one line
  b.asum = b.alpha + b.p.alpha_floats;
  // the numerator's arrays before the denominator's state region: that one is the workspace's tail, so a caller that never runs the split form
  // (chain_split_region_bytes) can leave it out
  b.la = (double *)(((uintptr_t)(b.asum + b.p.asum_floats) + 7) & ~(uintptr_t)7);
  b.lb = b.la + (b.p.la_floats - 2) / 4;
  b.gstate = (float *)(((uintptr_t)(b.la + b.p.la_floats / 2) + 127) & ~(uintptr_t)127);
  return b;
}
SupDev sup_dev(const tdnnf_supervision *sp) {
  return SupDev{sp->B, sp->T, sp->weight, sp->seq_state_begin, sp->frame_state_begin, sp->final_logprob, sp->in_begin, sp->in_src,
                sp->in_pdf, sp->in_lp, sp->out_begin, sp->out_dst, sp->out_pdf, sp->out_lp};
}
}  // namespace

// A second stream for the backward recursion and the fork / join events: one set per DEVICE and calling thread (created on first
// use on that device; a process that drives several devices, or several host threads, gets a set each -- the entry points are
// still not re-entrant for one thread).  They live as long as the process: HIP tears them down with the context.
// The stream is created only when the caller has none to lend (aux == nullptr: events only): a stream holds a share of one of the
// device's four hardware queues for as long as it exists, used or not, and the trainer passes its own.
int den_aux_stream(hipStream_t *aux, hipEvent_t *ev_fork, hipEvent_t *ev_join) {
  struct Set {
    hipStream_t st = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
  };
  constexpr int kMaxDev = 64;
  static thread_local Set sets[kMaxDev];
  int dev = 0;
  TDNNF_HIP(hipGetDevice(&dev));
  TDNNF_REQUIRE(dev >= 0 && dev < kMaxDev, "chain: device index %d out of range", dev);
  Set &S = sets[dev];
  if (!S.ef) {
    TDNNF_HIP(hipEventCreateWithFlags(&S.ef, hipEventDisableTiming));
    TDNNF_HIP(hipEventCreateWithFlags(&S.ej, hipEventDisableTiming));
  }
  if (aux && !S.st) TDNNF_HIP(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
  if (aux) *aux = S.st;
  *ev_fork = S.ef;
  *ev_join = S.ej;
  return TDNNF_OK;
}

// The three parts of ComputeChainObjfAndDeriv, separately launchable so that the trainer can run the
// denominator (one workgroup per sequence: half the CUs at 128 sequences) on a second stream beside the xent head.
// (1) denominator forward + backward: deriv = -weight * gamma_den (whole matrix overwritten), den log-probs -> workspace
int chain_den(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float leaky, tdnnf_mat *deriv, void *ws,
              hipStream_t s, bool beside_other_work, hipStream_t caller_aux, hipEvent_t ev_recursions, bool *ev_recorded) {
  if (ev_recorded) *ev_recorded = false;
  const int B = sp->B, T = sp->T;
  ChainBufs b = chain_bufs(g, B, T, ws);
  DenDev gd{g->H, g->P, g->by_dst, g->by_src, g->by_pdf, g->init, g->init_sum};
  MatView yv = view(y), dv = view(deriv);
  // algorithmic bytes per (frame, sequence): the arcs once per recursion (8 bytes each forward, twice that backward: two arrays), the
  // output row read by both recursions and the derivative row written; the range covers both streams (fork .. join)
  ProfHbmRange prof(6, (double)B * T * (24.0 * g->A + 12.0 * g->P), s);
  if (b.p.wide) {
    const int Hs = b.p.Hs, P = g->P;
    const WideDims d{B, b.p.NG, b.p.SG};
    const size_t Bw = (size_t)d.NG * d.SG, frame = Bw * Hs, prow = Bw * b.p.wide_blocks;
    const int nsl = kWideSlices;
    auto blocks = [&](const tdnnf_den_graph::Sell &t) { return (t.nslices + nsl - 1) / nsl; };
    const int nb_dst = blocks(g->by_dst), nb_src = blocks(g->by_src);
    float *alphaT = b.alpha, *asum = b.asum, *S = asum + (size_t)(T + 1) * B, *Zd = S + (size_t)(T + 1) * B;
    float *bT = b.gstate, *part = bT + frame * (T + 1), *part2 = part + 2 * prow, *xT = part2 + 2 * prow;
    const dim3 blk(256), tr((P + 63) / 64, (B + 63) / 64, T), ini(grid_for((long long)frame, 256));
    hipStream_t aux_stream;
    hipEvent_t ev_fork, ev_join;
    {
      int rc = den_aux_stream(&aux_stream, &ev_fork, &ev_join);
      if (rc) return rc;
    }
    hipStream_t aux = aux_stream;
    hipLaunchKernelGGL(den_wide_prep_kernel, tr, blk, 0, s, yv, d, P, xT);
    TDNNF_HIP(hipEventRecord(ev_fork, s));
    TDNNF_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
#define WIDE_LAUNCH(kernel, ...)                                      \
  do {                                                                \
    if (d.SG == 16) hipLaunchKernelGGL(kernel<16>, __VA_ARGS__);      \
    else hipLaunchKernelGGL(kernel<32>, __VA_ARGS__);                 \
  } while (0)
    // forward recursion (this stream): launch t sums launch t-1's partial rows itself
    hipLaunchKernelGGL(den_wide_init_kernel, ini, blk, 0, s, gd, d, Hs, false, alphaT, asum);
    for (int t = 1; t <= T; t++)
      WIDE_LAUNCH(den_wide_fwd_kernel, dim3(nb_dst * d.NG), blk, 0, s, gd, d, t, leaky, xT, alphaT, asum, Hs,
                  t > 1 ? part + (size_t)((t - 1) & 1) * prow : (const float *)nullptr, nb_dst, nsl, part + (size_t)(t & 1) * prow);
    WIDE_LAUNCH(den_wide_sum_kernel, dim3(d.NG), blk, 0, s, part + (size_t)(T & 1) * prow, nb_dst, d, asum + (size_t)T * B);
    hipLaunchKernelGGL(den_wide_total_kernel, dim3((B + 255) / 256), blk, 0, s, gd, B, T, leaky, asum, b.den_lp);
    // backward recursion, self-normalised (the other stream)
    hipLaunchKernelGGL(den_wide_init_kernel, ini, blk, 0, aux, gd, d, Hs, true, bT + (size_t)T * frame, S + (size_t)T * B);
    for (int t = T - 1; t >= 0; t--)
      WIDE_LAUNCH(den_wide_beta_kernel, dim3(nb_src * d.NG), blk, 0, aux, gd, d, t, leaky, xT, bT + (size_t)(t + 1) * frame, S, Hs, bT + (size_t)t * frame,
                  t < T - 1 ? part2 + (size_t)((t + 1) & 1) * prow : (const float *)nullptr, nb_src, nsl, part2 + (size_t)(t & 1) * prow);
    WIDE_LAUNCH(den_wide_sum_kernel, dim3(d.NG), blk, 0, aux, part2, nb_src, d, S);
    TDNNF_HIP(hipEventRecord(ev_join, aux));
    TDNNF_HIP(hipStreamWaitEvent(s, ev_join, 0));
    // occupancies of every frame at once
    WIDE_LAUNCH(den_wide_dot_kernel, dim3(T, d.NG), blk, 0, s, gd, d, leaky, alphaT, asum, Hs, bT, Zd);
    WIDE_LAUNCH(den_wide_gamma_kernel, dim3(g->by_pdf.nslices * d.NG, T), blk, 0, s, gd, d, leaky, xT, alphaT, asum, Hs, bT, S, Zd, -sp->weight);
#undef WIDE_LAUNCH
    hipLaunchKernelGGL(den_wide_unprep_kernel, tr, blk, 0, s, xT, d, P, dv);
    TDNNF_LAUNCH_CHECK();
    return TDNNF_OK;
  }
  if (b.p.split && !beside_other_work) {
    const int P4 = (g->P + 3) & ~3, H4 = (g->H + 3) & ~3;
    const size_t lds_beta = sizeof(float) * (P4 + H4), lds_gamma = sizeof(float) * (P4 + 2 * H4);
    float *b_all = b.gstate, *S_all = b_all + (size_t)B * (T + 1) * b.p.Hs;
    hipEvent_t ev_fork, ev_join;
    hipStream_t aux = caller_aux;  // a stream the caller has idle (beyond four streams in flight they share hardware queues)
    int rc = den_aux_stream(caller_aux ? nullptr : &aux, &ev_fork, &ev_join);
    if (rc) return rc;
    TDNNF_HIP(hipFuncSetAttribute((const void *)den_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b.p.lds_fwd));
    TDNNF_HIP(hipFuncSetAttribute((const void *)den_beta_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_beta));
    TDNNF_HIP(hipFuncSetAttribute((const void *)den_gamma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gamma));
    TDNNF_HIP(hipEventRecord(ev_fork, s));
    TDNNF_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
    int G = mw_groups(g, B, T);
    MwDev *md = mw_dev();
    if (!md) G = 0;
    if (G > 0 && !md->fallbacks) {  // (first use on this device: the host-visible fallback counter)
      if (hipHostMalloc((void **)&md->fallbacks, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        md->fallbacks = nullptr;
        G = 0;
      } else {
        *md->fallbacks = 0;
      }
    }
    if (G > 0 && *(volatile unsigned *)md->fallbacks != 0) {
      fprintf(stderr, "tdnnf: the multi-workgroup denominator recursion timed out (its workgroups were not co-resident); the one-workgroup kernels redid the "
                      "minibatch and are used from here on (this device)\n");
      md->off = true;
      G = 0;
    }
    const size_t stage_b = G > 0 ? 256 * (size_t)((std::max(g->by_dst.nslices, g->by_src.nslices) + G - 1) / G) : 0;
    const size_t lds_f = G > 0 ? sizeof(float) * (P4 + H4) + 8 * (size_t)g->by_dst.mw_max_arcs[G] + stage_b : 0;
    const size_t lds_b = G > 0 ? sizeof(float) * (P4 + H4) + 8 * (size_t)g->by_src.mw_max_arcs[G] + stage_b : 0;
    if (G > 0) {
      // Every workgroup of a recursion polls for its G - 1 partners: the two launches must be resident together.  HIP gives no such guarantee
      // (a CU mask, another process), so ask the occupancy calculator with the actual LDS sizes, and keep the bounded poll as the last resort.
      TDNNF_HIP(hipFuncSetAttribute((const void *)den_mw_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      TDNNF_HIP(hipFuncSetAttribute((const void *)den_mw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
      int occ_f = 0, occ_b = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_f, den_mw_kernel<0>, kDenThreads, lds_f) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, den_mw_kernel<1>, kDenThreads, lds_b) != hipSuccess || occ_f < 1 || occ_b < 1 ||
          md->cus <= 0 || 2 * B * G > md->cus) {
        (void)hipGetLastError();
        G = 0;
      }
    }
    if (G > 0) {  // several workgroups per sequence
      float *mw = S_all + (size_t)B * (T + 1) + 32;
      MwCtl ctl;
      ctl.NSp = (int)mw_slots(g);
      ctl.ctr = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(mw) + 15) & ~(uintptr_t)15);
      ctl.abort_flag = reinterpret_cast<unsigned *>(ctl.ctr + 2 * B);
      ctl.xf = reinterpret_cast<float *>(ctl.ctr + 2 * B + 2);
      ctl.xb = ctl.xf + (size_t)B * 2 * ctl.NSp;
      TDNNF_HIP(hipMemsetAsync(ctl.ctr, 0, sizeof(unsigned long long) * (2 * B + 2), s));  // (in front of the fork: both streams see it)
      if (options().den_mw_test_abort) TDNNF_HIP(hipMemsetAsync(ctl.abort_flag, 1, 1, s));  // (tests: the recursions give up at their first poll)
      TDNNF_HIP(hipEventRecord(ev_fork, s));
      TDNNF_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
      hipLaunchKernelGGL(den_mw_kernel<0>, dim3(B * G), dim3(kDenThreads), lds_f, s, gd, ctl, G, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp);
      hipLaunchKernelGGL(den_mw_kernel<1>, dim3(B * G), dim3(kDenThreads), lds_b, aux, gd, ctl, G, yv, B, T, leaky, b_all, S_all, b.p.Hs, (double *)nullptr);
      // behind them, the one-workgroup kernels: they return at once unless the abort word is set (then they redo the recursion, so that
      // the occupancy pass never reads half-written vectors and the minibatch is not lost)
      hipLaunchKernelGGL(den_forward_kernel<true>, dim3(B), dim3(kDenThreads), b.p.lds_fwd, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate,
                         (const unsigned *)ctl.abort_flag, 0);
      hipLaunchKernelGGL(den_beta_kernel<false>, dim3(B), dim3(kDenThreads), lds_beta, aux, gd, yv, B, T, leaky, b_all, S_all, b.p.Hs, (const unsigned *)ctl.abort_flag, 0);
      hipLaunchKernelGGL(den_mw_check_kernel, dim3(1), dim3(64), 0, s, (const unsigned *)ctl.abort_flag, md->fallbacks);
    } else {
      // FAST kernels where a thread's rows and states fit its registers (den_forward_kernel); and when the two launches hold every CU (one
      // 1024-thread workgroup each), the LDS nothing else can use keeps the tables' widest slices
      const bool fast = g->by_dst.nslices * 64 <= kDenFastSlots * kDenThreads && g->by_src.nslices * 64 <= kDenFastSlots * kDenThreads && g->H <= kDenFastStates * kDenThreads;
      const size_t lf = b.p.lds_fwd + (fast ? sizeof(float) * b.p.Hs : 0), lb = lds_beta + (fast ? sizeof(float) * H4 : 0);
      int res_f = 0, res_b = 0;
      if (md && md->cus > 0 && 2 * B >= md->cus) {
        const size_t budget = 150 * 1024;
        if (lf < budget) res_f = (int)std::min<long long>(g->by_dst.entries, (long long)((budget - lf) / 8)) & ~63;
        if (lb < budget) res_b = (int)std::min<long long>(g->by_src.entries, (long long)((budget - lb) / 8)) & ~63;
      }
      (void)hipGetLastError();
      const size_t lds_f2 = lf + 8 * (size_t)res_f, lds_b2 = lb + 8 * (size_t)res_b;
      if (fast) {
        TDNNF_HIP(hipFuncSetAttribute((const void *)den_forward_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f2));
        TDNNF_HIP(hipFuncSetAttribute((const void *)den_beta_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b2));
        hipLaunchKernelGGL((den_forward_kernel<true, true>), dim3(B), dim3(kDenThreads), lds_f2, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate,
                           (const unsigned *)nullptr, res_f);
        hipLaunchKernelGGL(den_beta_kernel<true>, dim3(B), dim3(kDenThreads), lds_b2, aux, gd, yv, B, T, leaky, b_all, S_all, b.p.Hs, (const unsigned *)nullptr, res_b);
      } else {
        TDNNF_HIP(hipFuncSetAttribute((const void *)den_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f2));
        TDNNF_HIP(hipFuncSetAttribute((const void *)den_beta_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b2));
        hipLaunchKernelGGL(den_forward_kernel<true>, dim3(B), dim3(kDenThreads), lds_f2, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate,
                           (const unsigned *)nullptr, res_f);
        hipLaunchKernelGGL(den_beta_kernel<false>, dim3(B), dim3(kDenThreads), lds_b2, aux, gd, yv, B, T, leaky, b_all, S_all, b.p.Hs, (const unsigned *)nullptr, res_b);
      }
    }
    TDNNF_HIP(hipEventRecord(ev_join, aux));
    TDNNF_HIP(hipStreamWaitEvent(s, ev_join, 0));
    if (ev_recursions) {  // both recursions are done here, the occupancies (a launch that fills the chip) come next
      TDNNF_HIP(hipEventRecord(ev_recursions, s));
      if (ev_recorded) *ev_recorded = true;
    }
    hipLaunchKernelGGL(den_gamma_kernel, dim3(T, B), dim3(kGammaThreads), lds_gamma, s, gd, yv, B, T, leaky, b.alpha, b_all, S_all, b.p.Hs, -sp->weight, dv);
  } else if (b.p.lds_state) {
    TDNNF_HIP(hipFuncSetAttribute((const void *)den_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b.p.lds_bwd));
    if (g->by_dst.nslices * 64 <= kDenFastSlots * kDenThreads && g->H <= kDenFastStates * kDenThreads) {
      const size_t lf = b.p.lds_fwd + sizeof(float) * b.p.Hs;
      TDNNF_HIP(hipFuncSetAttribute((const void *)den_forward_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf));
      hipLaunchKernelGGL((den_forward_kernel<true, true>), dim3(B), dim3(kDenThreads), lf, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate, (const unsigned *)nullptr, 0);
    } else {
      TDNNF_HIP(hipFuncSetAttribute((const void *)den_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b.p.lds_fwd));
      hipLaunchKernelGGL(den_forward_kernel<true>, dim3(B), dim3(kDenThreads), b.p.lds_fwd, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate, (const unsigned *)nullptr, 0);
    }
    hipLaunchKernelGGL(den_backward_kernel<true>, dim3(B), dim3(kDenThreads), b.p.lds_bwd, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, -sp->weight, dv, b.gstate);
  } else {
    hipLaunchKernelGGL(den_forward_kernel<false>, dim3(B), dim3(kDenThreads), b.p.lds_fwd, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, b.den_lp, b.gstate, (const unsigned *)nullptr, 0);
    hipLaunchKernelGGL(den_backward_kernel<false>, dim3(B), dim3(kDenThreads), b.p.lds_bwd, s, gd, yv, B, T, leaky, b.alpha, b.asum, b.p.Hs, -sp->weight, dv, b.gstate);
  }
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
float chain_supervision_weight(const tdnnf_supervision *sp) { return sp->weight; }
// (2) numerator recursion; xent_deriv = xent_regularize * gamma_num, xent objective -> workspace.  Does not touch deriv.
// In two launchable halves: the recursion needs the chain output only (the trainer starts it beside the denominator, under the
// xent head's forward pass: 1.35 ms of one wave per sequence walking 2 x 500 dependent frames that the step otherwise waited for),
// the xent posteriors need the recursion and the xent head's log-softmax.
int chain_num_recursion(const tdnnf_supervision *sp, const tdnnf_den_graph *g, const tdnnf_mat *y, void *ws, hipStream_t s) {
  const int B = sp->B, T = sp->T;
  ChainBufs b = chain_bufs(g, B, T, ws);
  const MatView none{nullptr, 0, 0, 0};
  hipLaunchKernelGGL(numerator_kernel, dim3(B), dim3(64), 0, s, sup_dev(sp), view(y), none, b.la, b.lb, b.num_lp, b.xent, none, none, 0.f, 1);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int chain_num_xent(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, const tdnnf_mat *xent_output, float xent_regularize,
                   tdnnf_mat *xent_deriv, void *ws, hipStream_t s, bool xent_deriv_initialised) {
  const int B = sp->B, T = sp->T;
  ChainBufs b = chain_bufs(g, B, T, ws);
  MatView yv = view(y);
  MatView xdv = xent_deriv ? view(xent_deriv) : MatView{nullptr, 0, 0, 0};
  MatView xov = xent_output ? view(xent_output) : MatView{nullptr, 0, 0, 0};
  if (xent_deriv && !xent_deriv_initialised) {  // (initialised: the posteriors are added onto what the caller put there)
    if (xdv.stride == xdv.cols) TDNNF_HIP(hipMemsetAsync(xdv.data, 0, sizeof(float) * (size_t)xdv.rows * xdv.cols, s));  // one contiguous fill
    else hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for((long long)xdv.rows * xdv.cols, 256)), dim3(256), 0, s, xdv);
  }
  hipLaunchKernelGGL(numerator_kernel, dim3(B), dim3(64), 0, s, sup_dev(sp), yv, xov, b.la, b.lb, b.num_lp, b.xent, MatView{nullptr, 0, 0, 0},
                     xdv, xent_regularize, 4);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
int chain_num(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, const tdnnf_mat *xent_output,
              float xent_regularize, tdnnf_mat *xent_deriv, void *ws, hipStream_t s, bool xent_deriv_initialised) {
  int rc = chain_num_recursion(sp, g, y, ws, s);
  if (rc) return rc;
  return chain_num_xent(g, sp, y, xent_output, xent_regularize, xent_deriv, ws, s, xent_deriv_initialised);
}
// (3) after (1) and (2): deriv += weight * gamma_num; objective, l2 term, failure handling
int chain_finish(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y, float l2_regularize, double *results,
                 tdnnf_mat *deriv, tdnnf_mat *xent_deriv, void *ws, hipStream_t s) {
  const int B = sp->B, T = sp->T;
  ChainBufs b = chain_bufs(g, B, T, ws);
  MatView yv = view(y), dv = view(deriv);
  MatView xdv = xent_deriv ? view(xent_deriv) : MatView{nullptr, 0, 0, 0};
  hipLaunchKernelGGL(numerator_kernel, dim3(B), dim3(64), 0, s, sup_dev(sp), yv, MatView{nullptr, 0, 0, 0}, b.la, b.lb, b.num_lp, b.xent, dv,
                     MatView{nullptr, 0, 0, 0}, 0.f, 2);
  if (l2_regularize != 0.f) {
    TDNNF_HIP(hipMemsetAsync(b.l2sum, 0, sizeof(double), s));
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((long long)yv.rows * yv.cols, 256, 1024)), dim3(256), 0, s, yv, b.l2sum);
  }
  hipLaunchKernelGGL(chain_finalize_kernel, dim3(1), dim3(64), 0, s, b.num_lp, b.den_lp, b.xent, l2_regularize != 0.f ? b.l2sum : nullptr,
                     B, T, sp->weight, l2_regularize, results);
  hipLaunchKernelGGL(chain_guard_kernel, dim3(grid_for((long long)dv.rows * dv.cols, 256)), dim3(256), 0, s, results, yv,
                     sp->weight * l2_regularize, dv, xdv);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}
}  // namespace tdnnf

extern "C" {

int tdnnf_chain_objf_and_deriv(const tdnnf_den_graph *g, const tdnnf_supervision *sp, const tdnnf_mat *y,
                               const tdnnf_mat *xent_output, float leaky, float l2_regularize, float xent_regularize,
                               double *results, tdnnf_mat *deriv, tdnnf_mat *xent_deriv, void *ws, size_t ws_bytes,
                               tdnnf_stream stream) {
  TDNNF_REQUIRE(g && sp && mat_ok(y) && mat_ok(deriv) && results, "chain_objf_and_deriv: bad arguments");
  const int B = sp->B, T = sp->T;
  TDNNF_REQUIRE(y->rows == B * T && y->cols == g->P && same_dim(y, deriv), "chain_objf_and_deriv: nnet_output must be (B*T) x num_pdfs, t-major");
  TDNNF_REQUIRE(!xent_deriv || (mat_ok(xent_deriv) && same_dim(y, xent_deriv)), "chain_objf_and_deriv: bad xent_deriv");
  TDNNF_REQUIRE(!xent_output || (mat_ok(xent_output) && same_dim(y, xent_output)), "chain_objf_and_deriv: bad xent_output");
  TDNNF_REQUIRE(sp->num_states <= B * 4 * (T + 1), "chain_objf_and_deriv: supervision has more than 4*(T+1) states per sequence on average");
  TDNNF_REQUIRE(ws && ws_bytes >= tdnnf_chain_workspace_bytes(g, B, T), "chain_objf_and_deriv: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = chain_den(g, sp, y, leaky, deriv, ws, s))) return rc;
  if ((rc = chain_num(g, sp, y, xent_output, xent_regularize, xent_deriv, ws, s))) return rc;
  return chain_finish(g, sp, y, l2_regularize, results, deriv, xent_deriv, ws, s);
}

}  // extern "C"
