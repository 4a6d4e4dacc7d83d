// gemm_f32.hip -- exact-f32 MFMA GEMM kernels for gfx950 (MI355X, CDNA4).
//
// These replace the per-tap cuBLAS SGEMMs the reference issues from
// TdnnDARTSV3Component::Propagate / Backprop / UpdateSimple
// (/root/reference/src/nnet3/nnet-tdnn-component.cc:302-324, :378-411, :452) and the
// AffineComponent / LinearComponent GEMMs (nnet-simple-component.cc:1235-1279).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf
// chain, so parity with the reference's fp32 BaseFloat path is limited only by
// summation order).  One launch covers ALL taps of a layer:
//   rows_gemm : C[m][n] (+)= sum_taps c_i * A_i[m][:] . B_i[:][n]   (fwd and bwd-data,
//               bwd-data in gather form so overlapping taps need no atomics)
//   wgrad     : G[o][i*Di+d] += lr * c_i * sum_rows dY[r][o] X_i[r][d] (split over rows,
//               deterministic slab reduction)
// Tiles: 4 waves / 256 threads, each wave owns TM x TN blocks of 32x32 accumulators;
// A/B tiles are staged global -> registers -> LDS (double buffered, one barrier per
// K-step); LDS rows are padded by 4 floats so the ds_read_b128 fragment reads are
// bank-conflict free (stride 36 / 20 dwords).
#include "gemm_f32.h"
#include "planes_gemm.h"
#include "gemm_ring.h"

#include "common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tdnnf_hip.h"

namespace tdnnf {

// ---- optional event timing of every GEMM launch (tdnnf_profile_*)
struct ProfClass {
  const char *name;
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  double flops = 0;
  double bytes = 0;  // algorithmic HBM bytes: every operand element touched once
};
constexpr int kProfClasses = 8;
static ProfClass g_prof[kProfClasses] = {{"rows_gemm_f32_128x128"}, {"rows_gemm_f32_128x160"}, {"wgrad_f32"}, {"ng_skinny_gemm_f32"},
                                         {"bn_apply_bypass"}, {"bn_relu_bwd"}, {"denominator"}, {"planes_split"}};
static bool g_prof_on = false;
static int g_prof_override = -1;
static double g_prof_flops_scale = 1.0;
static double g_prof_next_bytes = 0;  // algorithmic bytes of the whole GEMM the next ProfScope(s) belong to (set by rows_gemm / wgrad)
static double g_prof_next_flops = 0;
ProfFlopsScale::ProfFlopsScale(double f) : prev(g_prof_flops_scale) { g_prof_flops_scale = f; }
ProfFlopsScale::~ProfFlopsScale() { g_prof_flops_scale = prev; }
namespace {
extern float *g_scratch_override;
extern size_t g_scratch_override_bytes;
}  // namespace
SplitKScratchOverride::SplitKScratchOverride(float *buf, size_t bytes) : prev_buf(g_scratch_override), prev_bytes(g_scratch_override_bytes) {
  g_scratch_override = buf;
  g_scratch_override_bytes = bytes;
}
static int g_gemm_prec = 0;
GemmPrecisionScope::GemmPrecisionScope(int prec) : prev(g_gemm_prec) { g_gemm_prec = prec; }
GemmPrecisionScope::~GemmPrecisionScope() { g_gemm_prec = prev; }
static const float *g_tw_w = nullptr, *g_tw_wt = nullptr;
static long long g_tw_n = 0;
TransposedWeightsScope::TransposedWeightsScope(const float *w_base, const float *wt_base, long long n) : prev_w(g_tw_w), prev_wt(g_tw_wt), prev_n(g_tw_n) {
  g_tw_w = w_base;
  g_tw_wt = wt_base;
  g_tw_n = n;
}
TransposedWeightsScope::~TransposedWeightsScope() {
  g_tw_w = prev_w;
  g_tw_wt = prev_wt;
  g_tw_n = prev_n;
}
const float *transposed_weights(const float *W) {
  if (!g_tw_w || !g_tw_wt || W < g_tw_w || W >= g_tw_w + g_tw_n) return nullptr;
  return g_tw_wt + (W - g_tw_w);
}
SplitKScratchOverride::~SplitKScratchOverride() {
  g_scratch_override = prev_buf;
  g_scratch_override_bytes = prev_bytes;
}
ProfClassOverride::ProfClassOverride(int cls) : prev(g_prof_override) { g_prof_override = cls; }
ProfClassOverride::~ProfClassOverride() { g_prof_override = prev; }
constexpr size_t kProfMaxLaunches = 1 << 15;

struct ProfScope {
  ProfClass *c = nullptr;
  hipStream_t s;
  ProfScope(int cls, double flops, hipStream_t stream) : s(stream) {
    if (!g_prof_on) return;
    ProfClass &p = g_prof[g_prof_override >= 0 ? g_prof_override : cls];
    if (p.used + 2 > p.ev.size()) return;
    c = &p;
    p.flops += flops * g_prof_flops_scale;
    // a launch that covers a fraction of the GEMM's rows (main + split-K tail) gets that fraction of its bytes
    if (g_prof_next_flops > 0) p.bytes += g_prof_next_bytes * (flops / g_prof_next_flops);
    hipEventRecord(p.ev[p.used], s);
  }
  ~ProfScope() {
    if (!c) return;
    hipEventRecord(c->ev[c->used + 1], s);
    c->used += 2;
  }
};

// event pair around the launches of an HBM-bound pass (classes 4..7) with its algorithmic bytes
ProfHbmRange::ProfHbmRange(int cls, double bytes, hipStream_t stream) : c(nullptr), s(stream) {
  if (!g_prof_on || cls < 4 || cls >= kProfClasses) return;
  ProfClass &p = g_prof[cls];
  if (p.used + 2 > p.ev.size()) return;
  c = &p;
  p.bytes += bytes;
  hipEventRecord(p.ev[p.used], s);
}
ProfHbmRange::~ProfHbmRange() {
  if (!c) return;
  ProfClass *p = static_cast<ProfClass *>(c);
  hipEventRecord(p->ev[p->used + 1], s);
  p->used += 2;
}

ProfGemmRange::ProfGemmRange(int cls, double flops, double bytes, hipStream_t stream) : c(nullptr), s(stream) {
  if (!g_prof_on || cls < 0 || cls >= 4) return;
  ProfClass &p = g_prof[cls];
  if (p.used + 2 > p.ev.size()) return;
  c = &p;
  p.flops += flops;
  p.bytes += bytes;
  hipEventRecord(p.ev[p.used], s);
}
ProfGemmRange::~ProfGemmRange() {
  if (!c) return;
  ProfClass *p = static_cast<ProfClass *>(c);
  hipEventRecord(p->ev[p->used + 1], s);
  p->used += 2;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// 16 zero bytes: predicated-off float4 loads of the fast path read here instead of branching
__device__ float4 g_zero4 = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float4 ld4(const float *p, bool v0, bool v1, bool v2, bool v3, bool vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && v3) {  // whole float4 in range (validity is monotone in the element index)
    r = *reinterpret_cast<const float4 *>(p);
  } else {
    if (v0) r.x = p[0];
    if (v1) r.y = p[1];
    if (v2) r.z = p[2];
    if (v3) r.w = p[3];
  }
  return r;
}

// ------------------------------------------------------------------------ rows_gemm
// TAG only gives the launches of the natural-gradient statistics (ProfClassOverride(3)) their own kernel symbol, so
// that per-kernel profiler summaries keep them apart from the TDNN-F GEMMs; the code is identical.
// (the body of a block: block `bx` of the `gx` blocks that work on `p` -- the whole grid of a plain launch, one task's share of a grouped one)
template <int WM, int WN, int TM, int TN, int BK, bool B_KC, int VEC>
__device__ __forceinline__ void rows_gemm_block(const RowsGemmArgs &p, int ntm, int ntn, int bx, int gx) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDAS = BK + 4;
  constexpr int LDBS = B_KC ? BK + 4 : BN + 4;
  constexpr int A_TILE = BM * LDAS;
  constexpr int B_TILE = B_KC ? BN * LDBS : BK * LDBS;
  constexpr int A_F4 = (BM * BK / 4 + 255) / 256;
  constexpr int B_F4 = (BN * BK / 4 + 255) / 256;
  constexpr int KF4 = BK / 4;  // float4 per k-row of a k-contiguous tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;               // [2][A_TILE]
  float *Bs = smem + 2 * A_TILE;  // [2][B_TILE]

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of logical tile ids; within it tile_n varies fastest so the blocks
  // that re-read the same A rows (and the taps' neighbouring rows) hit the same L2.
  const int nblk = ntm * ntn;
  int bid = bx;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int sp = 0;
  if (p.ksplit > 1) {  // split-K launch: consecutive block ids share a tile
    sp = bx % p.ksplit;
    bid = bx / p.ksplit;
  }
  const long long k_begin = (long long)sp * p.kchunk, k_end = p.ksplit > 1 ? k_begin + p.kchunk : (1LL << 60);
  const int tile_m = bid / ntn, tile_n = bid % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // ---- K iterator over (segment, chunk), skipping zero-coefficient segments
  // (a split-K block only visits the part of each segment inside its [k_begin, k_end) slice)
  // (alt_seg_order: odd row tiles visit the segments in reverse.  The taps of a TDNN-F layer are row shifts of ONE matrix by a tile's worth
  // of rows, so row block i is the tap-0 operand of tile i and the tap-(-1) operand of tile i + 1: with every tile going tap by tap in the
  // same order the two reads are half a launch apart and both come from HBM -- PMC traffic 2.07 x the algorithmic bytes for the 160-wide
  // class in rounds 2-4; in alternating order both consumers of a row block read it in the same phase, at the same K step, on one XCD)
  const bool rev_seg = p.alt_seg_order && (tile_m & 1);
  int seg = -1, sgi = 0, kc = 0, klen = 0;  // seg: position in this tile's visiting order; sgi: the segment's index in p.seg
  long long seg_kstart = 0, seg_knext = 0;
  float cf = 1.f;
  auto next_seg = [&]() {
    for (++seg; seg < p.nseg; ++seg) {
      sgi = rev_seg ? p.nseg - 1 - seg : seg;
      seg_kstart = seg_knext;
      seg_knext += p.seg[sgi].klen;
      cf = p.coef ? p.coef[sgi] : 1.f;
      const long long lo = k_begin > seg_kstart ? k_begin - seg_kstart : 0;
      const long long hi = k_end < seg_knext ? k_end - seg_kstart : p.seg[sgi].klen;
      if (cf != 0.f && hi > lo) {
        kc = (int)lo;
        klen = (int)hi;
        return;
      }
    }
    kc = 0;
    klen = 0;
  };
  next_seg();

  float4 ra[A_F4], rb[B_F4];
  // The tap coefficient (and the sum of squares of p.sumsq) is applied when a staged tile goes to LDS, not when it is
  // loaded: anything that touches ra/rb right after the loads would wait for them in front of the MFMAs they are
  // supposed to overlap with.
  float cf_tile = 1.f;  // coefficient of the segment the tile in ra/rb was loaded from
  float ssq = 0.f;      // p.sumsq: running sum of (coef * a)^2 over everything this thread stages
  auto add_ssq = [&]() {
    // (plain v_fmac_f32 from inline assembly: written as a sum of products the compiler packs it into v_pk_mul_f32 / v_pk_add_f32, which
    // beside MFMAs cost several times their plain forms -- 622 against 482 us for the pass with and without this by-product)
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      asm volatile("v_fmac_f32 %0, %2, %2\n v_fmac_f32 %1, %3, %3\n v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %5, %5"
                   : "+v"(q0), "+v"(q1)
                   : "v"(ra[j].x), "v"(ra[j].y), "v"(ra[j].z), "v"(ra[j].w));
    }
    ssq += cf_tile * cf_tile * (q0 + q1);
  };
  // Per-segment, per-thread source pointers for the fast path (full K-step inside the segment, float4 loads):
  // rows/columns that are out of range read 16 zero bytes instead of branching.
  const float *aptr[A_F4], *bptr[B_F4];
  int astep[A_F4], bstep[B_F4];
  int ptr_seg = -1;
  auto setup_ptrs = [&]() {
    const GemmSeg sg = p.seg[sgi];
    const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j, row = idx / KF4, m = m0 + row;
      const bool rv = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && m < p.M && m >= sg.m_lo && m < sg.m_hi;
      aptr[j] = rv ? p.A + sg.a_off + (long long)m * p.lda + (idx % KF4) * 4 : zero;
      astep[j] = rv ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      bool rv;
      if (B_KC) {
        const int n = n0 + idx / KF4;
        rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n < p.N;
        bptr[j] = rv ? p.B + sg.b_off + (long long)n * p.ldb + (idx % KF4) * 4 : zero;
        bstep[j] = rv ? 1 : 0;
      } else {
        const int kr = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
        rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n + 3 < p.N;
        bptr[j] = rv ? p.B + sg.b_off + (long long)kr * p.ldb + n : zero;
        bstep[j] = rv ? (int)p.ldb : 0;
      }
    }
    ptr_seg = seg;
  };
  // !B_KC: a ragged last column group (n + 3 >= N) needs the general path for the whole launch
  const bool fast_ok = VEC == 4 && (B_KC || p.N % 4 == 0);
  auto load_tile = [&]() {  // global -> registers for chunk (seg, kc)
    if (fast_ok && kc + BK <= klen) {
      if (ptr_seg != seg) setup_ptrs();
#pragma unroll
      for (int j = 0; j < A_F4; j++) ra[j] = *reinterpret_cast<const float4 *>(aptr[j] + (long long)kc * astep[j]);
#pragma unroll
      for (int j = 0; j < B_F4; j++) rb[j] = *reinterpret_cast<const float4 *>(bptr[j] + (long long)kc * bstep[j]);
      cf_tile = cf;
      return;
    }
    const GemmSeg sg = p.seg[sgi];
    const float *Ab = p.A + sg.a_off;
    const float *Bb = p.B + sg.b_off;
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      const int row = idx / KF4, k = kc + (idx % KF4) * 4;
      const int m = m0 + row;
      const bool rv = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && m < p.M && m >= sg.m_lo && m < sg.m_hi;
      const float *ptr = Ab + (long long)m * p.lda + k;
      ra[j] = ld4(ptr, rv && k < klen, rv && k + 1 < klen, rv && k + 2 < klen, rv && k + 3 < klen, VEC == 4);
    }
    if (B_KC) {
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        const int idx = t + 256 * j;
        const int row = idx / KF4, k = kc + (idx % KF4) * 4;
        const int n = n0 + row;
        const bool rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n < p.N;
        const float *ptr = Bb + (long long)n * p.ldb + k;
        rb[j] = ld4(ptr, rv && k < klen, rv && k + 1 < klen, rv && k + 2 < klen, rv && k + 3 < klen, VEC == 4);
      }
    } else {
      constexpr int NF4 = BN / 4;
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        const int idx = t + 256 * j;
        const int kr = idx / NF4, n = n0 + (idx % NF4) * 4;
        const bool rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && kc + kr < klen;
        const float *ptr = Bb + (long long)(kc + kr) * p.ldb + n;
        rb[j] = ld4(ptr, rv && n < p.N, rv && n + 1 < p.N, rv && n + 2 < p.N, rv && n + 3 < p.N, VEC == 4);
      }
    }
    cf_tile = cf;
  };
  auto store_tile = [&](int buf) {  // registers -> LDS
    float *as = As + buf * A_TILE, *bs = Bs + buf * B_TILE;
    if (p.sumsq) add_ssq();
    if (p.coef) {
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        rb[j].x *= cf_tile; rb[j].y *= cf_tile; rb[j].z *= cf_tile; rb[j].w *= cf_tile;
      }
    }
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      if (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4)
        *reinterpret_cast<float4 *>(as + (idx / KF4) * LDAS + (idx % KF4) * 4) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      if (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) {
        if (B_KC)
          *reinterpret_cast<float4 *>(bs + (idx / KF4) * LDBS + (idx % KF4) * 4) = rb[j];
        else
          *reinterpret_cast<float4 *>(bs + (idx / (BN / 4)) * LDBS + (idx % (BN / 4)) * 4) = rb[j];
      }
    }
  };
  auto compute = [&](int buf) {
    const float *as = As + buf * A_TILE + (wm * TM * 32 + li) * LDAS + lh * 4;
    const float *bs = B_KC ? Bs + buf * B_TILE + (wn * TN * 32 + li) * LDBS + lh * 4
                           : Bs + buf * B_TILE + (lh * 4) * LDBS + wn * TN * 32 + li;
#pragma unroll
    for (int kg = 0; kg < BK / 8; kg++) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LDAS + kg * 8);
#pragma unroll
      for (int i = 0; i < TN; i++) {
        if (B_KC) {
          b[i] = *reinterpret_cast<const float4 *>(bs + i * 32 * LDBS + kg * 8);
        } else {
          const float *q = bs + (kg * 8) * LDBS + i * 32;
          b[i] = make_float4(q[0], q[LDBS], q[2 * LDBS], q[3 * LDBS]);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  if (seg < p.nseg) {
    load_tile();
    store_tile(0);
    __syncthreads();
    int buf = 0;
    while (true) {
      kc += BK;
      if (kc >= klen) next_seg();
      const bool more = seg < p.nseg;
      if (more) load_tile();  // in flight while the MFMAs run
      compute(buf);
      if (!more) break;
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The accumulators go through LDS (reusing the staging buffers) so that C is read/written as whole
  // 16-byte-per-lane row segments instead of 64 four-byte accesses per lane.
  constexpr int LDCS = BN + 4;
  constexpr int SMEM_FLOATS = 2 * (A_TILE + B_TILE);
  constexpr int HALF = (BM * LDCS <= SMEM_FLOATS) ? BM : ((BM / 2) * LDCS <= SMEM_FLOATS ? BM / 2 : BM / 4);
  static_assert(HALF * LDCS <= SMEM_FLOATS, "epilogue tile does not fit the staging LDS");
  static_assert(HALF % (TM * 32) == 0, "a wave's rows must not straddle epilogue passes");
  float *Cs = smem;
  const bool cvec = p.c_vec != 0;
  __syncthreads();
  if (p.sumsq) {  // block total through LDS (the staging buffers are free now)
    double v = ssq;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    double *red = reinterpret_cast<double *>(smem);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (t == 0) p.sumsq[p.ksplit > 1 ? bx : bid] = (red[0] + red[1]) + (red[2] + red[3]);  // (no K split: entry = row tile)
    if (bx == 0)  // entries no block owns (the array is sized for a split-K launch)
      for (int i = gx + t; i < p.sumsq_cap; i += 256) p.sumsq[i] = 0.0;
    __syncthreads();
  }
  constexpr bool kColStats = 256 % (BN / 4) == 0 && (HALF * (BN / 4)) % 256 == 0;  // a thread keeps one float4 column group
  const bool colstats = kColStats && p.colstats && p.ksplit <= 1;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < BM / HALF; pass++) {
    if ((wm * TM * 32) / HALF == pass) {
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh - pass * HALF;
            Cs[row * LDCS + (wn * TN + j) * 32 + li] = acc[i][j][r];
          }
    }
    // Interior tiles (the whole tile inside M x N, 16-byte accesses): the values this pass adds to the accumulators -- the old C of an
    // accumulating launch, the fused addend -- are requested for several row segments at a time (eight for the 128-wide tile) before anything waits for them, and the
    // stores are not waited for either.  (The general loop below asks for one segment, waits -- also for the previous segment's store,
    // which shares the counter --, stores, and so on: sixteen trips to memory in a row per tile, most of what a K = 320 tile spent
    // outside its MFMAs when it carried an addend.)
    constexpr int kSeg = HALF * (BN / 4) / 256;  // row segments per thread and pass
    constexpr int kGrp = kSeg % 8 == 0 ? 8 : (kSeg % 5 == 0 ? 5 : (kSeg % 4 == 0 ? 4 : (kSeg % 2 == 0 ? 2 : 1)));  // requested together
    constexpr bool kFastShape = (HALF * (BN / 4)) % 256 == 0;
    const bool fast_tile = kFastShape && cvec && p.ksplit <= 1 && !p.serial_epilogue && m0 + BM <= p.M && n0 + BN <= p.N;
    if (fast_tile) {
      __syncthreads();
#pragma unroll
      for (int g = 0; g < kSeg; g += kGrp) {
        // bias or old C; the fused addend.  Branch-free (a segment with nothing to add reads 16 zero bytes), so that the requests
        // of a group leave back to back.  (Requested before the accumulators go to LDS, the first group's trip to memory would run
        // under the transposition -- but the extra live registers take the BK 16 kernel from three resident blocks per CU to two.)
        float4 pre[kGrp], addv[kGrp];
        const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
        for (int u = 0; u < kGrp; u++) {
          const int idx = t + 256 * (g + u), m = m0 + pass * HALF + idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
          const float *pp = p.init_mode == 0 ? p.C + (long long)m * p.ldc + n : (p.init_mode == 1 ? p.bias + n : zero);
          const float *pa = (p.add && m >= p.add_lo && m < p.add_hi) ? p.add + (long long)(m - p.add_lo) * p.ldadd + n : zero;
          pre[u] = *reinterpret_cast<const float4 *>(pp);
          addv[u] = *reinterpret_cast<const float4 *>(pa);
        }
#pragma unroll
        for (int u = 0; u < kGrp; u++) {
          const int idx = t + 256 * (g + u), row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
          const int m = m0 + pass * HALF + row, n = n0 + c4;
          float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDCS + c4);
          v.x += pre[u].x + p.add_scale * addv[u].x;
          v.y += pre[u].y + p.add_scale * addv[u].y;
          v.z += pre[u].z + p.add_scale * addv[u].z;
          v.w += pre[u].w + p.add_scale * addv[u].w;
          if (p.relu) { v.x = floor_keep_nan(v.x, 0.f); v.y = floor_keep_nan(v.y, 0.f); v.z = floor_keep_nan(v.z, 0.f); v.w = floor_keep_nan(v.w, 0.f); }
          *reinterpret_cast<float4 *>(p.C + (long long)m * p.ldc + n) = v;
          if (kColStats && colstats) {  // (kColStats: the thread's column group is the same in every segment)
            cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
            cq[0] += v.x * v.x; cq[1] += v.y * v.y; cq[2] += v.z * v.z; cq[3] += v.w * v.w;
          }
        }
      }
      if (pass + 1 < BM / HALF) __syncthreads();
      continue;
    }
    __syncthreads();
    for (int idx = t; idx < HALF * (BN / 4); idx += 256) {
      const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int m = m0 + pass * HALF + row, n = n0 + c4;
      if (m >= p.M || n >= p.N) continue;
      float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDCS + c4);
      if (p.ksplit > 1) {  // raw partial tile; the reduce kernel applies the epilogue
        const int ldp = (p.N + 3) & ~3;
        *reinterpret_cast<float4 *>(p.partial + ((long long)sp * p.M + m) * ldp + n) = v;
        continue;
      }
      float *c = p.C + (long long)m * p.ldc + n;
      if (cvec && n + 3 < p.N) {
        if (p.init_mode == 1) {
          const float4 b = *reinterpret_cast<const float4 *>(p.bias + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        } else if (p.init_mode == 0) {
          const float4 o = *reinterpret_cast<const float4 *>(c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (p.add && m >= p.add_lo && m < p.add_hi) {
          const float4 o = *reinterpret_cast<const float4 *>(p.add + (long long)(m - p.add_lo) * p.ldadd + n);
          v.x += p.add_scale * o.x; v.y += p.add_scale * o.y; v.z += p.add_scale * o.z; v.w += p.add_scale * o.w;
        }
        if (p.relu) { v.x = floor_keep_nan(v.x, 0.f); v.y = floor_keep_nan(v.y, 0.f); v.z = floor_keep_nan(v.z, 0.f); v.w = floor_keep_nan(v.w, 0.f); }
        *reinterpret_cast<float4 *>(c) = v;
        if (colstats) {
          cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
          cq[0] += v.x * v.x; cq[1] += v.y * v.y; cq[2] += v.z * v.z; cq[3] += v.w * v.w;
        }
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (n + e < p.N) {
            float x = vv[e];
            if (p.init_mode == 1) x += p.bias[n + e];
            else if (p.init_mode == 0) x += c[e];
            if (p.add && m >= p.add_lo && m < p.add_hi) x += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n + e];
            if (p.relu) x = floor_keep_nan(x, 0.f);
            c[e] = x;
            if (colstats) { cs[e] += x; cq[e] += x * x; }
          }
        }
      }
    }
    if (pass + 1 < BM / HALF) __syncthreads();
  }
  if (kColStats && colstats) {
    // thread t owns columns 4 (t % (BN/4)) .. +3 of every row it stored: lanes 32 apart share them when BN = 128, then the
    // four waves; one partial row per row tile
    constexpr int G = BN / 4;
    __syncthreads();
    float *red = smem;  // [256 / G][BN][2]
#pragma unroll
    for (int e = 0; e < 4; e++) {
      red[((t / G) * BN + (t % G) * 4 + e) * 2] = cs[e];
      red[((t / G) * BN + (t % G) * 4 + e) * 2 + 1] = cq[e];
    }
    __syncthreads();
    if (t < BN && n0 + t < p.N) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int g = 0; g < 256 / G; g++) {
        a0 += red[(g * BN + t) * 2];
        a1 += red[(g * BN + t) * 2 + 1];
      }
      p.colstats[(long long)tile_m * p.N + n0 + t] = a0;
      p.colstats[((long long)p.colstats_stride + tile_m) * p.N + n0 + t] = a1;
    }
  }
}

template <int WM, int WN, int TM, int TN, int BK, bool B_KC, int VEC, int TAG = 0>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const RowsGemmArgs p, int ntm, int ntn) {
  rows_gemm_block<WM, WN, TM, TN, BK, B_KC, VEC>(p, ntm, ntn, (int)blockIdx.x, (int)gridDim.x);
}
// Grouped launch: task i owns the blocks [first[i], first[i + 1]) and runs them exactly as a launch of its own would (one column tile,
// no K split: its arguments say so).  The natural-gradient input-side statistics of a whole net at the recipes' minibatch: 33 launches
// of 26 .. 78 blocks each as one.
struct RowsGemmTasks {
  const RowsGemmArgs *args;  // device, ntasks
  const int *first;          // device, ntasks + 1
  int ntasks;
};
template <int WM, int WN, int TM, int TN, int BK, bool B_KC, int VEC>
__global__ __launch_bounds__(256) void rows_gemm_group_kernel(RowsGemmTasks g) {
  int lo = 0, hi = g.ntasks;  // the task whose block range holds blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= g.first[mid]) lo = mid;
    else hi = mid;
  }
  const int b0 = g.first[lo], nb = g.first[lo + 1] - b0;
  rows_gemm_block<WM, WN, TM, TN, BK, B_KC, VEC>(g.args[lo], nb, 1, (int)blockIdx.x - b0, nb);
}

// ------------------------------------------------------------------------ rows_gemm, split-bf16 arithmetic
// The same GEMM (arguments, K-segment iterator, epilogue) computed as a = a_hi + a_lo, b = b_hi + b_lo in bf16 with
//   a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi            (three v_mfma_f32_32x32x16_bf16 per 16 k, f32 accumulate):
// 16 mantissa bits per operand, products accurate to ~2^-16 relative, at 3/16 of the f32 MFMA's cycles per flop.
// The f32 operands are split when the staged tile goes to LDS (two bf16 planes per operand, 80-byte rows: conflict-free
// 16-byte fragment reads); lane (r, h) of a fragment holds k = 8h..8h+7 of row r (MI355X guide, bf16 operand maps).
// B must be k-contiguous (B_KC) and everything 16-byte aligned; rows_gemm() falls back to the f32 kernel otherwise.
// compile-time loop: f(IntC<0>{}), f(IntC<1>{}), ... -- register arrays indexed by the counter stay in registers without
// depending on the loop unroller
template <int V>
struct IntC {
  static constexpr int value = V;
};
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(IntC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// x = pl[0] + pl[1] (+ pl[2]) + O(2^-8NP |x|): each plane is the bf16 rounding of what the planes above it left.
template <int NP>
__device__ __forceinline__ void split_bf16(const float4 v, bf16x4 (&pl)[NP]) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float r = x[i];
#pragma unroll
    for (int q = 0; q < NP; q++) {
      const __bf16 h = (__bf16)r;
      pl[q][i] = h;
      r -= (float)h;
    }
  }
}

template <int WM, int WN, int TM, int TN, int BK, int NP, int D, int TAG = 0>
__global__ __launch_bounds__(256, 2) void rows_gemm_x3_kernel(const RowsGemmArgs p, int ntm, int ntn) {
  constexpr int VEC = 4;
  constexpr bool B_KC = true;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDH = BK + 8;                 // bf16 per LDS row (80 bytes at BK = 32)
  constexpr int A_TILE = NP * BM * LDH / 2;   // floats per A buffer (NP planes)
  constexpr int B_TILE = NP * BN * LDH / 2;
  constexpr int A_F4 = (BM * BK / 4 + 255) / 256;
  constexpr int B_F4 = (BN * BK / 4 + 255) / 256;
  constexpr int KF4 = BK / 4;  // float4 per k-row of a k-contiguous tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;               // [2][A_TILE]
  float *Bs = smem + 2 * A_TILE;  // [2][B_TILE]

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of logical tile ids; within it tile_n varies fastest so the blocks
  // that re-read the same A rows (and the taps' neighbouring rows) hit the same L2.
  const int nblk = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int sp = 0;
  if (p.ksplit > 1) {  // split-K launch: consecutive block ids share a tile
    sp = blockIdx.x % p.ksplit;
    bid = blockIdx.x / p.ksplit;
  }
  const long long k_begin = (long long)sp * p.kchunk, k_end = p.ksplit > 1 ? k_begin + p.kchunk : (1LL << 60);
  const int tile_m = bid / ntn, tile_n = bid % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // ---- K iterator over (segment, chunk), skipping zero-coefficient segments
  // (a split-K block only visits the part of each segment inside its [k_begin, k_end) slice)
  int seg = -1, kc = 0, klen = 0;
  long long seg_kstart = 0, seg_knext = 0;
  float cf = 1.f;
  auto next_seg = [&]() {
    for (++seg; seg < p.nseg; ++seg) {
      seg_kstart = seg_knext;
      seg_knext += p.seg[seg].klen;
      cf = p.coef ? p.coef[seg] : 1.f;
      const long long lo = k_begin > seg_kstart ? k_begin - seg_kstart : 0;
      const long long hi = k_end < seg_knext ? k_end - seg_kstart : p.seg[seg].klen;
      if (cf != 0.f && hi > lo) {
        kc = (int)lo;
        klen = (int)hi;
        return;
      }
    }
    kc = 0;
    klen = 0;
  };
  next_seg();

  // D staged K-steps in registers: one being split into LDS, D - 1 in flight behind it.  A bf16 K-step is 4..5x shorter
  // than the f32 kernel's, far shorter than a global load's latency, so one step of prefetch leaves the MFMAs waiting.
  float4 ra[D][A_F4], rb[D][B_F4];
  // The tap coefficient (and the sum of squares of p.sumsq) is applied when a staged tile goes to LDS, not when it is
  // loaded: anything that touches ra/rb right after the loads would wait for them in front of the MFMAs they are
  // supposed to overlap with.
  float cf_tile[D];     // coefficient of the segment the tile in ra/rb[slot] was loaded from
  float ssq = 0.f;      // p.sumsq: running sum of (coef * a)^2 over everything this thread stages
  auto add_ssq = [&](const float4 (&xa)[A_F4], float c) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < A_F4; j++) q += xa[j].x * xa[j].x + xa[j].y * xa[j].y + xa[j].z * xa[j].z + xa[j].w * xa[j].w;
    ssq += c * c * q;
  };
  // Per-segment, per-thread source pointers for the fast path (full K-step inside the segment, float4 loads):
  // rows/columns that are out of range read 16 zero bytes instead of branching.
  const float *aptr[A_F4], *bptr[B_F4];
  int astep[A_F4], bstep[B_F4];
  int ptr_seg = -1;
  auto setup_ptrs = [&]() {
    const GemmSeg sg = p.seg[seg];
    const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j, row = idx / KF4, m = m0 + row;
      const bool rv = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && m < p.M && m >= sg.m_lo && m < sg.m_hi;
      aptr[j] = rv ? p.A + sg.a_off + (long long)m * p.lda + (idx % KF4) * 4 : zero;
      astep[j] = rv ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      bool rv;
      if (B_KC) {
        const int n = n0 + idx / KF4;
        rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n < p.N;
        bptr[j] = rv ? p.B + sg.b_off + (long long)n * p.ldb + (idx % KF4) * 4 : zero;
        bstep[j] = rv ? 1 : 0;
      } else {
        const int kr = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
        rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n + 3 < p.N;
        bptr[j] = rv ? p.B + sg.b_off + (long long)kr * p.ldb + n : zero;
        bstep[j] = rv ? (int)p.ldb : 0;
      }
    }
    ptr_seg = seg;
  };
  // !B_KC: a ragged last column group (n + 3 >= N) needs the general path for the whole launch
  const bool fast_ok = VEC == 4 && (B_KC || p.N % 4 == 0);
  auto load_tile = [&](float4 (&ra)[A_F4], float4 (&rb)[B_F4], float &cf_tile) {  // global -> registers for chunk (seg, kc)
    if (fast_ok && kc + BK <= klen) {
      if (ptr_seg != seg) setup_ptrs();
#pragma unroll
      for (int j = 0; j < A_F4; j++) ra[j] = *reinterpret_cast<const float4 *>(aptr[j] + (long long)kc * astep[j]);
#pragma unroll
      for (int j = 0; j < B_F4; j++) rb[j] = *reinterpret_cast<const float4 *>(bptr[j] + (long long)kc * bstep[j]);
      cf_tile = cf;
      return;
    }
    const GemmSeg sg = p.seg[seg];
    const float *Ab = p.A + sg.a_off;
    const float *Bb = p.B + sg.b_off;
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      const int row = idx / KF4, k = kc + (idx % KF4) * 4;
      const int m = m0 + row;
      const bool rv = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && m < p.M && m >= sg.m_lo && m < sg.m_hi;
      const float *ptr = Ab + (long long)m * p.lda + k;
      ra[j] = ld4(ptr, rv && k < klen, rv && k + 1 < klen, rv && k + 2 < klen, rv && k + 3 < klen, VEC == 4);
    }
    if (B_KC) {
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        const int idx = t + 256 * j;
        const int row = idx / KF4, k = kc + (idx % KF4) * 4;
        const int n = n0 + row;
        const bool rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n < p.N;
        const float *ptr = Bb + (long long)n * p.ldb + k;
        rb[j] = ld4(ptr, rv && k < klen, rv && k + 1 < klen, rv && k + 2 < klen, rv && k + 3 < klen, VEC == 4);
      }
    } else {
      constexpr int NF4 = BN / 4;
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        const int idx = t + 256 * j;
        const int kr = idx / NF4, n = n0 + (idx % NF4) * 4;
        const bool rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && kc + kr < klen;
        const float *ptr = Bb + (long long)(kc + kr) * p.ldb + n;
        rb[j] = ld4(ptr, rv && n < p.N, rv && n + 1 < p.N, rv && n + 2 < p.N, rv && n + 3 < p.N, VEC == 4);
      }
    }
    cf_tile = cf;
  };
  auto store_tile = [&](float4 (&ra)[A_F4], float4 (&rb)[B_F4], const float cf_tile, int buf) {  // registers -> LDS
    float *as = As + buf * A_TILE, *bs = Bs + buf * B_TILE;
    if (p.sumsq) add_ssq(ra, cf_tile);
    if (p.coef) {
#pragma unroll
      for (int j = 0; j < B_F4; j++) {
        rb[j].x *= cf_tile; rb[j].y *= cf_tile; rb[j].z *= cf_tile; rb[j].w *= cf_tile;
      }
    }
    __bf16 *ah = reinterpret_cast<__bf16 *>(as), *bh = reinterpret_cast<__bf16 *>(bs);
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      if (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) {
        bf16x4 pl[NP];
        split_bf16<NP>(ra[j], pl);
        const int o = (idx / KF4) * LDH + (idx % KF4) * 4;
#pragma unroll
        for (int q = 0; q < NP; q++) *reinterpret_cast<bf16x4 *>(ah + q * BM * LDH + o) = pl[q];
      }
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      if (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) {
        bf16x4 pl[NP];
        split_bf16<NP>(rb[j], pl);
        const int o = (idx / KF4) * LDH + (idx % KF4) * 4;
#pragma unroll
        for (int q = 0; q < NP; q++) *reinterpret_cast<bf16x4 *>(bh + q * BN * LDH + o) = pl[q];
      }
    }
  };
  auto compute = [&](int buf) {
    const __bf16 *ah = reinterpret_cast<const __bf16 *>(As + buf * A_TILE) + (wm * TM * 32 + li) * LDH + lh * 8;
    const __bf16 *bh = reinterpret_cast<const __bf16 *>(Bs + buf * B_TILE) + (wn * TN * 32 + li) * LDH + lh * 8;
#pragma unroll
    for (int c = 0; c < BK / 16; c++) {
      bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[q][i] = *reinterpret_cast<const bf16x8 *>(ah + q * BM * LDH + i * 32 * LDH + c * 16);
#pragma unroll
        for (int i = 0; i < TN; i++) b[q][i] = *reinterpret_cast<const bf16x8 *>(bh + q * BN * LDH + i * 32 * LDH + c * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          // every plane product a_q b_r with q + r < NP, smallest terms first and the leading term last
#pragma unroll
          for (int d = NP - 1; d >= 0; d--)
#pragma unroll
            for (int q = 0; q <= d; q++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[d - q][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // One staged K-step beyond the one in LDS.  Measured on MI355X: a deeper register ring (2-4 steps, counted vmcnt waits
  // in a straight-line steady state) costs 60+ registers -> one block per CU or scratch, and ran 10-20 % slower than this
  // loop with two blocks per CU covering each other's load latency.
  if (seg < p.nseg) {
    load_tile(ra[0], rb[0], cf_tile[0]);
    store_tile(ra[0], rb[0], cf_tile[0], 0);
    __syncthreads();
    int buf = 0;
    while (true) {
      kc += BK;
      if (kc >= klen) next_seg();
      const bool more = seg < p.nseg;
      if (more) load_tile(ra[0], rb[0], cf_tile[0]);  // in flight while the MFMAs run
      compute(buf);
      if (!more) break;
      store_tile(ra[0], rb[0], cf_tile[0], buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The accumulators go through LDS (reusing the staging buffers) so that C is read/written as whole
  // 16-byte-per-lane row segments instead of 64 four-byte accesses per lane.
  constexpr int LDCS = BN + 4;
  constexpr int SMEM_FLOATS = 2 * (A_TILE + B_TILE);
  constexpr int HALF = (BM * LDCS <= SMEM_FLOATS) ? BM : ((BM / 2) * LDCS <= SMEM_FLOATS ? BM / 2 : BM / 4);
  static_assert(HALF * LDCS <= SMEM_FLOATS, "epilogue tile does not fit the staging LDS");
  static_assert(HALF % (TM * 32) == 0, "a wave's rows must not straddle epilogue passes");
  float *Cs = smem;
  const bool cvec = p.c_vec != 0;
  __syncthreads();
  if (p.sumsq) {  // block total through LDS (the staging buffers are free now)
    double v = ssq;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    double *red = reinterpret_cast<double *>(smem);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (t == 0) p.sumsq[p.ksplit > 1 ? (int)blockIdx.x : bid] = (red[0] + red[1]) + (red[2] + red[3]);  // (no K split: entry = row tile)
    if (blockIdx.x == 0)  // entries no block owns (the array is sized for a split-K launch)
      for (int i = gridDim.x + t; i < p.sumsq_cap; i += 256) p.sumsq[i] = 0.0;
    __syncthreads();
  }
#pragma unroll
  for (int pass = 0; pass < BM / HALF; pass++) {
    if ((wm * TM * 32) / HALF == pass) {
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh - pass * HALF;
            Cs[row * LDCS + (wn * TN + j) * 32 + li] = acc[i][j][r];
          }
    }
    {  // interior tiles: as in rows_gemm_kernel, the old C / bias / addend of several row segments requested together
      constexpr int kSeg = HALF * (BN / 4) / 256;
      constexpr int kGrp = kSeg % 8 == 0 ? 8 : (kSeg % 5 == 0 ? 5 : (kSeg % 4 == 0 ? 4 : (kSeg % 2 == 0 ? 2 : 1)));
      constexpr bool kFastShape = (HALF * (BN / 4)) % 256 == 0;
      if (kFastShape && cvec && p.ksplit <= 1 && !p.serial_epilogue && m0 + BM <= p.M && n0 + BN <= p.N) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < kSeg; g += kGrp) {
          float4 pre[kGrp], addv[kGrp];
          const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
          for (int u = 0; u < kGrp; u++) {
            const int idx = t + 256 * (g + u), m = m0 + pass * HALF + idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
            const float *pp = p.init_mode == 0 ? p.C + (long long)m * p.ldc + n : (p.init_mode == 1 ? p.bias + n : zero);
            const float *pa = (p.add && m >= p.add_lo && m < p.add_hi) ? p.add + (long long)(m - p.add_lo) * p.ldadd + n : zero;
            pre[u] = *reinterpret_cast<const float4 *>(pp);
            addv[u] = *reinterpret_cast<const float4 *>(pa);
          }
#pragma unroll
          for (int u = 0; u < kGrp; u++) {
            const int idx = t + 256 * (g + u), row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
            const int m = m0 + pass * HALF + row, n = n0 + c4;
            float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDCS + c4);
            v.x += pre[u].x + p.add_scale * addv[u].x;
            v.y += pre[u].y + p.add_scale * addv[u].y;
            v.z += pre[u].z + p.add_scale * addv[u].z;
            v.w += pre[u].w + p.add_scale * addv[u].w;
            if (p.relu) { v.x = floor_keep_nan(v.x, 0.f); v.y = floor_keep_nan(v.y, 0.f); v.z = floor_keep_nan(v.z, 0.f); v.w = floor_keep_nan(v.w, 0.f); }
            *reinterpret_cast<float4 *>(p.C + (long long)m * p.ldc + n) = v;
          }
        }
        if (pass + 1 < BM / HALF) __syncthreads();
        continue;
      }
    }
    __syncthreads();
    for (int idx = t; idx < HALF * (BN / 4); idx += 256) {
      const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int m = m0 + pass * HALF + row, n = n0 + c4;
      if (m >= p.M || n >= p.N) continue;
      float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDCS + c4);
      if (p.ksplit > 1) {  // raw partial tile; the reduce kernel applies the epilogue
        const int ldp = (p.N + 3) & ~3;
        *reinterpret_cast<float4 *>(p.partial + ((long long)sp * p.M + m) * ldp + n) = v;
        continue;
      }
      float *c = p.C + (long long)m * p.ldc + n;
      if (cvec && n + 3 < p.N) {
        if (p.init_mode == 1) {
          const float4 b = *reinterpret_cast<const float4 *>(p.bias + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        } else if (p.init_mode == 0) {
          const float4 o = *reinterpret_cast<const float4 *>(c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (p.add && m >= p.add_lo && m < p.add_hi) {
          const float4 o = *reinterpret_cast<const float4 *>(p.add + (long long)(m - p.add_lo) * p.ldadd + n);
          v.x += p.add_scale * o.x; v.y += p.add_scale * o.y; v.z += p.add_scale * o.z; v.w += p.add_scale * o.w;
        }
        if (p.relu) { v.x = floor_keep_nan(v.x, 0.f); v.y = floor_keep_nan(v.y, 0.f); v.z = floor_keep_nan(v.z, 0.f); v.w = floor_keep_nan(v.w, 0.f); }
        *reinterpret_cast<float4 *>(c) = v;
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (n + e < p.N) {
            float x = vv[e];
            if (p.init_mode == 1) x += p.bias[n + e];
            else if (p.init_mode == 0) x += c[e];
            if (p.add && m >= p.add_lo && m < p.add_hi) x += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n + e];
            if (p.relu) x = floor_keep_nan(x, 0.f);
            c[e] = x;
          }
        }
      }
    }
    if (pass + 1 < BM / HALF) __syncthreads();
  }
}

template <int WM, int WN, int TM, int TN, int BK, int NP, int TAG>
void launch_rows_x3_tagged(dim3 grid, const RowsGemmArgs &a, int ntm, int ntn, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int D = 1;  // staged K-steps beyond the one in LDS.  Measured: D = 2 needs > 256 registers (one block per CU, or
                        // scratch) and runs 20 % slower than D = 1 with two blocks per CU covering each other's load latency
  constexpr size_t lds = sizeof(__bf16) * 2 * NP * (size_t)(BM + BN) * (BK + 8);  // double buffer x NP planes
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void *)rows_gemm_x3_kernel<WM, WN, TM, TN, BK, NP, D, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((rows_gemm_x3_kernel<WM, WN, TM, TN, BK, NP, D, TAG>), grid, dim3(256), lds, s, a, ntm, ntn);
}

template <int WM, int WN, int TM, int TN, int BK, int TAG>
void rows_attr() {  // > 64 KiB of dynamic LDS must be opted into, once per instantiation
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds_kc = sizeof(float) * 2 * (BM * (BK + 4) + BN * (BK + 4));
  const size_t lds_nc = sizeof(float) * 2 * (BM * (BK + 4) + BK * (BN + 4));
  static bool attr_done = false;
  if (attr_done) return;
  hipFuncSetAttribute((const void *)rows_gemm_kernel<WM, WN, TM, TN, BK, true, 4, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kc);
  hipFuncSetAttribute((const void *)rows_gemm_kernel<WM, WN, TM, TN, BK, true, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kc);
  hipFuncSetAttribute((const void *)rows_gemm_kernel<WM, WN, TM, TN, BK, false, 4, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nc);
  hipFuncSetAttribute((const void *)rows_gemm_kernel<WM, WN, TM, TN, BK, false, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nc);
  attr_done = true;
}

template <int WM, int WN, int TM, int TN, int BK, int TAG>
void launch_rows_kernel_tagged(dim3 grid, const RowsGemmArgs &a, int ntm, int ntn, bool b_kc, bool vec, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds_kc = sizeof(float) * 2 * (BM * (BK + 4) + BN * (BK + 4));
  const size_t lds_nc = sizeof(float) * 2 * (BM * (BK + 4) + BK * (BN + 4));
  rows_attr<WM, WN, TM, TN, BK, TAG>();
  const dim3 block(256);
  if (b_kc) {
    if (vec) hipLaunchKernelGGL((rows_gemm_kernel<WM, WN, TM, TN, BK, true, 4, TAG>), grid, block, lds_kc, s, a, ntm, ntn);
    else hipLaunchKernelGGL((rows_gemm_kernel<WM, WN, TM, TN, BK, true, 1, TAG>), grid, block, lds_kc, s, a, ntm, ntn);
  } else {
    if (vec) hipLaunchKernelGGL((rows_gemm_kernel<WM, WN, TM, TN, BK, false, 4, TAG>), grid, block, lds_nc, s, a, ntm, ntn);
    else hipLaunchKernelGGL((rows_gemm_kernel<WM, WN, TM, TN, BK, false, 1, TAG>), grid, block, lds_nc, s, a, ntm, ntn);
  }
}
// every rows_gemm_kernel launch goes through here
template <int WM, int WN, int TM, int TN, int BK>
void launch_rows_kernel(dim3 grid, const RowsGemmArgs &a, int ntm, int ntn, bool b_kc, bool vec, hipStream_t s) {
  if (a.prec == 1 && b_kc && vec) {  // split-bf16 arithmetic, two planes (three products)
    if (g_prof_override == 3) launch_rows_x3_tagged<WM, WN, TM, TN, BK, 2, 1>(grid, a, ntm, ntn, s);
    else launch_rows_x3_tagged<WM, WN, TM, TN, BK, 2, 0>(grid, a, ntm, ntn, s);
    return;
  }
  if (a.prec == 3 && b_kc && vec) {  // three planes (six products): f32-equivalent
    if constexpr (BK == 16) {
      if (g_prof_override == 3) launch_rows_x3_tagged<WM, WN, TM, TN, BK, 3, 1>(grid, a, ntm, ntn, s);
      else launch_rows_x3_tagged<WM, WN, TM, TN, BK, 3, 0>(grid, a, ntm, ntn, s);
      return;
    }
  }
  if (g_prof_override == 3) launch_rows_kernel_tagged<WM, WN, TM, TN, BK, 1>(grid, a, ntm, ntn, b_kc, vec, s);
  else launch_rows_kernel_tagged<WM, WN, TM, TN, BK, 0>(grid, a, ntm, ntn, b_kc, vec, s);
}

template <int WM, int WN, int TM, int TN>
constexpr bool kRingTile = (WM == 2 && WN == 2 && TM == 2 && TN == 2) || (WM == 4 && WN == 1 && TM == 1 && TN == 5);
inline bool ring_applies(const RowsGemmArgs &a, bool b_kc, bool vec, int tile_cols) {
  return g_prof_override != 3 && (tile_cols == 128 || rows_gemm_ring_mode() >= 2) && rows_gemm_ring_ok(a, b_kc, vec);
}

template <int WM, int WN, int TM, int TN, int BK>
hipError_t launch_rows(const RowsGemmArgs &a, bool b_kc, bool vec, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  if constexpr (kRingTile<WM, WN, TM, TN>) {  // the persistent LDS-DMA-ring form (gemm_ring.hip) where it applies
    if (ring_applies(a, b_kc, vec, BN)) return rows_gemm_ring(a, b_kc, BN, s);
  }
  launch_rows_kernel<WM, WN, TM, TN, BK>(dim3(ntm * ntn), a, ntm, ntn, b_kc, vec, s);
  return hipGetLastError();
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace {

// epilogue of a split-K tail: C[m][n] = f(sum_sp partial[sp][m][n]) with the same init/bias/addend/ReLU rules
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const RowsGemmArgs p) {
  const int ldp = (p.N + 3) & ~3;
  const long long total = (long long)p.M * p.N;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const int m = (int)(e / p.N), n = (int)(e % p.N);
    float v = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < p.ksplit; sp++) v += p.partial[((long long)sp * p.M + m) * ldp + n];  // (unrolled: four requests in flight)
    float *c = p.C + (long long)m * p.ldc + n;
    if (p.init_mode == 1) v += p.bias[n];
    else if (p.init_mode == 0) v += *c;
    if (p.add && m >= p.add_lo && m < p.add_hi) v += p.add_scale * p.add[(long long)(m - p.add_lo) * p.ldadd + n];
    if (p.relu) v = floor_keep_nan(v, 0.f);
    *c = v;
  }
}

inline int device_cus() {  // compute units of the current device (cached per device)
  static int cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus_of[dev] == 0) {
    hipDeviceProp_t prop;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    (void)hipGetLastError();
  }
  return cus_of[dev];
}
template <int WM, int WN, int TM, int TN, int BK>
int rows_slots_per_cu(int prec) {
  int per_cu = (WM * TM == 4 && WN * TN == 4 && BK == 16 && prec == 0) ? 3 : 2;
  if (prec == 3 && WN * TN == 5) per_cu = 1;
  if (WM * TM == 2 && prec == 0) per_cu = 4;  // 64-row tiles (27 KiB of LDS, 32 accumulator registers)
  return per_cu;
}
template <int WM, int WN, int TM, int TN, int BK>
int rows_slots(bool b_kc, int prec) {  // resident blocks on the chip for this tile variant
  // Blocks per CU are MEASURED (tools/residency_probe.py: time of a plain launch steps up when one more tile needs one
  // more round), not queried: hipOccupancyMaxActiveBlocksPerMultiprocessor answers 3 for the 128x160 tile (168
  // registers, 46 KiB LDS) where the steps sit at 513 and 1025 tiles, i.e. 2 per CU (MI355X_MICROARCH.md warns that the
  // query can over-report).  128x128 BK 32: 2 (73.7 KiB LDS); 128x128 BK 16: 3 (41 KiB, 154 registers; steps at 513,
  // 769, 1025); 128x160: 2.
  // Split-bf16 variants are LDS-bound: 2 planes 128x128 BK 32 80 KiB (2), 128x160 BK 16 54 KiB (2); 3 planes 128x128
  // BK 16 72 KiB (2), 128x160 BK 16 81 KiB (1).
  (void)b_kc;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    (void)hipGetLastError();
  }
  return rows_slots_per_cu<WM, WN, TM, TN, BK>(prec) * cus;
}

// scratch for split-K partial tiles: allocated once, on first use (64 MiB covers slots x BM x BN floats)
constexpr size_t kScratchDefaultBytes = 64u << 20;
float *g_scratch_override = nullptr;
size_t g_scratch_override_bytes = 0;
float *splitk_scratch(size_t *bytes) {
  if (g_scratch_override) {
    *bytes = g_scratch_override_bytes;
    return g_scratch_override;
  }
  static float *buf = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    if (hipMalloc((void **)&buf, kScratchDefaultBytes) != hipSuccess) buf = nullptr;
    (void)hipGetLastError();
  }
  *bytes = kScratchDefaultBytes;
  return buf;
}

template <int WM, int WN, int TM, int TN, int BK>
hipError_t launch_rows_balanced(const RowsGemmArgs &a, bool b_kc, bool vec, int cls, double flops, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  int slots = rows_slots<WM, WN, TM, TN, BK>(b_kc, (b_kc && vec) ? a.prec : 0);
  if constexpr (kRingTile<WM, WN, TM, TN>) {
    if (ring_applies(a, b_kc, vec, BN)) slots = rows_gemm_ring_slots(BN);
  }
  const int tiles = ntm * ntn;
  const int q = tiles / slots, r = tiles % slots;
  const bool stats = a.colstats != nullptr;  // (rows_gemm() leaves it set only for the exact-f32 128 x 128 tile)
  long long ktot = 0;
  bool k4 = true;
  for (int i = 0; i < a.nseg; i++) {
    ktot += a.seg[i].klen;
    k4 = k4 && (a.seg[i].klen % 4 == 0 || a.nseg == 1);  // one segment: only the very end of K is ragged (slow path of the last chunk)
  }
  // All blocks run equally long, so q*slots + r tiles cost q+1 rounds.  When the last round would be less than
  // half full, finish the last rows with a split-K launch that spreads them over every CU instead.
  const int main_mt = (q * slots) / ntn;  // full tile rows handled by the plain launch
  float *scratch = nullptr;
  size_t scratch_bytes = 0;
  // A handful of tiles with a long reduction (the R x D products of the natural-gradient state, P = M M^T of the
  // orthonormal constraint): one block per tile would crawl through K at load latency on a few CUs, so split K
  // over the idle ones.
  // (and a launch of more than slots / 4 but fewer than `slots` tiles -- one partly filled round: 188 row tiles of the 1500 x 16 shard's
  // full-rate .linear layers leave 68 of 256 CUs idle for the whole launch -- splits K by the factor S that fills whole rounds of CUs
  // best, ceil(tiles S / CUs) / S smallest: 188 tiles -> S = 4, 752 quarter tiles = 2.94 rounds of 256 instead of 4 quarters on 188 CUs)
  int S_partial = 0;
  if (!stats && q == 0 && tiles * 4 > slots && k4 && options().splitk_partial_round) {
    const int cus = device_cus();
    const long long kt = (ktot + BK - 1) / BK;
    double best = (double)((tiles + cus - 1) / cus);  // S = 1
    for (int S = 2; S <= 8; S++) {
      if (kt / S < 24) break;  // at least 24 K steps per slice
      const double cost = (double)(((long long)tiles * S + cus - 1) / cus) / S + 0.02 * S;  // (+ the partial tiles' round trip)
      if (cost < best - 1e-9) {
        best = cost;
        S_partial = S;
      }
    }
  }
  if (!stats && (tiles * 4 <= slots || S_partial >= 2) && k4 && ktot >= 16 * BK && (scratch = splitk_scratch(&scratch_bytes))) {
    const long long kt = (ktot + BK - 1) / BK;
    int S = S_partial >= 2 ? S_partial : slots / tiles;
    if (S_partial < 2 && options().splitk_per_cu == 1) S = std::max(2, device_cus() / tiles);  // (one slice per CU: half the partial tiles)
    if (S > kt / 4) S = (int)(kt / 4);
    const size_t need = sizeof(float) * (size_t)S * a.M * ((a.N + 3) & ~3);
    if (S >= 2 && need <= scratch_bytes) {
      RowsGemmArgs at = a;
      at.kchunk = (int)(((kt + S - 1) / S) * BK);
      at.ksplit = (int)((ktot + at.kchunk - 1) / at.kchunk);
      at.partial = scratch;
      ProfScope ps(cls, flops, s);
      launch_rows_kernel<WM, WN, TM, TN, BK>(dim3(tiles * at.ksplit), at, ntm, ntn, b_kc, vec, s);
      const long long total = (long long)at.M * at.N;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, at);
      return hipGetLastError();
    }
  }
  if (q >= 1 && r > 0 && 2 * r <= slots && main_mt > 0 && main_mt < ntm && k4 && ktot >= 8 * BK && (scratch = splitk_scratch(&scratch_bytes))) {
    const int m_main = main_mt * BM;
    RowsGemmArgs am = a;
    am.M = m_main;
    // column statistics: the main launch writes one partial row per row tile, the tail rows follow as the chunks of a
    // column-reduction pass over the finished C (split-K tail) or as the tail launch's own row tiles (plain tail)
    const int tail_rows = a.M - m_main;
    const long long kt_tail = (ktot + BK - 1) / BK;
    const int tail_tiles0 = ((tail_rows + BM - 1) / BM) * ntn;
    int S_tail = slots / tail_tiles0;
    if (S_tail > kt_tail / 2) S_tail = (int)(kt_tail / 2);
    const bool tail_split = S_tail >= 2 && sizeof(float) * (size_t)S_tail * tail_rows * ((a.N + 3) & ~3) <= scratch_bytes;
    ColReducePlan tail_plan = colreduce_plan(tail_rows, a.N);
    if (stats) {
      am.colstats_stride = main_mt + (tail_split ? tail_plan.chunks : (tail_rows + BM - 1) / BM);
      *a.colstats_rows = am.colstats_stride;
    }
    {
      ProfScope ps(cls, flops * m_main / a.M, s);
      hipError_t e = launch_rows<WM, WN, TM, TN, BK>(am, b_kc, vec, s);
      if (e != hipSuccess) return e;
    }
    RowsGemmArgs at = a;
    at.M = a.M - m_main;
    at.A = a.A + (long long)m_main * a.lda;
    at.C = a.C + (long long)m_main * a.ldc;
    for (int i = 0; i < at.nseg; i++) {
      at.seg[i].m_lo -= m_main;
      at.seg[i].m_hi -= m_main;
    }
    at.add_lo -= m_main;
    at.add_hi -= m_main;
    if (stats) {  // partial rows main_mt .. of the same array (the sums of squares sit colstats_stride rows further in both views)
      at.colstats = a.colstats + (long long)main_mt * a.N;
      at.colstats_stride = am.colstats_stride;
    }
    const int tail_tiles = tail_tiles0;
    const long long kt = kt_tail;
    const int S = S_tail;
    if (tail_split) {
      at.kchunk = (int)(((kt + S - 1) / S) * BK);
      at.ksplit = (int)((ktot + at.kchunk - 1) / at.kchunk);
      at.partial = scratch;
      {
        ProfScope ps(cls, flops * at.M / a.M, s);
        const int tntm = (at.M + BM - 1) / BM;
        launch_rows_kernel<WM, WN, TM, TN, BK>(dim3(tail_tiles * at.ksplit), at, tntm, ntn, b_kc, vec, s);
        const long long total = (long long)at.M * at.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, at);
      }
      if (stats) {
        MatView ct{at.C, at.M, at.N, (int)at.ldc};
        hipError_t e = colreduce_partial_into(1, ct, ct, tail_plan.chunks, tail_plan.rows_per_chunk, am.colstats_stride, at.colstats, s);
        if (e != hipSuccess) return e;
      }
      return hipGetLastError();
    }
    // tail too small to split: plain launch of the remaining rows
    ProfScope ps(cls, flops * at.M / a.M, s);
    return launch_rows<WM, WN, TM, TN, BK>(at, b_kc, vec, s);
  }
  ProfScope ps(cls, flops, s);
  if (stats) {
    RowsGemmArgs as = a;
    as.colstats_stride = ntm;
    *a.colstats_rows = ntm;
    return launch_rows<WM, WN, TM, TN, BK>(as, b_kc, vec, s);
  }
  return launch_rows<WM, WN, TM, TN, BK>(a, b_kc, vec, s);
}

// A launch with p.sumsq: one column tile, block b owns rows [BM b, BM b + BM).  Few row blocks and a long reduction (the
// natural-gradient H = X W^T of a small minibatch: 30 blocks for 256 CUs, each crawling through K at load latency): split K
// over the idle CUs; every (block, slice) then writes the sum of squares of its own slice.
template <int WM, int WN, int TM, int TN, int BK>
hipError_t launch_rows_sumsq(const RowsGemmArgs &a, bool b_kc, bool vec, hipStream_t s) {
  constexpr int BM = WM * TM * 32;
  const int tiles = (a.M + BM - 1) / BM, slots = rows_slots<WM, WN, TM, TN, BK>(b_kc, 0);
  long long ktot = 0;
  bool k4 = true;
  for (int i = 0; i < a.nseg; i++) {
    ktot += a.seg[i].klen;
    k4 = k4 && (a.seg[i].klen % 4 == 0 || a.nseg == 1);
  }
  float *scratch = nullptr;
  size_t scratch_bytes = 0;
  if (tiles * 4 <= slots && k4 && ktot >= 16 * BK && (scratch = splitk_scratch(&scratch_bytes))) {
    const long long kt = (ktot + BK - 1) / BK;
    int S = slots / tiles;
    if (S > kt / 4) S = (int)(kt / 4);
    if (S > 16) S = 16;
    const size_t need = sizeof(float) * (size_t)S * a.M * ((a.N + 3) & ~3);
    if (S >= 2 && need <= scratch_bytes && tiles * S <= a.sumsq_cap) {
      RowsGemmArgs at = a;
      at.kchunk = (int)(((kt + S - 1) / S) * BK);
      at.ksplit = (int)((ktot + at.kchunk - 1) / at.kchunk);
      at.partial = scratch;
      launch_rows_kernel<WM, WN, TM, TN, BK>(dim3(tiles * at.ksplit), at, tiles, 1, b_kc, vec, s);
      const long long total = (long long)at.M * at.N;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, at);
      return hipGetLastError();
    }
  }
  return launch_rows<WM, WN, TM, TN, BK>(a, b_kc, vec, s);
}

}  // namespace

// floor division / modulus for element offsets that may be negative (row shifts of the backward-data gather)
static inline long long floordiv(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// The rows GEMM on the pre-split plane kernels (planes_gemm.hip) when the caller's hint describes these operands.  Everything is
// checked against the hint (base pointers, leading dimensions, 16-column alignment of the K segments, zero rows wherever a segment's
// view leaves [m_lo, m_hi)); false = not applicable, nothing launched.
static bool planes_try_rows(const RowsGemmArgs &a, bool b_kc, int np, double flops, hipStream_t s, hipError_t *err) {
  const PlanesOperand *ha = planes_hint_a(), *hb = planes_hint_b();
  if (!ha || !hb || ha->np != np || hb->np != np || !ha->P || a.sumsq || a.ksplit > 1 || a.nseg > 16) return false;
  // tap coefficients: only when they are exactly the vector folded into the weight planes, segment s = tap s (then the product needs none)
  if (ha->coef || (a.coef == nullptr) != (hb->coef == nullptr)) return false;
  long long coef_t0 = 0;  // segment s carries the coefficient of tap coef_t0 + s: the planes must hold exactly that tap there
  if (a.coef) {
    coef_t0 = a.coef - hb->coef;
    if (coef_t0 < 0 || coef_t0 >= 16 || hb->coef_period <= 0) return false;
  }
  if (a.lda != ha->ld || a.A < ha->base || (a.A - ha->base) % ha->ld != 0) return false;
  const long long arow0 = (a.A - ha->base) / ha->ld;
  const int BM = planes_gemm_tile_rows(a.N), BN = planes_gemm_tile_cols(a.N);
  const long long m_pad = (long long)((a.M + BM - 1) / BM) * BM, n_pad = (long long)((a.N + BN - 1) / BN) * BN;
  PlanesGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.np = np;
  g.A = ha->P; g.RA = ha->R; g.scale_a = ha->scale; g.scale_b = hb->scale;
  if (a.ldb != hb->ld || a.B < hb->base) return false;
  const long long boff0 = a.B - hb->base;
  if (b_kc) {
    if (!hb->P) return false;
    g.B = hb->P; g.RB = hb->R;
  } else {
    if (!hb->PT) return false;
    g.B = hb->PT; g.RB = hb->Rt;
  }
  for (int i = 0; i < a.nseg; i++) {
    const GemmSeg &sg = a.seg[i];
    // A_s[m][k] = base[(arow0 + shift + m) * ld + c0 + k]
    const long long shift = floordiv(sg.a_off, ha->ld), c0 = sg.a_off - shift * ha->ld;
    if (c0 % 16 != 0 || c0 + sg.klen > ha->cols || sg.klen <= 0) return false;
    const long long first = arow0 + shift;  // matrix row of output row 0
    const int lo = sg.m_lo > 0 ? sg.m_lo : 0, hi = sg.m_hi < a.M ? sg.m_hi : a.M;
    if (hi <= lo) return false;
    if (first + lo < 0 || first + hi > ha->rows) return false;       // rows that count must be the matrix's
    if (lo > 0 && first + lo != 0) return false;                      // rows below m_lo must fall into the lead zeros ...
    if (hi < a.M && first + hi != ha->rows) return false;             // ... and rows from m_hi on into the tail zeros
    if (ha->lead + first < 0 || ha->lead + first + m_pad > ha->R) return false;
    g.seg[i].a_row = ha->lead + first;
    g.seg[i].a_kb0 = (int)(c0 / 16);
    g.seg[i].nkb = (sg.klen + 15) / 16;
    if (sg.klen % 16 != 0 && c0 + sg.klen != ha->cols) return false;  // a ragged K block is zero-padded only at the matrix's end
    const long long bo = boff0 + sg.b_off;
    if (b_kc) {  // B[n][k] at base[(brow + n) * ld + bc0 + k]: the row-major planes of the hinted matrix
      const long long brow = bo / hb->ld, bc0 = bo % hb->ld;
      if (bo < 0 || bc0 % 16 != 0 || bc0 + sg.klen > hb->cols || brow + a.N > hb->rows || brow + n_pad > hb->R) return false;
      if (sg.klen % 16 != 0 && bc0 + sg.klen != hb->cols) return false;
      if (a.coef && (bc0 % hb->coef_period != 0 || bc0 / hb->coef_period != coef_t0 + i || sg.klen > hb->coef_period)) return false;
      g.seg[i].b_row = brow;
      g.seg[i].b_kb0 = (int)(bc0 / 16);
    } else {     // B[k][n] at base[(krow + k) * ld + ncol + n]: the transposed planes (k = row of the hinted matrix)
      const long long krow = bo / hb->ld, ncol = bo % hb->ld;
      if (bo < 0 || krow % 16 != 0 || krow + sg.klen > hb->rows || ncol + a.N > hb->cols || ncol + n_pad > hb->Rt) return false;
      if (sg.klen % 16 != 0 && krow + sg.klen != hb->rows) return false;
      if (a.coef && (ncol % hb->coef_period != 0 || ncol / hb->coef_period != coef_t0 + i || a.N > hb->coef_period)) return false;
      g.seg[i].b_row = ncol;
      g.seg[i].b_kb0 = (int)(krow / 16);
    }
  }
  g.nseg = a.nseg;
  g.alt_seg_order = a.alt_seg_order && a.nseg == 2 && g.seg[0].nkb == g.seg[1].nkb && g.seg[0].a_kb0 == g.seg[1].a_kb0 && g.seg[0].a_row != g.seg[1].a_row;
  if (g.alt_seg_order && options().gemm_alt_taps == 2 && g.seg[0].nkb >= 24 && !a.coef) {
    // the two taps in chunks of a few K blocks, alternating: a row block's two reads (as tap 0 by one tile, as the other tap by that tile or its
    // neighbour -- 256-row tiles, 128-row shift) are then a chunk apart instead of half a launch
    const PlanesSeg s0 = g.seg[0], s1 = g.seg[1];
    const int nchunk = std::min(16, s0.nkb / 6), per = (s0.nkb + nchunk - 1) / nchunk;
    int n = 0;
    for (int c = 0; c < nchunk; c++) {
      const int k0 = c * per, kn = std::min(per, s0.nkb - k0);
      if (kn <= 0) break;
      for (const PlanesSeg *sp : {&s0, &s1}) {
        PlanesSeg q = *sp;
        q.a_kb0 += k0;
        q.b_kb0 += k0;
        q.nkb = kn;
        g.seg[n++] = q;
      }
    }
    g.nseg = n;
    g.alt_seg_order = 0;
  }
  g.skip_coef = a.coef;
  g.C = a.C; g.ldc = a.ldc; g.M = a.M; g.N = a.N;
  g.bias = a.bias; g.init_mode = a.init_mode; g.relu = a.relu;
  g.add = a.add; g.ldadd = a.ldadd; g.add_scale = a.add_scale; g.add_lo = a.add_lo; g.add_hi = a.add_hi;
  ProfScope ps(BN == 160 ? 1 : 0, flops, s);
  g_planes_routed_rows++;
  // One block per CU, equal block durations: a launch of q * CUs + r tiles costs q + 1 rounds.  When the last round would be less than
  // half full, the whole rounds run as one launch and the r tail tiles as a second one that splits K over the idle CUs (slabs in the
  // split-K scratch, then the epilogue): 260 row tiles of the 1/3-rate .linear layers on 256 CUs = 1.02 rounds instead of 2.
  {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      (void)hipGetLastError();
    }
    g.M = a.M;
    g.N = a.N;
    const int BMl = planes_gemm_launch_tile_rows(g);  // (128-row tiles for short reductions: two blocks per CU)
    const int ntm = (a.M + BMl - 1) / BMl, ntn = (a.N + BN - 1) / BN, tiles = ntm * ntn, slots = cus * (BMl == 128 ? 2 : 1), q = tiles / slots, r = tiles % slots;
    int nkb = 0;
    for (int i = 0; i < g.nseg; i++) nkb += g.seg[i].nkb;
    const int main_mt = (q * slots) / ntn;  // whole row tiles inside the full rounds
    // BatchNorm statistics of the stored output from the epilogue (RowsGemmArgs::colstats): one partial row per row tile
    const bool stats = a.colstats && a.colstats_rows && g.init_mode != 0;
    if (stats) {
      g.colstats = a.colstats;
      g.colstats_stride = ntm;
      *a.colstats_rows = ntm;
    }
    size_t scratch_bytes = 0;
    float *scratch = nullptr;
    // (a long K range only: with K = 320 the round that is saved is as short as the tail's two extra launches)
    if (q >= 1 && r > 0 && 2 * r <= slots && main_mt > 0 && main_mt < ntm && nkb >= 48 && BMl == BM && (scratch = splitk_scratch(&scratch_bytes))) {
      const int m_main = main_mt * BM, tail_rows = a.M - m_main, tail_tiles = ((tail_rows + BM - 1) / BM) * ntn;
      int S = std::min(cus / tail_tiles, nkb / 4);
      const long long ldp = (a.N + 3) & ~3;
      if (S >= 2 && sizeof(float) * (size_t)S * tail_rows * ldp <= scratch_bytes) {
        PlanesGemmArgs gm = g;
        gm.M = m_main;
        const ColReducePlan tail_plan = colreduce_plan(tail_rows, a.N);
        if (stats) {  // the main launch's row tiles, then the chunks of a column-reduction pass over the finished tail rows
          gm.colstats_stride = main_mt + tail_plan.chunks;
          *a.colstats_rows = gm.colstats_stride;
        }
        *err = planes_gemm(gm, s);
        if (*err != hipSuccess) return true;
        PlanesGemmArgs gt = g;
        gt.colstats = nullptr;
        gt.M = tail_rows;
        gt.C = g.C + (long long)m_main * g.ldc;
        for (int i = 0; i < gt.nseg; i++) gt.seg[i].a_row += m_main;
        if (gt.add) {  // (addend rows are relative to the launch's first output row)
          gt.add_lo = g.add_lo - m_main;
          gt.add_hi = g.add_hi - m_main;
          gt.add = g.add;
        }
        const int kbps = (nkb + S - 1) / S;
        gt.ksplit = (nkb + kbps - 1) / kbps;
        gt.kb_per_split = kbps;
        gt.partial = scratch;
        gt.partial_stride = (long long)tail_rows * ldp;
        gt.ldp_m = ldp;
        gt.ldp_n = 1;
        if (gt.ksplit >= 2) {
          *err = planes_gemm(gt, s);
          if (*err == hipSuccess) *err = planes_splitk_finish(gt, s);
        } else {
          gt.ksplit = 0;
          gt.partial = nullptr;
          *err = planes_gemm(gt, s);  // (a K range too short to split: the tail as a plain launch)
        }
        if (stats && *err == hipSuccess) {
          MatView ct{gt.C, tail_rows, a.N, (int)gt.ldc};
          *err = colreduce_partial_into(1, ct, ct, tail_plan.chunks, tail_plan.rows_per_chunk, gm.colstats_stride, a.colstats + (long long)main_mt * a.N, s);
        }
        return true;
      }
    }
  }
  *err = planes_gemm(g, s);
  return true;
}

hipError_t rows_gemm(const RowsGemmArgs &a_in, bool b_kc, hipStream_t s) {
  if (a_in.M <= 0 || a_in.N <= 0 || a_in.nseg <= 0) return hipSuccess;
  RowsGemmArgs a = a_in;
  a.c_vec = aligned16(a.C) && a.ldc % 4 == 0 && (a.init_mode != 1 || aligned16(a.bias)) && (!a.add || (aligned16(a.add) && a.ldadd % 4 == 0));
  // float4 path needs 16-byte aligned rows and segment starts; ragged tails fall back per float4
  bool vec = aligned16(a.A) && aligned16(a.B) && a.lda % 4 == 0 && a.ldb % 4 == 0;
  for (int i = 0; i < a.nseg; i++) vec = vec && a.seg[i].a_off % 4 == 0 && a.seg[i].b_off % 4 == 0;
  // N == 160 (the TDNN-F bottleneck) gets a 128x160 tile so no column is wasted
  const int waste128 = ((a.N + 127) / 128) * 128 - a.N, waste160 = ((a.N + 159) / 160) * 160 - a.N;
  double flops = 0;
  for (int i = 0; i < a.nseg; i++) {
    const int lo = a.seg[i].m_lo > 0 ? a.seg[i].m_lo : 0, hi = a.seg[i].m_hi < a.M ? a.seg[i].m_hi : a.M;
    if (hi > lo) flops += 2.0 * (hi - lo) * a.N * a.seg[i].klen;
  }
  if (g_prof_on) {
    // algorithmic bytes, each operand element once: the distinct A rows the segments read (taps of one matrix are row
    // shifts of it, so they share rows), the B blocks, the C tile written (and read when it is added to), the addend
    double a_elems = 0, b_elems = 0;
    bool taps = true;  // all segments: same reduction length, A offsets whole rows apart -> one matrix, shifted
    for (int i = 1; i < a.nseg; i++)
      taps = taps && a.seg[i].klen == a.seg[0].klen && a.lda > 0 && (a.seg[i].a_off - a.seg[0].a_off) % a.lda == 0;
    if (taps) {
      long long lo = a.seg[0].a_off, hi = a.seg[0].a_off;
      for (int i = 1; i < a.nseg; i++) {
        lo = std::min(lo, a.seg[i].a_off);
        hi = std::max(hi, a.seg[i].a_off);
      }
      a_elems = ((double)a.M + (double)(hi - lo) / (double)a.lda) * a.seg[0].klen;
    }
    for (int i = 0; i < a.nseg; i++) {
      const int lo = a.seg[i].m_lo > 0 ? a.seg[i].m_lo : 0, hi = a.seg[i].m_hi < a.M ? a.seg[i].m_hi : a.M;
      if (!taps && hi > lo) a_elems += (double)(hi - lo) * a.seg[i].klen;
      b_elems += (double)a.seg[i].klen * a.N;
    }
    double c_elems = (double)a.M * a.N * (a.init_mode == 0 ? 2.0 : 1.0);
    if (a.add) c_elems += (double)(std::min(a.add_hi, a.M) - std::max(a.add_lo, 0)) * a.N;
    g_prof_next_bytes = 4.0 * (a_elems + b_elems + c_elems);
    g_prof_next_flops = flops;
  }
  a.serial_epilogue = 0;
  // two taps of one matrix (same reduction length, A offsets whole rows apart): alternate the order row tile by row tile (rows_gemm_kernel)
  a.alt_seg_order = options().gemm_alt_taps && a.nseg == 2 && a.seg[0].klen == a.seg[1].klen && a.lda > 0 && a.seg[0].a_off != a.seg[1].a_off &&
                    (a.seg[1].a_off - a.seg[0].a_off) % a.lda == 0;
  if (a.prec == 0) a.prec = g_gemm_prec;
  if (a.prec == 2) a.prec = 0;  // 2 = exact f32 regardless of the default
  if (a.prec == 4 || (a.prec == 3 && options().planes)) {  // pre-split planes (f16x3 / bf16x6) when the caller hinted them for these operands
    hipError_t pe = hipSuccess;
    if (planes_try_rows(a, b_kc, a.prec == 4 ? 2 : 3, flops, s, &pe)) return pe;
    if (a.prec == 4) a.prec = 0;  // no planes for this call: exact f32
  }
  if (!(b_kc && vec)) a.prec = 0;  // the split-bf16 kernels need a k-contiguous B and 16-byte alignment
  if (a.colstats_rows) *a.colstats_rows = 0;
  if (!a.colstats_rows || a.prec != 0 || a.sumsq || a.N <= 32 || waste160 < waste128 || a.ksplit > 1) a.colstats = nullptr;
  if (a.sumsq) {  // one column tile, no split-K tail: block b owns rows [128 b, 128 b + 128)
    if (a.N > 128) return hipErrorInvalidValue;
    a.sumsq_cap = rows_gemm_sumsq_blocks(a.M);
    ProfScope ps(0, flops, s);
    // (option ng_bk: 1 = K steps twice as long for these HBM-bound passes -- twice the bytes in flight per resident block)
    if (a.N <= 32) return (options().ng_bk & 1) ? launch_rows_sumsq<4, 1, 1, 1, 64>(a, b_kc, vec, s) : launch_rows_sumsq<4, 1, 1, 1, 32>(a, b_kc, vec, s);
    if (a.N <= 64 && a.N > 32 && (options().ng_bk & 8) == 0) return launch_rows_sumsq<4, 1, 1, 2, 32>(a, b_kc, vec, s);  // both taps' rank-20 products side by side (ng.hip, P form)
    {  // long reductions (the rank-80 pass over the 6034-wide output derivative): the 128 x 128 tile, 48 idle columns and all, 1291 -> 1144 us;
       // short ones (160 columns) lose by it, 79 -> 106 (option ng_bk 4 forces it for both)
      long long ktot = 0;
      for (int i = 0; i < a.nseg; i++) ktot += a.seg[i].klen;
      if (a.N <= 96 && a.N > 64 && ((options().ng_bk & 4) || ktot >= 2048)) return launch_rows_sumsq<2, 2, 2, 2, 32>(a, b_kc, vec, s);
    }
    if (a.N <= 96) return (options().ng_bk & 2) ? launch_rows_sumsq<4, 1, 1, 3, 32>(a, b_kc, vec, s) : launch_rows_sumsq<4, 1, 1, 3, 16>(a, b_kc, vec, s);  // rank-80 preconditioners: 96 of 96 columns, not 80 of 128
    return launch_rows_sumsq<2, 2, 2, 2, 32>(a, b_kc, vec, s);
  }
  // skinny outputs (the natural-gradient projections X W^T, rank <= 32): a 128x32 tile wastes no MFMA columns and
  // keeps three blocks per CU resident to pull the A operand at HBM rate
  if (a.N <= 32 && a.M >= 1024) {
    ProfScope ps(0, flops, s);
    return (options().ng_bk & 1) ? launch_rows<4, 1, 1, 1, 64>(a, b_kc, vec, s) : launch_rows<4, 1, 1, 1, 32>(a, b_kc, vec, s);
  }
  if (waste160 < waste128) return launch_rows_balanced<4, 1, 1, 5, 16>(a, b_kc, vec, 1, flops, s);
  {
    // short reductions (K <= 512: affine forward, linear backward, prefinal layers): BK 16 halves the LDS footprint,
    // three blocks per CU cover each other's prologue / epilogue (+6..13 % measured); long reductions keep BK 32
    long long kt = 0;
    for (int i = 0; i < a.nseg; i++) kt += a.seg[i].klen;
    // launches that leave most of the chip's block slots empty with 128 x 128 tiles (the recipes' minibatch: 3 200 rows) take
    // 64 x 128 tiles: twice the blocks, so the busiest CU carries 3 half tiles instead of 2 whole ones
    const long long tiles128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const bool small = tiles128 < 768;  // measured at 150 x 64: 13.45 -> 13.28 ms per step; 1500 x 16 unchanged
    if (small && a.prec == 0 && !a.sumsq) return launch_rows_balanced<2, 2, 1, 2, 16>(a, b_kc, vec, 0, flops, s);
    if ((kt <= 512 && a.prec == 0) || a.prec == 3) return launch_rows_balanced<2, 2, 2, 2, 16>(a, b_kc, vec, 0, flops, s);
  }
  return launch_rows_balanced<2, 2, 2, 2, 32>(a, b_kc, vec, 0, flops, s);
}

// ---- grouped launch of skinny statistics passes (rows_gemm_group_kernel)
struct RowsGemmGroup {
  std::vector<RowsGemmArgs> args;  // as uploaded
  std::vector<int> first;
  RowsGemmArgs *d_args = nullptr;
  int *d_first = nullptr;
  int capacity = 0;
};
void rows_gemm_group_destroy(RowsGemmGroup *g) {
  if (!g) return;
  if (g->d_args) hipFree(g->d_args);
  if (g->d_first) hipFree(g->d_first);
  delete g;
}
// what the group kernel takes: exact f32, one 32-column tile with the ||A||^2 by-product, k-contiguous B, 16-byte aligned operands
bool rows_gemm_group_ok(const RowsGemmArgs &a) {
  if (a.M <= 0 || a.N <= 0 || a.N > 32 || !a.sumsq || a.nseg <= 0 || a.ksplit > 1 || a.colstats || a.add) return false;
  bool vec = aligned16(a.A) && aligned16(a.B) && a.lda % 4 == 0 && a.ldb % 4 == 0;
  for (int i = 0; i < a.nseg; i++) vec = vec && a.seg[i].a_off % 4 == 0 && a.seg[i].b_off % 4 == 0;
  return vec;
}
hipError_t rows_gemm_group(const std::vector<RowsGemmArgs> &calls, RowsGemmGroup **cache, hipStream_t s) {
  if (calls.empty()) return hipSuccess;
  if (!*cache) *cache = new RowsGemmGroup();
  RowsGemmGroup &g = **cache;
  std::vector<RowsGemmArgs> prep(calls.size());
  std::vector<int> first(calls.size() + 1, 0);
  double flops = 0, bytes = 0;
  for (size_t i = 0; i < calls.size(); i++) {
    RowsGemmArgs a = calls[i];
    if (!rows_gemm_group_ok(a)) return hipErrorInvalidValue;
    a.c_vec = aligned16(a.C) && a.ldc % 4 == 0 && (a.init_mode != 1 || aligned16(a.bias));
    a.sumsq_cap = rows_gemm_sumsq_blocks(a.M);
    a.serial_epilogue = 0;
    a.ksplit = 0;
    a.partial = nullptr;
    a.prec = 0;
    a.colstats = nullptr;
    a.colstats_rows = nullptr;
    a.alt_seg_order = options().gemm_alt_taps && a.nseg == 2 && a.seg[0].klen == a.seg[1].klen && a.lda > 0 && a.seg[0].a_off != a.seg[1].a_off &&
                      (a.seg[1].a_off - a.seg[0].a_off) % a.lda == 0;
    prep[i] = a;
    first[i + 1] = first[i] + (a.M + 127) / 128;
    double kt = 0;
    for (int j = 0; j < a.nseg; j++) kt += a.seg[j].klen;
    flops += 2.0 * a.M * a.N * kt;
    bytes += 4.0 * ((double)a.M * kt + kt * a.N + (double)a.M * a.N);
  }
  const bool same = g.args.size() == prep.size() && memcmp(g.args.data(), prep.data(), sizeof(RowsGemmArgs) * prep.size()) == 0 && g.first == first;
  if (!same) {  // (fixed shapes and buffers: uploaded once; the copy is ordered on the launch's stream)
    if (g.capacity < (int)prep.size()) {
      if (g.d_args) hipFree(g.d_args);
      if (g.d_first) hipFree(g.d_first);
      hipError_t e = hipMalloc((void **)&g.d_args, sizeof(RowsGemmArgs) * prep.size());
      if (e != hipSuccess) return e;
      e = hipMalloc((void **)&g.d_first, sizeof(int) * (prep.size() + 1));
      if (e != hipSuccess) return e;
      g.capacity = (int)prep.size();
    }
    g.args = prep;
    g.first = first;
    hipError_t e = hipMemcpyAsync(g.d_args, g.args.data(), sizeof(RowsGemmArgs) * prep.size(), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(g.d_first, g.first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
  }
  constexpr int BK = 32;
  const size_t lds = sizeof(float) * 2 * (128 * (BK + 4) + 32 * (BK + 4));
  ProfGemmRange prof(3, flops, bytes, s);
  RowsGemmTasks t{g.d_args, g.d_first, (int)prep.size()};
  hipLaunchKernelGGL((rows_gemm_group_kernel<4, 1, 1, 1, BK, true, 4>), dim3(first.back()), dim3(256), lds, s, t);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- wgrad
namespace {

// A = dY (k = row, m = output dim contiguous), B = X_tap (k = row, n = input dim contiguous).
template <int WM, int WN, int TM, int TN, int VEC, int TAG = 0>  // TAG: as for rows_gemm_kernel
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p, int ntm, int ntn_tap, int rows_per_split,
                                                    float *partial) {
  constexpr int BK = 32;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDAS = BM + 4, LDBS = BN + 4;
  constexpr int A_TILE = BK * LDAS, B_TILE = BK * LDBS;
  constexpr int A_F4 = (BM * BK / 4 + 255) / 256, B_F4 = (BN * BK / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem, *Bs = smem + 2 * A_TILE;

  // Block -> (row split, tile, tap).  The taps of a component are row shifts of ONE matrix, so the blocks that differ only in the
  // tap stream the same slab of the big operand (X for the .linear components, dY for the .affine ones): they are given
  // neighbouring logical ids, and every XCD (workgroups are dealt to the eight XCDs round-robin) a contiguous run of logical ids,
  // so that a slab's second reader finds it in the L2 the first one filled instead of fetching it over the fabric again.
  const int tiles = ntm * (int)gridDim.x / ntm;  // = gridDim.x: ntm * taps launched * ntn_tap
  int bid = blockIdx.x, split = blockIdx.y;
  int tile_m, tile_n, tap;
  if (p.xcd_order) {
    const int nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nb / 8, r = nb % 8, xcd = b % 8, j = b / 8;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int ktaps = (int)gridDim.x / (ntm * ntn_tap);
    split = L / tiles;
    const int w = L % tiles;
    tap = w % ktaps;
    tile_m = (w / ktaps) % ntm;
    tile_n = (w / ktaps) / ntm;
  } else {
    tile_m = bid % ntm;
    tap = (bid / ntm) / ntn_tap;
    tile_n = (bid / ntm) % ntn_tap;
  }
  if (p.active) {  // compacted tap list: slots beyond the active count have nothing to do
    if (tap >= p.active[0]) return;
    tap = p.active[1 + tap];
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int r_begin = split * rows_per_split;
  const int r_end = min(p.N, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const float cf = p.coef ? p.coef[tap] : 1.f;
  const float *Xb = p.X + (long long)p.row_offsets[tap] * p.ldx;
  const long long xrow = (long long)p.row_stride * p.ldx;

  float4 ra[A_F4], rb[B_F4];
  // Fast path (whole K-step inside the split, 16-byte accesses, Do and Di multiples of 4): per-thread source pointers
  // set up once; a column group that is out of range reads 16 zero bytes with step 0 instead of branching, which
  // keeps the steady-state loop a single basic block (accumulators stay in AGPRs).
  const bool fast_ok = VEC == 4 && p.Do % 4 == 0 && p.Di % 4 == 0;
  const float *aptr[A_F4], *bptr[B_F4];
  long long astep[A_F4], bstep[B_F4];
  {
    const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j, kr = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
      const bool v = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && m < p.Do;
      aptr[j] = v ? p.dY + (long long)kr * p.lddy + m : zero;
      astep[j] = v ? p.lddy : 0;
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j, kr = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
      const bool v = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && n < p.Di;
      bptr[j] = v ? Xb + (long long)kr * xrow + n : zero;
      bstep[j] = v ? xrow : 0;
    }
  }
  auto load_tile = [&](int r0) {
    if (fast_ok && r0 + BK <= r_end) {
#pragma unroll
      for (int j = 0; j < A_F4; j++) ra[j] = *reinterpret_cast<const float4 *>(aptr[j] + (long long)r0 * astep[j]);
#pragma unroll
      for (int j = 0; j < B_F4; j++) rb[j] = *reinterpret_cast<const float4 *>(bptr[j] + (long long)r0 * bstep[j]);
      return;
    }
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      const int kr = idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
      const bool rv = (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4) && r0 + kr < r_end;
      const float *ptr = p.dY + (long long)(r0 + kr) * p.lddy + m;
      ra[j] = ld4(ptr, rv && m < p.Do, rv && m + 1 < p.Do, rv && m + 2 < p.Do, rv && m + 3 < p.Do, VEC == 4);
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      const int kr = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
      const bool rv = (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4) && r0 + kr < r_end;
      const float *ptr = Xb + (long long)(r0 + kr) * xrow + n;
      rb[j] = ld4(ptr, rv && n < p.Di, rv && n + 1 < p.Di, rv && n + 2 < p.Di, rv && n + 3 < p.Di, VEC == 4);
    }
  };
  auto store_tile = [&](int buf) {
    float *as = As + buf * A_TILE, *bs = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < A_F4; j++) {
      const int idx = t + 256 * j;
      if (BM * BK / 4 % 256 == 0 || idx < BM * BK / 4)
        *reinterpret_cast<float4 *>(as + (idx / (BM / 4)) * LDAS + (idx % (BM / 4)) * 4) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_F4; j++) {
      const int idx = t + 256 * j;
      if (BN * BK / 4 % 256 == 0 || idx < BN * BK / 4)
        *reinterpret_cast<float4 *>(bs + (idx / (BN / 4)) * LDBS + (idx % (BN / 4)) * 4) = rb[j];
    }
  };
  auto compute = [&](int buf) {
    const float *as = As + buf * A_TILE + lh * LDAS + wm * TM * 32 + li;
    const float *bs = Bs + buf * B_TILE + lh * LDBS + wn * TN * 32 + li;
    // fragments of k-pair k2 + 1 are requested before the MFMAs of k-pair k2 are issued (two register sets), so the
    // LDS latency hides behind TM * TN MFMAs instead of stalling every other one
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) a[0][i] = as[i * 32];
#pragma unroll
    for (int i = 0; i < TN; i++) b[0][i] = bs[i * 32];
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; k2++) {
      const int cur = k2 & 1, nxt = cur ^ 1;
      if (k2 + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[nxt][i] = as[(2 * k2 + 2) * LDAS + i * 32];
#pragma unroll
        for (int i = 0; i < TN; i++) b[nxt][i] = bs[(2 * k2 + 2) * LDBS + i * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
  };

  // partial slab [split][Do][K*Di]
  float *P = partial + (long long)split * p.Do * (p.K * p.Di);
  if (cf == 0.f) return;  // skipped tap: the reduce kernel does not read its slab
  if (r_begin >= r_end) {  // (cannot happen with the host's split plan; keep the slab defined anyway)
    for (int e = t; e < BM * BN; e += 256) {
      const int m = m0 + e / BN, n = n0 + e % BN;
      if (m < p.Do && n < p.Di) P[(long long)m * (p.K * p.Di) + tap * p.Di + n] = 0.f;
    }
    return;
  }
  // straight-line prologue / loop / epilogue: with the loop under a condition the register allocator kept the 64-80
  // accumulators in VGPRs across the back edge and copied them to AGPRs every K-step (288 registers, one wave per SIMD)
  load_tile(r_begin);
  store_tile(0);
  __syncthreads();
  {
    int buf = 0;
    for (int r0 = r_begin;; r0 += BK) {
      const bool more = r0 + BK < r_end;
      if (more) load_tile(r0 + BK);
      compute(buf);
      if (!more) break;
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + (wn * TN + j) * 32 + li;
      if (n >= p.Di) continue;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.Do) P[(long long)m * (p.K * p.Di) + tap * p.Di + n] = acc[i][j][r];
      }
    }
}

// The weight gradient in split-bf16 arithmetic (see rows_gemm_x3_kernel).  Both operands are reduced over rows, so a
// fragment needs 8 consecutive ROWS of one column: every thread loads one column of 8 rows (dword loads, consecutive
// lanes on consecutive columns: coalesced), splits it and writes one 16-byte [column][k] piece per plane.
template <int WM, int WN, int TM, int TN, int NP, int D, int TAG = 0>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(const WgradArgs p, int ntm, int ntn_tap, int rows_per_split, float *partial) {
  constexpr int BK = NP == 2 ? 32 : 16, LDH = BK + 8;  // three planes: half the K-step, the same LDS budget
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_IT = (BM * BK / 8 + 255) / 256, B_IT = (BN * BK / 8 + 255) / 256;  // (column, 8-row group) items per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *As = reinterpret_cast<__bf16 *>(smem);       // [2 buffers][NP planes][BM][LDH]
  __bf16 *Bs = As + 2 * NP * BM * LDH;                 // [2 buffers][NP planes][BN][LDH]

  int bid = blockIdx.x;
  const int tile_m = bid % ntm;
  int tap = (bid / ntm) / ntn_tap;
  const int tile_n = (bid / ntm) % ntn_tap;
  if (p.active) {
    if (tap >= p.active[0]) return;
    tap = p.active[1 + tap];
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const int r_begin = split * rows_per_split;
  const int r_end = min(p.N, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const float cf = p.coef ? p.coef[tap] : 1.f;
  const float *Xb = p.X + (long long)p.row_offsets[tap] * p.ldx;
  const long long xrow = (long long)p.row_stride * p.ldx;

  float ra[D][A_IT][8], rb[D][B_IT][8];  // D staged K-steps (see rows_gemm_x3_kernel)
  // per-item source: column pointer at row 0 of the item's 8-row group; an out-of-range column reads zeros (step 0)
  const float *aptr[A_IT], *bptr[B_IT];
  long long astep[A_IT], bstep[B_IT];
  int akg[A_IT], bkg[B_IT];
  {
    const float *zero = reinterpret_cast<const float *>(&g_zero4);
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
      const int item = t + 256 * j, col = item % BM, kg = item / BM;
      const bool v = (BM * BK / 8 % 256 == 0 || item < BM * BK / 8) && m0 + col < p.Do;
      akg[j] = kg;
      aptr[j] = v ? p.dY + (long long)(kg * 8) * p.lddy + m0 + col : zero;
      astep[j] = v ? p.lddy : 0;
    }
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
      const int item = t + 256 * j, col = item % BN, kg = item / BN;
      const bool v = (BN * BK / 8 % 256 == 0 || item < BN * BK / 8) && n0 + col < p.Di;
      bkg[j] = kg;
      bptr[j] = v ? Xb + (long long)(kg * 8) * xrow + n0 + col : zero;
      bstep[j] = v ? xrow : 0;
    }
  }
  // whole K-step inside the split: unconditional loads (a branch-free steady state, see rows_gemm_x3_kernel)
  auto load_fast = [&](float (&ra)[A_IT][8], float (&rb)[B_IT][8], int r0) {
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
      const float *q = aptr[j] + (long long)r0 * astep[j];
#pragma unroll
      for (int i = 0; i < 8; i++) ra[j][i] = q[(long long)i * astep[j]];
    }
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
      const float *q = bptr[j] + (long long)r0 * bstep[j];
#pragma unroll
      for (int i = 0; i < 8; i++) rb[j][i] = q[(long long)i * bstep[j]];
    }
  };
  auto load_ragged = [&](float (&ra)[A_IT][8], float (&rb)[B_IT][8], int r0) {  // the split's last, partial K-step
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
      const float *q = aptr[j] + (long long)r0 * astep[j];
#pragma unroll
      for (int i = 0; i < 8; i++) ra[j][i] = r0 + akg[j] * 8 + i < r_end ? q[(long long)i * astep[j]] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
      const float *q = bptr[j] + (long long)r0 * bstep[j];
#pragma unroll
      for (int i = 0; i < 8; i++) rb[j][i] = r0 + bkg[j] * 8 + i < r_end ? q[(long long)i * bstep[j]] : 0.f;
    }
  };
  auto split8 = [](const float *x, bf16x8 (&pl)[NP]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float r = x[i];
#pragma unroll
      for (int q = 0; q < NP; q++) {
        const __bf16 h = (__bf16)r;
        pl[q][i] = h;
        r -= (float)h;
      }
    }
  };
  auto store_tile = [&](const float (&ra)[A_IT][8], const float (&rb)[B_IT][8], int buf) {
    __bf16 *ah = As + buf * NP * BM * LDH, *bh = Bs + buf * NP * BN * LDH;
#pragma unroll
    for (int j = 0; j < A_IT; j++) {
      const int item = t + 256 * j;
      if (BM * BK / 8 % 256 == 0 || item < BM * BK / 8) {
        bf16x8 pl[NP];
        split8(ra[j], pl);
        const int o = (item % BM) * LDH + (item / BM) * 8;
#pragma unroll
        for (int q = 0; q < NP; q++) *reinterpret_cast<bf16x8 *>(ah + q * BM * LDH + o) = pl[q];
      }
    }
#pragma unroll
    for (int j = 0; j < B_IT; j++) {
      const int item = t + 256 * j;
      if (BN * BK / 8 % 256 == 0 || item < BN * BK / 8) {
        bf16x8 pl[NP];
        split8(rb[j], pl);
        const int o = (item % BN) * LDH + (item / BN) * 8;
#pragma unroll
        for (int q = 0; q < NP; q++) *reinterpret_cast<bf16x8 *>(bh + q * BN * LDH + o) = pl[q];
      }
    }
  };
  auto compute = [&](int buf) {
    const __bf16 *ah = As + buf * NP * BM * LDH + (wm * TM * 32 + li) * LDH + lh * 8;
    const __bf16 *bh = Bs + buf * NP * BN * LDH + (wn * TN * 32 + li) * LDH + lh * 8;
#pragma unroll
    for (int c = 0; c < BK / 16; c++) {
      bf16x8 a[NP][TM], b[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[q][i] = *reinterpret_cast<const bf16x8 *>(ah + q * BM * LDH + i * 32 * LDH + c * 16);
#pragma unroll
        for (int i = 0; i < TN; i++) b[q][i] = *reinterpret_cast<const bf16x8 *>(bh + q * BN * LDH + i * 32 * LDH + c * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
#pragma unroll
          for (int d = NP - 1; d >= 0; d--)
#pragma unroll
            for (int q = 0; q <= d; q++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[d - q][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  float *P = partial + (long long)split * p.Do * (p.K * p.Di);
  if (cf == 0.f) return;
  if (r_begin >= r_end) {
    for (int e = t; e < BM * BN; e += 256) {
      const int m = m0 + e / BN, n = n0 + e % BN;
      if (m < p.Do && n < p.Di) P[(long long)m * (p.K * p.Di) + tap * p.Di + n] = 0.f;
    }
    return;
  }
  {
    bool have[D];
    int rnext = r_begin;  // first row of the next K-step to stage
    int buf = 0;
    static_for<D>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      have[u] = rnext + BK <= r_end;
      if (have[u]) {
        load_fast(ra[u], rb[u], rnext);
        rnext += BK;
      }
    });
    if (have[0]) {
      store_tile(ra[0], rb[0], 0);
      __syncthreads();
      if (have[D - 1]) {
        // steady state: LDS buffer `buf` holds the step staged in slot u, slots u+1 .. u+D-1 the next D - 1 steps
        bool run = true;
        while (run) {
          static_for<D>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (!run) return;
            if (rnext + BK > r_end) {  // no full step left to stage: drain
              static_for<D - 1>([&](auto jc) {
                constexpr int nx = (u + 1 + decltype(jc)::value) % D;
                compute(buf);
                store_tile(ra[nx], rb[nx], buf ^ 1);
                __syncthreads();
                buf ^= 1;
              });
              compute(buf);
              run = false;
              return;
            }
            load_fast(ra[u], rb[u], rnext);
            rnext += BK;
            compute(buf);
            constexpr int nx = (u + 1) % D;
            store_tile(ra[nx], rb[nx], buf ^ 1);
            __syncthreads();
            buf ^= 1;
          });
        }
      } else {  // fewer than D full steps
        bool run = true;
        static_for<D>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          if (!run) return;
          compute(buf);
          if constexpr (u + 1 >= D) {
            run = false;
          } else {
            if (!have[u + 1]) {
              run = false;
              return;
            }
            store_tile(ra[u + 1], rb[u + 1], buf ^ 1);
            __syncthreads();
            buf ^= 1;
          }
        });
      }
    }
    if (rnext < r_end) {  // ragged end; buffer buf ^ 1 is free (last read before the last barrier)
      load_ragged(ra[0], rb[0], rnext);
      store_tile(ra[0], rb[0], buf ^ 1);
      __syncthreads();
      compute(buf ^ 1);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + (wn * TN + j) * 32 + li;
      if (n >= p.Di) continue;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.Do) P[(long long)m * (p.K * p.Di) + tap * p.Di + n] = acc[i][j][r];
      }
    }
}

// G[o][c] (+)= scale * coef[tap(c)] * sum_split partial[split][o][c]
__global__ void wgrad_reduce_kernel(const float *partial, int splits, int Do, int KDi, int Di, const float *coef,
                                    float scale, float *G, long long ldg, int accumulate, const float *ds1 = nullptr, const float *ds2 = nullptr) {
  if (ds1) scale *= ds1[1] * ds2[1];  // plane operands: the reciprocals of the scales they were split with
  const long long total = (long long)Do * KDi;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(e / KDi), c = (int)(e % KDi);
    float s = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < splits; sp++) s += partial[(long long)sp * total + e];
    const float cf = coef ? coef[c / Di] : 1.f;
    if (cf == 0.f) continue;  // skipped tap: its slab may not have been written
    float *g = G + (long long)o * ldg + c;
    const float v = scale * cf * s;
    *g = accumulate ? *g + v : v;
  }
}

// The same for small outputs (the R x R products of the natural-gradient statistics: one tile, hundreds of row
// splits): 16 split groups per element so the serial chain is splits/16 loads long.
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float *partial, int splits, int Do, int KDi, int Di, const float *coef,
                                                                 float scale, float *G, long long ldg, int accumulate) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long long total = (long long)Do * KDi, e = blockIdx.x * 16LL + el;
  float s = 0.f;
  if (e < total)
    for (int sp = grp; sp < splits; sp += 16) s += partial[(long long)sp * total + e];
  red[grp][el] = s;
  __syncthreads();
  if (grp != 0 || e >= total) return;
  s = 0.f;
#pragma unroll
  for (int g = 0; g < 16; g++) s += red[g][el];
  const int o = (int)(e / KDi), c = (int)(e % KDi);
  const float cf = coef ? coef[c / Di] : 1.f;
  if (cf == 0.f) return;
  float *g = G + (long long)o * ldg + c;
  const float v = scale * cf * s;
  *g = accumulate ? *g + v : v;
}

// column sums of dY in two deterministic stages: partial[chunk][col] then bias_acc[col] += scale*sum
__global__ void colsum_partial_kernel(const float *Y, long long ld, int rows, int cols, int rows_per_chunk,
                                      float *partial) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  if (col >= cols) return;
  float s = 0.f;
  for (int r = r0; r < r1; r++) s += Y[(long long)r * ld + col];
  partial[(long long)blockIdx.y * cols + col] = s;
}
__global__ void colsum_final_kernel(const float *partial, int chunks, int cols, float scale, float *acc) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  float s = 0.f;
  for (int c = 0; c < chunks; c++) s += partial[(long long)c * cols + col];
  acc[col] += scale * s;
}

struct WgradPlan {
  int splits, rows_per_split, chunks, rows_per_chunk;
  size_t slab_floats, colsum_floats;
};

// tile shape of the weight-gradient kernel: 160-wide variants for the TDNN-F bottleneck dimension
struct WgradTile {
  int BM, BN, variant;  // variant 0: 128x128, 1: 160x128 (Do == 160-ish), 2: 128x160 (Di == 160-ish), 3: 32x128 (Do <= 32)
};
inline int waste_of(int n, int t) { return ((n + t - 1) / t) * t - n; }
WgradTile wgrad_tile(int Do, int Di, bool x3 = false, int N = 1 << 30) {
  if (Do <= 32) return {32, 128, 3};
  // few rows (the recipes' minibatch: 3 000 - 10 000): with the 160 x 128 / 128 x 160 tiles the reduction is so short that a block is mostly
  // prologue and a 80 KB partial tile (13 slabs of 256 rows: 25 MB of partials for 43 MB of operands); 64 x 64 tiles give six times the
  // tiles, so two or three slabs fill the chip (option wgrad_small)
  if (!x3 && options().wgrad_small && N <= options().wgrad_small) return {64, 64, 4};
  // split-bf16 arithmetic is not MFMA bound: the 160-wide tiles (92 KiB of LDS in bf16 planes, one block per CU) lose to
  // plain 128x128 tiles with a few wasted columns
  if (x3) return {128, 128, 0};  // J = H^T X of a rank <= 32 preconditioner: HBM-bound on X, no wasted MFMA rows
  if (waste_of(Do, 160) * 128 < waste_of(Do, 128) * 160 && waste_of(Do, 160) < waste_of(Do, 128)) return {160, 128, 1};
  if (waste_of(Di, 160) < waste_of(Di, 128)) return {128, 160, 2};
  return {128, 128, 0};
}

// resident wgrad blocks on the whole chip (blocks/CU from the occupancy API x CUs), queried once per variant
template <int WM, int WN, int TM, int TN>
int wgrad_slots_of() {
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 256, occ = 2;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    const size_t lds = sizeof(float) * 2 * (32 * (WM * TM * 32 + 4) + 32 * (WN * TN * 32 + 4));
    hipFuncSetAttribute((const void *)wgrad_kernel<WM, WN, TM, TN, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)wgrad_kernel<WM, WN, TM, TN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)wgrad_kernel<WM, WN, TM, TN, 4>, 256, lds) != hipSuccess || occ < 1) occ = 2;
    slots = std::min(occ * cus, 1024);  // wgrad_workspace_bytes() assumes at most 1024 resident blocks
    (void)hipGetLastError();
  }
  return slots;
}
int wgrad_slots(int variant) {
  if (variant == 4) return wgrad_slots_of<2, 2, 1, 1>();
  return variant == 1 ? wgrad_slots_of<1, 4, 5, 1>() : variant == 2 ? wgrad_slots_of<4, 1, 1, 5>() : variant == 3 ? wgrad_slots_of<1, 4, 1, 1>()
                                                                                                   : wgrad_slots_of<2, 2, 2, 2>();
}

inline int wgrad_min_rounds() { return 2; }

// Split the row (reduction) range so that tiles * splits fills whole rounds of resident blocks: every block
// runs equally long, so a grid of q*slots + r blocks costs q+1 rounds; we want r == 0 (just under a multiple).
WgradPlan wgrad_plan(int Do, int Di, int K, int N, int slots, int ktaps = 0, bool x3 = false) {
  if (ktaps > 0) { WgradPlan p2 = wgrad_plan(Do, Di, ktaps, N, slots, 0, x3); p2.slab_floats = (size_t)p2.splits * Do * K * Di; return p2; }
  WgradPlan pl;
  const WgradTile wt = wgrad_tile(Do, Di, x3, N);
  const int tiles = ((Do + wt.BM - 1) / wt.BM) * K * ((Di + wt.BN - 1) / wt.BN);
  const int max_splits = std::max(1, (N + 255) / 256);  // at least 256 rows per split
  int splits = 1;
  // at least two rounds of resident blocks: a launch sized for exactly one round runs two as soon as a few slots are taken by
  // another stream's kernels (the denominator, the natural-gradient side stream); with shorter blocks the stragglers cost half
  // (weight-gradient class 30.7 -> 29.1 ms per step; TDNNF_WGRAD_ROUNDS for experiments)
  const int min_rounds = wgrad_min_rounds();
  for (int rounds = min_rounds; rounds <= std::max(4, min_rounds); rounds++) {
    splits = (rounds * slots) / tiles;
    if (splits >= 1 && (rounds * slots) / tiles * tiles >= rounds * slots * 3 / 4) break;  // >= 75 % of the round used
  }
  if (wt.variant == 4) splits = std::max(1, (2 * device_cus()) / tiles);  // two blocks per CU, long slabs
  if (splits < 1) splits = 1;
  if (splits > max_splits) splits = max_splits;
  int rps = (N + splits - 1) / splits;
  rps = ((rps + 31) / 32) * 32;
  pl.splits = (N + rps - 1) / rps;
  pl.rows_per_split = rps;
  pl.slab_floats = (size_t)pl.splits * Do * K * Di;
  pl.rows_per_chunk = 512;
  pl.chunks = (N + 511) / 512;
  pl.colsum_floats = (size_t)pl.chunks * Do;
  return pl;
}

}  // namespace

size_t wgrad_workspace_bytes(int Do, int Di, int K, int N) {
  // sized for the largest split count any device can ask for (4 rounds of 8 blocks on 304 CUs), so the
  // answer does not depend on the GPU being present
  const WgradTile wt0 = wgrad_tile(Do, Di, false), wt1 = wgrad_tile(Do, Di, true);
  const int tiles = std::min(((Do + wt0.BM - 1) / wt0.BM) * ((Di + wt0.BN - 1) / wt0.BN),
                             ((Do + wt1.BM - 1) / wt1.BM) * ((Di + wt1.BN - 1) / wt1.BN));  // worst case: one active tap, either arithmetic
  // wgrad_plan picks one round when tiles <= slots/4 (splits = slots/tiles) and at most 4 rounds otherwise (< 16 splits);
  // slots <= 1024 on any gfx950 part
  // (with r = wgrad_min_rounds(): r rounds when tiles <= r slots / 4, i.e. splits = r slots / tiles, else < 16 splits)
  size_t max_splits = std::max<size_t>(1, std::min<size_t>((N + 255) / 256, std::max<size_t>((size_t)std::max(wgrad_min_rounds(), 4) * 1024 / tiles + 1, 16)));
  return sizeof(float) * (max_splits * Do * K * Di) + colreduce_bytes(N, Do) + 64;
}

static hipError_t wgrad_finish(const WgradArgs &a, const float *partial, int splits, float *cs_partial, const float *ds1, const float *ds2, hipStream_t s);

// The weight gradient on the pre-split plane kernels: G = dY^T X_tap reduces over ROWS, so its operands are the transposed planes of
// dY and X (k = row); the taps are K-block offsets of X.  The operand with more columns gives the tile rows (Do < Di: the transposed
// product, stored through strides), the rows are split over the CUs into slabs that wgrad_finish() adds.  false = not applicable.
static bool planes_try_wgrad(const WgradArgs &a, void *workspace, size_t workspace_bytes, int np, hipStream_t s, hipError_t *err) {
  const PlanesOperand *hy = planes_hint_a(), *hx = planes_hint_b();
  if (!hy || !hx || hy->np != np || hx->np != np) return false;
  if (a.row_stride != 1 || a.K > 16 || a.K < 1) return false;  // (tap coefficients: applied by the reduce, zero ones skipped in the kernel -- the compacted tap list is not needed)
  if (a.dY != hy->base || a.lddy != hy->ld || a.N != hy->rows || a.Do != hy->cols) return false;
  if (a.ldx != hx->ld || a.X < hx->base || (a.X - hx->base) % hx->ld != 0 || a.Di != hx->cols) return false;
  const long long xrow0 = (a.X - hx->base) / hx->ld;
  const int nkb = (a.N + 15) / 16;
  if (nkb < 16) return false;  // (too few rows to split over the chip: the f32 kernels)
  const bool normal = a.Do >= a.Di;  // the operand with more columns gives the tile rows
  const int M = normal ? a.Do : a.Di, Nn = normal ? a.Di : a.Do;
  const int BM = planes_gemm_tile_rows(Nn), BN = planes_gemm_tile_cols(Nn);
  const int ntm = (M + BM - 1) / BM, ntn = (Nn + BN - 1) / BN;
  const PlanesOperand *hm = normal ? hy : hx, *hn = normal ? hx : hy;  // operands giving the tile rows / columns
  if (!hn->PT || (long long)ntn * BN > hn->Rt) return false;
  for (int i = 0; i < a.K; i++)
    if (a.row_offsets[i] < 0 || a.row_offsets[i] % 16 != 0 || xrow0 + a.row_offsets[i] + a.N > hx->rows) return false;
  PlanesGemmArgs g;
  memset(&g, 0, sizeof(g));
  // the tile-row operand: its row-major planes through transposing LDS reads when they are there (no planes of the transpose needed
  // for the big matrix), else its transposed planes
  const long long m_first = hm->lead + (normal ? 0 : xrow0);  // first matrix row of the K range in the row-major buffer
  const bool atr = hm->P && m_first % 16 == 0 && (long long)ntm * (BM / 16) <= hm->kb_alloc && m_first + 16LL * nkb + (normal ? 0 : a.row_offsets[a.K - 1]) <= hm->R;
  if (atr) {
    g.A = hm->P; g.RA = hm->R; g.a_rows_as_k = 1;
    g.seg[0].a_row = m_first;
  } else {
    if (!hm->PT || (long long)ntm * BM > hm->Rt || (!normal && xrow0 % 16 != 0)) return false;
    g.A = hm->PT; g.RA = hm->Rt;
  }
  for (int i = 0; i < a.K; i++) {
    if (normal) g.tap_b_kb[i] = (int)((xrow0 + a.row_offsets[i]) / 16);
    else g.tap_a_kb[i] = (int)((a.row_offsets[i] + (atr ? 0 : xrow0)) / 16);
  }
  if (normal && (xrow0 % 16 != 0)) return false;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    (void)hipGetLastError();
  }
  const int tiles = ntm * ntn * a.K;
  const size_t slab = (size_t)a.Do * a.K * a.Di;
  int splits = std::max(2, (2 * cus) / tiles);  // two rounds of one block per CU
  splits = std::min(splits, nkb / 8);
  splits = (int)std::min<size_t>((size_t)splits, workspace_bytes / sizeof(float) / slab);
  if (splits < 2) return false;
  const int kbps = (nkb + splits - 1) / splits;
  splits = (nkb + kbps - 1) / kbps;
  if (splits < 2) return false;
  if (sizeof(float) * slab * splits + colreduce_bytes(a.N, a.Do) > workspace_bytes) return false;
  g.np = np;
  g.B = hn->PT; g.RB = hn->Rt;
  g.M = M; g.N = Nn;
  g.nseg = 1;
  g.seg[0].nkb = nkb;
  g.ntap = a.K;
  g.ksplit = splits; g.kb_per_split = kbps;
  g.partial = reinterpret_cast<float *>(workspace);
  g.partial_stride = (long long)slab;
  g.tap_off_p = a.Di;
  g.skip_coef = a.coef;
  if (normal) { g.ldp_m = (long long)a.K * a.Di; g.ldp_n = 1; }
  else { g.ldp_m = 1; g.ldp_n = (long long)a.K * a.Di; }
  {
    if (g_prof_on) {
      g_prof_next_flops = 2.0 * a.N * a.Do * a.K * a.Di;
      g_prof_next_bytes = 4.0 * ((double)a.N * a.Do + ((double)a.N + a.row_offsets[a.K - 1] - a.row_offsets[0]) * a.Di + (double)a.Do * a.K * a.Di * (a.accumulate ? 2.0 : 1.0));
    }
    ProfScope ps(2, 2.0 * a.N * a.Do * a.K * a.Di, s);
    *err = planes_gemm(g, s);
  }
  if (*err != hipSuccess) return true;
  *err = wgrad_finish(a, g.partial, splits, g.partial + slab * splits, hy->scale, hx->scale, s);
  g_planes_routed_wgrad++;
  return true;
}

hipError_t wgrad(const WgradArgs &a, void *workspace, size_t workspace_bytes, hipStream_t s) {
  if (a.N <= 0 || a.Do <= 0 || a.Di <= 0) return hipSuccess;
  if (workspace_bytes < wgrad_workspace_bytes(a.Do, a.Di, a.K, a.N)) return hipErrorInvalidValue;
  int planes = 0;  // 0: f32 MFMA; 2 / 3: split-bf16 with that many planes per operand
  {
    int prec = a.prec;
    if (prec == 0) prec = g_gemm_prec;
    if (prec == 4 || (prec == 3 && options().planes)) {  // pre-split planes when the caller hinted them for these operands
      hipError_t pe = hipSuccess;
      if (planes_try_wgrad(a, workspace, workspace_bytes, prec == 4 ? 2 : 3, s, &pe)) return pe;
    }
    planes = prec == 1 ? 2 : prec == 3 ? 3 : 0;
  }
  const bool use_x3 = planes != 0;
  const WgradTile wt = wgrad_tile(a.Do, a.Di, use_x3, a.N);
  const int ktaps = a.active && a.max_active > 0 && a.max_active < a.K ? a.max_active : a.K;
  WgradArgs a_x = a;
  a_x.xcd_order = ktaps > 1 ? 1 : 0;  // the taps of a tile side by side on one XCD
  WgradPlan pl = wgrad_plan(a.Do, a.Di, a.K, a.N, wgrad_slots(wt.variant), ktaps == a.K ? 0 : ktaps, use_x3);
  if (sizeof(float) * pl.slab_floats > workspace_bytes) return hipErrorInvalidValue;
  float *partial = reinterpret_cast<float *>(workspace);
  float *cs_partial = partial + pl.slab_floats;
  const bool vec = aligned16(a.dY) && aligned16(a.X) && a.lddy % 4 == 0 && a.ldx % 4 == 0;
  const int ntm = (a.Do + wt.BM - 1) / wt.BM, ntn = (a.Di + wt.BN - 1) / wt.BN;
  dim3 grid(ntm * ktaps * ntn, pl.splits), block(256);
  const size_t lds = sizeof(float) * 2 * (32 * (wt.BM + 4) + 32 * (wt.BN + 4));
  {
    ProfFlopsScale exact(ktaps != a.K ? 1.0 : g_prof_flops_scale);  // a compacted launch already counts only its taps
    if (g_prof_on) {  // algorithmic bytes: dY once, the distinct input rows once, the gradient block written (and read)
      int lo = a.row_offsets[0], hi = a.row_offsets[0];
      for (int i = 1; i < a.K; i++) {
        lo = std::min(lo, a.row_offsets[i]);
        hi = std::max(hi, a.row_offsets[i]);
      }
      g_prof_next_flops = 2.0 * a.N * a.Do * ktaps * a.Di;
      g_prof_next_bytes = 4.0 * ((double)a.N * a.Do + std::min((double)a.N * ktaps, (double)a.N * a.row_stride + (hi - lo)) * a.Di +
                                 (double)a.Do * ktaps * a.Di * (a.accumulate ? 2.0 : 1.0));
    }
    ProfScope ps(2, 2.0 * a.N * a.Do * ktaps * a.Di, s);
#define WG_LAUNCH_T(WM, WN, TM, TN, TAG)                                                                                            \
  {                                                                                                                                \
    static bool attr_done = false;                                                                                                 \
    if (!attr_done) {                                                                                                              \
      hipFuncSetAttribute((const void *)wgrad_kernel<WM, WN, TM, TN, 4, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipFuncSetAttribute((const void *)wgrad_kernel<WM, WN, TM, TN, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_done = true;                                                                                                            \
    }                                                                                                                              \
    if (vec) hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, 4, TAG>), grid, block, lds, s, a_x, ntm, ntn, pl.rows_per_split, partial); \
    else hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, 1, TAG>), grid, block, lds, s, a_x, ntm, ntn, pl.rows_per_split, partial);     \
  }
#define WG_LAUNCH_X3(WM, WN, TM, TN, NP, TAG)                                                                                  \
  {                                                                                                                             \
    constexpr size_t lds3 = sizeof(__bf16) * 2 * NP * (size_t)(WM * TM * 32 + WN * TN * 32) * (NP == 2 ? 40 : 24);              \
    static bool attr_done3 = false;                                                                                             \
    if (!attr_done3) {                                                                                                          \
      hipFuncSetAttribute((const void *)wgrad_x3_kernel<WM, WN, TM, TN, NP, 1, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
      attr_done3 = true;                                                                                                        \
    }                                                                                                                           \
    hipLaunchKernelGGL((wgrad_x3_kernel<WM, WN, TM, TN, NP, 1, TAG>), grid, block, lds3, s, a, ntm, ntn, pl.rows_per_split, partial);  \
  }
#define WG_LAUNCH(WM, WN, TM, TN)                                  \
  if (planes == 2) {                                               \
    if (g_prof_override == 3) WG_LAUNCH_X3(WM, WN, TM, TN, 2, 1)   \
    else WG_LAUNCH_X3(WM, WN, TM, TN, 2, 0)                        \
  } else if (planes == 3) {                                        \
    if (g_prof_override == 3) WG_LAUNCH_X3(WM, WN, TM, TN, 3, 1)   \
    else WG_LAUNCH_X3(WM, WN, TM, TN, 3, 0)                        \
  } else if (g_prof_override == 3) WG_LAUNCH_T(WM, WN, TM, TN, 1)  \
  else WG_LAUNCH_T(WM, WN, TM, TN, 0)
#define WG_LAUNCH_F32(WM, WN, TM, TN)                              \
  if (g_prof_override == 3) WG_LAUNCH_T(WM, WN, TM, TN, 1)         \
  else WG_LAUNCH_T(WM, WN, TM, TN, 0)
    // (wgrad_tile() gives the split-bf16 arithmetic the 128x128 and 32x128 tiles only: the 160-wide ones need 92 KiB of LDS)
    if (wt.variant == 4) { WG_LAUNCH_F32(2, 2, 1, 1) }
    else if (wt.variant == 1) { WG_LAUNCH_F32(1, 4, 5, 1) }
    else if (wt.variant == 2) { WG_LAUNCH_F32(4, 1, 1, 5) }
    else if (wt.variant == 3) { WG_LAUNCH(1, 4, 1, 1) }
    else { WG_LAUNCH(2, 2, 2, 2) }
#undef WG_LAUNCH_T
#undef WG_LAUNCH_F32
#undef WG_LAUNCH_X3
#undef WG_LAUNCH
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return wgrad_finish(a, partial, pl.splits, cs_partial, nullptr, nullptr, s);
}

// sum of the split slabs into G (scaled, accumulated) and the bias column sums
static hipError_t wgrad_finish(const WgradArgs &a, const float *partial, int splits, float *cs_partial, const float *ds1, const float *ds2, hipStream_t s) {
  hipError_t e;
  struct { int splits; } pl{splits};
  const long long total = (long long)a.Do * a.K * a.Di;
  int rb = (int)((total + 255) / 256);
  if (rb > 2048) rb = 2048;
  if (total <= 32768 && pl.splits >= 8 && !ds1)
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, partial, pl.splits, a.Do, a.K * a.Di,
                       a.Di, a.coef, a.scale, a.G, a.ldg, a.accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, s, partial, pl.splits, a.Do, a.K * a.Di, a.Di, a.coef,
                       a.scale, a.G, a.ldg, a.accumulate, ds1, ds2);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.bias_acc) {
    MatView dyv{const_cast<float *>(a.dY), a.N, a.Do, (int)a.lddy};
    e = colsum_add(dyv, a.scale, a.bias_acc, cs_partial, s);
  }
  return e;
}

}  // namespace tdnnf

extern "C" {
int tdnnf_profile_enable(int on) {
  using namespace tdnnf;
  if (on) {
    for (auto &p : g_prof) {
      if (p.ev.empty()) {
        p.ev.resize(2 * kProfMaxLaunches);
        for (auto &e : p.ev)
          if (hipEventCreate(&e) != hipSuccess) return TDNNF_EHIP;
      }
      p.used = 0;
      p.flops = 0;
      p.bytes = 0;
    }
  }
  g_prof_on = on != 0;
  return TDNNF_OK;
}
int tdnnf_profile_read(int cls, double *launches, double *total_ms, double *total_flops) {
  using namespace tdnnf;
  if (cls < 0 || cls >= kProfClasses) return TDNNF_EINVAL;
  ProfClass &p = g_prof[cls];
  double ms = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) return TDNNF_EHIP;
    float t = 0;
    if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) != hipSuccess) return TDNNF_EHIP;
    ms += t;
  }
  if (launches) *launches = (double)(p.used / 2);
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = p.flops;
  return TDNNF_OK;
}
int tdnnf_profile_read_bytes(int cls, double *algorithmic_bytes) {
  if (cls < 0 || cls >= tdnnf::kProfClasses || !algorithmic_bytes) return TDNNF_EINVAL;
  *algorithmic_bytes = tdnnf::g_prof[cls].bytes;
  return TDNNF_OK;
}
const char *tdnnf_profile_class_name(int cls) { return cls >= 0 && cls < tdnnf::kProfClasses ? tdnnf::g_prof[cls].name : ""; }
}
