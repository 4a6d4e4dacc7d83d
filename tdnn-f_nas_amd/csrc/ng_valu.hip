// ng_valu.hip -- the N-sized statistics pass of OnlineNaturalGradient, H = X~ W^T (rank R = 20 .. 80), on the VECTOR ALUs.
//
// Call sites of the reference: PreconditionDirections at /root/reference/src/nnet3/nnet-tdnn-component.cc:598-599 and
// nnet-simple-component.cc:3001-3002 (UPSTREAM natural-gradient-online.cc forms X W^T first of all); ng.hip restates the algorithm.
//
// Why not the matrix cores.  On gfx950 the f32 MFMA (v_mfma_f32_32x32x2_f32) and the f32 VALU (v_fma_f32, 2 cycles per wave64) have the SAME peak,
// 64 FLOP / clock / SIMD, and they are separate pipes: a vector-only wave and a matrix-only wave of one SIMD run side by side.  Every GEMM of
// an exact-f32 training step is bound by the f32 matrix pipe, and so were these passes (a rank of 20 on 32-column MFMA tiles, 80 on 96:
// docs/experiments.md r5-g): one TFLOP of issued tile work per step that came straight out of the main stream's GEMMs -- skipping the
// passes (timing only) gave 7.0 ms of a 119.5 ms step back.  The same products on the vector pipe cost the matrix pipe nothing and have
// no padding (20 is 20).
//
// Shape of the kernel.  A 256-thread block owns 256 rows of X~.  A thread holds a 4 x 20 register tile: rows lane, lane + 64, lane + 128,
// lane + 192 of the block's rows against 20 of the R outputs, so that the 20 values of W^T[k][.] it reads from LDS (five broadcast
// ds_read_b128) meet four x values -- 80 v_fmac_f32 per k.  The four waves split what is left:
// R = 80 -> four groups of 20 outputs; R = 20 -> four quarters of every K step (their partial sums are added through LDS at the end of the
// tile); R = 40 -> two by two.  K steps of 32 columns (128-byte row segments; 16 for R > 20): the X tile goes global -> registers -> LDS (row
// pitch K step + 4 floats: conflict-free ds_read_b128 down a column of rows), one K step ahead in registers, two LDS buffers, one barrier per
// K step.  LDS array cycles per wave and k: (5 + 1) reads x 4 = 24 against 160 VALU cycles -- four SIMDs stay under the array's rate.
// A tap's Di % (K step) last columns do not go through the tile: each thread adds them from global memory at the end.
// (First form, kept in docs/experiments.md r5-n: one row per thread, W^T from SGPRs through s_load -- instruction-perfect, and 8 x slower
// than its VALU time: every s_load of a 245 KB W^T misses the 16 KB scalar cache.)
// ||X~||_F^2 (the trace the preconditioner's scale needs) is a by-product: the waves of output group 0 also square what they read.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "gemm_f32.h"
#include "ng.h"

namespace tdnnf {
namespace {

constexpr int kRW = 20;  // outputs per thread
constexpr int kTM = 4;   // rows per thread
constexpr int kRows = 64 * kTM;  // rows per block
static_assert(kWtPadRows >= 32, "a K step is at most 32 rows of W^T");

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// The inner product is issued from inline assembly: the compiler, given the same loop in C++, hoists every LDS read of a K step above its
// first FMA and spills what no longer fits (122 .. 431 VGPR spills in six attempts to talk it out of that).  `asm volatile` statements keep
// their order, so the order below is the order on the SIMD; LDS reads are waited for by hand (lgkmcnt(0): also covers any s_load in flight).
//   lds_w(w, addr, OFF): w[0 .. 9] = the 20 floats at LDS byte address addr + OFF (five ds_read_b128)
//   fma20(acc, w, x): acc[r] += x * w[r], r = 0 .. 19
template <int OFF>
__device__ __forceinline__ void lds_w(float (&w)[kRW], unsigned addr) {
  v4f t0, t1, t2, t3, t4;
  asm volatile(
      "ds_read_b128 %0, %5 offset:%6\n ds_read_b128 %1, %5 offset:%7\n ds_read_b128 %2, %5 offset:%8\n ds_read_b128 %3, %5 offset:%9\n"
      "ds_read_b128 %4, %5 offset:%10"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
      : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48), "n"(OFF + 64)
      : "memory");
  // (the dwords of the 128-bit registers: sub-register views, no copies)
#pragma unroll
  for (int e = 0; e < 4; e++) {
    w[e] = t0[e]; w[4 + e] = t1[e]; w[8 + e] = t2[e]; w[12 + e] = t3[e]; w[16 + e] = t4[e];
  }
}
template <int OFF>
__device__ __forceinline__ void lds_x(v2f &x, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(x) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// acc[0 .. 19] += x * w[0 .. 19]: twenty v_fmac_f32 (PLAIN f32 FMAs: the SIMD is 32 lanes wide and retires one wave-instruction per 2
// cycles, 64 FLOP / clock; v_pk_fma_f32 is no faster per FLOP -- measured here 10 .. 16 cycles per instruction -- see MI355X_MICROARCH.md)
__device__ __forceinline__ void fma10(float *acc, const float *w, float x) {
  asm volatile("v_fmac_f32 %0, %20, %10\n v_fmac_f32 %1, %20, %11\n v_fmac_f32 %2, %20, %12\n v_fmac_f32 %3, %20, %13\n v_fmac_f32 %4, %20, %14\n"
               "v_fmac_f32 %5, %20, %15\n v_fmac_f32 %6, %20, %16\n v_fmac_f32 %7, %20, %17\n v_fmac_f32 %8, %20, %18\n v_fmac_f32 %9, %20, %19"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]),
                 "+v"(acc[9])
               : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(x));
}
__device__ __forceinline__ void fma20(float (&acc)[kRW], const float (&w)[kRW], float x) {
  fma10(acc, w, x);
  fma10(acc + 10, w + 10, x);
}

// KS x RG = 4 waves: wave w multiplies quarter / half `w % KS` of every K step into output group `w / KS`
template <int KS, int RG, bool EFF>
__global__ __launch_bounds__(256, 2) void ng_rowdot_kernel(NgRowdotArgs a, int ntiles, int part_cap) {
  constexpr int KC = KS == 4 ? 32 : 16, XP = KC + 4, Q = KC / 4, KW = KC / KS, RT = kRW * RG;
  constexpr int XF4 = kRows * Q / 256;             // float4 of the X tile per thread and K step (8 or 4)
  constexpr int WF4 = (KC * RT / 4 + 255) / 256;   // float4 of the W^T tile per thread and K step (1 or 2)
  __shared__ __attribute__((aligned(16))) float xs[2][kRows * XP];
  __shared__ __attribute__((aligned(16))) float ws[2][KC * RT];
  __shared__ double red[4];
  static_assert((KS - 1) * RG * kTM * kRW * 64 <= 2 * kRows * XP, "the partial sums of the K slices fit the X buffers");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = wave % KS, rg = wave / KS;
  const int spk = a.Di / KC, S = a.nseg * spk, ktail = spk * KC;  // whole K steps per tap; columns [ktail, Di) of each tap: the tail
  double sq_total = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * kRows;
    const float *xrow[XF4];
#pragma unroll
    for (int j = 0; j < XF4; j++) {
      const int f = tid + 256 * j, lrow = f / Q, lq = f % Q;
      const int lm = min(m0 + lrow, a.N - 1);  // (rows past the end: a valid row is loaded, its results are not stored)
      xrow[j] = a.X + (long long)lm * a.row_stride * a.ldx + 4 * lq;
    }
    // K step -> this thread's float4 of the X tile and of the W^T tile (KC consecutive rows of W^T = KC * Rp consecutive floats).  Only
    // WHOLE K steps of a tap go through the tile; the Di % KC columns left of each tap are added afterwards, straight from global memory.
    v4f sx[XF4], sw[WF4];  // (plain arrays of native vectors, filled and parked by unrolled loops: a struct behind lambda references ended up in scratch)
#define NG_LOAD(sg, kb)                                                                             \
  do {                                                                                               \
    const long long off__ = a.seg_off[sg] + (kb);                                                    \
    _Pragma("unroll") for (int j = 0; j < XF4; j++) sx[j] = *reinterpret_cast<const v4f *>(xrow[j] + off__); \
    const float *wsrc__ = a.WT + ((long long)(sg) * a.Di + (kb)) * a.Rp;                             \
    _Pragma("unroll") for (int j = 0; j < WF4; j++) {                                                \
      const int f = tid + 256 * j;                                                                   \
      sw[j] = *reinterpret_cast<const v4f *>(wsrc__ + 4 * (f < KC * RT / 4 ? f : 0));             \
    }                                                                                                \
  } while (0)
#define NG_PARK(b)                                                                                   \
  do {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < XF4; j++) {                                                \
      const int f = tid + 256 * j;                                                                   \
      *reinterpret_cast<v4f *>(&xs[b][(f / Q) * XP + 4 * (f % Q)]) = sx[j];                       \
    }                                                                                                \
    _Pragma("unroll") for (int j = 0; j < WF4; j++) {                                                \
      const int f = tid + 256 * j;                                                                   \
      if (f < KC * RT / 4) *reinterpret_cast<v4f *>(&ws[b][4 * f]) = sw[j];                       \
    }                                                                                                \
  } while (0)
    auto advance = [&](int &sg, int &kb) {
      kb += KC;
      if (kb >= ktail) {
        kb = 0;
        sg++;
      }
    };
    float acc[kTM][kRW];
#pragma unroll
    for (int i = 0; i < kTM; i++)
#pragma unroll
      for (int j = 0; j < kRW; j++) acc[i][j] = 0.f;
    v2f sq2[kTM];
#pragma unroll
    for (int i = 0; i < kTM; i++) sq2[i] = (v2f){0.f, 0.f};
    int seg = 0;            // the tap of the K step being multiplied
    int k0 = 0;
    int lseg = 0, lk0 = 0;  // the K step being loaded (one ahead in registers)
    if (S > 0) {
      NG_LOAD(lseg, lk0);
      advance(lseg, lk0);
      NG_PARK(0);
    }
    if (S > 1) {
      NG_LOAD(lseg, lk0);
      advance(lseg, lk0);
    }
    __syncthreads();
    for (int s = 0; s < S; s++) {
      float cf = 1.0f;
      if (EFF) cf = a.eff[seg];
      const float *xb = &xs[s & 1][lane * XP + kq * KW];
      const float *wb = &ws[s & 1][kq * KW * RT + rg * kRW];
      // This wave's KW columns of the step, software-pipelined by hand: while the 20 packed FMAs of column k issue, the 20 values of
      // W^T[k + 1] (and, every other column, the next pair of x values of both rows) are on their way from LDS.
      {
        const unsigned xa = (unsigned)(size_t)xb, wa = (unsigned)(size_t)wb;  // LDS byte addresses (the low 32 bits of a __shared__ pointer)
        float wA[kRW], wB[kRW];
        v2f xA[kTM], xB[kTM];
        lds_w<0>(wA, wa);
#define NG_X(K, X)                                                                  \
  do {                                                                              \
    lds_x<(K) * 4>(X[0], xa);                                                       \
    lds_x<64 * XP * 4 + (K) * 4>(X[1], xa);                                         \
    lds_x<128 * XP * 4 + (K) * 4>(X[2], xa);                                        \
    lds_x<192 * XP * 4 + (K) * 4>(X[3], xa);                                        \
  } while (0)
        NG_X(0, xA);
#define NG_PAIR(K, XC, XN)                                                          \
  do {                                                                              \
    lds_wait();                                                                     \
    lds_w<((K) + 1) * RT * 4>(wB, wa);                                              \
    if ((K) + 2 < KW) NG_X((K) + 2, XN);                                            \
    if (EFF) { XC[0] *= cf; XC[1] *= cf; XC[2] *= cf; XC[3] *= cf; }                \
    fma20(acc[0], wA, XC[0].x);                                                     \
    fma20(acc[1], wA, XC[1].x);                                                     \
    fma20(acc[2], wA, XC[2].x);                                                     \
    fma20(acc[3], wA, XC[3].x);                                                     \
    lds_wait();                                                                     \
    if ((K) + 2 < KW) lds_w<((K) + 2) * RT * 4>(wA, wa);                            \
    fma20(acc[0], wB, XC[0].y);                                                     \
    fma20(acc[1], wB, XC[1].y);                                                     \
    fma20(acc[2], wB, XC[2].y);                                                     \
    fma20(acc[3], wB, XC[3].y);                                                     \
    sq2[0] = __builtin_elementwise_fma(XC[0], XC[0], sq2[0]); /* (every wave: a select per register costs more than the FMAs) */ \
    sq2[1] = __builtin_elementwise_fma(XC[1], XC[1], sq2[1]);                       \
    sq2[2] = __builtin_elementwise_fma(XC[2], XC[2], sq2[2]);                       \
    sq2[3] = __builtin_elementwise_fma(XC[3], XC[3], sq2[3]);                       \
  } while (0)
        static_assert(kTM == 4 && KW % 4 == 0, "the unrolled pairs below");
#pragma unroll
        for (int kk = 0; kk < KW; kk += 4) {
          if (kk == 0) { NG_PAIR(0, xA, xB); NG_PAIR(2, xB, xA); }
          if (kk == 4) { NG_PAIR(4, xA, xB); NG_PAIR(6, xB, xA); }
          if (kk == 8) { NG_PAIR(8, xA, xB); NG_PAIR(10, xB, xA); }
          if (kk == 12) { NG_PAIR(12, xA, xB); NG_PAIR(14, xB, xA); }
        }
#undef NG_X
#undef NG_PAIR
      }
      // the step loaded during the last iteration goes to the other buffer (its readers finished before the last barrier); then the
      // load of the step after it is issued, to land during the next iteration's products
      if (s + 1 < S) NG_PARK((s + 1) & 1);
      if (s + 2 < S) {
        NG_LOAD(lseg, lk0);
        advance(lseg, lk0);
      }
      advance(seg, k0);
      __syncthreads();
    }
    // the Di % KC columns behind each tap's last whole K step: four rows x 20 outputs per thread straight from global memory, the columns
    // dealt round-robin to the K slices
    if (ktail < a.Di) {
      for (int sg = 0; sg < a.nseg; sg++) {
        const float cfs = EFF ? a.eff[sg] : 1.0f;
        for (int k = ktail + kq; k < a.Di; k += KS) {
          const float *wr = a.WT + ((long long)sg * a.Di + k) * a.Rp + rg * kRW;
          float w[kRW];
#pragma unroll
          for (int q = 0; q < kRW; q++) w[q] = wr[q];
#pragma unroll
          for (int i = 0; i < kTM; i++) {
            const int m = min(m0 + lane + 64 * i, a.N - 1);
            const float x = a.X[(long long)m * a.row_stride * a.ldx + a.seg_off[sg] + k] * cfs;
#pragma unroll
            for (int q = 0; q < kRW; q++) acc[i][q] = fmaf(x, w[q], acc[i][q]);
            sq2[i].x = fmaf(x, x, sq2[i].x);
          }
        }
      }
    }
    // partial sums of the K slices 1 .. KS - 1 -> LDS (the X buffers, free now) -> added by slice 0, which stores the rows
    if (KS > 1) {
      float *pr = &xs[0][0];
      if (kq > 0) {
        float *p = pr + ((long long)((kq - 1) * RG + rg) * kTM * kRW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < kTM; i++)
#pragma unroll
          for (int q = 0; q < kRW; q++) p[(i * kRW + q) * 64] = acc[i][q];
      }
      __syncthreads();
      if (kq == 0) {
        for (int o = 1; o < KS; o++) {
          const float *p = pr + ((long long)((o - 1) * RG + rg) * kTM * kRW) * 64 + lane;
#pragma unroll
          for (int i = 0; i < kTM; i++)
#pragma unroll
            for (int q = 0; q < kRW; q++) acc[i][q] += p[(i * kRW + q) * 64];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kTM; i++) {
      const int m = m0 + lane + 64 * i;
      if (m >= a.N) continue;  // (rows past the end were loaded as copies of the last row: neither stored nor counted)
      if (rg == 0) sq_total += (double)sq2[i].x + (double)sq2[i].y;
      if (kq != 0) continue;
      float *h = a.H + (long long)m * a.ldh + rg * kRW;
#pragma unroll
      for (int q = 0; q < kRW / 4; q++) {
        float4 o = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
        if (a.bias) {
          const float *b = a.bias + rg * kRW + 4 * q;
          o.x += b[0]; o.y += b[1]; o.z += b[2]; o.w += b[3];
        }
        *reinterpret_cast<float4 *>(h + 4 * q) = o;
      }
    }
    if (KS > 1) __syncthreads();  // (the X buffers, which held the partial sums, are written again by the next tile's first park)
  }
  if (!a.part) return;
  // block sum of the rows' squared norms -> part[blockIdx.x]; the entries no block owns are zeroed (consumers add up all part_cap of them)
  for (int o = 32; o > 0; o >>= 1) sq_total += __shfl_xor(sq_total, o, 64);
  if (lane == 0) red[wave] = sq_total;
  __syncthreads();
  if (tid == 0) a.part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  for (int i = gridDim.x + blockIdx.x * 256 + tid; i < part_cap; i += gridDim.x * 256) a.part[i] = 0.0;
}

#undef NG_LOAD
#undef NG_PARK

template <int KS, int RG>
void launch_rowdot(const NgRowdotArgs &a, int grid, int ntiles, int cap, hipStream_t s) {
  if (a.eff) hipLaunchKernelGGL((ng_rowdot_kernel<KS, RG, true>), dim3(grid), dim3(256), 0, s, a, ntiles, cap);
  else hipLaunchKernelGGL((ng_rowdot_kernel<KS, RG, false>), dim3(grid), dim3(256), 0, s, a, ntiles, cap);
}

}  // namespace

bool ng_rowdot_ok(const NgRowdotArgs &a) {
  if (!(a.Rp == 20 || a.Rp == 40 || a.Rp == 80) || a.ldh % 4 != 0 || a.N <= 0 || a.nseg < 1 || a.nseg > kMaxSeg || a.Di < 4) return false;
  if (a.ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(a.X) & 15) || (reinterpret_cast<uintptr_t>(a.H) & 15) || (reinterpret_cast<uintptr_t>(a.WT) & 15)) return false;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
  for (int i = 0; i < a.nseg; i++)
    if (a.seg_off[i] % 4 != 0) return false;
  return true;
}

hipError_t ng_rowdot(const NgRowdotArgs &a, hipStream_t s) {
  const int RG = a.Rp / kRW, ntiles = (a.N + kRows - 1) / kRows;
  const int cap = a.part ? a.part_cap : 1024;
  const int grid = std::max(1, std::min(ntiles, std::min(cap, 1024)));
  // algorithmic work of the pass for the event-timed class (bench.py roofline_secondary): 2 N D R FLOPs; X once, H once
  // (taps = row shifts of one matrix share its rows, as gemm_f32.hip counts them)
  const double D = (double)a.nseg * a.Di;
  long long lo = a.seg_off[0], hi = a.seg_off[0];
  for (int i = 1; i < a.nseg; i++) {
    lo = std::min(lo, a.seg_off[i]);
    hi = std::max(hi, a.seg_off[i]);
  }
  ProfGemmRange prof(3, 2.0 * a.N * D * a.Rp, 4.0 * (((double)a.N + (double)(hi - lo) / (double)std::max(1LL, a.ldx)) * a.Di + D * a.Rp + (double)a.N * a.Rp), s);
  if (RG == 1) launch_rowdot<4, 1>(a, grid, ntiles, cap, s);
  else if (RG == 2) launch_rowdot<2, 2>(a, grid, ntiles, cap, s);
  else launch_rowdot<1, 4>(a, grid, ntiles, cap, s);
  return hipGetLastError();
}

}  // namespace tdnnf
