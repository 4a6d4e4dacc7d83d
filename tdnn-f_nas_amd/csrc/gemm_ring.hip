// gemm_ring.hip -- the exact-f32 rows GEMM as a PERSISTENT kernel fed by an LDS-DMA ring (gfx950).
//
// Same arithmetic and arguments as rows_gemm_kernel (gemm_f32.hip): C[m][n] (+)= sum over K-segments (the taps of
// TdnnComponent::Propagate / Backprop, /root/reference/src/nnet3/nnet-tdnn-component.cc:302-324, :378-411) of A_s[m][k] B_s[k][n] with
// v_mfma_f32_32x32x2_f32, bias / old C / fused addend / ReLU / column statistics in the epilogue.  What differs is how a tile is fed
// and what happens between tiles -- the two things the K = 320 shapes of a TDNN-F layer (20 K steps per tile) spent their time on
// ("no global loads after the first tile / no LDS refill and barrier: 120 -> 132 -> 141 TFLOP/s"):
//
//   * operands go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write, nothing to wait for in front of the
//     MFMAs) into a ring of three K steps of 16, two steps ahead of the one being multiplied, counted vmcnt waits and ONE raw
//     barrier per K step;
//   * a block does not end with its tile: it walks a list of tiles (grid = the resident block slots of the chip), and the ring does
//     not know about tile boundaries -- while a tile's accumulators are stored, the first two K steps of the block's next tile are
//     already in flight.  The epilogue reads and writes straight from the accumulator layout (a store instruction = two 128-byte row
//     segments), so it needs no LDS and no barrier, and the ring is not disturbed;
//   * an LDS row is 64 bytes (16 k); a DMA instruction fills 1 KB lane-linearly, so padding is not available: the four 16-byte chunks
//     of a row are permuted by bits 2-3 of the row index instead (the lane picks its source address), which makes the 16-byte
//     fragment reads of 16 consecutive rows fall on 16 different 16-byte columns of the 256-byte bank row.
//
// Taken by rows_gemm() for exact-f32 launches whose segments are whole K steps, without tap coefficients / sumsq / split-K; everything
// else stays on rows_gemm_kernel.  TDNNF_GEMM_RING=0 turns it off (A/B runs).
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm_f32.h"
#include "gemm_ring.h"

namespace tdnnf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 3;  // K steps in LDS: one being multiplied, two in flight

__device__ float4 g_ring_zero = {0.f, 0.f, 0.f, 0.f};  // rows outside M / N / a segment's row range read these 16 bytes

// The epilogue's loads and stores are inline assembly: hipcc (ROCm 7.2) answers ANY ordinary vector-memory instruction inside the K loop with
// an s_waitcnt vmcnt(0) in front of the first fragment read of every K step (it then sees LDS-DMA and register loads pending on one
// counter and drains it), which takes the ring apart.  Hidden from its wait tracker, they are counted by hand: the loads are followed
// by one vmcnt(0) that names their destination registers.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void *base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));  // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ float hidden_buffer_load(i32x4 rsrc, unsigned voff) {
  float v;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
  return v;
}
__device__ __forceinline__ void hidden_buffer_store(float v, i32x4 rsrc, unsigned voff) {
  asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void hidden_wait16(float (&x)[16]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]),
                 "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
               :
               : "memory");
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, TN == 5 ? 2 : 3) void rows_gemm_ring_kernel(const RowsGemmArgs p, int ntm, int ntn) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
  constexpr int A_CH = BM * 4, B_CH = BN * 4, CH = A_CH + B_CH;  // 16-byte chunks of a stage: [A rows x 4 | B]
  constexpr int PPT = (CH + 255) / 256;                          // chunks a thread requests per stage (surplus ones repeat early chunks)
  constexpr int STAGE = PPT * 256 * 16;                          // bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *red = reinterpret_cast<float *>(smem + kRing * STAGE);  // [WM][BN][2] column statistics of a tile

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);  // (wave-uniform, and known to be)
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int G = gridDim.x, ntiles = ntm * ntn;
  // logical tile of this block in round i: every round hands each XCD (blockIdx % 8, round-robin dispatch) a contiguous run of
  // tile ids, tile_n fastest, so the blocks that re-read the same A rows share an L2
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  auto tile_of_round = [&](int i) -> int {
    const int base = i * G;
    const int R = ntiles - base < G ? ntiles - base : G;
    if (R <= 0) return -1;
    const int q = R / 8, r = R % 8;
    const int mine = q + (xcd < r ? 1 : 0);
    if (jx >= mine) return -1;
    return base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jx;
  };
  int my_tiles = 0;
  while (tile_of_round(my_tiles) >= 0) my_tiles++;
  if (my_tiles == 0) return;
  int KT = 0;  // K steps per tile
  for (int s = 0; s < p.nseg; s++) KT += p.seg[s].klen / BK;
  const int total = my_tiles * KT;

  // ---- requests: chunk q = t + 256 j of the stage image
  const char *srcp[PPT];
  unsigned live = 0;  // bit j: chunk j reads memory (advances per K step); clear: it reads the 16 zero bytes
  int rq_round = -1, rq_seg = 0, rq_left = 0, rq_m0 = 0, rq_n0 = 0;
  const char *zero = reinterpret_cast<const char *>(&g_ring_zero);
  auto setup_sources = [&]() {  // for (rq_round's tile, rq_seg), K offset 0
    const GemmSeg sg = p.seg[rq_seg];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      int q = t + 256 * j;
      if (q >= CH) q -= CH;  // surplus chunk (its LDS destination is the unused pad behind the stage image): an early chunk again
      bool ok;
      const float *src;
      if (q < A_CH) {
        const int row = q >> 2, c = (q & 3) ^ ((row >> 2) & 3), m = rq_m0 + row;
        ok = m < p.M && m >= sg.m_lo && m < sg.m_hi;
        src = p.A + sg.a_off + (long long)m * p.lda + c * 4;
      } else {
        const int qb = q - A_CH, row = qb >> 2, c = (qb & 3) ^ ((row >> 2) & 3), n = rq_n0 + row;
        ok = n < p.N;
        src = p.B + sg.b_off + (long long)n * p.ldb + c * 4;
      }
      srcp[j] = ok ? reinterpret_cast<const char *>(src) : zero;
      live = ok ? live | (1u << j) : live & ~(1u << j);
    }
  };
  auto next_request_span = [&]() {  // next segment, or the first segment of the block's next tile
    if (rq_round >= 0 && rq_seg + 1 < p.nseg) {
      rq_seg++;
    } else {
      rq_round++;
      rq_seg = 0;
      if (rq_round >= my_tiles) return;
      const int id = tile_of_round(rq_round);
      rq_m0 = (id / ntn) * BM;
      rq_n0 = (id % ntn) * BN;
    }
    rq_left = p.seg[rq_seg].klen / BK;
    setup_sources();
  };
  next_request_span();
  auto request = [&](int slot) {
    char *dst = smem + slot * STAGE + t * 16;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcp[j]), (__attribute__((address_space(3))) void *)(dst + j * 4096), 16, 0, 0);
      srcp[j] += ((live >> j) & 1u) ? (long long)(BK * 4) : 0ll;
    }
    if (--rq_left == 0) next_request_span();
  };

  // ---- fragments: lane (li, lh) holds k = 8 kg + 4 lh .. + 3 of row li (MFMA x of the group multiplies k = 8 kg + x and 8 kg + 4 + x)
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int row = (wm * TM + i) * 32 + li;
    a_off[i] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);  // kg = 1: the same with chunk bit 1 flipped (^ 32)
  }
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int row = (wn * TN + j) * 32 + li;
    b_off[j] = A_CH * 16 + row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
  }

  f32x16 acc[TM][TN];
  auto clear = [&]() {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  };
  clear();
  // Fragment reads are inline assembly as well: hipcc answers an LDS read that follows a global_load_lds with s_waitcnt vmcnt(0) (every
  // LDS-DMA is a pending LDS write to it), i.e. it would wait for the two K steps just requested.  The reads of both halves of the K
  // step are issued together; LDS returns in order, so the first half's MFMAs start when the second half's reads are still out.
  auto multiply = [&](int slot) {
    const unsigned st = (unsigned)(slot * STAGE);
    f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int kg = 0; kg < 2; kg++) {
#pragma unroll
      for (int i = 0; i < TM; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(a[kg][i]) : "v"(st + (unsigned)(a_off[i] ^ (kg * 32))) : "memory");
#pragma unroll
      for (int j = 0; j < TN; j++) asm volatile("ds_read_b128 %0, %1" : "=v"(b[kg][j]) : "v"(st + (unsigned)(b_off[j] ^ (kg * 32))) : "memory");
    }
#pragma unroll
    for (int kg = 0; kg < 2; kg++) {
      if constexpr (TM == 2 && TN == 2) {
        if (kg == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1])::"memory");
        else {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[1][0]), "+v"(b[1][1])::"memory");
        }
      } else {
        static_assert((TM == 2 && TN == 2) || (TM == 1 && TN == 5), "fragment waits are written out per tile shape");
        if (kg == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0][0]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[0][4])::"memory");
        else {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]), "+v"(b[1][4])::"memory");
        }
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][i].x, b[kg][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][i].y, b[kg][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][i].z, b[kg][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][i].w, b[kg][j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  // ---- epilogue of one tile, from the accumulator layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  // One branch-free path through buffer instructions: the descriptors cover exactly the tile's valid rows of C and of the addend, so
  // rows past M (or outside the addend's row range: their offsets wrap) fail the range check -- loads give 0, stores are dropped --
  // and a lane whose column is past N carries an offset of 2^31.  A store instruction writes two 128-byte row segments.
  auto epilogue = [&](int id) {
    const int tile_m = id / ntn, m0 = tile_m * BM, n0 = (id % ntn) * BN;
    const bool stats = p.colstats != nullptr;
    const int rows_c = p.M - m0 < BM ? p.M - m0 : BM;
    const i32x4 rc = make_rsrc(p.C + (long long)m0 * p.ldc, (unsigned)(rows_c * p.ldc * 4));
    // addend rows [a0, a1) of this tile (tile-relative); its descriptor starts at row a0, so a row below it gets a wrapped offset
    int a0 = p.add_lo - m0, a1 = p.add_hi - m0;
    a0 = a0 < 0 ? 0 : a0;
    a1 = a1 > rows_c ? rows_c : a1;
    const bool adds = p.add && a1 > a0;
    const i32x4 ra = make_rsrc(adds ? p.add + (long long)(m0 + a0 - p.add_lo) * p.ldadd : nullptr, adds ? (unsigned)((a1 - a0) * p.ldadd * 4) : 0u);
    const i32x4 rbias = make_rsrc(p.bias, p.init_mode == 1 ? (unsigned)p.N * 4u : 0u);
    unsigned ldc4 = (unsigned)p.ldc * 4u, lda4 = (unsigned)p.ldadd * 4u;
    // (opaque to the optimiser: the row offsets derived from these are loop invariants of the K loop, and hoisted out of it they
    // occupied ~60 scalar and ~30 vector registers for the whole kernel)
    asm volatile("" : "+s"(ldc4), "+s"(lda4));
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int nl = (wn * TN + j) * 32 + li, n = n0 + nl;
      const bool nv = n < p.N;
      float cs = 0.f, cq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int rb = (wm * TM + i) * 32;  // wave-uniform first row of the block, tile-relative; this lane's rows start 4 lh further
        unsigned c_lane = nv ? (unsigned)(4 * lh) * ldc4 + (unsigned)n * 4u : 0x80000000u;
        unsigned a_lane = nv ? (unsigned)(4 * lh - a0) * lda4 + (unsigned)n * 4u : 0x80000000u;  // (wraps for rows below a0)
        asm volatile("" : "+v"(c_lane), "+v"(a_lane));
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = acc[i][j][r];
        if (p.init_mode == 1) {  // (a lane past N reads 0)
          float pre[16];
          pre[0] = hidden_buffer_load(rbias, (unsigned)n * 4u);
#pragma unroll
          for (int r = 1; r < 16; r++) pre[r] = 0.f;
          hidden_wait16(pre);
#pragma unroll
          for (int r = 0; r < 16; r++) v[r] += pre[0];
        }
        if (p.init_mode == 0) {
          float pre[16];
#pragma unroll
          for (int r = 0; r < 16; r++) pre[r] = hidden_buffer_load(rc, c_lane + (unsigned)(rb + (r & 3) + 8 * (r >> 2)) * ldc4);
          hidden_wait16(pre);
#pragma unroll
          for (int r = 0; r < 16; r++) v[r] += pre[r];
        }
        if (adds) {
          float addv[16];
#pragma unroll
          for (int r = 0; r < 16; r++) addv[r] = hidden_buffer_load(ra, a_lane + (unsigned)(rb + (r & 3) + 8 * (r >> 2)) * lda4);
          hidden_wait16(addv);
#pragma unroll
          for (int r = 0; r < 16; r++) v[r] += p.add_scale * addv[r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          if (p.relu) v[r] = floor_keep_nan(v[r], 0.f);
          hidden_buffer_store(v[r], rc, c_lane + (unsigned)(rb + (r & 3) + 8 * (r >> 2)) * ldc4);
        }
        if (stats) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float x = (nv && rb + 4 * lh + (r & 3) + 8 * (r >> 2) < rows_c) ? v[r] : 0.f;
            cs += x;
            cq += x * x;
          }
        }
      }
      if (stats) {  // rows of the other half wave, then of the other waves of this column block (through LDS, fixed order)
        cs += __shfl_xor(cs, 32, 64);
        cq += __shfl_xor(cq, 32, 64);
        if (lh == 0) {
          red[(wm * BN + nl) * 2] = cs;
          red[(wm * BN + nl) * 2 + 1] = cq;
        }
      }
    }
    if (stats) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (not __syncthreads(): its fence would wait for the stores above and the ring)
      __builtin_amdgcn_s_barrier();
      if (t < BN && n0 + t < p.N) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; w++) {
          s0 += red[(w * BN + t) * 2];
          s1 += red[(w * BN + t) * 2 + 1];
        }
        float *q0 = p.colstats + (long long)tile_m * p.N + n0 + t, *q1 = p.colstats + ((long long)p.colstats_stride + tile_m) * p.N + n0 + t;
        asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %2, %3, off" ::"v"(q0), "v"(s0), "v"(q1), "v"(s1) : "memory");
      }
      // (the next tile's statistics are written KT barriers from here)
    }
  };

  // ---- the ring.  Step g sits in slot g % 3; when step g is multiplied the requests of steps g + 1 and g + 2 are in flight.
  // A wait for "my chunks of step g" is vmcnt(PPT): at most step g + 1's requests outstanding (requests complete in order among
  // themselves; whatever else is older -- an epilogue's loads and stores -- must then be complete as well, which is safe and, one K step
  // after they were issued, free).  For the step that follows a tile's last one the wait is made BEFORE the epilogue's stores are issued,
  // so that it does not wait for them.
  int done_in_tile = 0, round = 0, rq_slot = 0, mul_slot = 0;
  bool waited = false;
  for (int g = -2; g < total; g++) {
    if (g >= 0) {
      if (!waited) {
        if (g + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      waited = false;
      __builtin_amdgcn_s_barrier();  // everybody's chunks of step g are in LDS; everybody has multiplied step g - 1, whose slot is free
    }
    if (g + 2 < total) {
      request(rq_slot);
      rq_slot = rq_slot == kRing - 1 ? 0 : rq_slot + 1;
    }
    if (g >= 0) {
      multiply(mul_slot);
      mul_slot = mul_slot == kRing - 1 ? 0 : mul_slot + 1;
      if (++done_in_tile == KT) {
        if (g + 1 < total) {  // step g + 1 (the next tile's first) landed?  asked here, in front of the stores
          if (g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          waited = true;
        }
        epilogue(tile_of_round(round));
        clear();
        done_in_tile = 0;
        round++;
      }
    }
  }
}

int g_ring_cus = 0;
int ring_cus() {
  if (g_ring_cus == 0) {
    int dev = 0;
    g_ring_cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_ring_cus = prop.multiProcessorCount;
    (void)hipGetLastError();
  }
  return g_ring_cus;
}

template <int WM, int WN, int TM, int TN>
hipError_t launch_ring(const RowsGemmArgs &a, int blocks, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int CH = (BM + BN) * 4, PPT = (CH + 255) / 256;
  constexpr size_t lds = (size_t)kRing * PPT * 256 * 16 + sizeof(float) * WM * BN * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void *)rows_gemm_ring_kernel<WM, WN, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  const int tiles = ntm * ntn;
  hipLaunchKernelGGL((rows_gemm_ring_kernel<WM, WN, TM, TN>), dim3(tiles < blocks ? tiles : blocks), dim3(256), lds, s, a, ntm, ntn);
  return hipGetLastError();
}

}  // namespace

// option "gemm_ring": 0 off, 1 (default) the 128 x 128 tile, 2 the 128 x 160 tile as well.  Same-box A/B of the 7q step (ms per step, two
// runs each): off 127.18 / 127.33; 128-wide tile 126.59 / 126.30; both tiles and transposed weights for the backward-data GEMMs 127.80 / 127.65
// -- alone the kernel is 3-7 % faster on seven of nine layer shapes, in the step that is what is left of it.
int rows_gemm_ring_mode() { return options().gemm_ring; }
bool rows_gemm_ring_enabled() { return rows_gemm_ring_mode() != 0; }

bool rows_gemm_ring_ok(const RowsGemmArgs &a, bool b_kc, bool vec) {
  if (!rows_gemm_ring_enabled() || a.prec != 0 || !vec || a.coef || a.sumsq || a.ksplit > 1 || a.nseg <= 0) return false;
  for (int i = 0; i < a.nseg; i++)
    if (a.seg[i].klen <= 0 || a.seg[i].klen % 16 != 0) return false;
  if (!b_kc) return false;  // (a [k][n] form of B was written too: 4-byte fragment reads, spills at three blocks per CU -- removed; TDNNF_WT=1 hands the backward-data GEMMs W^T)
  if ((a.init_mode == 1 && !a.bias) || a.M <= 0 || a.N <= 0) return false;
  return true;
}

int rows_gemm_ring_slots(int tile_cols) { return (tile_cols == 160 ? 2 : 3) * ring_cus(); }

hipError_t rows_gemm_ring(const RowsGemmArgs &a, bool b_kc, int tile_cols, hipStream_t s) {
  const int blocks = rows_gemm_ring_slots(tile_cols);
  if (!b_kc) return hipErrorInvalidValue;
  if (tile_cols == 160) return launch_ring<4, 1, 1, 5>(a, blocks, s);
  return launch_ring<2, 2, 2, 2>(a, blocks, s);
}

}  // namespace tdnnf
