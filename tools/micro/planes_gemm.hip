// planes_gemm.hip -- experiment: f32-equivalent GEMM on the bf16 matrix cores with operands PRE-SPLIT into three bf16 planes
// in HBM (x = p0 + p1 + p2, 24 mantissa bits), six products p_i q_j (i + j <= 2) per 16 k on v_mfma_f32_32x32x16_bf16.
// C[m][n] = sum_k A[m][k] B[n][k]  (both operands k-contiguous, the layout of the forward and backward-data GEMMs).
// The library's split-bf16 kernels convert f32 -> planes when a tile goes to LDS and were bound by that staging path (K-step
// of 16, one barrier per 24 MFMAs); here the kernel only moves planes: K-step 32, ONE LDS buffer of 60 KB (two blocks per
// CU), fragments of the second 16-k chunk requested before the MFMAs of the first, next tile's global loads in flight during
// the MFMAs.
// usage: planes_gemm M N K [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(e)                                                          \
  do {                                                                 \
    hipError_t err__ = (e);                                            \
    if (err__ != hipSuccess) {                                         \
      printf("%s: %s\n", #e, hipGetErrorString(err__));                \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

// planes[q][r][k] (ld = ldp bf16 per row), q = 0..2
__global__ void split_kernel(const float *x, long long rows, int cols, int ld, __bf16 *planes, int ldp) {
  const long long total = rows * cols;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += gridDim.x * 256LL) {
    const long long r = e / cols;
    const int c = (int)(e % cols);
    float v = x[r * ld + c];
    for (int q = 0; q < 3; q++) {
      const __bf16 h = (__bf16)v;
      planes[(q * rows + r) * ldp + c] = h;
      v -= (float)h;
    }
  }
}

constexpr int BM = 128, BN = 128, BK = 32, LDH = BK + 8;  // 80-byte LDS rows: conflict-free 16-byte fragment reads
constexpr int PLANE_A = BM * LDH, PLANE_B = BN * LDH;      // bf16 elements per plane image

template <int NP, int ABL>
__global__ __launch_bounds__(256, 2) void planes_gemm_kernel(const __bf16 *A, long long a_plane, int lda, const __bf16 *B, long long b_plane, int ldb,
                                                             float *C, int ldc, int M, int N, int K, int ntn) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16 *As = smem, *Bs = smem + NP * PLANE_A;
  int bid = blockIdx.x;
  {  // XCD-aware order (as the library's kernels): blocks of one XCD get a contiguous run of tiles
    const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m = bid / ntn, tile_n = bid % ntn, m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  // staging: per plane and operand 128 rows x 64 bytes = 512 x 16 B -> two 16-byte loads per thread
  const int srow[2] = {t >> 2, (t + 256) >> 2}, skq = t & 3;
  const __bf16 *ag[2], *bg[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int m = min(m0 + srow[j], M - 1), n = min(n0 + srow[j], N - 1);  // (rows beyond the edge are computed and not stored)
    ag[j] = A + (long long)m * lda + skq * 8;
    bg[j] = B + (long long)n * ldb + skq * 8;
  }
  uint4 ra[NP][2], rb[NP][2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        ra[q][j] = *reinterpret_cast<const uint4 *>(ag[j] + q * a_plane + k0);
        rb[q][j] = *reinterpret_cast<const uint4 *>(bg[j] + q * b_plane + k0);
      }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        *reinterpret_cast<uint4 *>(As + q * PLANE_A + srow[j] * LDH + skq * 8) = ra[q][j];
        *reinterpret_cast<uint4 *>(Bs + q * PLANE_B + srow[j] * LDH + skq * 8) = rb[q][j];
      }
  };
  const __bf16 *af = As + (wm * 64 + li) * LDH + lh * 8, *bf = Bs + (wn * 64 + li) * LDH + lh * 8;
  bf16x8 fa[2][NP][2], fb[2][NP][2];  // [chunk parity][plane][tile]
  auto load_frags = [&](int c, int par) {
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
      for (int i = 0; i < 2; i++) {
        fa[par][q][i] = *reinterpret_cast<const bf16x8 *>(af + q * PLANE_A + i * 32 * LDH + c * 16);
        fb[par][q][i] = *reinterpret_cast<const bf16x8 *>(bf + q * PLANE_B + i * 32 * LDH + c * 16);
      }
  };
  auto mfmas = [&](int par) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int d = NP - 1; d >= 0; d--)  // smallest terms first, the leading product last
#pragma unroll
          for (int q = 0; q <= d; q++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[par][q][i], fb[par][d - q][j], acc[i][j], 0, 0, 0);
  };
  const int nk = K / BK;  // (K a multiple of 32 in this experiment)
  load_tile(0);
  store_tile();
  __syncthreads();
  if (ABL >= 2) { load_frags(0, 0); load_frags(1, 1); }
  for (int kt = 0; kt < nk; kt++) {
    if (ABL < 1 && kt + 1 < nk) load_tile((kt + 1) * BK);  // in flight during the MFMAs
    if (ABL < 2) { load_frags(0, 0); load_frags(1, 1); }
    mfmas(0);
    mfmas(1);
    if (ABL < 1 && kt + 1 < nk) {
      __syncthreads();  // everyone has read this tile's fragments
      store_tile();
      __syncthreads();
    }
    if (ABL == 1) __syncthreads();
  }
  // epilogue (experiment: straight from the accumulators; C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + wn * 64 + j * 32 + li;
        if (m < M && n < N) C[(long long)m * ldc + n] = acc[i][j][r];
      }
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 192000, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 320, reps = argc > 4 ? atoi(argv[4]) : 10;
  if (K % BK) { printf("K must be a multiple of %d\n", BK); return 1; }
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  for (auto &v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  float *dA, *dB, *dC;
  __bf16 *pA, *pB;
  CK(hipMalloc(&dA, sizeof(float) * hA.size()));
  CK(hipMalloc(&dB, sizeof(float) * hB.size()));
  CK(hipMalloc(&dC, sizeof(float) * (size_t)M * N));
  CK(hipMalloc(&pA, sizeof(__bf16) * 3 * hA.size()));
  CK(hipMalloc(&pB, sizeof(__bf16) * 3 * hB.size()));
  CK(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  hipEventRecord(e0);
  hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, 0, dA, (long long)M, K, K, pA, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("split A (%d x %d): %.3f ms = %.2f TB/s of (4 + 6) bytes per element\n", M, K, ms, 10.0 * M * K / ms / 1e9);
  hipLaunchKernelGGL(split_kernel, dim3(1024), dim3(256), 0, 0, dB, (long long)N, K, K, pB, K);
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  for (int v = 0; v < 4; v++) {
    const int np = v < 3 ? 3 : 2;
    const size_t lds = sizeof(__bf16) * np * (PLANE_A + PLANE_B);
    auto kern = v == 0 ? planes_gemm_kernel<3, 0> : v == 1 ? planes_gemm_kernel<3, 1> : v == 2 ? planes_gemm_kernel<3, 2> : planes_gemm_kernel<2, 0>;
    printf("variant %d (0 full, 1 no global loads / LDS refill, 2 MFMAs only, 3 two planes)\n", v);
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), lds, 0, pA, (long long)M * K, K, pB, (long long)N * K, K, dC, N, M, N, K, ntn);
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      hipEventElapsedTime(&ms, e0, e1);
      ms /= reps;
      printf("planes %d (%d products): M %d N %d K %d: %.1f us  %.1f TFLOP/s f32-equivalent, %.0f TFLOP/s of bf16 MFMA\n", np, np == 3 ? 6 : 3, M, N, K, ms * 1e3,
             2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K * (np == 3 ? 6 : 3) / ms / 1e9);
    }
    // accuracy on a sample of rows against float64
    std::vector<float> hC((size_t)256 * N);
    CK(hipMemcpy(hC.data(), dC + (size_t)(M / 2) * N, sizeof(float) * hC.size(), hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int r = 0; r < 256; r++)
      for (int n = 0; n < N; n += 7) {
        double s = 0;
        for (int k = 0; k < K; k++) s += (double)hA[(size_t)(M / 2 + r) * K + k] * hB[(size_t)n * K + k];
        const double d = hC[(size_t)r * N + n] - s;
        num += d * d;
        den += s * s;
      }
    printf("   relative L2 error against float64 on 256 rows: %.3e\n", sqrt(num / den));
  }
  return 0;
}
