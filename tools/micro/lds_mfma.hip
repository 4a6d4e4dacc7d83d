// What rate does the K-loop body of rows_gemm_kernel reach with everything but the LDS fragment reads and the MFMAs
// removed?  Variants of the fragment / MFMA ordering.  usage: lds_mfma [variant] [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

constexpr int BK = 32, LD = BK + 4;

// V0: as the product kernel (TM = TN = 2): per kg 2 A + 2 B float4 reads, 16 MFMAs
template <int VAR>
__global__ __launch_bounds__(256) void kern(float *out, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  for (int i = t; i < 2 * 2 * 128 * LD; i += 256) smem[i] = (float)((i * 7 + 3) % 13) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  const float *As = smem, *Bs = smem + 2 * 128 * LD;
  int buf = 0;
  if (VAR == 0) {
    for (int ks = 0; ks < ksteps; ks++) {
      const float *as = As + buf * 128 * LD + (wm * 64 + li) * LD + lh * 4;
      const float *bs = Bs + buf * 128 * LD + (wn * 64 + li) * LD + lh * 4;
#pragma unroll
      for (int kg = 0; kg < 4; kg++) {
        float4 a[2], b[2];
        for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const float4 *>(as + i * 32 * LD + kg * 8);
        for (int i = 0; i < 2; i++) b[i] = *reinterpret_cast<const float4 *>(bs + i * 32 * LD + kg * 8);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            MFMA(a[i].x, b[j].x, acc[i][j]);
            MFMA(a[i].y, b[j].y, acc[i][j]);
            MFMA(a[i].z, b[j].z, acc[i][j]);
            MFMA(a[i].w, b[j].w, acc[i][j]);
          }
      }
      buf ^= 1;
    }
  } else if (VAR == 1) {
    // k-major issue order: for each of the 4 k's of a float4 all four accumulators in turn (every MFMA depends on
    // the one 4 instructions back), fragments of the next kg requested before the current kg's MFMAs
    float4 a[2][2], b[2][2];
    {
      const float *as = As + (wm * 64 + li) * LD + lh * 4, *bs = Bs + (wn * 64 + li) * LD + lh * 4;
      for (int i = 0; i < 2; i++) a[0][i] = *reinterpret_cast<const float4 *>(as + i * 32 * LD);
      for (int i = 0; i < 2; i++) b[0][i] = *reinterpret_cast<const float4 *>(bs + i * 32 * LD);
    }
    for (int ks = 0; ks < ksteps; ks++) {
      const float *as = As + buf * 128 * LD + (wm * 64 + li) * LD + lh * 4;
      const float *bs = Bs + buf * 128 * LD + (wn * 64 + li) * LD + lh * 4;
      const float *asn = As + (buf ^ 1) * 128 * LD + (wm * 64 + li) * LD + lh * 4;
      const float *bsn = Bs + (buf ^ 1) * 128 * LD + (wn * 64 + li) * LD + lh * 4;
#pragma unroll
      for (int kg = 0; kg < 4; kg++) {
        const int cur = kg & 1, nxt = cur ^ 1;
        const float *pa = kg < 3 ? as + (kg + 1) * 8 : asn, *pb = kg < 3 ? bs + (kg + 1) * 8 : bsn;
        for (int i = 0; i < 2; i++) a[nxt][i] = *reinterpret_cast<const float4 *>(pa + i * 32 * LD);
        for (int i = 0; i < 2; i++) b[nxt][i] = *reinterpret_cast<const float4 *>(pb + i * 32 * LD);
#define STEP(f)                                 \
  MFMA(a[cur][0].f, b[cur][0].f, acc[0][0]);    \
  MFMA(a[cur][0].f, b[cur][1].f, acc[0][1]);    \
  MFMA(a[cur][1].f, b[cur][0].f, acc[1][0]);    \
  MFMA(a[cur][1].f, b[cur][1].f, acc[1][1]);
        STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
      }
      buf ^= 1;
    }
  }
  float s = 0;
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      for (int r = 0; r < 16; r++) s += acc[a][b][r];
  out[blockIdx.x * 256 + t] = s;
}

int main(int argc, char **argv) {
  const int var = argc > 1 ? atoi(argv[1]) : 0, bpc = argc > 2 ? atoi(argv[2]) : 2;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * bpc, ksteps = 4000;
  const size_t lds = sizeof(float) * 2 * 2 * 128 * LD;
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipFuncSetAttribute((const void *)kern<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void *)kern<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    if (var == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(256), lds, 0, out, ksteps);
    else hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(256), lds, 0, out, ksteps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * ksteps * 64.0 * 4096.0;
    printf("variant %d blocks/CU %d: %.2f ms  %.1f TFLOP/s\n", var, bpc, ms, flops / ms / 1e9);
  }
  return 0;
}
