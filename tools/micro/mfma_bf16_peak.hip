// Practical peak of v_mfma_f32_32x32x16_bf16 (and of the three-plane split-bf16 pattern: six MFMAs on shared operands) on
// this GPU: independent accumulator chains, operands in registers, no memory traffic.
// usage: mfma_bf16_peak [waves_per_simd=2] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  bf16x8 a[3], b[3];
  for (int q = 0; q < 3; q++)
    for (int i = 0; i < 8; i++) {
      a[q][i] = (__bf16)(a0 + threadIdx.x * 1e-3f + q);
      b[q][i] = (__bf16)(1e-3f * (q + 1));
    }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int d = 2; d >= 0; d--)
#pragma unroll
        for (int q = 0; q <= d; q++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[d - q], acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * wps, iters = argc > 2 ? atoi(argv[2]) : 20000;
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 24.0;
    printf("CUs %d  waves/SIMD %d: %.2f ms  %.1f TFLOP/s bf16  (%.1f cycles per MFMA per SIMD at %d MHz)\n", cus, wps, ms, mfmas * 32768.0 / ms / 1e9,
           ms * 1e-3 * prop.clockRate * 1e3 / (mfmas / (cus * 4.0)), prop.clockRate / 1000);
  }
  return 0;
}
