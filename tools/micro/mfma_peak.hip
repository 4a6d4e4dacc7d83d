// Practical peak of v_mfma_f32_32x32x2_f32 on this GPU: independent accumulator chains, operands in registers,
// no memory traffic.  usage: mfma_peak [waves_per_simd=2] [chains=4] [iters] [reps] [random_operands=0]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0, int rnd) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  // rnd != 0: operands with random mantissas that change every MFMA (data toggling costs power, and the chip trades power
  // for clock: MI355X_MICROARCH.md "DVFS give-back") -- the rate a real GEMM on real data can see
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float av[8], bv[8];
  for (int u = 0; u < 8; u++) {
    h = h * 1664525u + 1013904223u;
    av[u] = rnd ? __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f : a;
    h = h * 1664525u + 1013904223u;
    bv[u] = rnd ? __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f : b;
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u * CH + c) & 7], bv[(u + c) & 7], acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2, ch = argc > 2 ? atoi(argv[2]) : 4;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * wps, iters = argc > 3 ? atoi(argv[3]) : 20000;
  const int rnd = argc > 5 ? atoi(argv[5]) : 0;
  float *out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < (argc > 4 ? atoi(argv[4]) : 3); rep++) {
    hipEventRecord(e0);
    if (ch == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f, rnd);
    else if (ch == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f, rnd);
    else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4.0 * ch * 4096.0;
    printf("CUs %d clock %d MHz  waves/SIMD %d chains %d: %.2f ms  %.1f TFLOP/s\n", cus, prop.clockRate / 1000, wps, ch, ms, flops / ms / 1e9);
  }
  return 0;
}
