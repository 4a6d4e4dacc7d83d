// Micro-benchmark behind the wide denominator (chain): what one "arc step" of a wave costs a CU --
// 3 ds_bpermute (the arc's fields out of another lane's registers) + 2 gathers of SG-float runs out of an L2-resident table.
// usage: ./gather_bperm MODE SG WAVES_PER_SIMD [table_KB_per_group]   MODE 0 bpermute only, 1 gathers only, 2 both, 3 readlane+select instead of bpermute
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int SG>
__global__ __launch_bounds__(256) void k(const float *table, const unsigned *keys, int rows, int iters, int ngroups, float *out) {
  constexpr int RL = 64 / SG, IT = 16 / RL;
  const int lane = threadIdx.x & 63, sl = lane % SG, rl = lane / SG;
  const int grp = blockIdx.x % ngroups;
  const float *tab = table + (size_t)grp * rows * SG;
  const unsigned *kp = keys + ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane) * 4;
  float acc[IT];
  for (int i = 0; i < IT; i++) acc[i] = 0.f;
  unsigned ax = kp[0], ay = kp[1], az = kp[2];
  for (int n = 0; n < iters; n++) {
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int row4 = (it * RL + rl) * 4;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        unsigned key, p, q;
        if (MODE == 0 || MODE == 2) {
          key = __builtin_amdgcn_ds_bpermute(row4 + jj * 64, (int)ax);
          p = __builtin_amdgcn_ds_bpermute(row4 + jj * 64, (int)ay);
          q = __builtin_amdgcn_ds_bpermute(row4 + jj * 64, (int)az);
        } else if (MODE == 3) {
          unsigned k0 = __builtin_amdgcn_readlane((int)ax, jj * 16 + it * RL), k1 = __builtin_amdgcn_readlane((int)ax, jj * 16 + it * RL + (RL > 1 ? 1 : 0));
          unsigned p0 = __builtin_amdgcn_readlane((int)ay, jj * 16 + it * RL), p1 = __builtin_amdgcn_readlane((int)ay, jj * 16 + it * RL + (RL > 1 ? 1 : 0));
          unsigned q0 = __builtin_amdgcn_readlane((int)az, jj * 16 + it * RL), q1 = __builtin_amdgcn_readlane((int)az, jj * 16 + it * RL + (RL > 1 ? 1 : 0));
          key = rl ? k1 : k0;
          p = rl ? p1 : p0;
          q = rl ? q1 : q0;
        } else {
          key = ax + (it * 4 + jj) * 977u + rl * 131u;
          p = ay;
          q = az;
        }
        const unsigned src = (key & 0xffffu) & (rows - 1), dst = (key >> 16) & (rows - 1);  // rows: a power of two
        if (MODE >= 1) acc[it] += tab[src * SG + sl] * __uint_as_float(p) + tab[dst * SG + sl] * __uint_as_float(q);
        else acc[it] += __uint_as_float(p) * (float)src + __uint_as_float(q) * (float)dst;
      }
    }
    ax = ax * 1664525u + 1013904223u;  // next "arcs"
  }
  float s = 0.f;
  for (int i = 0; i < IT; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int SG>
float run(const float *table, const unsigned *keys, int rows, int iters, int ngroups, float *out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, SG>), dim3(blocks), dim3(256), 0, 0, table, keys, rows, 4, ngroups, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, SG>), dim3(blocks), dim3(256), 0, 0, table, keys, rows, iters, ngroups, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 2, SG = argc > 2 ? atoi(argv[2]) : 32, wps = argc > 3 ? atoi(argv[3]) : 3;
  const int table_kb = argc > 4 ? atoi(argv[4]) : 2048;
  const int ngroups = 128 / SG, rows0 = table_kb * 1024 / (SG * 4), blocks = 256 * wps, iters = 200;
  int rows = 1;
  while (rows * 2 <= rows0) rows *= 2;
  float *table, *out;
  unsigned *keys;
  CK(hipMalloc(&table, (size_t)ngroups * rows * SG * 4));
  CK(hipMemset(table, 0, (size_t)ngroups * rows * SG * 4));
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  std::vector<unsigned> h((size_t)blocks * 256 * 4);
  unsigned r = 12345;
  for (auto &v : h) v = (r = r * 1664525u + 1013904223u);
  CK(hipMalloc(&keys, h.size() * 4));
  CK(hipMemcpy(keys, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  float ms = 0;
#define GO(M, S) if (mode == M && SG == S) ms = run<M, S>(table, keys, rows, iters, ngroups, out, blocks);
  GO(0, 16) GO(1, 16) GO(2, 16) GO(3, 16) GO(0, 32) GO(1, 32) GO(2, 32) GO(3, 32) GO(0, 64) GO(1, 64) GO(2, 64) GO(3, 64)
  const double steps_per_cu = (double)blocks * 4 * iters * (16 * 16 / (64 / SG)) / 256;  // wave arc steps per CU
  printf("mode %d SG %d waves/SIMD %d table %d KB/group: %.3f ms  %.1f ns = %.1f cycles@2.4GHz per wave arc step per CU\n", mode, SG, wps, table_kb, ms,
         ms * 1e6 / steps_per_cu, ms * 1e6 / steps_per_cu * 2.4);
  return 0;
}
