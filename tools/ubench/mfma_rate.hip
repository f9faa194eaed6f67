// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 in the MLP kernel's dependency pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool SHARE_A>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  half8 a[4], b[NACC];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)in[(threadIdx.x + i * 7 + j) & 255];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)in[(threadIdx.x * 3 + i + j) & 255];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[SHARE_A ? (kc & 3) : ((kc + n) & 3)], b[n], acc[n], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool SHARE_A>
void run(const char* name, int threads, int blocks_per_cu) {
  float *in, *out;
  hipMalloc(&in, 1024); hipMemset(in, 0, 1024);
  hipMalloc(&out, 256 * 8 * 512 * 4);
  const int iters = 20000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, SHARE_A>), dim3(grid), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, SHARE_A>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = double(threads) / 64 * blocks_per_cu / 4;
  const double mfma_per_simd = double(iters) * 8 * NACC * waves_per_simd;
  const double tf = double(iters) * 8 * NACC * (threads / 64) * grid * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-34s waves/SIMD %.0f  %.3f ms  %.1f TF/s  -> %.1f cycles/MFMA/SIMD @2.4GHz (%.1f @2.0)\n", name, waves_per_simd, ms, tf,
         ms * 1e-3 * 2.4e9 / mfma_per_simd, ms * 1e-3 * 2.0e9 / mfma_per_simd);
  hipFree(in); hipFree(out);
}
int main() {
  run<1, true>("1 acc, 1 wave/SIMD", 256, 1);
  run<2, true>("2 acc shared A, 1 wave/SIMD", 256, 1);
  run<2, true>("2 acc shared A, 2 waves/SIMD", 512, 1);
  run<2, false>("2 acc distinct A, 2 waves/SIMD", 512, 1);
  run<3, true>("3 acc shared A, 1 wave/SIMD", 256, 1);
  run<4, true>("4 acc shared A, 1 wave/SIMD", 256, 1);
  run<4, true>("4 acc shared A, 2 waves/SIMD", 512, 1);
  run<2, true>("2 acc shared A, 4 waves/SIMD", 512, 2);
  run<1, true>("1 acc, 2 waves/SIMD", 512, 1);
  run<1, true>("1 acc, 4 waves/SIMD", 512, 2);
  return 0;
}
