// Micro-benchmark: LDS read throughput per CU by instruction width (ds_read_b128 / 2 x ds_read_b64 / 4 x ds_read_b32), lane-linear
// addresses as the MLP kernels' fragment reads, 8 waves per CU, nothing else running.  Bytes per clock per CU at the measured time
// and a nominal 2.4 GHz (the loop is short of any power limit).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = float(i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const char* base = smem + ((it * 7 + wave) % 12) * 8192;
    if (MODE == 0) {   // eight independent reads in flight per wave, then one wait
      f32x4 v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = *reinterpret_cast<const f32x4*>(base + t * 1024 + lane * 16);
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(v[t]));
    } else if (MODE == 1) {
      f32x2 v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = *reinterpret_cast<const f32x2*>(base + t * 512 + lane * 8);
#pragma unroll
      for (int t = 0; t < 16; ++t) asm volatile("" ::"v"(v[t]));
    } else {
      float v[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) v[t] = *reinterpret_cast<const float*>(base + t * 256 + lane * 4);
#pragma unroll
      for (int t = 0; t < 32; ++t) asm volatile("" ::"v"(v[t]));
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 96 * 1024, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 96 * 1024, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = double(iters) * 8 * 1024 * 8;
  printf("%-22s %.3f ms  %.1f bytes per clock per CU at 2.4 GHz\n", name, ms, bytes_per_cu / (ms * 1e-3 * 2.4e9));
  hipFree(out);
}
int main() {
  run<0>("ds_read_b128"); run<1>("2 x ds_read_b64"); run<2>("4 x ds_read_b32"); run<0>("ds_read_b128 (again)");
  return 0;
}
