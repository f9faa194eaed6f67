// Micro-benchmark: how many VALU fillers hide behind v_mfma_f32_32x32x16_f16 (same wave / 2 waves per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>  // VALU pairs (cvt_pk + pk_max) per MFMA
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  half8 a, b0, b1;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[(threadIdx.x + j) & 255]; b0[j] = (_Float16)in[(threadIdx.x * 3 + j) & 255]; b1[j] = (_Float16)in[(threadIdx.x * 5 + j) & 255]; }
  f32x16 acc0 = {0}, acc1 = {0};
  float x0 = in[threadIdx.x & 255], x1 = in[(threadIdx.x + 1) & 255];
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b0));
      if (NV >= 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r0) : "v"(x0), "v"(x1));
      if (NV >= 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r1) : "v"(x0), "v"(x1));
      if (NV >= 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r2) : "v"(x0), "v"(x1));
      if (NV >= 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r3) : "v"(x0), "v"(x1));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b1));
      if (NV >= 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r0) : "v"(x0), "v"(x1));
      if (NV >= 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r1) : "v"(x0), "v"(x1));
      if (NV >= 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r2) : "v"(x0), "v"(x1));
      if (NV >= 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_pk_max_f16 %0, %0, 0" : "=v"(r3) : "v"(x0), "v"(x1));
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(r0 + r1 + r2 + r3);
}
template <int NV> void run(int threads) {
  float *in, *out;
  hipMalloc(&in, 1024); hipMemset(in, 0, 1024);
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = double(iters) * 16 * (threads / 256);
  printf("%d waves/SIMD, %d VALU per MFMA: %.3f ms -> %.1f cycles per MFMA per SIMD @2.4GHz\n", threads / 256, 2 * NV, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
  hipFree(in); hipFree(out);
}
int main() {
  run<0>(256); run<1>(256); run<2>(256); run<3>(256); run<4>(256);
  run<0>(512); run<1>(512); run<2>(512); run<3>(512); run<4>(512);
  return 0;
}
