// Micro-benchmark: what v_mfma_f32_32x32x16_f16 rate the part SUSTAINS (power management included) with operands that toggle like
// real data -- nothing but back-to-back independent MFMAs, four accumulators per wave, two waves per SIMD, ~0.5 s per case.
// zero operands vs uniform random f16 in [-1, 1) vs random with the accumulators fed back (values grow).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) {
    a[i][j] = (_Float16)in[(threadIdx.x * 8 + i * 4096 + j) & 32767];
    b[i][j] = (_Float16)in[(threadIdx.x * 8 + i * 4096 + j + 16384) & 32767];
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(n + r) & 3], b[n], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k32(const float* in, float* out, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x + i * 4096) & 32767]; b[i] = in[(threadIdx.x + i * 4096 + 16384) & 32767]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(n + r) & 3], b[n], acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run32(const char* name, const std::vector<float>& host) {
  float *in, *out;
  hipMalloc(&in, 32768 * 4); hipMemcpy(in, host.data(), 32768 * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, in, out, 1000);
  hipDeviceSynchronize();
  const int iters = 200000;   // a 32x32x2 fp32 MFMA is 16 passes = 64 cycles
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(e0);
    for (int l = 0; l < 3; ++l) hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f ms  %7.1f TFLOP/s fp32 MFMA (peak 157.3 at 2.4 GHz)  = %.2f GHz if the pipe never idles\n", name, ms,
           3.0 * iters * 16 * 8 * 256 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12, 3.0 * iters * 16 * 2 * 64 / (ms * 1e-3) / 1e9);
  }
  hipFree(in); hipFree(out);
}
void run(const char* name, const std::vector<float>& host) {
  float *in, *out;
  hipMalloc(&in, 32768 * 4); hipMemcpy(in, host.data(), 32768 * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, 1000);
  hipDeviceSynchronize();
  const int iters = 400000, reps = 3;   // 16 MFMAs per iteration and wave: ~0.1 s per launch at full rate
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    for (int l = 0; l < 3; ++l) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 3.0 * iters * 16 * 2;
    const double tf = 3.0 * iters * 16 * 8 * 256 * 32768.0 * 2 / 2 / (ms * 1e-3) / 1e12 * 1.0;
    printf("%-28s %8.1f ms  %7.1f TFLOP/s dense f16 (peak 2516 at 2.4 GHz)  = %.2f GHz if the pipe never idles\n", name, ms,
           3.0 * iters * 16 * 8 * 256 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12, mfma_per_simd * 32 / (ms * 1e-3) / 1e9);
    (void)tf;
  }
  hipFree(in); hipFree(out);
}
int main() {
  std::vector<float> z(32768, 0.f), r(32768), s(32768);
  srand(1);
  for (auto& v : r) v = (rand() / float(RAND_MAX)) * 2.f - 1.f;
  for (auto& v : s) v = ((rand() / float(RAND_MAX)) * 2.f - 1.f) * 0.01f;
  run("zero operands", z);
  run("uniform random [-1, 1)", r);
  run("uniform random x 0.01", s);
  run("zero operands (again)", z);
  run32("fp32 MFMA, zero operands", z);
  run32("fp32 MFMA, uniform random", r);
  return 0;
}
