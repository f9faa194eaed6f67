// Micro-benchmark: cost of streaming weight units L2 -> LDS while 8 waves read A fragments from LDS and run MFMAs.
// MODE 0: no staging; 1: global_load_lds_dwordx4 (direct-to-LDS DMA); 2: global_load_dwordx4 + ds_write_b128;
// 3: DMA with dword (4 B/lane) granularity.  Prints cycles-equivalent time per unit (32 fragments x 2 MFMAs per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int UNIT = 32 * 1024;  // 32 fragments of 1 KiB

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const char* w, float* out, int units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  half8 b0, b1;
  for (int j = 0; j < 8; ++j) { b0[j] = (_Float16)(lane + j); b1[j] = (_Float16)(lane * 3 + j); }
  f32x16 acc0 = {0}, acc1 = {0};
  uint32_t cur = 0, nxt = UNIT;
  for (int u = 0; u < units; ++u) {
    const char* src = w + (size_t)(u & 7) * UNIT + lane * 16;
    f32x4 stage[4];
    if (MODE == 1) {
      for (int p = wave * 1024; p < UNIT; p += 8 * 1024)
        __builtin_amdgcn_global_load_lds((const void*)(src + p), LDSP(smem + nxt + p), 16, 0, 0);
    } else if (MODE == 3) {
      for (int p = wave * 256; p < UNIT; p += 8 * 256)
        __builtin_amdgcn_global_load_lds((const void*)(w + (size_t)(u & 7) * UNIT + lane * 4 + p), LDSP(smem + nxt + p), 4, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage[i] = *reinterpret_cast<const f32x4*>(src + (wave + 8 * i) * 1024);
    }
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
      const half8 a = *reinterpret_cast<const half8*>(smem + cur + t * 1024 + lane * 16);
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b0));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b1));
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(smem + nxt + (wave + 8 * i) * 1024 + lane * 16) = stage[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t tswap = cur; cur = nxt; nxt = tswap;
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  char* w; float* out;
  hipMalloc(&w, 8 * UNIT); hipMemset(w, 0, 8 * UNIT);
  hipMalloc(&out, 256 * 512 * 4);
  const int units = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * UNIT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 2 * UNIT, 0, w, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 2 * UNIT, 0, w, out, units);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us_per_unit = ms * 1e3 / units;
  printf("%-34s %.3f us/unit  (ideal MFMA: 2 waves x 64 MFMA x 32 clk = 4096 clk = %.3f us at 2.1 GHz)  -> %.0f clk\n", name, us_per_unit,
         4096 / 2100.0, us_per_unit * 2100);
}
int main() {
  run<0>("no staging");
  run<1>("LDS-DMA dwordx4");
  run<2>("global_load_dwordx4 + ds_write_b128");
  run<3>("LDS-DMA dword");
  return 0;
}
