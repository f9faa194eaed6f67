// Probe: what ds_read_b64_tr_b16 returns.  LDS holds half-word i = i (u16); lane l supplies byte address addr(l) (three address
// patterns); prints, per lane, the four u16 it received.  Used to pin the operand-fragment reads of the weight-gradient kernel
// (nerfh_train_fused.hip: the [point][feature] activation image is read as [feature][point] MFMA operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = uint16_t(i);
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = l * 8;                                   // lane-linear 8 bytes
  else if (mode == 1) addr = (l & 15) * 32 + (l >> 4) * 8;       // stride 32 within a 16-lane group
  else addr = ((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 1024;  // rows of 64 B, groups 1 KiB apart
  addr += (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16;
  out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (values = half-word index in LDS)\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
