// Micro-benchmark: issue cadence of DEPENDENT v_mfma_f32_32x32x16_f16 chains (same accumulator) with other instructions in
// between, one and two waves per SIMD.  Whole sequences live inside ONE asm statement so hipcc adds nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define M0 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n"
#define M1 "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
#define N1 "s_nop 0\n"
#define N4 "s_nop 3\n"
#define V1 "v_mul_f32 %4, %5, %5\n"
#define V2 V1 "v_mul_f32 %4, %5, %5\n"
#define V4 V2 V2
#define X4(s) s s s s
#define RUN1(seq) asm volatile(seq : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b), "v"(t), "v"(x))
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[(threadIdx.x + j) & 255]; b[j] = (_Float16)in[(threadIdx.x * 3 + j) & 255]; }
  f32x16 acc0 = {0}, acc1 = {0};
  float t = 0, x = in[threadIdx.x & 255];
  float live[MODE >= 100 ? 160 : 1];
  if (MODE >= 100) for (int i = 0; i < 160; ++i) live[i] = in[(threadIdx.x + i) & 255];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 100) { for (int i = 0; i < 160; ++i) asm volatile("" : "+v"(live[i])); RUN1(X4(M0 N1 M0 N1)); }
#define RUN(seq) asm volatile(seq : "+v"(acc0), "+v"(acc1), "+v"(t) : "v"(a), "v"(b), "v"(x))
#undef RUN
#define RUN(seq) asm volatile(seq : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b), "v"(t), "v"(x))
    if (MODE == 0) RUN(X4(M0 M0));
    if (MODE == 1) RUN(X4(M0 N1 M0 N1));
    if (MODE == 2) RUN(X4(M0 N4 M0 N4));
    if (MODE == 3) RUN(X4(M0 V1 M0 V1));
    if (MODE == 4) RUN(X4(M0 V4 M0 V4));
    if (MODE == 5) RUN(X4(M0 M1));
    if (MODE == 6) RUN(X4(M0 N1 M1 N1));
    if (MODE == 7) RUN(X4(M0 V2 M1 V2));
    if (MODE == 8) RUN(X4(M0 V4 M1 V4));
    if (MODE == 9) RUN(X4(M0 M0 M0 V4 V4 M1 M1 M1 V4 V4));   // 24 MFMAs per RUN
    if (MODE == 10) RUN(X4(M0 V2 M0 V2 M0 V4 M1 V2 M1 V2 M1 V4));
    if (MODE == 11) RUN(X4(M0 V2 M1 V2 M0 V2 M1 V2 M0 V4 M1 V4));
  }
  float s = t;
  if (MODE >= 100) for (int i = 0; i < 160; ++i) s += live[i];
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int threads, int per) {
  float *in, *out;
  hipMalloc(&in, 1024); hipMemset(in, 0, 1024);
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %d wave/SIMD: %6.1f cycles per MFMA per SIMD @2.4GHz\n", name, threads / 256, ms * 1e-3 * 2.4e9 / (double(iters) * per * (threads / 256)));
  hipFree(in); hipFree(out);
}
#define BOTH(M, name, per) run<M>(name, 256, per); run<M>(name, 512, per);
int main(int argc, char** argv) {
  if (argc > 2) { for (int i = 0; i < 10; ++i) { run<1>("few VGPRs: same acc, s_nop 0 between", 512, 8); run<100>("160 live VGPRs: same acc, s_nop 0 between", 512, 8); } return 0; }
  if (argc > 1) { for (int i = 0; i < 100; ++i) run<1>("sustained: same acc, s_nop 0 between", 512, 8); return 0; }
  BOTH(0, "same acc, back-to-back", 8)
  BOTH(1, "same acc, s_nop 0 between", 8)
  BOTH(2, "same acc, s_nop 3 between", 8)
  BOTH(3, "same acc, 1 VALU between", 8)
  BOTH(4, "same acc, 4 VALU between", 8)
  BOTH(5, "two accs alternating", 8)
  BOTH(6, "two accs alternating, s_nop 0 between", 8)
  BOTH(7, "two accs alternating, 2 VALU between", 8)
  BOTH(8, "two accs alternating, 4 VALU between", 8)
  BOTH(9, "3x acc0, 8 VALU, 3x acc1, 8 VALU", 24)
  BOTH(10, "acc0 V2 acc0 V2 acc0 V4, then acc1 same", 24)
  BOTH(11, "acc0 V2 acc1 V2 ... (2 chains, 2.67 VALU/MFMA)", 24)
  return 0;
}
