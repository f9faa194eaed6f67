// Micro-benchmark: what one vector instruction of the split-f16 conversion piece costs on gfx950 -- alone (64 back to back) and as eight
// of them between the MFMAs of a dependent v_mfma_f32_32x32x16_f16 chain (the shape of the fine kernel's K-chunk), one and two waves per
// SIMD.  Whole sequences live inside ONE asm statement so hipcc adds nothing.  Build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// operands: %0 acc (MFMA chain), %1 t (scratch dest), %2 t2 (64-bit dest), %3 r (running max, read-modify-write), %4 a, %5 b (MFMA in),
//           %6 x (f32), %7 x2 (f32 pair), %8 h (packed f16)
#define MF "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n"
#define OP_MUL "v_mul_f32 %1, %6, %6\n"
#define OP_MAXF "v_max_f32 %1, %6, 0\n"
#define OP_PKMUL "v_pk_mul_f32 %2, %7, %7\n"
#define OP_FMAMIX "v_fma_mix_f32 %1, %8, -1.0, %6 op_sel_hi:[1,0,0]\n"
#define OP_CVTRTZ "v_cvt_pkrtz_f16_f32 %1, %6, %6\n"
#define OP_CVTF32 "v_cvt_f32_f16 %1, %8\n"
#define OP_PKMAXU "v_pk_max_u16 %3, %3, %8\n"
#define OP_PKMAXI "v_pk_max_i16 %1, %8, 0\n"
#define OP_MAX3 "v_pk_maximum3_f16 %3, %3, %8, %8\n"
#define OP_PKMAXF "v_pk_max_f16 %1, %8, 0\n"
#define X8(s) s s s s s s s s
#define RUN(seq) asm volatile(seq : "+v"(acc), "+v"(t), "+v"(t2), "+v"(r) : "v"(a), "v"(b), "v"(x), "v"(x2), "v"(h))
template <int OP, int MODE>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[(threadIdx.x + j) & 255]; b[j] = (_Float16)in[(threadIdx.x * 3 + j) & 255]; }
  f32x16 acc = {0};
  float t = 0, x = in[threadIdx.x & 255];
  f32x2 t2 = {0, 0}, x2 = {x, x};
  unsigned r = 0, h = __float_as_uint(in[(threadIdx.x + 7) & 255]);
  for (int it = 0; it < iters; ++it) {
#define CASE(N, O) if (OP == N) { if (MODE == 0) RUN(X8(X8(O))); else if (MODE == 1) RUN(X8(MF X8(O))); else RUN(X8(MF O O O O)); }
    CASE(0, OP_MUL) CASE(1, OP_MAXF) CASE(2, OP_PKMUL) CASE(3, OP_FMAMIX) CASE(4, OP_CVTRTZ) CASE(5, OP_CVTF32)
    CASE(6, OP_PKMAXU) CASE(7, OP_PKMAXI) CASE(8, OP_MAX3) CASE(9, OP_PKMAXF)
    if (OP == 10) RUN(X8(MF));
  }
  float s = t + t2[0] + t2[1] + __uint_as_float(r);
  for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int MODE> double run(int threads) {
  float *in, *out;
  hipMalloc(&in, 1024); hipMemset(in, 0, 1024);
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, MODE>), dim3(256), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, MODE>), dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(in); hipFree(out);
  return ms * 1e-3 * 2.4e9 / iters;   // cycles per loop iteration at a nominal 2.4 GHz
}
template <int OP> void row(const char* name) {
  // alone: 64 instructions per iteration and wave; between: 8 MFMAs per iteration and wave
  printf("%-22s alone %5.1f / %5.1f cycles per instruction per SIMD (1 / 2 waves);  8 between dependent MFMAs: %5.1f / %5.1f cycles per MFMA per SIMD;  4 between: %5.1f / %5.1f\n",
         name, run<OP, 0>(256) / 64, run<OP, 0>(512) / 128, run<OP, 1>(256) / 8, run<OP, 1>(512) / 16, run<OP, 2>(256) / 8, run<OP, 2>(512) / 16);
}
int main() {
  printf("bare dependent MFMA chain: %5.1f / %5.1f cycles per MFMA per SIMD (1 / 2 waves)\n", run<10, 0>(256) / 8, run<10, 0>(512) / 16);
  row<0>("v_mul_f32"); row<1>("v_max_f32"); row<2>("v_pk_mul_f32"); row<3>("v_fma_mix_f32"); row<4>("v_cvt_pkrtz_f16_f32");
  row<5>("v_cvt_f32_f16"); row<6>("v_pk_max_u16 (chain)"); row<7>("v_pk_max_i16"); row<8>("v_pk_maximum3_f16 (chain)"); row<9>("v_pk_max_f16");
  return 0;
}
