// Micro-benchmark of the split-f16 MLP inner loop (one "unit" = 2 M-blocks x 8 chunks x 3 MFMAs per wave, 8 waves per CU):
// which ingredient costs what — LDS fragment reads, the 8-VALU split epilogue piece (block / spread / none), one or two
// accumulators, the unit barrier, the weight DMA (L2 -> LDS) and who issues it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void dma128(const void* g, const char* l) {
  const uint32_t off = (uint32_t)(size_t)LDSP(l);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(off) : "memory");
}
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PA(t0, t1, x0, x1, os) asm volatile("v_mul_f32 %0, %2, %4\n v_mul_f32 %1, %3, %4\n v_max_f32 %0, %0, 0\n v_max_f32 %1, %1, 0" : "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(os))
#define PB(h, t0, t1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_fma_mix_f32 %1, %0, -1.0, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %0, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(h), "+v"(t0), "+v"(t1))
#define PC(l, t0, t1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(l) : "v"(t0), "v"(t1))

// MODE bits: 1 = fragments from LDS, 2 = epilogue piece as a block after the 3 MFMAs, 4 = epilogue spread between the MFMAs,
// 8 = two accumulators (two M-blocks interleaved), 16 = barrier per unit, 32 = DMA by all 8 waves, 64 = DMA by waves 0-3 only
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* in, const char* blob, float* out, int units, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8 bh[8], bl[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const _Float16 vh = (_Float16)in[(threadIdx.x * 3 + c) & 255], vl = (_Float16)in[(threadIdx.x * 5 + c) & 255];
    bh[c] = half8{vh, vh, vh, vh, vh, vh, vh, vh}; bl[c] = half8{vl, vl, vl, vl, vl, vl, vl, vl};
  }
  half8 ah = bh[0], al = bl[1];
  f32x16 acc0 = {0}, acc1 = {0};
  float x0 = in[lane], x1 = in[lane + 1], os = in[lane + 2] + 1.f, t0 = 0, t1 = 0;
  unsigned h = 0, l = 0, sink = 0;
  const unsigned long long c_begin = __builtin_amdgcn_s_memtime();
  if (MODE & 1) { for (int i = threadIdx.x; i < 3 * 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = in[i & 255]; __syncthreads(); }
  if (MODE & 512) {
    for (int u = 0; u < units * 6; ++u)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\n"
                   "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\n" : "+v"(acc0) : "v"(ah), "v"(al));
  } else
  for (int u = 0; u < units; ++u) {
    const char* ub = smem + (u % 3) * 32768 + lane * 16;
#pragma unroll
    for (int mb = 0; mb < ((MODE & 8) ? 1 : 2); ++mb) {
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        if ((MODE & 16) && mb == 0 && kc == 4) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __builtin_amdgcn_s_barrier();
          if (MODE & (32 | 64)) {
            const int dw = (MODE & 64) ? 4 : 8;
            if (wave < dw)
              for (uint32_t p = wave * 1024; p < 32768; p += dw * 1024)
                dma128(blob + (size_t)((u * 7 + blockIdx.x) % 24) * 32768 + p + lane * 16, smem + ((u + 2) % 3) * 32768 + p);
          }
        }
        half8 a0h = ah, a0l = al, a1h = ah, a1l = al;
        if (MODE & 1) {
          a0h = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc) * 2048);
          a0l = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc) * 2048 + 1024);
          if (MODE & 8) { a1h = *reinterpret_cast<const half8*>(ub + (8 + kc) * 2048); a1l = *reinterpret_cast<const half8*>(ub + (8 + kc) * 2048 + 1024); }
        }
        const int kb = (MODE & 128) ? 0 : kc;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 8) {
          MFMA(acc0, a0h, bh[kb]); if (MODE & 4) PA(t0, t1, x0, x1, os);
          MFMA(acc1, a1h, bh[kb]); if (MODE & 4) PB(h, t0, t1);
          MFMA(acc0, a0h, bl[kb]); if (MODE & 4) { PC(l, t0, t1); sink += h + l; }
          MFMA(acc1, a1h, bl[kb]); if (MODE & 4) PA(t0, t1, x0, x1, os);
          MFMA(acc0, a0l, bh[kb]); if (MODE & 4) PB(h, t0, t1);
          MFMA(acc1, a1l, bh[kb]); if (MODE & 4) { PC(l, t0, t1); sink += h + l; }
          if (MODE & 2) { PA(t0, t1, x0, x1, os); PB(h, t0, t1); PC(l, t0, t1); sink += h + l; PA(t0, t1, x1, x0, os); PB(h, t0, t1); PC(l, t0, t1); sink += h + l; }
        } else if (MODE & 1024) {
          MFMA(acc0, a0h, bh[kb]); if (MODE & 4) PA(t0, t1, x0, x1, os);
          MFMA(acc1, a0h, bl[kb]); if (MODE & 4) PB(h, t0, t1);
          MFMA(acc1, a0l, bh[kb]); if (MODE & 4) { PC(l, t0, t1); sink += h + l; }
          if (MODE & 2) { PA(t0, t1, x0, x1, os); PB(h, t0, t1); PC(l, t0, t1); sink += h + l; }
          if (kc == 7) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc0[j]) : "v"(acc1[j]));
          }
        } else {
          MFMA(acc0, a0h, bh[kb]); if (MODE & 4) PA(t0, t1, x0, x1, os);
          MFMA(acc0, a0h, bl[kb]); if (MODE & 4) PB(h, t0, t1);
          MFMA(acc0, a0l, bh[kb]); if (MODE & 4) { PC(l, t0, t1); sink += h + l; }
          if (MODE & 2) { PA(t0, t1, x0, x1, os); PB(h, t0, t1); PC(l, t0, t1); sink += h + l; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(sink);
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - c_begin;
}
__global__ __launch_bounds__(512) void k2(const float* in, float* out, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[(threadIdx.x + j) & 255]; b[j] = (_Float16)in[(threadIdx.x * 3 + j) & 255]; }
  f32x16 acc0 = {0};
  for (int it = 0; it < iters; ++it)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\n"
                   "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\nv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\ns_nop 0\n" : "+v"(acc0) : "v"(a), "v"(b));
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run2(int insize) {
  float *in, *out;
  hipMalloc(&in, insize); hipMemset(in, 0, insize);
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 24000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("k2 (mfma_dep kernel), in %d bytes: %.3f ms -> %.1f cycles per MFMA per SIMD\n", insize, ms, ms * 1e-3 * 2.4e9 / (double(iters) * 8 * 2));
  hipFree(in); hipFree(out);
}

// One wave per SIMD (4 waves per CU, 512 registers each): NB point blocks per wave share every fragment read.
template <int NB, int MODE>
__global__ __launch_bounds__(256, 1) void k3(const float* in, const char* blob, float* out, int units, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8 bh[NB][8], bl[NB][8];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const _Float16 vh = (_Float16)in[(threadIdx.x * 3 + c + n) & 255], vl = (_Float16)in[(threadIdx.x * 5 + c + n) & 255];
      bh[n][c] = half8{vh, vh, vh, vh, vh, vh, vh, vh}; bl[n][c] = half8{vl, vl, vl, vl, vl, vl, vl, vl};
    }
  f32x16 acc[NB];
  for (int n = 0; n < NB; ++n) acc[n] = f32x16{0};
  float x0 = in[lane], x1 = in[lane + 1], os = in[lane + 2] + 1.f, t0 = 0, t1 = 0;
  unsigned h = 0, l = 0, sink = 0;
  for (int i = threadIdx.x; i < 3 * 32768 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = in[i & 255];
  __syncthreads();
  const unsigned long long c_begin = __builtin_amdgcn_s_memtime();
  for (int u = 0; u < units; ++u) {
    const char* ub = smem + (u % 3) * 32768 + lane * 16;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        if ((MODE & 16) && mb == 0 && kc == 4) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __builtin_amdgcn_s_barrier();
          if (MODE & 32)
            for (uint32_t p = wave * 1024; p < 32768; p += 4 * 1024)
              dma128(blob + (size_t)((u * 7 + blockIdx.x) % 24) * 32768 + p + lane * 16, smem + ((u + 2) % 3) * 32768 + p);
        }
        const half8 ah = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc) * 2048);
        const half8 al = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc) * 2048 + 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) { MFMA(acc[n], ah, bh[n][kc]); if (MODE & 4) PA(t0, t1, x0, x1, os); }
#pragma unroll
        for (int n = 0; n < NB; ++n) { MFMA(acc[n], ah, bl[n][kc]); if (MODE & 4) PB(h, t0, t1); }
#pragma unroll
        for (int n = 0; n < NB; ++n) { MFMA(acc[n], al, bh[n][kc]); if (MODE & 4) { PC(l, t0, t1); sink += h + l; } }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int n = 0; n < NB; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(sink);
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - c_begin;
}
template <int NB, int MODE> void run3(const char* name) {
  float *in, *out; char* blob; unsigned long long* cyc;
  hipMalloc(&cyc, 256 * 8 * 8); hipMemset(cyc, 0, 256 * 8 * 8);
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipMalloc(&blob, 24 * 32768); hipMemset(blob, 0, 24 * 32768);
  hipMalloc(&out, 256 * 512 * 4);
  const int units = 4000, lds = 3 * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k3<NB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k3<NB, MODE>), dim3(256), dim3(256), lds, 0, in, blob, out, 400, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k3<NB, MODE>), dim3(256), dim3(256), lds, 0, in, blob, out, units, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = double(units) * 48 * NB;
  printf("%-58s %.3f ms  wall %.1f cycles per MFMA per SIMD (ideal 32)\n", name, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
  hipFree(in); hipFree(out); hipFree(blob); hipFree(cyc);
}

// Ping-pong: waves 0-3 and 4-7 of a workgroup (the two waves of each SIMD) alternate a matrix segment (one M-block: 8 chunks x 3 MFMAs,
// fragments from LDS) with a vector segment (the 64-VALU split epilogue, the weight DMA), one s_barrier between segments.
// MODE bit 1: DMA (by the lagging group, in its vector segment at unit boundaries); bit 2: fragment prefetch depth 2; bit 4: no VALU
template <int MODE>
__global__ __launch_bounds__(512, 2) void k4(const float* in, const char* blob, float* out, int units, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool lag = wave >= 4;
  half8 bh[8], bl[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const _Float16 vh = (_Float16)in[(threadIdx.x * 3 + c) & 255], vl = (_Float16)in[(threadIdx.x * 5 + c) & 255];
    bh[c] = half8{vh, vh, vh, vh, vh, vh, vh, vh}; bl[c] = half8{vl, vl, vl, vl, vl, vl, vl, vl};
  }
  f32x16 acc0 = {0};
  float x0 = in[lane], x1 = in[lane + 1], os = in[lane + 2] + 1.f, t0 = 0, t1 = 0;
  unsigned h = 0, l = 0, sink = 0;
  for (int i = threadIdx.x; i < 3 * 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = in[i & 255];
  __syncthreads();
  const unsigned long long c_begin = __builtin_amdgcn_s_memtime();
  if (lag) __builtin_amdgcn_s_barrier();
  for (int u = 0; u < units; ++u) {
    const char* ub = smem + (u % 3) * 32768 + lane * 16;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      // matrix segment
      half8 ah[2], al[2];
      ah[0] = *reinterpret_cast<const half8*>(ub + (mb * 8) * 2048);
      al[0] = *reinterpret_cast<const half8*>(ub + (mb * 8) * 2048 + 1024);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        if (kc + 1 < 8) {
          ah[(kc + 1) & 1] = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc + 1) * 2048);
          al[(kc + 1) & 1] = *reinterpret_cast<const half8*>(ub + (mb * 8 + kc + 1) * 2048 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kc & 1], bh[kc], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kc & 1], bl[kc], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kc & 1], bh[kc], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_barrier();
      // vector segment
      if (!(MODE & 4)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { PA(t0, t1, acc0[2 * i], acc0[2 * i + 1], os); PB(h, t0, t1); PC(l, t0, t1); sink += h + l; }
      }
      if ((MODE & 1) && mb == 1 && lag) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (uint32_t p = (wave - 4) * 1024; p < 32768; p += 4 * 1024)
          dma128(blob + (size_t)((u * 7 + blockIdx.x) % 24) * 32768 + p + lane * 16, smem + ((u + 2) % 3) * 32768 + p);
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  if (!lag) __builtin_amdgcn_s_barrier();
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc0[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + float(sink);
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - c_begin;
}
template <int MODE> void run4(const char* name) {
  float *in, *out; char* blob; unsigned long long* cyc;
  hipMalloc(&cyc, 256 * 8 * 8); hipMemset(cyc, 0, 256 * 8 * 8);
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipMalloc(&blob, 24 * 32768); hipMemset(blob, 0, 24 * 32768);
  hipMalloc(&out, 256 * 512 * 4);
  const int units = 4000, lds = 3 * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k4<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k4<MODE>), dim3(256), dim3(512), lds, 0, in, blob, out, 400, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k4<MODE>), dim3(256), dim3(512), lds, 0, in, blob, out, units, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = double(units) * 48 * 2;
  printf("%-58s %.3f ms  wall %.1f cycles per MFMA per SIMD (ideal 32)\n", name, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
  hipFree(in); hipFree(out); hipFree(blob); hipFree(cyc);
}
template <int MODE> void run(const char* name) {
  float *in, *out; char* blob; unsigned long long* cyc;
  hipMalloc(&cyc, 256 * 8 * 8);
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipMalloc(&blob, 24 * 32768); hipMemset(blob, 0, 24 * 32768);
  hipMalloc(&out, 256 * 512 * 4);
  const int units = 4000, lds = (MODE & 256) ? 0 : 3 * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, in, blob, out, 400, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, in, blob, out, units, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = double(units) * 48 * 2;
  static unsigned long long hc[2048];
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  double tot = 0, old4 = 0, young4 = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { tot += hc[b * 8 + w]; }
  if (name) printf("%-58s %.3f ms  wall %.1f  shader-clock %.1f cycles per MFMA per SIMD (ideal 32)  clock %.2f GHz\n", name, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd,
         tot / 2048 / mfma_per_simd, tot / 2048 / (ms * 1e-3) / 1e9);
  hipFree(cyc);
  hipFree(in); hipFree(out); hipFree(blob);
}
int main() {
  for (int i = 0; i < 10; ++i) run<0>(nullptr);
  run4<0>("ping-pong, VALU, no DMA");
  run4<1>("ping-pong, VALU, DMA");
  run4<4>("ping-pong, no VALU, no DMA");
  run4<5>("ping-pong, no VALU, DMA");
  run4<1>("ping-pong, VALU, DMA (again)");
  run<1 | 2 | 16 | 32>("1 acc, VALU block, DMA 8 waves (today)");
  run<1 | 2 | 16 | 64>("1 acc, VALU block, DMA 4 waves");
  run<1 | 2 | 16 | 64 | 1024>("main+corr accs, VALU block, DMA 4 waves");
  run<1 | 4 | 16 | 64 | 1024>("main+corr accs, VALU spread, DMA 4 waves");
  run<1 | 2 | 8 | 16 | 64>("2 M-blocks interleaved, VALU block, DMA 4 waves");
  run<1 | 4 | 8 | 16 | 64>("2 M-blocks interleaved, VALU spread, DMA 4 waves");
  run<1 | 8 | 16 | 64>("2 M-blocks interleaved, no VALU, DMA 4 waves");
  run<1 | 16 | 64>("1 acc, no VALU, DMA 4 waves");
  run<1 | 16>("1 acc, no VALU, no DMA");
  run<1 | 8 | 16>("2 M-blocks interleaved, no VALU, no DMA");
  return 0;
  run<0>("regs, no VALU, 1 acc");
  run<1>("LDS frags, no VALU, 1 acc");
  run<1 | 2>("LDS frags, VALU block, 1 acc");
  run<1 | 4>("LDS frags, VALU spread, 1 acc");
  run<1 | 8>("LDS frags, no VALU, 2 acc");
  run<1 | 2 | 8>("LDS frags, VALU block, 2 acc");
  run<1 | 4 | 8>("LDS frags, VALU spread, 2 acc");
  run<1 | 2 | 16>("LDS frags, VALU block, 1 acc, barrier");
  run<1 | 4 | 16>("LDS frags, VALU spread, 1 acc, barrier");
  run<1 | 2 | 16 | 32>("LDS frags, VALU block, 1 acc, barrier, DMA 8 waves");
  run<1 | 4 | 16 | 32>("LDS frags, VALU spread, 1 acc, barrier, DMA 8 waves");
  run<1 | 4 | 16 | 64>("LDS frags, VALU spread, 1 acc, barrier, DMA 4 waves");
  run<1 | 4 | 8 | 16 | 32>("LDS frags, VALU spread, 2 acc, barrier, DMA 8 waves");
  run<1 | 4 | 8 | 16 | 64>("LDS frags, VALU spread, 2 acc, barrier, DMA 4 waves");
  run<1 | 16 | 32>("LDS frags, no VALU, 1 acc, barrier, DMA 8 waves");
  return 0;
}
