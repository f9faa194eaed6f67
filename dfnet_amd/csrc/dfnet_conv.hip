// dfnet_conv.hip — DFNet's VGG-style feature pyramid on the CDNA4 matrix cores (gfx950 only).
//
// Implicit-GEMM convolution, stride 1, "same" zero padding, KS in {1,3,5}: the weights are the MFMA
// A operand (M = 32 output channels per M-block), 32 consecutive output pixels of an image row are
// the B operand (N = 32), the contraction runs over (input-channel block, ky, kx, channel chunk).
// A workgroup of 4 wavefronts computes an 8x32-pixel output tile for MB M-blocks; each wavefront
// owns two image rows (NB = 2) so an A fragment feeds 2 MFMAs and a B fragment feeds MB.
// The input patch (tile + halo) of one 32-channel block sits in LDS with a padded pixel stride
// (conflict-free ds_read_b128); the weights of one (block, ky) slice stream L2 -> LDS by
// direct-to-LDS DMA.  Bias is preloaded into the accumulators; the epilogue writes the pre-ReLU
// hypercolumn tap and/or the ReLU'd activation in the blocked-permuted NHWC layout
// (dfnet_kernels.h) as one contiguous 32/64-byte run per lane.
//
// Replaces (reference): torchvision VGG16 `features` convs driven by
// /root/reference/script/feature/dfnet.py:121-136 and the AdaptLayers convs (:57-62, BatchNorm
// folded into the 5x5 weights on the host), the maxpools, UpsamplingBilinear2d (:145) and the
// GAP+FC pose head (:168-170).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include "dfnet_kernels.h"
#include "mfma_frag.h"

namespace dfn {

#define DFN_DEV_INLINE __device__ __forceinline__

template <class P> struct ConvGeom;
template <> struct ConvGeom<PrecF16> { static constexpr int kPad = 16, kVec = 16; };
template <> struct ConvGeom<PrecF32> { static constexpr int kPad = 4, kVec = 4; };

template <class P, int KS, int SB>
constexpr int conv_patch_bytes() {
  constexpr int ps = 2 * SB * int(sizeof(typename FragOf<P>::elem)) + ConvGeom<P>::kPad;
  return ((kConvTileH + KS - 1) * (kConvTileW + KS - 1) * ps + 15) & ~15;
}
template <class P, int KS, int SB, int MB>
constexpr int conv_wstage_bytes() { return MB * KS * (SB / P::kSlotsPerChunk) * 64 * P::kLaneBytes; }

// fp32 NCHW planes straight from the accumulators (ConvArgs::out_nchw): for a fixed accumulator register the 32 lanes of a
// half-wave hold 32 consecutive pixels of ONE channel — a 128-byte run of that channel's plane.  T = the activation type
// the unfused path would have rounded through.
// (yrow, x) = the lane's pixel in fragment 0; fragment 1 lies RF rows below.
template <int MB, class T, int RF>
DFN_DEV_INLINE void store_nchw(const ConvArgs& a, const f32x16 (&acc)[MB][2], float scale, int b, int cg, int yrow, int x, int h) {
  const size_t hw = (size_t)a.H * a.W;
  float* img = a.out_nchw + (size_t)(b / a.nchw_split) * a.nchw_group_stride + (size_t)(b % a.nchw_split) * a.cout_blocks * 32 * hw;
  if (x >= a.W) return;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int y = yrow + nb * RF;
    if (y >= a.H) continue;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float* o = img + ((size_t)(cg * MB + mb) * 32 + 4 * h) * hw + (size_t)y * a.W + x;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * hw] = (float)(T)(acc[mb][nb][r] * scale);
    }
  }
}

template <class P, int KS, int SB, int MB>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvArgs a) {
  using T = typename FragOf<P>::elem;
  using F = typename FragOf<P>::type;
  constexpr int SPC = P::kSlotsPerChunk, LB = P::kLaneBytes;
  constexpr int KCB = SB / SPC;
  constexpr int TH = kConvTileH, TW = kConvTileW, R = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int PIXB = 2 * SB * int(sizeof(T));
  constexpr int PS = PIXB + ConvGeom<P>::kPad;
  constexpr int VEC = ConvGeom<P>::kVec, SEG = PIXB / VEC;
  constexpr int WST = conv_wstage_bytes<P, KS, SB, MB>();
  static_assert(WST % 1024 == 0, "weight stage must be whole 1 KiB DMA pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wst = smem + conv_patch_bytes<P, KS, SB>();

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int tiles_x = (a.W + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int cg = blockIdx.y, b = blockIdx.z;

  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const f32x4* bq = reinterpret_cast<const f32x4*>(a.bias + ((cg * MB + mb) * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = bq[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[mb][0][4 * q + i] = v[i]; acc[mb][1][4 * q + i] = v[i]; }
    }
  }

  const char* in = static_cast<const char*>(a.in);
  for (int blk = 0; blk < a.nblk_in; ++blk) {
    __syncthreads();  // everyone is done with the previous patch and weight slice
    for (int e = tid; e < PH * PW * SEG; e += 256) {
      const int pix = e / SEG, seg = e - pix * SEG;
      const int py = pix / PW, px = pix - py * PW;
      const int gy = y0 + py - R, gx = x0 + px - R;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const size_t src = ((((size_t)b * a.H + (ok ? gy : 0)) * a.W + (ok ? gx : 0)) * a.nblk_in + blk) * PIXB + seg * VEC;
      if constexpr (VEC == 16) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(in + src);
        *reinterpret_cast<f32x4*>(patch + pix * PS + seg * VEC) = v;
      } else {
        float v = 0.f;
        if (ok) v = *reinterpret_cast<const float*>(in + src);
        *reinterpret_cast<float*>(patch + pix * PS + seg * VEC) = v;
      }
    }
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {
      if (ky) __syncthreads();  // previous slice fully consumed
      const char* wsrc = a.w + (((size_t)cg * a.nblk_in + blk) * KS + ky) * WST + lane * 16;
      for (int q = wave * 1024; q < WST; q += 4096)
        __builtin_amdgcn_global_load_lds((const void*)(wsrc + q), DFN_LDS_PTR(wst + q), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int kc = 0; kc < KCB; ++kc) {
          F bf[2];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            bf[nb] = *reinterpret_cast<const F*>(patch + ((2 * wave + nb + ky) * PW + p + kx) * PS +
                                                 (h * SB + kc * SPC) * int(sizeof(T)));
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const F af = *reinterpret_cast<const F*>(wst + (((mb * KS + kx) * KCB + kc) * 64 + lane) * LB);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma<P>(af, bf[nb], acc[mb][nb]);
          }
        }
      }
    }
  }

  // Epilogue: lane (p, h) owns pixel (y0 + 2*wave + nb, x0 + p), 16 results per M-block = half of the pixel's channel line.
  // Stored straight from the accumulators every instruction would touch 32 lines partially; each wave turns one
  // (row, M-block) at a time through LDS (the staging buffers are dead) and stores whole lines (see conv_x3_kernel).
  constexpr int LINEB = 32 * int(sizeof(T)), ROWB = LINEB + 16, LPL = LINEB / 16;   // lanes per line
  if (a.out_nchw) store_nchw<MB, T, 1>(a, acc, 1.f, b, cg, y0 + 2 * wave, x0 + p, h);
  if (!a.out_act && !a.out_pre) return;
  __syncthreads();
  char* turn = smem + wave * (32 * ROWB);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int y = y0 + 2 * wave + nb;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      alignas(16) T pre[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pre[r] = (T)acc[mb][nb][r];
      f32x4* d = reinterpret_cast<f32x4*>(turn + p * ROWB + h * (LINEB / 2));
#pragma unroll
      for (int q = 0; q < LINEB / 32; ++q) d[q] = reinterpret_cast<const f32x4*>(pre)[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (y < a.H) {
        const size_t rowoff = ((((size_t)b * a.H + y) * a.W + x0) * a.cout_blocks + (cg * MB + mb)) * 32;   // elements
#pragma unroll
        for (int i = 0; i < 32 * LPL / 64; ++i) {
          const int px = i * (64 / LPL) + lane / LPL, chunk = lane % LPL;
          if (x0 + px < a.W) {
            alignas(16) T v[16 / sizeof(T)];
            *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(turn + px * ROWB + chunk * 16);
            const size_t off = rowoff + (size_t)px * a.cout_blocks * 32 + chunk * (16 / sizeof(T));
            if (a.out_pre) *reinterpret_cast<f32x4*>(static_cast<T*>(a.out_pre) + off) = *reinterpret_cast<const f32x4*>(v);
            if (a.out_act) {
              if (a.relu) {
#pragma unroll
                for (int k = 0; k < int(16 / sizeof(T)); ++k) v[k] = (T)fmaxf((float)v[k], 0.f);
              }
              *reinterpret_cast<f32x4*>(static_cast<T*>(a.out_act) + off) = *reinterpret_cast<const f32x4*>(v);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <class P, int KS, int SB, int MB>
static hipError_t launch_conv_t(const ConvArgs& a, hipStream_t stream) {
  if (a.cout_blocks % MB) return hipErrorInvalidValue;
  constexpr int lds = conv_patch_bytes<P, KS, SB>() + conv_wstage_bytes<P, KS, SB, MB>();
  auto kern = conv_kernel<P, KS, SB, MB>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = ((a.H + kConvTileH - 1) / kConvTileH) * ((a.W + kConvTileW - 1) / kConvTileW);
  hipLaunchKernelGGL(kern, dim3(tiles, a.cout_blocks / MB, a.B), dim3(256), lds, stream, a);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------ split-f16 ("x3") convolution
// fp32-grade results at f16 MFMA rate (v_mfma_f32_32x32x16_f16 is 16x faster than v_mfma_f32_32x32x2_f32): every
// operand is split x = hi + lo with hi = f16(x), lo = f16(x - hi), and a product is accumulated in fp32 as
// hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-22 relative).  Activations live in HBM as fp32 in the
// blocked layout and are split while the input patch is staged into two f16 LDS planes; weights are split on the
// host and staged as [hi fragments][lo fragments].  Both operands are pre-scaled by powers of two (activations
// x kConvActScale, weights x2^s per layer, ConvArgs::out_scale undoes it exactly) so the lo parts stay normal f16.
// The weight sub-slices (one per input block, kernel row and K-chunk: 12 KB for a 3x3) are double-buffered — small enough
// that TWO workgroups still fit a CU —: the DMA of sub-slice s+1 is in flight
// while slice s is multiplied.
#ifdef DFN_TIMING
// Cycle accounting of conv_x3_kernel (timing build only, tools/gpu_conv_timing.py): lane 0 of every wave adds the
// cycles it spent in [0] DMA/patch wait, [1] barrier, [2] patch split + stores, [3] fragment reads + MFMA issue,
// [4] epilogue, [5] whole kernel, [6] waves counted.
__device__ unsigned long long g_conv_cycles[8];
#define CONV_T_DECL unsigned long long ct_last = __builtin_amdgcn_s_memtime(), ct_begin = ct_last, ct[5] = {0, 0, 0, 0, 0}
#define CONV_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ct[i] += now_ - ct_last; ct_last = now_; } while (0)
#define CONV_T_FLUSH do { if (lane == 0) { for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&g_conv_cycles[i_], ct[i_]); \
    atomicAdd(&g_conv_cycles[5], __builtin_amdgcn_s_memtime() - ct_begin); atomicAdd(&g_conv_cycles[6], 1ull); } } while (0)
#else
#define CONV_T_DECL
#define CONV_T(i)
#define CONV_T_FLUSH
#endif

// s_waitcnt vmcnt(n) with lgkmcnt / expcnt left alone (gfx9 encoding: vmcnt = bits [3:0] and [15:14])
#define DFN_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))

DFN_DEV_INLINE void vmcnt_upto(int n) {     // s_waitcnt vmcnt(n) for a value known after unrolling
  switch (n) {
#define DFN_VMC(k) case k: DFN_VMCNT(k); break;
    DFN_VMC(0) DFN_VMC(1) DFN_VMC(2) DFN_VMC(3) DFN_VMC(4) DFN_VMC(5) DFN_VMC(6) DFN_VMC(7) DFN_VMC(8) DFN_VMC(9) DFN_VMC(10) DFN_VMC(11)
    DFN_VMC(12) DFN_VMC(13) DFN_VMC(14) DFN_VMC(15) DFN_VMC(16) DFN_VMC(17) DFN_VMC(18) DFN_VMC(19) DFN_VMC(20) DFN_VMC(21) DFN_VMC(22)
    DFN_VMC(23) DFN_VMC(24) DFN_VMC(25) DFN_VMC(26) DFN_VMC(27) DFN_VMC(28) DFN_VMC(29) DFN_VMC(30) DFN_VMC(31) DFN_VMC(32) DFN_VMC(33)
    DFN_VMC(34) DFN_VMC(35) DFN_VMC(36) DFN_VMC(37) DFN_VMC(38) DFN_VMC(39) DFN_VMC(40)
#undef DFN_VMC
    default: DFN_VMCNT(0);
  }
}

DFN_DEV_INLINE void conv_lds_dma_b128(const void* gptr, const char* lds_dst) {
  const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)DFN_LDS_PTR(lds_dst));   // wave-uniform by construction
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(off) : "memory");
}

// Split-f16 activation storage (ConvArgs::in_split / out_split): the same bytes as the fp32 tensor, but every value is kept as the
// two f16 operands the consuming convolution multiplies — hi = f16(x * kConvActScale), lo = f16(x * kConvActScale - hi) — in ROW-PLANAR
// order [B][H][C/32][K-chunk][hi | lo][lane half h][W][8 slots]: for one image row, (block, K-chunk, plane, half) is a run of W 16-byte
// pieces.  A consumer stages its patch by LDS-DMA with no conversion, each DMA instruction fetching runs of consecutive pixels
// (conv_x3s_kernel); a producer's 32 lanes of one half store 512 contiguous bytes per instruction straight from the accumulators.
// The split is done once by the producer instead of once per (output-channel group, halo overlap) by the consumers.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
DFN_DEV_INLINE void split4(const f32x4 v, float scale, half4_t& hi, half4_t& lo) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xs = v[k] * scale;
    hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);   // saturate instead of producing inf: the lo half
    lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);  // then carries up to another 65 000
  }
}
// byte offset of the 16-byte piece (block, K-chunk kc, plane, half h) of pixel (b, y, x) in a split tensor of nblk blocks, kcb K-chunks
DFN_DEV_INLINE size_t split_piece(int H, int W, int nblk, int kcb, int b, int y, int x, int blk, int kc, int plane, int h) {
  return ((((((size_t)b * H + y) * nblk + blk) * kcb + kc) * 4 + plane * 2 + h) * W + x) * 16;
}
// one 16-byte fp32 chunk (4 of a pixel-block's 32 floats: lane half chunk / 4, slots (chunk % 4) * 4 ..) into a split tensor
DFN_DEV_INLINE void store_split_chunk(void* base, int H, int W, int nblk, int b, int y, int x, int blk, int chunk, const f32x4 v) {
  half4_t hi, lo;
  split4(v, kConvActScale, hi, lo);
  const int h = chunk >> 2, s0 = (chunk & 3) * 4;
  char* d = static_cast<char*>(base) + split_piece(H, W, nblk, 2, b, y, x, blk, s0 >> 3, 0, h) + (s0 & 7) * 2;
  *reinterpret_cast<half4_t*>(d) = hi;
  *reinterpret_cast<half4_t*>(d + 2 * (size_t)W * 16) = lo;   // plane 1 lies two sub-planes (h = 0, 1) further
}
// a lane's 16 accumulator values (pixel (y, x), half h, block blk) into a split tensor: hi / lo of both K-chunks, 16 bytes each
DFN_DEV_INLINE void store_split_frag(void* base, int H, int W, int nblk, int b, int y, int x, int blk, int h, const f32x16& v, float scale,
                                     bool relu) {
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float xs = v[8 * kc + k] * scale;
      if (relu) xs = fmaxf(xs, 0.f);
      xs *= kConvActScale;
      hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);
      lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);
    }
    char* d = static_cast<char*>(base) + split_piece(H, W, nblk, 2, b, y, x, blk, kc, 0, h);
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + 2 * (size_t)W * 16) = lo;
  }
}

// Epilogue of the split-f16 convolutions: undo the operand scaling, write the pre-ReLU tap, the ReLU'd activation, the
// 2x2-pooled activation and / or fp32 NCHW planes.
template <int MB, int TW>
DFN_DEV_INLINE void x3_epilogue(const ConvArgs& a, const f32x16 (&acc)[MB][2], float out_scale, char* smem, int wave, int lane,
                                int b, int cg, int y0, int x0) {
  constexpr int RF = 32 / TW;
  const int p = lane & 31, h = lane >> 5, pr = p / TW, pc = p % TW;
  // A lane
  // owns 64 bytes of a pixel's 128-byte channel line, so storing straight from the accumulators issues 64 quarter-line
  // requests per instruction (measured: a fifth of the kernel's time).  Each wave therefore turns its output rows
  // through LDS — the staging buffers are dead by now — and stores whole lines, eight lanes per line.
  constexpr int ROWB = MB * 128 + 16;                 // padded bytes per pixel in the turn buffer
  if (a.absmax_out) {
    // max |output| over the pixels of the image (tile padding excluded): one atomicMax per wave on the float's bit pattern
    float m = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int y = y0 + (2 * wave + nb) * RF + pr, x = x0 + pc;
      if (y < a.H && x < a.W) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[mb][nb][r]));
      }
    }
    m *= fabsf(out_scale);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    // (the word only grows: a wave whose maximum is already covered skips the atomic — thousands of same-address atomics serialise)
    if (lane == 0 && m > __uint_as_float(__hip_atomic_load(a.absmax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
      atomicMax(a.absmax_out, __float_as_uint(m));
  }
  if (a.out_nchw) store_nchw<MB, float, RF>(a, acc, out_scale, b, cg, y0 + 2 * wave * RF + pr, x0 + pc, h);
  if constexpr (MB == 2) {
    if (a.fuse_out) {
      // Fused 1x1 (64 -> 64) + ReLU: the C fragments of the two M-blocks ARE the B operand of the next product (lane (pixel, half)
      // holds slots 8 kc .. 8 kc + 7 of block mb in acc[mb][nb][8 kc ..]), split exactly as the stored tap would have been.
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        half8 bh[2][2], bl[2][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float xs = (acc[blk][nb][8 * kc + k] * out_scale) * kConvActScale;
              const _Float16 hi = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);
              bh[blk][kc][k] = hi;
              bl[blk][kc][k] = (_Float16)fminf(fmaxf(xs - (float)hi, -65000.f), 65000.f);
            }
        f32x16 acc2[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const f32x4* bq = reinterpret_cast<const f32x4*>(a.fuse_bias + (mb * 2 + h) * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = bq[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[mb][4 * q + i] = v[i];
          }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            const char* sl = a.fuse_w + (blk * 2 + kc) * 4096 + lane * 16;    // sub-slice (block, K-chunk): [hi: mb 0, 1][lo: mb 0, 1]
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
              const half8 ah = *reinterpret_cast<const half8*>(sl + mb * 1024);
              const half8 al = *reinterpret_cast<const half8*>(sl + 2048 + mb * 1024);
              acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[blk][kc], acc2[mb], 0, 0, 0);
              acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[blk][kc], acc2[mb], 0, 0, 0);
              acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[blk][kc], acc2[mb], 0, 0, 0);
            }
          }
        const int y = y0 + (2 * wave + nb) * RF + pr, x = x0 + pc;
        if (y < a.H && x < a.W) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) store_split_frag(a.fuse_out, a.H, a.W, 2, b, y, x, mb, h, acc2[mb], a.fuse_scale, true);
        }
      }
    }
  }
  // split outputs go straight from the accumulators (512-byte runs per half-wave); fp32 outputs take the turn below
  void* act32 = (a.out_split & 1) ? nullptr : a.out_act;
  void* pre32 = (a.out_split & 2) ? nullptr : a.out_pre;
  if ((a.out_act && !act32) || (a.out_pre && !pre32)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int y = y0 + (2 * wave + nb) * RF + pr, x = x0 + pc;
      if (y < a.H && x < a.W) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (a.out_pre && !pre32) store_split_frag(a.out_pre, a.H, a.W, a.cout_blocks, b, y, x, cg * MB + mb, h, acc[mb][nb], out_scale, false);
          if (a.out_act && !act32) store_split_frag(a.out_act, a.H, a.W, a.cout_blocks, b, y, x, cg * MB + mb, h, acc[mb][nb], out_scale, a.relu);
        }
      }
    }
  }
  if (!act32 && !pre32 && !a.out_pool) return;
  __syncthreads();                                     // every wave is done with the planes and the weight buffers
  char* turn = smem + wave * (32 * ROWB);
  if (a.out_pool) {
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    auto pool_store = [&](int yp, int xp0, int npx, auto src) {   // src(j, mbl, chunk) = 2x2 maximum of pooled pixel j of this pass
      if (yp >= Hp) return;
      const size_t rowoff = (((size_t)b * Hp + yp) * Wp + xp0) * a.cout_blocks * 32 + (size_t)cg * MB * 32;   // floats
#pragma unroll
      for (int i = 0; i < npx * MB / 8; ++i) {
        const int line = i * 8 + (lane >> 3), px = line / MB, mbl = line - px * MB, chunk = lane & 7;
        if (xp0 + px < Wp) {
          f32x4 v = src(px, mbl, chunk);
          if (a.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          if (a.out_split & 4) store_split_chunk(a.out_pool, Hp, Wp, a.cout_blocks, b, yp, xp0 + px, cg * MB + mbl, chunk, v);
          else *reinterpret_cast<f32x4*>(static_cast<float*>(a.out_pool) + rowoff + ((size_t)px * a.cout_blocks + mbl) * 32 + chunk * 4) = v;
        }
      }
    };
    auto mx = [](f32x4 u, f32x4 w) { return f32x4{fmaxf(u[0], w[0]), fmaxf(u[1], w[1]), fmaxf(u[2], w[2]), fmaxf(u[3], w[3])}; };
    auto at = [&](int px, int mbl, int chunk) { return *reinterpret_cast<const f32x4*>(turn + px * ROWB + mbl * 128 + chunk * 16); };
    if constexpr (RF == 1) {
      // the wave's two rows are the two rows of its pooling windows: vertical maximum in registers, horizontal through the turn buffer
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        f32x4* d = reinterpret_cast<f32x4*>(turn + p * ROWB + mb * 128 + h * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[mb][0][4 * q + k], acc[mb][1][4 * q + k]) * out_scale;
          d[q] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      pool_store((y0 + 2 * wave) >> 1, x0 >> 1, 16, [&](int j, int mbl, int chunk) { return mx(at(2 * j, mbl, chunk), at(2 * j + 1, mbl, chunk)); });
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      // a fragment holds both rows of its windows (pixels j and j + TW of the turn buffer): one pooled row of TW / 2 pixels each
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          f32x4* d = reinterpret_cast<f32x4*>(turn + p * ROWB + mb * 128 + h * 64);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            d[q] = f32x4{acc[mb][nb][4 * q] * out_scale, acc[mb][nb][4 * q + 1] * out_scale, acc[mb][nb][4 * q + 2] * out_scale,
                         acc[mb][nb][4 * q + 3] * out_scale};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pool_store(((y0 + (2 * wave + nb) * RF) >> 1), x0 >> 1, TW / 2, [&](int j, int mbl, int chunk) {
          return mx(mx(at(2 * j, mbl, chunk), at(2 * j + 1, mbl, chunk)), mx(at(TW + 2 * j, mbl, chunk), at(TW + 2 * j + 1, mbl, chunk)));
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!act32 && !pre32) return;
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      f32x4* d = reinterpret_cast<f32x4*>(turn + p * ROWB + mb * 128 + h * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        d[q] = f32x4{acc[mb][nb][4 * q] * out_scale, acc[mb][nb][4 * q + 1] * out_scale, acc[mb][nb][4 * q + 2] * out_scale,
                     acc[mb][nb][4 * q + 3] * out_scale};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-local hand-over: LDS operations of a wave stay in order
    __builtin_amdgcn_wave_barrier();
    {
      const int yf = y0 + (2 * wave + nb) * RF;   // first image row of the fragment
#pragma unroll
      for (int i = 0; i < 4 * MB; ++i) {
        const int line = i * 8 + (lane >> 3), px = line / MB, mbl = line - px * MB, chunk = lane & 7;
        const int y = yf + px / TW, x = x0 + px % TW;
        if (y < a.H && x < a.W) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(turn + px * ROWB + mbl * 128 + chunk * 16);
          const size_t off = ((((size_t)b * a.H + y) * a.W + x) * a.cout_blocks + cg * MB + mbl) * 32;   // first float of the pixel's block
          if (pre32) *reinterpret_cast<f32x4*>(static_cast<float*>(pre32) + off + chunk * 4) = v;
          if (act32) {
            const f32x4 r = a.relu ? f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)} : v;
            *reinterpret_cast<f32x4*>(static_cast<float*>(act32) + off + chunk * 4) = r;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // the second row reuses the turn buffer
  }
}

template <int KS, int SB, int WAVES = 4, int TW = 32>
constexpr int x3_plane_bytes() { return (2 * WAVES * (32 / TW) + KS - 1) * (TW + KS - 1) * SB * 4; }   // [2][SB/8][pixels][8 x f16]
template <int KS, int SB, int MB>
constexpr int x3_wslice_bytes() { return 2 * MB * KS * 1024; }  // hi + lo fragments of one (block, ky, kc) sub-slice

// WAVES = 4: 8 x 32 output pixels per workgroup, two workgroups per CU.  WAVES = 8: 16 x 32 pixels, ONE workgroup per CU — the
// same eight waves share every weight slice (half the LDS-DMA per output pixel), the patch halo is amortised over twice
// the rows.
// Staging.  The patch of one 32-channel input block lives in two f16 planes (hi, lo), each [lane half h][K-chunk][pixel][8 slots]:
// the B fragment of lane (pixel, h) is 16 contiguous bytes and consecutive lanes read consecutive 16-byte pieces — conflict-free
// without padding.  Weight sub-slices stream L2 -> LDS by DMA through a RING of buffers: sub-slice s + RING - 1 is issued when
// sub-slice s starts being multiplied, so a DMA has RING - 1 multiply phases to land (with one phase it was the wave's main
// wait: DESIGN.md section 7).  Every wave issues the same number of DMA pieces and of patch loads in every iteration (tail
// iterations repeat the last sub-slice / block: same bytes, dead destination), so the in-order vmcnt can be awaited by count.
// Tile shape.  An MFMA B fragment is 32 pixels: TW = 32 takes them from one image row (a wave's two fragments = two rows, the
// workgroup 2*WAVES x 32 pixels), TW = 16 from two rows of 16 (a wave = four rows, the workgroup 4*WAVES x 16) — chosen per layer
// by which wastes fewer pixels on the image border (launch_conv): 60 x 80 (conv4_x) fills 78 % of its 8 x 32 tiles, 94 % of 16 x 16.
// Channel tile.  The weights are packed for MBP = 2 M-blocks (64 output channels) per sub-slice; MB = 1 takes one M-block's pieces out
// of it (32 output channels per workgroup: twice the workgroups, each with half the accumulators and 53 KB of LDS — three per CU).
// launch_conv picks it for the layers whose 64-channel workgroups would leave the chip's 512 slots badly filled (60 x 80 and below).
template <int KS, int SB, int MB, int RING, int WAVES = 4, int TW = 32>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? (MB == 1 ? 3 : 2) : 1) void conv_x3_kernel(ConvArgs a) {
  constexpr int MBP = 2;                  // M-blocks per packed sub-slice (pack_conv_x3)
  constexpr int KCB = SB / 8;
  constexpr int NT = WAVES * 64;
  constexpr int RF = 32 / TW;             // image rows per fragment
  constexpr int TH = 2 * WAVES * RF, R = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1, NPIX = PH * PW;
  constexpr int PIXG = 2 * SB * 4;        // bytes of a pixel's block in HBM (fp32)
  constexpr int SEG = PIXG / 16;          // 16-byte (4-float) segments per pixel
  constexpr int PLANE = x3_plane_bytes<KS, SB, WAVES, TW>();
  constexpr int WSL = x3_wslice_bytes<KS, SB, MB>();      // what this workgroup stages of a sub-slice
  constexpr int WSLP = x3_wslice_bytes<KS, SB, MBP>();    // the packed sub-slice
  constexpr int WHALF = WSL / 2;
  constexpr int NP = WSL / 1024;                      // 1 KB DMA pieces per sub-slice
  constexpr int PPW = (NP + WAVES - 1) / WAVES;       // pieces each wave issues per sub-slice
  constexpr int AHEAD = RING - 1;                     // sub-slices in flight beyond the one being multiplied
  constexpr int SPB = KS * KCB;                       // sub-slices per input block
  static_assert(RING >= 2 && AHEAD <= SPB, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* plane_hi = smem;
  char* plane_lo = smem + PLANE;
  char* wst = smem + 2 * PLANE;           // RING sub-slices

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int pr = p / TW, pc = p % TW;   // the lane's pixel inside a fragment: row, column
  const int tiles_x = (a.W + TW - 1) / TW;
  int tile = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
  if (a.xcd_groups > 0) {
    // XCD-aware 1-D grid (layers whose packed weights exceed an XCD's 4 MB L2).  Workgroup n is dispatched to XCD n % 8:
    // give XCD k the k-th contiguous eighth of the (output-channel group)-major order, so that each XCD's L2 holds only
    // ITS groups' weights (one eighth of the layer) instead of thrashing on all of them.  The grid is padded to a
    // multiple of eight; the padding workgroups leave at once.  Speed only: any placement computes the same result.
    const int n8 = (a.xcd_groups + 7) / 8;
    const int L = (blockIdx.x & 7) * n8 + (blockIdx.x >> 3);
    if (L >= a.xcd_groups) return;
    const int tiles = tiles_x * ((a.H + TH - 1) / TH), per_cg = tiles * a.B;
    cg = L / per_cg;
    const int rem = L - cg * per_cg;
    b = rem / tiles;
    tile = rem - b * tiles;
  }
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;

  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const f32x4* bq = reinterpret_cast<const f32x4*>(a.bias + ((cg * MB + mb) * 2 + h) * 16);  // pre-scaled bias
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = bq[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[mb][0][4 * q + i] = v[i]; acc[mb][1][4 * q + i] = v[i]; }
    }
  }
  const char* in = static_cast<const char*>(a.in);
  const int n_slices = a.nblk_in * SPB;   // one per (block, ky, K-chunk)
  // operand scale of the input tensor: the fixed activation scale, or (gradient tensors) a measured power of two
  const float act_scale = a.dyn_scale ? a.dyn_scale[0] : kConvActScale;
  const float out_scale = a.dyn_scale ? a.out_scale * kConvActScale * a.dyn_scale[1] : a.out_scale;
  int ring_w = 0;                          // ring buffer the next issued sub-slice goes to
  auto issue_slice = [&](int sl) {
    const char* wsrc = a.w + ((size_t)(cg * MB / MBP) * n_slices + min(sl, n_slices - 1)) * WSLP + lane * 16;
    const int mbsel = (cg * MB) % MBP;     // first packed M-block this workgroup computes
    char* dst = wst + ring_w * WSL;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = min(wave + i * WAVES, NP - 1);                 // piece q of the staged slice = (hi | lo, M-block, kx)
      const int half = q / (MB * KS), r = q - half * (MB * KS);
      conv_lds_dma_b128(wsrc + ((half * MBP + mbsel) * KS + r) * 1024, dst + q * 1024);
    }
    ring_w = ring_w + 1 == RING ? 0 : ring_w + 1;
  };
  // The fp32 patch of the NEXT input block is prefetched into registers while the current block is multiplied.
  constexpr int TOTAL = NPIX * SEG, NPRE = (TOTAL + NT - 1) / NT;
  static_assert(AHEAD * PPW + NPRE <= 63, "vmcnt range");
  f32x4 pre[NPRE];
  auto load_patch = [&](int blk) {
    blk = min(blk, a.nblk_in - 1);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = min(tid + i * NT, TOTAL - 1);
      const int pix = e / SEG, seg = e - pix * SEG;
      const int py = pix / PW, px = pix - py * PW;
      const int gy = min(max(y0 + py - R, 0), a.H - 1), gx = min(max(x0 + px - R, 0), a.W - 1);
      pre[i] = *reinterpret_cast<const f32x4*>(in + ((((size_t)b * a.H + gy) * a.W + gx) * a.nblk_in + blk) * PIXG + seg * 16);
    }
  };
  auto store_patch = [&]() {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = tid + i * NT;
      if (e < TOTAL) {
        const int pix = e / SEG, seg = e - pix * SEG;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - R, gx = x0 + px - R;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // zero padding outside the image
        half4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xs = ok ? pre[i][k] * act_scale : 0.f;
          hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);   // saturate instead of producing inf: the lo half
          lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);  // then carries up to another 65 000
        }
        // segment = 4 consecutive slots of half hh: slots s0..s0+3 -> K-chunk s0 / 8, byte (s0 % 8) * 2 of the pixel's 16
        const int hh = seg / (SB / 4), s0 = (seg - hh * (SB / 4)) * 4;
        const int o = ((hh * KCB + (s0 >> 3)) * NPIX + pix) * 16 + (s0 & 7) * 2;
        *reinterpret_cast<half4*>(plane_hi + o) = hi;
        *reinterpret_cast<half4*>(plane_lo + o) = lo;
      }
    }
  };
  CONV_T_DECL;
  load_patch(0);
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) issue_slice(i);
  int sl = 0, ring_r = 0;
  for (int blk = 0; blk < a.nblk_in; ++blk) {
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kc = 0; kc < KCB; ++kc, ++sl) {
        // Issue order of a block starting at sub-slice j: ... slice j+AHEAD, patch(next block), slice j+AHEAD+1, ...  Awaiting
        // sub-slice j+i may leave outstanding what was issued after it: AHEAD-1 sub-slices, and the patch loads while i <= AHEAD.
        // The block's first iteration needs the patch issued one block ago: only the sub-slices issued after THAT may be out.
        const int i = ky * KCB + kc;
        CONV_T(3);
        if (i == 0) {
          DFN_VMCNT((AHEAD - 1 < SPB - 1 ? AHEAD - 1 : SPB - 1) * PPW);
          asm volatile("" ::: "memory");
          CONV_T(0);
          __syncthreads();                   // everyone is done with the previous patch and sub-slice sl-1
          CONV_T(1);
          store_patch();
          CONV_T(2);
        } else if (i <= AHEAD) {
          DFN_VMCNT((AHEAD - 1) * PPW + NPRE);
        } else {
          DFN_VMCNT((AHEAD - 1) * PPW);
        }
        asm volatile("" ::: "memory");
        CONV_T(0);
        __syncthreads();                     // sub-slice sl (and the patch) visible; sub-slice sl-1 fully consumed
        CONV_T(1);
        issue_slice(sl + AHEAD);             // into the buffer sub-slice sl-1 has left
        if (i == 0) load_patch(blk + 1);
        const char* wb = wst + ring_r * WSL;
        ring_r = ring_r + 1 == RING ? 0 : ring_r + 1;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          half8 bh[2], bl[2];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int o = ((h * KCB + kc) * NPIX + ((2 * wave + nb) * RF + pr + ky) * PW + pc + kx) * 16;
            bh[nb] = *reinterpret_cast<const half8*>(plane_hi + o);
            bl[nb] = *reinterpret_cast<const half8*>(plane_lo + o);
          }
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int fo = ((mb * KS + kx) * 64 + lane) * 16;
            const half8 ah = *reinterpret_cast<const half8*>(wb + fo);
            const half8 al = *reinterpret_cast<const half8*>(wb + WHALF + fo);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nb], acc[mb][nb], 0, 0, 0);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nb], acc[mb][nb], 0, 0, 0);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nb], acc[mb][nb], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // the tail's repeated DMA pieces land before the buffers are reused by the epilogue
  CONV_T(3);
  x3_epilogue<MB, TW>(a, acc, out_scale, smem, wave, lane, b, cg, y0, x0);
  CONV_T(4);
  CONV_T_FLUSH;
}

// ------------------------------------------------------------------------------------------ split-f16 convolution, split input
// The same product as conv_x3_kernel on an input already stored as hi | lo f16 (the row-planar split storage above): the patch needs no conversion and
// is staged by LDS-DMA like the weights — no patch registers, no split VALU, no LDS writes by the waves (the ablations of DESIGN.md
// section 7 put those at a quarter of conv_x3_kernel's time, paid once per output-channel group and halo overlap).  The
// contraction runs over HALF-blocks (16 input channels = one K-chunk of both lane halves): the planes of one half-block are 21 KB
// for an 8 x 32 tile, so two of them ping-pong — half-block hb + 1 lands while hb is multiplied — in the LDS the fp32-input kernel
// needs for one block.  Planes: [hi | lo][lane half h][pixel][8 slots] f16; a DMA instruction fills 64 consecutive 16-byte slots,
// each lane fetching its pixel's piece — runs of up to a patch row of consecutive pieces in the row-planar storage — or 16 bytes of
// zeros outside the image.  One barrier per sub-slice,
// none extra per half-block.  Sub-slice order (block, K-chunk, ky) over the weights packed as (block, ky, K-chunk).
// RING = number of weight sub-slice buffers.  2: sub-slice s + 1 is issued while s is multiplied and awaited at the next barrier —
// its last pieces have the tail of one multiply phase to land, and the waves' cycle counters (make conv_timing) put 30 % of the 3x3
// layers' wave time into that wait.  3 (the 3x3 layers; exactly 80 KB with the two plane buffers, still two workgroups per CU):
// sub-slice s + 2 is issued during s, the planes of half-block hb + 1 during (hb, ky = 0 .. KS - 2), and a wave arrives at barrier s
// with everything its PREVIOUS iteration issued still allowed in flight (in-order vmcnt) — the wait halves (16 %).
template <int KS, int SB, int WAVES = 4, int TW = 32>
constexpr int x3s_patch_bytes() {
  return ((2 * 2 * (2 * WAVES * (32 / TW) + KS - 1) * (TW + KS - 1) * 16) + 1023) & ~1023;   // one half-block, whole 1 KB DMA pieces
}

template <int KS, int SB, int MB, int WAVES = 4, int TW = 32, bool SPREAD = true, int RING = 2>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void conv_x3s_kernel(ConvArgs a) {
  constexpr int KCB = SB / 8;
  constexpr int RF = 32 / TW;
  constexpr int TH = 2 * WAVES * RF, R = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1, NPIX = PH * PW;
  constexpr int PIXG = 2 * SB * 4;        // bytes of a pixel's block in HBM: [hi: 2 x SB f16][lo: 2 x SB f16]
  constexpr int PBUF = x3s_patch_bytes<KS, SB, WAVES, TW>();
  constexpr int PPIECES = PBUF / 1024, PPP = (PPIECES + WAVES - 1) / WAVES;   // DMA pieces of a half-block's planes, per wave
  constexpr int WSL = x3_wslice_bytes<KS, SB, MB>();
  constexpr int WHALF = WSL / 2;
  constexpr int NP = WSL / 1024, PPW = (NP + WAVES - 1) / WAVES;
  static_assert(PPP + PPW <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;                     // two half-block buffers
  char* wst = smem + 2 * PBUF;            // RING sub-slices
  static_assert(RING == 2 || (RING == 3 && KS >= 3), "RING = 3 spreads a half-block's planes over its first KS - 1 iterations");

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int pr = p / TW, pc = p % TW;
  const int tiles_x = (a.W + TW - 1) / TW;
  int tile = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
  const int S = a.ksplit > 1 ? a.ksplit : 1;   // K slices of a tile group (ConvArgs::ksplit)
  int ks = 0, group = 0;
  if (a.xcd_groups > 0) {                 // XCD-aware 1-D grid, see conv_x3_kernel
    // logical tile groups in contiguous runs per XCD (block id % 8, observed placement: speed only); the S slices of a group are
    // consecutive blocks of ONE XCD, so the reducing slice reads same-XCD slabs
    const int n8 = (a.xcd_groups + 7) / 8;
    const int q = blockIdx.x >> 3;
    ks = q % S;
    const int L = (blockIdx.x & 7) * n8 + q / S;
    if (L >= a.xcd_groups || q / S >= n8) return;
    group = L;
    const int tiles = tiles_x * ((a.H + TH - 1) / TH), per_cg = tiles * a.B;
    cg = L / per_cg;
    const int rem = L - cg * per_cg;
    b = rem / tiles;
    tile = rem - b * tiles;
  }
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;

  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const f32x4* bq = reinterpret_cast<const f32x4*>(a.bias + ((cg * MB + mb) * 2 + h) * 16);  // pre-scaled bias
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ks == 0 ? bq[q] : f32x4{0.f, 0.f, 0.f, 0.f};   // the bias enters once: with slice 0
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[mb][0][4 * q + i] = v[i]; acc[mb][1][4 * q + i] = v[i]; }
    }
  }
  const int NHB = a.nblk_in * KCB;        // half-blocks
  const int n_slices = NHB * KS;
  const int hb0 = ks * NHB / S, hb1 = (ks + 1) * NHB / S;   // this workgroup's half-blocks (all of them without a split)
  // Where each of this lane's DMA slots comes from (fixed for the tile): slot = piece * 64 + lane = (plane, h, pixel).
  const char* psrc[PPP];
  unsigned inside = 0;                    // bit i: slot i is a pixel of the image (its address advances with the half-block)
#pragma unroll
  for (int i = 0; i < PPP; ++i) {
    const int slot = min(wave + i * WAVES, PPIECES - 1) * 64 + lane;
    const int q = slot / NPIX, pix = slot - q * NPIX;          // q = plane * 2 + h; q >= 4: padding of the last piece
    const int py = pix / PW, px = pix - py * PW;
    const int gy = y0 + py - R, gx = x0 + px - R;
    const bool ok = q < 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    psrc[i] = ok ? static_cast<const char*>(a.in) + split_piece(a.H, a.W, a.nblk_in, KCB, b, gy, gx, 0, 0, q >> 1, q & 1)
                 : static_cast<const char*>(a.zeros);
    inside |= unsigned(ok) << i;
  }
  // DMA pieces are issued ONE AT A TIME between the MFMA groups of an iteration (an LDS-DMA instruction costs the issuing wave
  // 60+ cycles, 100-185 in a burst next to the fragment reads: MI355X_MICROARCH.md).  Iteration (hb, ky) issues, in this order, the
  // PPW pieces of sub-slice s + 1 and then up to PPI pieces of half-block hb + 1's planes.
  constexpr int PPI = RING == 3 ? (PPP + KS - 2) / (KS - 1) : (PPP + KS - 1) / KS;   // patch pieces per iteration (RING = 3: none in the last)
  constexpr int GAPS = KS * MB;                              // MFMA groups of an iteration
  static_assert(PPW + PPI <= 40, "vmcnt_upto range");
  auto patch_piece = [&](int hbn, int i) {                   // piece i (compile-time after unrolling) of half-block hbn's planes
    const size_t off = (size_t)min(hbn, NHB - 1) * 4 * a.W * 16;   // the half-block's four sub-planes of the row
    conv_lds_dma_b128(psrc[i] + (((inside >> i) & 1) ? off : (size_t)0), patch + (hbn & 1) * PBUF + min(wave + i * WAVES, PPIECES - 1) * 1024);
  };
  auto slice_piece = [&](int sl, int buf, int i) {
    sl = min(sl, n_slices - 1);
    const int hb = sl / KS, ky = sl - hb * KS;
    const int packed = ((hb / KCB) * KS + ky) * KCB + (hb % KCB);
    const int q = min(wave + i * WAVES, NP - 1) * 1024;
    conv_lds_dma_b128(a.w + ((size_t)cg * n_slices + packed) * WSL + lane * 16 + q, wst + buf * WSL + q);
  };
  // a gradient tensor was stored at its own measured power-of-two scale (dyn_scale[0]) instead of kConvActScale: undo that one
  const float out_scale = a.dyn_scale ? a.out_scale * kConvActScale * a.dyn_scale[1] : a.out_scale;
  CONV_T_DECL;
#pragma unroll
  for (int i = 0; i < PPP; ++i) patch_piece(hb0, i);
#pragma unroll
  for (int i = 0; i < PPW; ++i) slice_piece(hb0 * KS, 0, i);
  if (RING == 3) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) slice_piece(hb0 * KS + 1, 1, i);
  }
  int sl = hb0 * KS, rb = 0;                // rb: ring slot of sub-slice sl
  for (int hb = hb0; hb < hb1; ++hb) {
    const char* pb = patch + (hb & 1) * PBUF;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky, ++sl) {
      // the previous iteration issued [sub-slice s][its share of the next planes]: sub-slice s is home once only that share is out;
      // the half-block's first iteration needs the planes too
      // RING = 3: sub-slice s was issued TWO iterations ago (and the planes' last share at ky = KS - 2): everything the previous
      // iteration issued may still be in flight — a DMA has a whole multiply phase to land instead of the tail of one
      const int prev_patch = ky == 0 ? 0 : (min(ky * PPI, PPP) - min((ky - 1) * PPI, PPP));
      CONV_T(3);
      vmcnt_upto(RING == 3 ? PPW + prev_patch : prev_patch);
      asm volatile("" ::: "memory");
      CONV_T(0);
      __syncthreads();                     // sub-slice s and the planes visible; sub-slice s-1 (and at ky == 0 the other planes) consumed
      CONV_T(1);
      const char* wb = wst + rb * WSL;
      const int rbn = RING == 3 ? (rb == 0 ? 2 : rb - 1) : (rb ^ 1);   // slot of sub-slice s + RING - 1: read last in iteration s - 1
      rb = rb + 1 == RING ? 0 : rb + 1;
      const int p_lo = min(ky * PPI, PPP), p_n = min((ky + 1) * PPI, PPP) - p_lo;   // this iteration's share of the next planes
      auto issue = [&](int n) {            // n-th DMA piece of this iteration
        if (n < PPW) slice_piece(sl + RING - 1, rbn, n);
        else if (n - PPW < p_n) {
#pragma unroll
          for (int i = 0; i < PPP; ++i) if (i == p_lo + n - PPW) patch_piece(hb + 1, i);
        }
      };
      if (!SPREAD) {                       // all of the iteration's pieces up front (the 8-wave 5x5 tile measured better this way)
#pragma unroll
        for (int n = 0; n < PPW + PPI; ++n) issue(n);
      }
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        half8 bh[2], bl[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int o = (h * NPIX + ((2 * wave + nb) * RF + pr + ky) * PW + pc + kx) * 16;
          bh[nb] = *reinterpret_cast<const half8*>(pb + o);
          bl[nb] = *reinterpret_cast<const half8*>(pb + 2 * NPIX * 16 + o);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int fo = ((mb * KS + kx) * 64 + lane) * 16;
          const half8 ah = *reinterpret_cast<const half8*>(wb + fo);
          const half8 al = *reinterpret_cast<const half8*>(wb + WHALF + fo);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nb], acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nb], acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nb], acc[mb][nb], 0, 0, 0);
          }
          const int g = kx * MB + mb;      // after this MFMA group: its DMA piece (the last group takes what is left)
          if (SPREAD) issue(g);
          if (SPREAD && g == GAPS - 1) {
#pragma unroll
            for (int n = GAPS; n < PPW + PPI; ++n) issue(n);
          }
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // the tail's repeated DMA pieces land before the buffers are reused by the epilogue
  CONV_T(3);
  if (S > 1) {
    // K split: the slab leaves by WRITE-THROUGH (sc1) 16-byte stores — no release fence, whose L2 write-back under 512 workgroups cost
    // more than the split saved — every wave drains its stores, one lane draws the ticket; the last arriver acquires once (its L1) and
    // sums the slabs in slice order, two slabs' loads in flight at a time, then goes on to the epilogue (cdna_hip_programming.md,
    // Guideline 16, R1 in its counter form: correct wherever the slices ran).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int SLAB = MB * 2 * 16 * WAVES * 64;             // floats per slab: [mb][nb][q][thread][4]
    constexpr int NQ = MB * 8;                                  // 16-byte pieces per thread and slab
    {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.ks_slab + ((size_t)group * S + ks) * SLAB, 0, SLAB * 4, 0x00020000);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v{acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (((mb * 2 + nb) * 4 + q) * (WAVES * 64) + tid) * 16, 0, 16);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave
    __syncthreads();
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem);   // the one LDS array (the staging buffers are dead)
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.ks_count + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != unsigned(S - 1)) return;                    // not the last slice of this tile group: done
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(a.ks_count + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the stream's next launch
    }
    __syncthreads();
    const float* all = a.ks_slab + (size_t)group * S * SLAB + tid * 4;
    auto add_slab = [&](const f32x4 (&u)[NQ], bool first) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float x = u[(mb * 2 + nb) * 4 + q][i];
              acc[mb][nb][4 * q + i] = first ? x : acc[mb][nb][4 * q + i] + x;
            }
    };
    for (int s2 = 0; s2 < S; s2 += 2) {
      f32x4 u0[NQ], u1[NQ];
      const int s3 = s2 + 1 < S ? s2 + 1 : s2;
#pragma unroll
      for (int j = 0; j < NQ; ++j) u0[j] = *reinterpret_cast<const f32x4*>(all + (size_t)s2 * SLAB + j * (WAVES * 64 * 4));
#pragma unroll
      for (int j = 0; j < NQ; ++j) u1[j] = *reinterpret_cast<const f32x4*>(all + (size_t)s3 * SLAB + j * (WAVES * 64 * 4));
      add_slab(u0, s2 == 0);
      if (s2 + 1 < S) add_slab(u1, false);
    }
  }
  x3_epilogue<MB, TW>(a, acc, out_scale, smem, wave, lane, b, cg, y0, x0);
  CONV_T(4);
  CONV_T_FLUSH;
}

#ifdef DFN_TIMING
}  // namespace dfn
// timing build only: read (and optionally clear) the cycle counters of conv_x3_kernel
extern "C" int dfn_debug_conv_cycles(unsigned long long* out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(dfn::g_conv_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dfn::g_conv_cycles), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace dfn {
#endif

template <int KS, int SB, int MB, int RING, int WAVES = 4, int TW = 32>
static hipError_t launch_conv_x3_t(const ConvArgs& a, hipStream_t stream) {
  if (a.cout_blocks % MB) return hipErrorInvalidValue;
  constexpr int TH = 2 * WAVES * (32 / TW);
  constexpr int lds = 2 * x3_plane_bytes<KS, SB, WAVES, TW>() + RING * x3_wslice_bytes<KS, SB, MB>();
  static_assert(lds <= (WAVES == 4 ? (MB == 1 ? 160 * 1024 / 3 : 80 * 1024) : 160 * 1024), "x3 conv tile does not fit in LDS (4-wave tiles: two / three workgroups per CU)");
  static_assert(lds >= WAVES * 32 * (MB * 128 + 16), "epilogue turn buffers");
  auto kern = conv_x3_kernel<KS, SB, MB, RING, WAVES, TW>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
  // weights of the layer in the packed split-f16 form: hi + lo f16 per element
  const size_t wbytes = size_t(a.cout_blocks) * 32 * a.nblk_in * 32 * KS * KS * 4;
  if (wbytes > (3u << 20) && a.cout_blocks / MB >= 8) {
    ConvArgs ax = a;
    ax.xcd_groups = tiles * (a.cout_blocks / MB) * a.B;
    hipLaunchKernelGGL(kern, dim3((ax.xcd_groups + 7) / 8 * 8), dim3(WAVES * 64), lds, stream, ax);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3(tiles, a.cout_blocks / MB, a.B), dim3(WAVES * 64), lds, stream, a);
  return hipGetLastError();
}

// Workspace of the K-split launches of ONE stream (launches of a stream are ordered, so they share it; the side stream's concurrent
// convolutions get their own): slabs + one arrival counter per tile group, zeroed when allocated and reset by each group's reducer.
// Grown on demand (first launches only), released with the process.
struct KSplitWs { float* slab = nullptr; unsigned* count = nullptr; size_t slab_bytes = 0; int counters = 0; };
static KSplitWs* ksplit_workspace(hipStream_t stream, size_t slab_bytes, int counters) {
  static std::mutex mu;
  static std::map<hipStream_t, KSplitWs> table;
  std::lock_guard<std::mutex> lock(mu);
  KSplitWs& w = table[stream];
  if (w.slab_bytes < slab_bytes) {
    if (w.slab) { (void)hipStreamSynchronize(stream); (void)hipFree(w.slab); w.slab = nullptr; w.slab_bytes = 0; }
    const size_t want = slab_bytes < (size_t(40) << 20) ? (size_t(40) << 20) : slab_bytes;
    if (hipMalloc(&w.slab, want) != hipSuccess) return nullptr;
    w.slab_bytes = want;
  }
  if (w.counters < counters) {
    if (w.count) { (void)hipStreamSynchronize(stream); (void)hipFree(w.count); w.count = nullptr; w.counters = 0; }
    const int want = counters < 4096 ? 4096 : counters;
    if (hipMalloc(&w.count, size_t(want) * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(w.count, 0, size_t(want) * sizeof(unsigned)) != hipSuccess) return nullptr;
    w.counters = want;
  }
  return &w;
}

template <int KS, int SB, int MB, int WAVES = 4, int TW = 32, bool SPREAD = true, int RING = 2>
static hipError_t launch_conv_x3s_t(const ConvArgs& a, hipStream_t stream) {
  if (a.cout_blocks % MB || !a.zeros) return hipErrorInvalidValue;
  constexpr int TH = 2 * WAVES * (32 / TW);
  constexpr int lds = 2 * x3s_patch_bytes<KS, SB, WAVES, TW>() + RING * x3_wslice_bytes<KS, SB, MB>();
  static_assert(lds <= (WAVES == 4 ? 80 : 160) * 1024, "x3s conv tile does not fit in LDS (4-wave tiles: two workgroups per CU)");
  static_assert(lds >= WAVES * 32 * (MB * 128 + 16), "epilogue turn buffers");
  auto kern = conv_x3s_kernel<KS, SB, MB, WAVES, TW, SPREAD, RING>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
  const size_t wbytes = size_t(a.cout_blocks) * 32 * a.nblk_in * 32 * KS * KS * 4;
  const int groups = tiles * (a.cout_blocks / MB) * a.B;
  // K split (3x3, 4-wave tiles): for grids of at most a third of the chip's 512 workgroup slots (conv5_x on 15 x 20 maps: 64 groups),
  // up to four slices of at least four half-blocks.  Measured on 4 x 240x320 (LABBOOK R6.2): conv5_x 85 -> 42 us with four slices
  // (eight: 45; two: 59; the slices' arithmetic alone 34), conv4_x (192 groups = 0.75 per CU) unchanged with two and slower with four,
  // so grids above 170 groups are left alone.  DFN_CONV_KSPLIT=0 turns it off, =n forces n (A/B runs).
  int S = 1;
  if (KS == 3 && SB == 16 && WAVES == 4) {
    static const int forced = [] { const char* e = getenv("DFN_CONV_KSPLIT"); return e ? atoi(e) : -1; }();
    const int nhb = a.nblk_in * (SB / 8);
    if (forced >= 0) S = forced > 1 ? forced : 1;
    else if (groups * 3 <= 512) S = 512 / groups > 4 ? 4 : 512 / groups;
    if (S > nhb / 4) S = nhb / 4;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
  }
  if (S > 1) {
    ConvArgs ax = a;
    KSplitWs* ws = ksplit_workspace(stream, size_t(groups) * S * (MB * 2 * 16 * WAVES * 64) * sizeof(float), groups);
    if (!ws) return hipErrorOutOfMemory;
    ax.ksplit = S; ax.ks_slab = ws->slab; ax.ks_count = ws->count;
    ax.xcd_groups = groups;
    hipLaunchKernelGGL(kern, dim3((groups + 7) / 8 * 8 * S), dim3(WAVES * 64), lds, stream, ax);
    return hipGetLastError();
  }
  if (wbytes > (3u << 20) && a.cout_blocks / MB >= 8) {      // XCD-aware grid, as launch_conv_x3_t
    ConvArgs ax = a;
    ax.xcd_groups = groups;
    hipLaunchKernelGGL(kern, dim3((ax.xcd_groups + 7) / 8 * 8), dim3(WAVES * 64), lds, stream, ax);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3(tiles, a.cout_blocks / MB, a.B), dim3(WAVES * 64), lds, stream, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ tensor scale for split-f16 gradients
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
  __shared__ float red[256];
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void absmax_finalize_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, part[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mx = red[0];
    int k = 0;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
      k = 11 - e;              // mx * 2^k in [2^10, 2^11)
      k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    out[0] = ldexpf(1.f, k);
    out[1] = ldexpf(1.f, -k);
  }
}
hipError_t launch_absmax_finalize(const float* part, int n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(absmax_finalize_kernel, dim3(1), dim3(256), 0, s, part, n, out);
  return hipGetLastError();
}
hipError_t launch_absmax_scale(const float* x, size_t n, float* part, float* out, hipStream_t s) {
  const int blocks = int((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, x, n, part);
  hipLaunchKernelGGL(absmax_finalize_kernel, dim3(1), dim3(256), 0, s, part, blocks > 0 ? blocks : 1, out);
  return hipGetLastError();
}

#ifdef DFN_TIMING
}  // namespace dfn
// timing build only: resident workgroups per CU of the main split-f16 conv variants (runtime's answer for their LDS / registers)
extern "C" int dfn_debug_conv_occupancy(int* out, int n) {
  using namespace dfn;
  int k = 0;
  auto q = [&](auto kern, int threads, int lds) {
    int nb = -1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess) nb = -2;
    if (k < n) out[k++] = nb;
  };
  q(conv_x3_kernel<3, 16, 2, 2, 4, 32>, 256, 2 * x3_plane_bytes<3, 16, 4, 32>() + 2 * x3_wslice_bytes<3, 16, 2>());
  q(conv_x3s_kernel<3, 16, 2, 4, 32, true, 3>, 256, 2 * x3s_patch_bytes<3, 16, 4, 32>() + 3 * x3_wslice_bytes<3, 16, 2>());
  q(conv_x3_kernel<3, 16, 1, 2, 4, 16>, 256, 2 * x3_plane_bytes<3, 16, 4, 16>() + 2 * x3_wslice_bytes<3, 16, 1>());
  q(conv_x3s_kernel<1, 16, 2, 4, 32>, 256, 2 * x3s_patch_bytes<1, 16, 4, 32>() + 2 * x3_wslice_bytes<1, 16, 2>());
  q(conv_kernel<PrecF16, 3, 16, 4>, 256, conv_patch_bytes<PrecF16, 3, 16>() + conv_wstage_bytes<PrecF16, 3, 16, 4>());
  return k;
}
namespace dfn {
#endif

int conv_mb(int prec, int cout_blocks) { return (prec == 0 && cout_blocks % 4 == 0) ? 4 : 2; }
int prep_sb(int prec) { return prec == 1 ? 4 : 8; }

hipError_t launch_conv(int prec, int ks, int sb, const ConvArgs& a, hipStream_t stream) {
  const bool wide = a.cout_blocks % 4 == 0;
  if (prec == 0) {
    if (sb == 8 && ks == 3) return launch_conv_t<PrecF16, 3, 8, 2>(a, stream);
    if (sb != 16) return hipErrorInvalidValue;
    if (ks == 1) return wide ? launch_conv_t<PrecF16, 1, 16, 4>(a, stream) : launch_conv_t<PrecF16, 1, 16, 2>(a, stream);
    if (ks == 3) return wide ? launch_conv_t<PrecF16, 3, 16, 4>(a, stream) : launch_conv_t<PrecF16, 3, 16, 2>(a, stream);
    if (ks == 5) return wide ? launch_conv_t<PrecF16, 5, 16, 4>(a, stream) : launch_conv_t<PrecF16, 5, 16, 2>(a, stream);
  } else if (prec == 2) {
    // 8 x 32 or 16 x 16 pixel tiles for the 3x3 layers: whichever covers the image with fewer padding pixels (ties: the 128-byte rows
    // of 8 x 32).  Variants that were built, measured and dropped (DESIGN.md sections 6 / 7): a ring of three weight sub-slices
    // (equal), one 8-wave 16 x 32 workgroup per CU for 3x3 (4-11 % slower), 32-channel workgroups three to a CU (10 % slower),
    // A fragments straight from L2 into registers (12 % slower).
    auto padded = [&](int th, int tw) { return double((a.H + th - 1) / th * th) * ((a.W + tw - 1) / tw * tw); };
    const bool sq = padded(16, 16) < 0.97 * padded(8, 32);
    // (32 x 8 tiles — fragments of 4 rows x 8 columns, 94 % instead of 78 % of the 30 x 40 maps filled — measured equal for conv5_x,
    //  104-110 vs 103-107 us: those layers are bound by per-iteration latency, and the 10-pixel patch rows conflict in LDS: LABBOOK.md)
    if (a.in_split) {   // split storage in: patch staged by LDS-DMA (inference forward)
      if (sb == 8 && ks == 3) return launch_conv_x3s_t<3, 8, 2>(a, stream);
      if (sb != 16) return hipErrorInvalidValue;
      if (ks == 1) return launch_conv_x3s_t<1, 16, 2>(a, stream);
      // 3x3: ring of three weight sub-slices (a DMA piece has a whole multiply phase to land: -5 ... -12 % per layer, conv5_x most);
      // the 8-wave 5x5 tile measured 3 % slower with it and keeps two
      if (ks == 3) return sq ? launch_conv_x3s_t<3, 16, 2, 4, 16, true, 3>(a, stream) : launch_conv_x3s_t<3, 16, 2, 4, 32, true, 3>(a, stream);
      if (ks == 5) return launch_conv_x3s_t<5, 16, 2, 8, 32, false>(a, stream);   // 8-wave tile: DMA pieces in a burst measured better
    } else {            // fp32 in: patch converted while it is staged (training / gradient paths)
      if (sb == 8 && ks == 3) return launch_conv_x3_t<3, 8, 2, 2>(a, stream);
      if (sb != 16) return hipErrorInvalidValue;
      if (ks == 1) return launch_conv_x3_t<1, 16, 2, 2>(a, stream);
      if (ks == 3) return sq ? launch_conv_x3_t<3, 16, 2, 2, 4, 16>(a, stream) : launch_conv_x3_t<3, 16, 2, 2>(a, stream);
      // 5x5: 16 x 32-pixel tiles, eight waves — the 4-wave tile's planes + slices allow one workgroup = ONE wave per SIMD; eight waves
      // share each weight slice and give every SIMD two waves (measured 1.63 -> 1.22 ms on 4 x 480x640)
      if (ks == 5) return launch_conv_x3_t<5, 16, 2, 2, 8>(a, stream);
    }
  } else {
    if (sb == 4 && ks == 3) return launch_conv_t<PrecF32, 3, 4, 2>(a, stream);
    if (sb != 16) return hipErrorInvalidValue;
    if (ks == 1) return launch_conv_t<PrecF32, 1, 16, 2>(a, stream);
    if (ks == 3) return launch_conv_t<PrecF32, 3, 16, 2>(a, stream);
    if (ks == 5) return launch_conv_t<PrecF32, 5, 16, 2>(a, stream);
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------ input prep
template <class T, int SB>
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ x, int B, int H, int W, T* __restrict__ out) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};  // dfnet.py:79-80
  const size_t n = (size_t)B * H * W, plane = (size_t)H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / plane, r = i - b * plane;
    T* o = out + i * (2 * SB);
#pragma unroll
    for (int s = 0; s < 2 * SB; ++s) o[s] = (T)0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (T)((x[(b * 3 + c) * plane + r] - mean[c]) / stdv[c]);  // half 0, slots 0..2
  }
}
// the conv1_1 input in split-f16 storage (one block of 2 x 8 slots, one K-chunk): [B,H][hi | lo][h][W][8 f16], RGB in half 0 slots 0..2
__global__ __launch_bounds__(256) void prep_split_kernel(const float* __restrict__ x, int B, int H, int W, char* __restrict__ out) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};  // dfnet.py:79-80
  const size_t n = (size_t)B * H * W, plane = (size_t)H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / plane, r = i - b * plane;
    const int y = int(r / W), xx = int(r - (size_t)y * W);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (x[(b * 3 + c) * plane + r] - mean[c]) / stdv[c];
    half4_t hi, lo;
    split4(v, kConvActScale, hi, lo);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {           // q = plane * 2 + h
      char* d = out + split_piece(H, W, 1, 1, int(b), y, xx, 0, 0, q >> 1, q & 1);
      *reinterpret_cast<f32x4*>(d) = z;
      if (q == 0) *reinterpret_cast<half4_t*>(d) = hi;
      if (q == 2) *reinterpret_cast<half4_t*>(d) = lo;
    }
  }
}
// prec 3 = the split-f16 storage of prec 2 (inference forward)
hipError_t launch_dfnet_prep(int prec, const float* x, int B, int H, int W, void* out, hipStream_t stream) {
  const size_t n = (size_t)B * H * W;
  const int grid = int((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (prec == 3) hipLaunchKernelGGL(prep_split_kernel, dim3(grid), dim3(256), 0, stream, x, B, H, W, static_cast<char*>(out));
  else if (prec == 0) hipLaunchKernelGGL((prep_kernel<_Float16, 8>), dim3(grid), dim3(256), 0, stream, x, B, H, W, static_cast<_Float16*>(out));
  else if (prec == 2) hipLaunchKernelGGL((prep_kernel<float, 8>), dim3(grid), dim3(256), 0, stream, x, B, H, W, static_cast<float*>(out));
  else hipLaunchKernelGGL((prep_kernel<float, 4>), dim3(grid), dim3(256), 0, stream, x, B, H, W, static_cast<float*>(out));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ 2x2 max pool
template <class T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, int B, int H, int W, int C, T* __restrict__ out) {
  const int Ho = H / 2, Wo = W / 2;
  constexpr int V = 16 / int(sizeof(T));  // elements per 16-byte vector
  const int cv = C / V;
  const size_t n = (size_t)B * Ho * Wo * cv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = int(i % cv);
    size_t r = i / cv;
    const int x = int(r % Wo); r /= Wo;
    const int y = int(r % Ho);
    const size_t b = r / Ho;
    const T* s = in + (((b * H + 2 * y) * W + 2 * x) * (size_t)C) + c * V;
    alignas(16) T v[4][V];
    *reinterpret_cast<f32x4*>(v[0]) = *reinterpret_cast<const f32x4*>(s);
    *reinterpret_cast<f32x4*>(v[1]) = *reinterpret_cast<const f32x4*>(s + C);
    *reinterpret_cast<f32x4*>(v[2]) = *reinterpret_cast<const f32x4*>(s + (size_t)W * C);
    *reinterpret_cast<f32x4*>(v[3]) = *reinterpret_cast<const f32x4*>(s + (size_t)W * C + C);
    alignas(16) T o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float m = fmaxf(fmaxf((float)v[0][k], (float)v[1][k]), fmaxf((float)v[2][k], (float)v[3][k]));
      o[k] = (T)m;
    }
    *reinterpret_cast<f32x4*>(out + i * V) = *reinterpret_cast<const f32x4*>(o);
  }
}
hipError_t launch_maxpool(int prec, const void* in, int B, int H, int W, int nblk, void* out, hipStream_t stream) {
  const int C = nblk * 32;
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C / (prec == 0 ? 8 : 4);
  if (!n) return hipSuccess;
  const int grid = int((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  if (prec == 0) hipLaunchKernelGGL(maxpool_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, static_cast<const _Float16*>(in), B, H, W, C, static_cast<_Float16*>(out));
  else hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, stream, static_cast<const float*>(in), B, H, W, C, static_cast<float*>(out));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ bilinear upsample
// nn.UpsamplingBilinear2d(size) = align_corners=True: src = dst * (in-1)/(out-1).  One thread per
// output pixel and 32-channel block; lanes run along X so every fp32 NCHW plane row is written coalesced.
template <class T>
__global__ __launch_bounds__(256) void upsample_kernel(const T* __restrict__ in, int B, int h, int w, int UH, int UW,
                                                       float* __restrict__ out, size_t out_bstride, const float* __restrict__ affine) {
  const float sy = UH > 1 ? float(h - 1) / float(UH - 1) : 0.f;
  const float sx = UW > 1 ? float(w - 1) / float(UW - 1) : 0.f;
  const size_t n = (size_t)B * 4 * UH * UW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int X = int(i % UW);
    size_t r = i / UW;
    const int Y = int(r % UH); r /= UH;
    const int blk = int(r % 4);
    const size_t b = r / 4;
    const float fy = sy * float(Y), fx = sx * float(X);
    const int yA = int(fy), xA = int(fx);
    const int yB = yA + (yA < h - 1 ? 1 : 0), xB = xA + (xA < w - 1 ? 1 : 0);
    const float ly = fy - float(yA), lx = fx - float(xA);
    const float wy0 = 1.f - ly, wx0 = 1.f - lx;
    const T* base = in + (b * h * (size_t)w) * 128 + blk * 32;
    const T* p00 = base + ((size_t)yA * w + xA) * 128;
    // taps with zero weight alias the first one (same cache lines): the identity resize of level 0 reads each source once
    const T* p01 = lx == 0.f ? p00 : base + ((size_t)yA * w + xB) * 128;
    const T* p10 = ly == 0.f ? p00 : base + ((size_t)yB * w + xA) * 128;
    const T* p11 = ly == 0.f ? p01 : (lx == 0.f ? p10 : base + ((size_t)yB * w + xB) * 128);
    float* o = out + b * out_bstride + (size_t)blk * 32 * UH * UW + (size_t)Y * UW + X;
    constexpr int V = 16 / int(sizeof(T));  // elements per 16-byte vector load
#pragma unroll
    for (int q = 0; q < 32 / V; ++q) {
      alignas(16) T a00[V], a01[V], a10[V], a11[V];
      *reinterpret_cast<f32x4*>(a00) = reinterpret_cast<const f32x4*>(p00)[q];
      *reinterpret_cast<f32x4*>(a01) = reinterpret_cast<const f32x4*>(p01)[q];
      *reinterpret_cast<f32x4*>(a10) = reinterpret_cast<const f32x4*>(p10)[q];
      *reinterpret_cast<f32x4*>(a11) = reinterpret_cast<const f32x4*>(p11)[q];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const int e = q * V + k, hh = e >> 4, s = e & 15;
        const int ch = 4 * hh + (s & 3) + 8 * (s >> 2);
        float v = wy0 * (wx0 * (float)a00[k] + lx * (float)a01[k]) + ly * (wx0 * (float)a10[k] + lx * (float)a11[k]);
        if (affine) v = fmaf(v, affine[kBnSc + blk * 32 + e], affine[kBnSh + blk * 32 + e]);
        o[(size_t)ch * UH * UW] = v;
      }
    }
  }
}
// Row-tiled variant for source rows of at most kUpRowMax pixels (pyramid levels 1 and 2): a workgroup owns one output row of one
// 32-channel block.  The two source rows are staged once in LDS, channel-major; every wave then writes whole channel rows as 16-byte
// stores aligned in memory (1 KB contiguous per instruction; the unaligned head / tail of a row as scalars) — the kernel above
// writes 256-byte runs interleaved with its gathers and reaches half of this one's write rate.  Same arithmetic, same result.
constexpr int kUpRowMax = 640;   // 3 x 8 x (w + 1) floats stay under the 64 KB a kernel gets without opting in
// CPB = channels (stored positions) per workgroup: 32, or 16 / 8 for the longer source rows so that the staged rows stay small
// enough for eight workgroups per CU (the stores need the waves: 3 resident workgroups wrote at 3 TB/s, 8 at 5).
// RY = output rows per workgroup: 4 when the resize enlarges at least ~3x vertically — four consecutive output rows then read at most
// kUpRows = 3 source rows, staged once (0.75 instead of 2 staged rows per output row) — else 1.
constexpr int kUpRows = 3;
template <class T, int CPB, int RY>
__global__ __launch_bounds__(256) void upsample_rows_kernel(const T* __restrict__ in, int h, int w, int UH, int UW, float* __restrict__ out,
                                                            size_t out_bstride, const float* __restrict__ affine) {
  extern __shared__ float rows[];          // [NR][CPB][w + 1], NR = 2 (RY = 1) or kUpRows
  const int ws = w + 1;
  const int Y0 = blockIdx.x * RY, blk = blockIdx.y / (32 / CPB), e0 = (blockIdx.y % (32 / CPB)) * CPB;   // stored positions e0 .. e0 + CPB
  const size_t b = blockIdx.z;
  const float sy = UH > 1 ? float(h - 1) / float(UH - 1) : 0.f;
  const float sx = UW > 1 ? float(w - 1) / float(UW - 1) : 0.f;
  const int y_first = int(sy * float(Y0));                   // first staged source row
  constexpr int NR = RY == 1 ? 2 : kUpRows;
  constexpr int V = 16 / int(sizeof(T));   // elements per 16-byte vector
  for (int e = threadIdx.x; e < NR * w * (CPB / V); e += 256) {
    const int r = e / (w * (CPB / V)), rem = e - r * (w * (CPB / V)), px = rem / (CPB / V), q = rem - px * (CPB / V);
    alignas(16) T v[V];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(in + ((b * h + min(y_first + r, h - 1)) * (size_t)w + px) * 128 + blk * 32 + e0 + q * V);
#pragma unroll
    for (int k = 0; k < V; ++k) rows[(r * CPB + q * V + k) * ws + px] = (float)v[k];
  }
  if (threadIdx.x < NR * CPB) rows[threadIdx.x * ws + w] = 0.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // The lane's columns X = head + 4 * lane + 256 * it + k and their horizontal taps are the same for every channel whose row
  // starts at the same 16-byte phase `head` (all of them when UH * UW is a multiple of four): computed once, kept in registers.
  constexpr int kIt = 3;                   // up to 768 output columns on the fast path
  int xa[kIt][4];
  float lxs[kIt][4];
  int cached_head = -1;
  auto taps = [&](int X, int& xA, float& lx) {
    const float fx = sx * float(X);
    xA = int(fx);
    lx = fx - float(xA);
  };
  for (int job = wave; job < RY * CPB; job += 4) {   // (output row, stored position e of the block -> channel)
    const int Y = Y0 + job / CPB, el = job % CPB;
    if (Y >= UH) break;
    const float fy = sy * float(Y);
    const int yA = int(fy), yB = yA + (yA < h - 1 ? 1 : 0);
    const float ly = fy - float(yA), wy0 = 1.f - ly;
    const int e = e0 + el, hh = e >> 4, s = e & 15, ch = blk * 32 + 4 * hh + (s & 3) + 8 * (s >> 2);
    const float* rA = rows + ((yA - y_first) * CPB + el) * ws;
    const float* rB = rows + ((yB - y_first) * CPB + el) * ws;
    const float sc = affine ? affine[kBnSc + blk * 32 + e] : 1.f, sh = affine ? affine[kBnSh + blk * 32 + e] : 0.f;
    float* o = out + b * out_bstride + ((size_t)ch * UH + Y) * UW;
    // column w of a staged row is zero: the right tap of the last source pixel has weight 0 and reads it
    auto blend = [&](int xA, float lx) {
      const float wx0 = 1.f - lx;
      float v = wy0 * (wx0 * rA[xA] + lx * rA[xA + 1]) + ly * (wx0 * rB[xA] + lx * rB[xA + 1]);
      if (affine) v = fmaf(v, sc, sh);
      return v;
    };
    auto value = [&](int X) { int xA; float lx; taps(X, xA, lx); return blend(xA, lx); };
    const int head = int((4 - ((reinterpret_cast<size_t>(o) >> 2) & 3)) & 3);   // floats before the first 16-byte boundary
    if (lane < head && lane < UW) o[lane] = value(lane);
    if (head != cached_head) {
      cached_head = head;
#pragma unroll
      for (int it = 0; it < kIt; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) taps(min(head + 4 * lane + 256 * it + k, UW - 1), xa[it][k], lxs[it][k]);
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int X = head + 4 * lane + 256 * it;
      if (X + 3 < UW) {
        *reinterpret_cast<f32x4*>(o + X) = f32x4{blend(xa[it][0], lxs[it][0]), blend(xa[it][1], lxs[it][1]), blend(xa[it][2], lxs[it][2]),
                                                 blend(xa[it][3], lxs[it][3])};
      } else {
        for (int k = X; k < UW; ++k) o[k] = value(k);
      }
    }
    for (int X = head + 4 * lane + 256 * kIt; X < UW; X += 256)   // wider rows than the cached taps cover
      for (int k = X; k < min(X + 4, UW); ++k) o[k] = value(k);
  }
}

hipError_t launch_upsample(int prec, const void* in, int B, int h, int w, int UH, int UW, float* out,
                           size_t out_bstride, hipStream_t stream, const float* affine) {
  if (w <= kUpRowMax && B > 0 && UH > 0 && UW > 0) {
    // four output rows per workgroup when they share three source rows; channels per workgroup by the source row length so that the
    // staged rows stay <= ~20 KB: eight workgroups per CU
    const float sy = UH > 1 ? float(h - 1) / float(UH - 1) : 0.f;
    const bool ry4 = 3.f * sy < 0.99f;     // int(sy (Y0 + 3)) - int(sy Y0) <= 1: rows y_first .. y_first + 2 cover the four
    const int nr = ry4 ? kUpRows : 2;
    const int cpb = nr * (w + 1) * 4 * 32 <= 20480 ? 32 : (nr * (w + 1) * 4 * 16 <= 20480 ? 16 : 8);
    const size_t lds = size_t(nr) * cpb * (w + 1) * 4;
    if (lds > 65536) return hipErrorInvalidValue;
    const dim3 grid(ry4 ? (UH + 3) / 4 : UH, 128 / cpb, B);
#define DFN_UP_LAUNCH(T, C, R) hipLaunchKernelGGL((upsample_rows_kernel<T, C, R>), grid, dim3(256), lds, stream, static_cast<const T*>(in), h, w, UH, UW, out, out_bstride, affine)
#define DFN_UP_CPB(T, R) do { if (cpb == 8) DFN_UP_LAUNCH(T, 8, R); else if (cpb == 16) DFN_UP_LAUNCH(T, 16, R); else DFN_UP_LAUNCH(T, 32, R); } while (0)
    if (prec == 0) { if (ry4) DFN_UP_CPB(_Float16, 4); else DFN_UP_CPB(_Float16, 1); }
    else { if (ry4) DFN_UP_CPB(float, 4); else DFN_UP_CPB(float, 1); }
#undef DFN_UP_CPB
#undef DFN_UP_LAUNCH
    return hipGetLastError();
  }
  const size_t n = (size_t)B * 4 * UH * UW;
  if (!n) return hipSuccess;
  const int grid = int((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  if (prec == 0) hipLaunchKernelGGL(upsample_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, static_cast<const _Float16*>(in), B, h, w, UH, UW, out, out_bstride, affine);
  else hipLaunchKernelGGL(upsample_kernel<float>, dim3(grid), dim3(256), 0, stream, static_cast<const float*>(in), B, h, w, UH, UW, out, out_bstride, affine);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ pose head
// relu5_3 -> pool5 (2x2 max) -> AdaptiveAvgPool2d(1) -> Linear(512, feat_dim).  Two launches: one workgroup per (image, pooled row) sums
// the row's window maxima per stored channel position (512 threads) into part[B][h/2][512]; one workgroup per image adds the rows in
// order, divides, and runs the fc (weights indexed through the storage permutation).  (One workgroup per image walking all windows
// serially, as this was until round 6, took 0.1 ms for ONE 480x640 frame — a tenth of the pose regressor's forward at batch 1.)
template <class T>
__global__ __launch_bounds__(512) void pose_pool_rows_kernel(const T* __restrict__ act, int h, int w, float* __restrict__ part) {
  const int c = threadIdx.x;  // stored position
  const size_t b = blockIdx.x;
  const int yo = blockIdx.y, ho = gridDim.y, wo = w / 2;
  const T* s0 = act + ((b * h + 2 * yo) * (size_t)w) * 512 + c;
  float sum = 0.f;
  for (int x = 0; x < wo; ++x) {
    const T* s = s0 + (size_t)x * 1024;
    sum += fmaxf(fmaxf((float)s[0], (float)s[512]), fmaxf((float)s[(size_t)w * 512], (float)s[(size_t)w * 512 + 512]));
  }
  part[(b * ho + yo) * 512 + c] = sum;
}
__global__ __launch_bounds__(512) void pose_fc_kernel(const float* __restrict__ part, int ho, int wo, const float* __restrict__ fc_w,
                                                      const float* __restrict__ fc_b, int feat_dim, float* __restrict__ pose) {
  __shared__ float pooled[512];
  const int c = threadIdx.x;
  const size_t b = blockIdx.x;
  float sum = 0.f;
  for (int yo = 0; yo < ho; ++yo) sum += part[(b * ho + yo) * 512 + c];
  const int blk = c >> 5, e = c & 31, hh = e >> 4, s = e & 15;
  const int ch = blk * 32 + 4 * hh + (s & 3) + 8 * (s >> 2);
  pooled[ch] = sum / float(ho * wo);
  __syncthreads();
  if (c < feat_dim) {
    float accv = fc_b[c];
    for (int k = 0; k < 512; ++k) accv = fmaf(fc_w[c * 512 + k], pooled[k], accv);
    pose[b * feat_dim + c] = accv;
  }
}
// part: scratch of B * (h / 2) * 512 floats
hipError_t launch_pose_head(int prec, const void* act, int B, int h, int w, const float* fc_w, const float* fc_b,
                            int feat_dim, float* part, float* pose, hipStream_t stream) {
  if (!B) return hipSuccess;
  if (!part || h < 2 || w < 2) return hipErrorInvalidValue;
  const dim3 grid(B, h / 2);
  if (prec == 0) hipLaunchKernelGGL(pose_pool_rows_kernel<_Float16>, grid, dim3(512), 0, stream, static_cast<const _Float16*>(act), h, w, part);
  else hipLaunchKernelGGL(pose_pool_rows_kernel<float>, grid, dim3(512), 0, stream, static_cast<const float*>(act), h, w, part);
  hipLaunchKernelGGL(pose_fc_kernel, dim3(B), dim3(512), 0, stream, part, h / 2, w / 2, fc_w, fc_b, feat_dim, pose);
  return hipGetLastError();
}

}  // namespace dfn
