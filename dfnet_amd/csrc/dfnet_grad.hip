// dfnet_grad.hip — HBM-bound stages of DFNet's INPUT gradient (gfx950): what loss.backward() runs through
// feat_model(cat([data, rgb])) down to the rendered image in the DFNet_dm step
// (/root/reference/script/feature/direct_feature_matching.py:350-376).  The convolutions of the backward pass are
// the forward implicit-GEMM kernel (dfnet_conv.hip) on flipped / transposed weights (dfnet_api.hip: dgrad packing);
// here are the element-wise pieces between them, all on the blocked-permuted NHWC layout of dfnet_kernels.h:
// ReLU gating (+ the hypercolumn tap's gradient joining the trunk), max-pool routing, the adjoint of the
// align_corners bilinear upsample, and the adjoint of the input normalisation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfnet_kernels.h"
#include "mfma_frag.h"

namespace dfn {

static inline int grid_of(size_t n, int cap = 16384) {
  const size_t g = (n + 255) / 256;
  return int(g < 1 ? 1 : (g > size_t(cap) ? size_t(cap) : g));
}

// out = (act > 0 ? g : 0) + add   (g and/or add may be null: treated as zero).  ReLU'(0) = 0 as in torch.
template <class T>
__global__ __launch_bounds__(256) void relu_gate_kernel(const T* __restrict__ g, const T* __restrict__ act,
                                                        const T* __restrict__ add, size_t n, T* __restrict__ out) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    float v = (g && (float)act[i] > 0.f) ? (float)g[i] : 0.f;
    if (add) v += (float)add[i];
    out[i] = (T)v;
  }
}
hipError_t launch_relu_gate(int prec, const void* g, const void* act, const void* add, size_t n, void* out, hipStream_t s) {
  if (!n) return hipSuccess;
  if (prec == 0)
    hipLaunchKernelGGL(relu_gate_kernel<_Float16>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const _Float16*>(g),
                       static_cast<const _Float16*>(act), static_cast<const _Float16*>(add), n, static_cast<_Float16*>(out));
  else
    hipLaunchKernelGGL(relu_gate_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const float*>(g),
                       static_cast<const float*>(act), static_cast<const float*>(add), n, static_cast<float*>(out));
  return hipGetLastError();
}

// The same gate (fp32) that also leaves the tensor's |max| behind as the power-of-two operand scale [scale, 1/scale] of the split-f16
// gradient conv that reads `out` next (launch_absmax_scale's result without its extra pass over the tensor).
__global__ __launch_bounds__(256) void relu_gate_absmax_kernel(const float4* __restrict__ g, const float4* __restrict__ act,
                                                               const float4* __restrict__ add, size_t n4, float4* __restrict__ out,
                                                               float* __restrict__ part) {
  float m = 0.f;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g) {
      const float4 a = act[i], x = g[i];
      v = make_float4(a.x > 0.f ? x.x : 0.f, a.y > 0.f ? x.y : 0.f, a.z > 0.f ? x.z : 0.f, a.w > 0.f ? x.w : 0.f);
    }
    if (add) { const float4 t = add[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    out[i] = v;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
hipError_t launch_relu_gate_scale(const void* g, const void* act, const void* add, size_t n, void* out, float* part, float* scale,
                                  hipStream_t s) {
  if (!n) return hipSuccess;
  const bool vec = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(add) |
                                   reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (!vec) {
    if (hipError_t e = launch_relu_gate(1, g, act, add, n, out, s)) return e;
    return launch_absmax_scale(static_cast<const float*>(out), n, part, scale, s);
  }
  const size_t n4 = n / 4;
  const int blocks = int((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(relu_gate_absmax_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const float4*>(g), static_cast<const float4*>(act),
                     static_cast<const float4*>(add), n4, static_cast<float4*>(out), part);
  return launch_absmax_finalize(part, blocks, scale, s);
}

// 2x2/2 max-pool backward: the gradient of a pooled pixel goes to the FIRST maximum of its window in row-major
// order (torch's max_pool2d_with_indices).  act [B,H,W,C] (pre-pool), g [B,H/2,W/2,C] -> out [B,H,W,C]; rows /
// columns beyond 2*(H/2), 2*(W/2) get zero.
template <class T>
__global__ __launch_bounds__(256) void maxpool_backward_kernel(const T* __restrict__ act, const T* __restrict__ g, int B, int H,
                                                               int W, int C, T* __restrict__ out) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t n = (size_t)B * H * W * C;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const int c = int(i % C);
    size_t r = i / C;
    const int x = int(r % W); r /= W;
    const int y = int(r % H);
    const size_t b = r / H;
    const int yo = y >> 1, xo = x >> 1;
    float v = 0.f;
    if (yo < Ho && xo < Wo) {
      const T* s = act + ((b * H + 2 * yo) * (size_t)W + 2 * xo) * C + c;
      const float a0 = (float)s[0], a1 = (float)s[C], a2 = (float)s[(size_t)W * C], a3 = (float)s[(size_t)W * C + C];
      const float m = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
      const int first = a0 == m ? 0 : (a1 == m ? 1 : (a2 == m ? 2 : 3));
      if (first == ((y & 1) * 2 + (x & 1))) v = (float)g[((b * Ho + yo) * (size_t)Wo + xo) * C + c];
    }
    out[i] = (T)v;
  }
}
hipError_t launch_maxpool_backward(int prec, const void* act, const void* g, int B, int H, int W, int nblk, void* out,
                                   hipStream_t s) {
  const int C = nblk * 32;
  const size_t n = (size_t)B * H * W * C;
  if (!n) return hipSuccess;
  if (prec == 0)
    hipLaunchKernelGGL(maxpool_backward_kernel<_Float16>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const _Float16*>(act),
                       static_cast<const _Float16*>(g), B, H, W, C, static_cast<_Float16*>(out));
  else
    hipLaunchKernelGGL(maxpool_backward_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const float*>(act),
                       static_cast<const float*>(g), B, H, W, C, static_cast<float*>(out));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ gate into the split storage
// The backward chain of the split-f16 training path (dfnet_api.hip: backward_params_core): the gradient w.r.t. a conv's
// pre-activation is written ONCE, already split into the hi | lo f16 planes that both of its consumers multiply — the data-gradient
// conv (conv_x3s_kernel, LDS-DMA staged) and the weight-gradient stream (dfnet_wgrad_s.hip) — in the row-planar order of
// dfnet_conv.hip (split_piece):
//     out = split(2^k * ((act > 0 ? route(g) : 0) + add))
//   g     fp32 blocked [B, H, W, nblk, 32], or with POOL [B, H/2, W/2, nblk, 32]: the gradient w.r.t. the 2x2 max-pooled activation,
//         routed to the FIRST maximum of each window in row-major order (torch's max_pool2d_with_indices), zero elsewhere;
//   act   the conv's ReLU'd output at [B, H, W]: split storage (ACT_SPLIT) or fp32 blocked; null = no gate;
//   add   fp32 blocked [B, H, W, nblk, 32] or null (the hypercolumn tap's gradient joining the trunk);
//   2^k   from the bounds the producers left behind: absmax_g / absmax_add = bit patterns of max |g|, max |add| (upper bounds are
//         fine: 2^k * (|g|max + |add|max) lands in [2^10, 2^11), a factor 32 under the f16 range); [2^k, 2^-k] goes to scale_out.
// OUT_SPLIT = false writes fp32 blocked instead (conv1_1's weight gradient gathers RGB taps from an fp32 tensor).
// A thread owns 8 consecutive stored positions of one pixel: one 16-byte hi piece and one 16-byte lo piece.
namespace {
__device__ __forceinline__ float pow2_scale(const unsigned* absmax_g, const unsigned* absmax_add, float* inv) {
  float u = absmax_g ? __uint_as_float(*absmax_g) : 0.f;
  if (absmax_add) u += __uint_as_float(*absmax_add);
  int k = 0;
  if (u > 0.f && u < 3.0e38f) {
    int e;
    frexpf(u, &e);           // u = f * 2^e, f in [0.5, 1)
    k = 11 - e;              // u * 2^k in [2^10, 2^11)
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
  }
  *inv = ldexpf(1.f, -k);
  return ldexpf(1.f, k);
}
struct Piece8 { f32x4 a, b; };
// the 8 values (stored positions 16 h + 8 kc + 0..7 of block blk) of pixel (row, x) of a split tensor, hi + lo, unscaled by `inv`
__device__ __forceinline__ Piece8 load_split8(const char* t, size_t row, int W, int nblk, int blk, int kc, int h, int x, float inv) {
  const char* p = t + ((((row * nblk + blk) * 2 + kc) * 4 + h) * W + x) * 16;
  const half8 hi = *reinterpret_cast<const half8*>(p), lo = *reinterpret_cast<const half8*>(p + 2 * (size_t)W * 16);
  Piece8 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) { r.a[k] = ((float)hi[k] + (float)lo[k]) * inv; r.b[k] = ((float)hi[4 + k] + (float)lo[4 + k]) * inv; }
  return r;
}
__device__ __forceinline__ Piece8 load_f32x8(const float* t, size_t pix, int nblk, int blk, int kc, int h) {
  const float* p = t + (pix * nblk + blk) * 32 + 16 * h + 8 * kc;
  return Piece8{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
}
}  // namespace

// max |v| over the workgroup into the float-bits word `out` (bound of a gradient tensor for the split that follows, gate_split_kernel):
// one atomic per workgroup at most, skipped when the word already covers the value.  Every thread of the workgroup must call it.
__device__ __forceinline__ void publish_absmax(float m, unsigned* out) {
  __shared__ float s_amax[16];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < int((blockDim.x + 63) >> 6); ++i) m = fmaxf(m, s_amax[i]);
    if (m > 0.f && m > __uint_as_float(__hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(out, __float_as_uint(m));
  }
}

template <bool POOL, bool ACT_SPLIT, bool OUT_SPLIT>
__global__ __launch_bounds__(256) void gate_split_kernel(const float* __restrict__ g, const void* __restrict__ act, const float* __restrict__ add,
                                                         int B, int H, int W, int nblk, const unsigned* __restrict__ absmax_g,
                                                         const unsigned* __restrict__ absmax_add, void* __restrict__ out,
                                                         float* __restrict__ scale_out, int add_images) {
  float inv;
  const float sc = pow2_scale(absmax_g, absmax_add, &inv);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = sc; scale_out[1] = inv; }
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t n = (size_t)B * H * nblk * W * 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // i = ((row * nblk + blk) * W + x) * 4 + hk: a wave covers 16 pixels x the 4 (K-chunk, half) parts of one block — 2 KB of
    // contiguous fp32 reads, four 256-byte runs of split writes
    const int hk = int(i & 3), kc = hk >> 1, h = hk & 1;
    size_t r = i >> 2;
    const int x = int(r % W); r /= W;
    const int blk = int(r % nblk);
    const size_t row = r / nblk;
    const int y = int(row % H);
    const size_t b = row / H;
    Piece8 v{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (g) {
      if (!POOL) {
        v = load_f32x8(g, row * W + x, nblk, blk, kc, h);
        if (act) {
          const Piece8 a = ACT_SPLIT ? load_split8(static_cast<const char*>(act), row, W, nblk, blk, kc, h, x, 1.f)
                                     : load_f32x8(static_cast<const float*>(act), row * W + x, nblk, blk, kc, h);
#pragma unroll
          for (int k = 0; k < 4; ++k) { v.a[k] = a.a[k] > 0.f ? v.a[k] : 0.f; v.b[k] = a.b[k] > 0.f ? v.b[k] : 0.f; }
        }
      } else {
        const int yo = y >> 1, xo = x >> 1;
        if (yo < Ho && xo < Wo) {
          const Piece8 gp = load_f32x8(g, (b * Ho + yo) * (size_t)Wo + xo, nblk, blk, kc, h);
          Piece8 a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const size_t rq = b * H + 2 * yo + (q >> 1);
            const int xq = 2 * xo + (q & 1);
            a[q] = ACT_SPLIT ? load_split8(static_cast<const char*>(act), rq, W, nblk, blk, kc, h, xq, 1.f)
                             : load_f32x8(static_cast<const float*>(act), rq * W + xq, nblk, blk, kc, h);
          }
          const int me = (y & 1) * 2 + (x & 1);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float a0 = k < 4 ? a[0].a[k] : a[0].b[k - 4], a1 = k < 4 ? a[1].a[k] : a[1].b[k - 4];
            const float a2 = k < 4 ? a[2].a[k] : a[2].b[k - 4], a3 = k < 4 ? a[3].a[k] : a[3].b[k - 4];
            const float m = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            const int first = a0 == m ? 0 : (a1 == m ? 1 : (a2 == m ? 2 : 3));
            const float mine = me == 0 ? a0 : (me == 1 ? a1 : (me == 2 ? a2 : a3));
            const float gv = k < 4 ? gp.a[k] : gp.b[k - 4];
            const float o = (first == me && mine > 0.f) ? gv : 0.f;   // the window's first maximum, gated by ReLU'(act)
            if (k < 4) v.a[k] = o; else v.b[k - 4] = o;
          }
        }
      }
    }
    if (add && b < (size_t)add_images) {   // (the tap gradient covers the leading add_images frames: the siamese pair)
      const Piece8 t = load_f32x8(add, row * W + x, nblk, blk, kc, h);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v.a[k] += t.a[k]; v.b[k] += t.b[k]; }
    }
    if (OUT_SPLIT) {
      half8 hi, lo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xs = (k < 4 ? v.a[k] : v.b[k - 4]) * sc;
        hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);
        lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);
      }
      char* d = static_cast<char*>(out) + ((((row * nblk + blk) * 2 + kc) * 4 + h) * W + x) * 16;
      *reinterpret_cast<half8*>(d) = hi;
      *reinterpret_cast<half8*>(d + 2 * (size_t)W * 16) = lo;
    } else {
      float* d = static_cast<float*>(out) + ((row * W + x) * nblk + blk) * 32 + 16 * h + 8 * kc;
      *reinterpret_cast<f32x4*>(d) = v.a;
      *reinterpret_cast<f32x4*>(d + 4) = v.b;
    }
  }
}

hipError_t launch_gate_split(const float* g, int pooled_g, const void* act, int act_split, const float* add, int B, int H, int W, int nblk,
                             const unsigned* absmax_g, const unsigned* absmax_add, void* out, int out_split, float* scale_out,
                             hipStream_t s, int add_images) {
  const size_t n = (size_t)B * H * nblk * W * 4;
  if (!n) return hipSuccess;
  if (pooled_g && !act) return hipErrorInvalidValue;
  const dim3 grid(grid_of(n)), block(256);
#define DFN_GS(P, A, O) hipLaunchKernelGGL((gate_split_kernel<P, A, O>), grid, block, 0, s, g, act, add, B, H, W, nblk, absmax_g, absmax_add, out, scale_out, add_images < 0 ? B : add_images)
  if (pooled_g) {
    if (act_split) { if (out_split) DFN_GS(true, true, true); else DFN_GS(true, true, false); }
    else { if (out_split) DFN_GS(true, false, true); else DFN_GS(true, false, false); }
  } else {
    if (act_split) { if (out_split) DFN_GS(false, true, true); else DFN_GS(false, true, false); }
    else { if (out_split) DFN_GS(false, false, true); else DFN_GS(false, false, false); }
  }
#undef DFN_GS
  return hipGetLastError();
}

// Adjoint of upsample_kernel (bilinear, align_corners=True): g_up fp32 NCHW planes [b*bstride + c*UH*UW + Y*UW + X]
// -> blocked [B,h,w,4,32] T.  Gather form (deterministic): an input pixel collects from every output pixel whose
// two source rows / columns include it, with exactly the weights the forward kernel used.
template <class T>
__global__ __launch_bounds__(256) void upsample_backward_kernel(const float* __restrict__ gup, size_t bstride, int B, int h, int w,
                                                                int UH, int UW, T* __restrict__ out, unsigned* __restrict__ absmax_out) {
  float amax = 0.f;
  const float sy = UH > 1 ? float(h - 1) / float(UH - 1) : 0.f;
  const float sx = UW > 1 ? float(w - 1) / float(UW - 1) : 0.f;
  const size_t n = (size_t)B * h * w * 128;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    // x fastest so neighbouring lanes read neighbouring output columns
    const int x = int(i % w);
    size_t r = i / w;
    const int y = int(r % h); r /= h;
    const int e128 = int(r % 128);
    const size_t b = r / 128;
    const int blk = e128 >> 5, e = e128 & 31, hh = e >> 4, s = e & 15;
    const int ch = blk * 32 + 4 * hh + (s & 3) + 8 * (s >> 2);
    // candidate output rows: sy * Y in (y - 1, y + 1)
    int Y0 = 0, Y1 = UH - 1, X0 = 0, X1 = UW - 1;
    if (sy > 0.f) { Y0 = max(0, int(floorf(float(y - 1) / sy)) - 1); Y1 = min(UH - 1, int(ceilf(float(y + 1) / sy)) + 1); }
    if (sx > 0.f) { X0 = max(0, int(floorf(float(x - 1) / sx)) - 1); X1 = min(UW - 1, int(ceilf(float(x + 1) / sx)) + 1); }
    const float* plane = gup + b * bstride + (size_t)ch * UH * UW;
    if (h == UH && w == UW) {   // level 0: the resize is the identity, its adjoint a layout change
      const float v = plane[(size_t)y * UW + x];
      amax = fmaxf(amax, fabsf(v));
      out[((b * h + y) * (size_t)w + x) * 128 + e128] = (T)v;
      continue;
    }
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
      const float fy = sy * float(Y);
      const int yA = int(fy), yB = yA + (yA < h - 1 ? 1 : 0);
      const float ly = fy - float(yA);
      const float wy = (yA == y ? 1.f - ly : 0.f) + (yB == y ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const float fx = sx * float(X);
        const int xA = int(fx), xB = xA + (xA < w - 1 ? 1 : 0);
        const float lx = fx - float(xA);
        const float wx = (xA == x ? 1.f - lx : 0.f) + (xB == x ? lx : 0.f);
        if (wx != 0.f) row += wx * plane[(size_t)Y * UW + X];
      }
      acc += wy * row;
    }
    amax = fmaxf(amax, fabsf(acc));
    out[((b * h + y) * (size_t)w + x) * 128 + e128] = (T)acc;
  }
  if (absmax_out) publish_absmax(amax, absmax_out);
}
// The same adjoint, fp32, organised for bandwidth.  One workgroup per (image, input row y, 32-channel block); per channel
//   phase 1: tmp[X] = sum_Y wy(Y) g_up[Y][X]    over the output rows that touch input row y — coalesced along X
//   phase 2: row[x] = sum_X wx(X) tmp[X]         from LDS
// and the 32 channels of every pixel leave as one 128-byte store.  Each g_up element is read about twice in total
// (the gather above reads it (2 scale)^2 times through L2).  Weights are exactly the forward kernel's.
__global__ __launch_bounds__(256) void upsample_backward_rows_kernel(const float* __restrict__ gup, size_t bstride, int h, int w, int UH,
                                                                     int UW, float* __restrict__ out, unsigned* __restrict__ absmax_out) {
  extern __shared__ float lds[];
  float* tmp = lds;                    // [UW]
  float* wyv = tmp + UW;               // [UH] (only Y0..Y1 used)
  float* row = wyv + UH;               // [w][33]
  const int tid = threadIdx.x;
  const int y = blockIdx.x % h, blk = blockIdx.y;
  const size_t b = blockIdx.x / h;
  const float sy = UH > 1 ? float(h - 1) / float(UH - 1) : 0.f;
  const float sx = UW > 1 ? float(w - 1) / float(UW - 1) : 0.f;
  int Y0 = 0, Y1 = UH - 1;
  if (sy > 0.f) { Y0 = max(0, int(floorf(float(y - 1) / sy)) - 1); Y1 = min(UH - 1, int(ceilf(float(y + 1) / sy)) + 1); }
  for (int Y = Y0 + tid; Y <= Y1; Y += 256) {
    const float fy = sy * float(Y);
    const int yA = int(fy), yB = yA + (yA < h - 1 ? 1 : 0);
    const float ly = fy - float(yA);
    wyv[Y] = (yA == y ? 1.f - ly : 0.f) + (yB == y ? ly : 0.f);
  }
  __syncthreads();
  while (Y0 < Y1 && wyv[Y0] == 0.f) ++Y0;      // uniform: trim the slack rows
  while (Y1 > Y0 && wyv[Y1] == 0.f) --Y1;
  // gridDim.z splits the 32 channels of the block over workgroups (small levels: more workgroups than CUs)
  const int per_z = 32 / gridDim.z, e_lo = blockIdx.z * per_z;
  for (int e = e_lo; e < e_lo + per_z; ++e) {
    const int ch = blk * 32 + 4 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2);
    const float* plane = gup + b * bstride + (size_t)ch * UH * UW;
    for (int X = tid; X < UW; X += 256) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;   // four independent chains: four loads in flight per thread
      int Y = Y0;
      for (; Y + 3 <= Y1; Y += 4) {
        const float* pp = plane + (size_t)Y * UW + X;
        t0 = fmaf(wyv[Y], pp[0], t0);
        t1 = fmaf(wyv[Y + 1], pp[UW], t1);
        t2 = fmaf(wyv[Y + 2], pp[2 * (size_t)UW], t2);
        t3 = fmaf(wyv[Y + 3], pp[3 * (size_t)UW], t3);
      }
      for (; Y <= Y1; ++Y) t0 = fmaf(wyv[Y], plane[(size_t)Y * UW + X], t0);
      tmp[X] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    for (int x = tid; x < w; x += 256) {
      int X0 = 0, X1 = UW - 1;
      if (sx > 0.f) { X0 = max(0, int(floorf(float(x - 1) / sx)) - 1); X1 = min(UW - 1, int(ceilf(float(x + 1) / sx)) + 1); }
      float acc = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const float fx = sx * float(X);
        const int xA = int(fx), xB = xA + (xA < w - 1 ? 1 : 0);
        const float lx = fx - float(xA);
        const float wx = (xA == x ? 1.f - lx : 0.f) + (xB == x ? lx : 0.f);
        acc = fmaf(wx, tmp[X], acc);
      }
      row[x * 33 + e] = acc;
    }
    __syncthreads();
  }
  float* o = out + ((b * h + y) * (size_t)w) * 128 + blk * 32;
  float amax = 0.f;
  for (int i = tid; i < w * per_z; i += 256) {
    const int px = i / per_z, e = e_lo + i % per_z;
    const float v = row[px * 33 + e];
    amax = fmaxf(amax, fabsf(v));
    o[(size_t)px * 128 + e] = v;
  }
  if (absmax_out) publish_absmax(amax, absmax_out);
}

// Level 0 (no resize): NCHW planes -> blocked NHWC through an LDS tile of 128 channels x 64 pixels; 256-byte reads,
// 512-byte writes.
__global__ __launch_bounds__(256) void upsample_backward_identity_kernel(const float* __restrict__ gup, size_t bstride, size_t plane,
                                                                         float* __restrict__ out, unsigned* __restrict__ absmax_out) {
  __shared__ float tile[128][65];
  const size_t b = blockIdx.y, p0 = (size_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, t4 = threadIdx.x >> 6;
  const float* src = gup + b * bstride + p0;
  if (p0 + tx < plane)
    for (int ch = t4; ch < 128; ch += 4) tile[ch][tx] = src[(size_t)ch * plane + tx];
  __syncthreads();
  const int e128 = threadIdx.x & 127, t2 = threadIdx.x >> 7;
  const int blk = e128 >> 5, e = e128 & 31;
  const int ch = blk * 32 + 4 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2);
  float amax = 0.f;
  for (int px = t2; px < 64; px += 2)
    if (p0 + px < plane) {
      const float v = tile[ch][px];
      amax = fmaxf(amax, fabsf(v));
      out[(b * plane + p0 + px) * 128 + e128] = v;
    }
  if (absmax_out) publish_absmax(amax, absmax_out);
}

hipError_t launch_upsample_backward(int prec, const float* gup, size_t bstride, int B, int h, int w, int UH, int UW, void* out,
                                    hipStream_t s, unsigned* absmax_out) {
  const size_t n = (size_t)B * h * w * 128;
  if (!n) return hipSuccess;
  if (prec != 0 && h == UH && w == UW) {
    const size_t plane = (size_t)h * w;
    hipLaunchKernelGGL(upsample_backward_identity_kernel, dim3((unsigned)((plane + 63) / 64), B), dim3(256), 0, s, gup, bstride, plane,
                       static_cast<float*>(out), absmax_out);
    return hipGetLastError();
  }
  const size_t lds = ((size_t)UW + UH + (size_t)w * 33) * 4;
  if (prec != 0 && lds <= 64 * 1024) {
    const int zsplit = (size_t)B * h * 4 >= 2048 ? 1 : ((size_t)B * h * 4 >= 1024 ? 2 : 4);
    hipLaunchKernelGGL(upsample_backward_rows_kernel, dim3(B * h, 4, zsplit), dim3(256), lds, s, gup, bstride, h, w, UH, UW, static_cast<float*>(out), absmax_out);
    return hipGetLastError();
  }
  if (prec == 0)
    hipLaunchKernelGGL(upsample_backward_kernel<_Float16>, dim3(grid_of(n)), dim3(256), 0, s, gup, bstride, B, h, w, UH, UW,
                       static_cast<_Float16*>(out), absmax_out);
  else
    hipLaunchKernelGGL(upsample_backward_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, gup, bstride, B, h, w, UH, UW,
                       static_cast<float*>(out), absmax_out);
  return hipGetLastError();
}

// Adjoint of prep_kernel: g wrt the normalised input, blocked [B,H,W,nblk*32] (RGB = elements 0..2 of block 0),
// -> d L / d x [B,3,H,W] fp32 = g / std (dfnet.py:121-122).
template <class T>
__global__ __launch_bounds__(256) void unprep_kernel(const T* __restrict__ g, int B, int H, int W, int C, float* __restrict__ gx) {
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const size_t plane = (size_t)H * W, n = (size_t)B * 3 * plane;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const size_t r = i % plane;
    const int c = int((i / plane) % 3);
    const size_t b = i / (3 * plane);
    gx[i] = (float)g[(b * plane + r) * C + c] / stdv[c];
  }
}
hipError_t launch_unprep(int prec, const void* g, int B, int H, int W, int nblk, float* gx, hipStream_t s) {
  const size_t n = (size_t)B * 3 * H * W;
  if (!n) return hipSuccess;
  if (prec == 0)
    hipLaunchKernelGGL(unprep_kernel<_Float16>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const _Float16*>(g), B, H, W,
                       nblk * 32, gx);
  else
    hipLaunchKernelGGL(unprep_kernel<float>, dim3(grid_of(n)), dim3(256), 0, s, static_cast<const float*>(g), B, H, W, nblk * 32, gx);
  return hipGetLastError();
}

}  // namespace dfn
