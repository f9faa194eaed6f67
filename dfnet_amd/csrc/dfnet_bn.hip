// dfnet_bn.hip — BatchNorm2d of the adaptation layers when DFNet itself is being trained (gfx950).
// /root/reference/script/feature/dfnet.py:57-62 ends every adaptation layer with nn.BatchNorm2d(128); run_feature.py
// trains with the module in train() mode (batch statistics over the concatenated [target, render] batch) unless
// --freezeBN puts the BatchNorm layers back into eval() (utils.py:30-39).  Inference folds the layer into the 5x5
// convolution (dfnet_api.hip); training keeps the convolution unfolded and runs the normalisation here:
//   moments    per-channel sums over the Q = B*h*w pixels of a blocked [Q,4,32] fp32 tensor, fp64 accumulators,
//              fixed-order chunk partials (deterministic) -> finalize
//   finalize   batch or running statistics -> the affine map y = z * sc + sh applied inside the upsample kernel
//   backward   d L/d z from d L/d y (batch statistics: the mean / projection terms; frozen: a plain scale), and
//              d L/d gamma, d L/d beta
// All HBM-bound: one read of z (and g) per pass, 512-byte coalesced rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfnet_kernels.h"
#include "mfma_frag.h"

namespace dfn {

namespace {
__device__ __forceinline__ int chan_of_pos128(int e128) {   // stored position -> channel (dfnet_kernels.h layout)
  const int blk = e128 >> 5, e = e128 & 31, hh = e >> 4, s = e & 15;
  return blk * 32 + 4 * hh + (s & 3) + 8 * (s >> 2);
}
}  // namespace

// MODE 0: part[chunk][0][p] = sum z, [1][p] = sum z^2.
// MODE 1: part[chunk][0][p] = sum g, [1][p] = sum g * xhat, xhat = (z - mean[p]) * rstd[p]  (bw = BnWork of the level).
template <int MODE>
__global__ __launch_bounds__(256) void bn_moments_kernel(const float* __restrict__ a, const float* __restrict__ z,
                                                         const float* __restrict__ bw, long long Q, int n_chunks,
                                                         double* __restrict__ part) {
  // a thread owns 4 consecutive stored positions (one 16-byte load per pixel) of every 8th pixel of the chunk, four pixels' loads in
  // flight; the 8 pixel phases are combined through LDS in a fixed order (deterministic)
  __shared__ double red[7][2][128];
  const int p4 = (threadIdx.x & 31) * 4, ph = threadIdx.x >> 5;
  const long long per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = blockIdx.x * per, q1 = q0 + per < Q ? q0 + per : Q;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  f32x4 mean{0.f, 0.f, 0.f, 0.f}, rstd{0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) { mean = *reinterpret_cast<const f32x4*>(bw + kBnMean + p4); rstd = *reinterpret_cast<const f32x4*>(bw + kBnRstd + p4); }
  auto fold = [&](const f32x4& v, const f32x4& zz) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s0[k] += (double)v[k];
      if (MODE == 0) s1[k] += (double)v[k] * (double)v[k];
      else s1[k] += (double)v[k] * (double)((zz[k] - mean[k]) * rstd[k]);
    }
  };
  long long q = q0 + ph;
  for (; q + 24 < q1; q += 32) {
    f32x4 v[4], zz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = *reinterpret_cast<const f32x4*>(a + (q + 8 * u) * 128 + p4);
      if (MODE == 1) zz[u] = *reinterpret_cast<const f32x4*>(z + (q + 8 * u) * 128 + p4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) fold(v[u], zz[u]);
  }
  for (; q < q1; q += 8) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a + q * 128 + p4);
    f32x4 zz{0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) zz = *reinterpret_cast<const f32x4*>(z + q * 128 + p4);
    fold(v, zz);
  }
  if (ph) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[ph - 1][0][p4 + k] = s0[k]; red[ph - 1][1][p4 + k] = s1[k]; }
  }
  __syncthreads();
  if (!ph) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double t0 = s0[k], t1 = s1[k];
      for (int r = 0; r < 7; ++r) { t0 += red[r][0][p4 + k]; t1 += red[r][1][p4 + k]; }
      part[((size_t)blockIdx.x * 2 + 0) * 128 + p4 + k] = t0;
      part[((size_t)blockIdx.x * 2 + 1) * 128 + p4 + k] = t1;
    }
  }
}

// MODE 0 (batch statistics): mean / biased variance from the partials -> stats_out[0|1][channel] (optional), and the
//   work block: mean, rstd, sc = gamma * rstd, sh = beta - mean * sc   (position order).
// MODE 1 (backward): dgamma[ch] = sum g xhat, dbeta[ch] = sum g; work block: mg = sum g / Q, mgx = sum g xhat / Q.
// MODE 2 (frozen): work block from the running statistics, no partials.
// bn: device [4][128] = gamma, beta, running_mean, running_var (channel order).
template <int MODE>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ part, int n_chunks, long long Q,
                                                           const float* __restrict__ bn, float eps, float* __restrict__ bw,
                                                           float* __restrict__ out0, float* __restrict__ out1) {
  // a workgroup owns 32 positions (grid = 4): 32 chunk segments per position, four chunks' loads in flight per thread, the segments
  // combined in a fixed order (deterministic)
  __shared__ double seg[2][32][32];
  const int pl = threadIdx.x & 31, sg = threadIdx.x >> 5, p = blockIdx.x * 32 + pl, ch = chan_of_pos128(p);
  double s0 = 0.0, s1 = 0.0;
  if (MODE != 2) {
    const int per = (n_chunks + 31) / 32, c0 = sg * per, c1 = min(n_chunks, c0 + per);
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    int c = c0;
    for (; c + 4 <= c1; c += 4)
#pragma unroll
      for (int k = 0; k < 4; ++k) { a0[k] += part[((size_t)(c + k) * 2 + 0) * 128 + p]; a1[k] += part[((size_t)(c + k) * 2 + 1) * 128 + p]; }
    for (; c < c1; ++c) { a0[0] += part[((size_t)c * 2 + 0) * 128 + p]; a1[0] += part[((size_t)c * 2 + 1) * 128 + p]; }
    seg[0][sg][pl] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
    seg[1][sg][pl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    __syncthreads();
    if (sg != 0) return;
    for (int k = 0; k < 32; ++k) { s0 += seg[0][k][pl]; s1 += seg[1][k][pl]; }
  }
  if (sg != 0) return;
  if (MODE == 1) {
    out0[ch] = (float)s1;   // d gamma
    out1[ch] = (float)s0;   // d beta
    bw[kBnMg + p] = (float)(s0 / (double)Q);
    bw[kBnMgx + p] = (float)(s1 / (double)Q);
    return;
  }
  double mean, var;
  if (MODE == 0) {
    mean = s0 / (double)Q;
    var = s1 / (double)Q - mean * mean;
    if (var < 0.0) var = 0.0;
    if (out0) { out0[ch] = (float)mean; out1[ch] = (float)var; }
  } else {
    mean = (double)bn[2 * 128 + ch];
    var = (double)bn[3 * 128 + ch];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)bn[ch] * rstd;
  bw[kBnMean + p] = (float)mean;
  bw[kBnRstd + p] = (float)rstd;
  bw[kBnSc + p] = (float)sc;
  bw[kBnSh + p] = (float)((double)bn[128 + ch] - mean * sc);
}

static inline int chunks_for(long long Q) {
  long long c = (Q + 255) / 256;   // >= 256 pixels per workgroup
  return int(c < 1 ? 1 : (c > kBnMaxChunks ? kBnMaxChunks : c));
}

hipError_t launch_bn_batch_stats(const float* z, long long Q, const float* bn, float eps, double* part, float* bw, float* mean_out,
                                 float* var_out, hipStream_t s) {
  const int nc = chunks_for(Q);
  hipLaunchKernelGGL(bn_moments_kernel<0>, dim3(nc), dim3(256), 0, s, z, nullptr, nullptr, Q, nc, part);
  hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(4), dim3(1024), 0, s, part, nc, Q, bn, eps, bw, mean_out, var_out);
  return hipGetLastError();
}
hipError_t launch_bn_running_stats(const float* bn, float eps, float* bw, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel<2>, dim3(4), dim3(1024), 0, s, nullptr, 0, 1, bn, eps, bw, nullptr, nullptr);
  return hipGetLastError();
}

// d L/d z in place of g (blocked [Q,128]).  batch: sc * (g - mg - xhat * mgx); frozen: sc * g.
template <bool BATCH>
__global__ __launch_bounds__(256) void bn_backward_kernel(float* __restrict__ g, const float* __restrict__ z,
                                                          const float* __restrict__ bw, size_t n, unsigned* __restrict__ absmax_out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int p = int(i & 127);
    float v = g[i];
    if (BATCH) v = v - bw[kBnMg + p] - (z[i] - bw[kBnMean + p]) * bw[kBnRstd + p] * bw[kBnMgx + p];
    v *= bw[kBnSc + p];
    g[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (absmax_out) {   // max |d L / d z| for the split of this tensor (gate_split): at most one atomicMax per workgroup on the float's
                      // bit pattern, skipped when the word already covers it (65 536 same-address atomics cost 1.3 ms per step)
    __shared__ float red[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (m > __uint_as_float(__hip_atomic_load(absmax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(absmax_out, __float_as_uint(m));
    }
  }
}

hipError_t launch_bn_backward(int batch, float* g, const float* z, long long Q, double* part, float* bw, float* dgamma, float* dbeta,
                              hipStream_t s, unsigned* absmax_out) {
  const size_t n = (size_t)Q * 128;
  if (!n) return hipSuccess;
  const int grid = int((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (batch) {
    const int nc = chunks_for(Q);
    hipLaunchKernelGGL(bn_moments_kernel<1>, dim3(nc), dim3(256), 0, s, g, z, bw, Q, nc, part);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(4), dim3(1024), 0, s, part, nc, Q, nullptr, 0.f, bw, dgamma, dbeta);
    hipLaunchKernelGGL(bn_backward_kernel<true>, dim3(grid), dim3(256), 0, s, g, z, bw, n, absmax_out);
  } else {
    hipLaunchKernelGGL(bn_backward_kernel<false>, dim3(grid), dim3(256), 0, s, g, z, bw, n, absmax_out);
  }
  return hipGetLastError();
}

}  // namespace dfn
