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
  __shared__ double red[2][128];
  const int p = threadIdx.x & 127, lane = threadIdx.x >> 7;
  const long long per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = blockIdx.x * per, q1 = q0 + per < Q ? q0 + per : Q;
  double s0 = 0.0, s1 = 0.0;
  const float mean = MODE == 1 ? bw[kBnMean + p] : 0.f, rstd = MODE == 1 ? bw[kBnRstd + p] : 0.f;
  for (long long q = q0 + lane; q < q1; q += 2) {
    const float v = a[q * 128 + p];
    s0 += (double)v;
    if (MODE == 0) s1 += (double)v * (double)v;
    else s1 += (double)v * (double)((z[q * 128 + p] - mean) * rstd);
  }
  if (lane == 1) { red[0][p] = s0; red[1][p] = s1; }
  __syncthreads();
  if (lane == 0) {
    part[((size_t)blockIdx.x * 2 + 0) * 128 + p] = s0 + red[0][p];
    part[((size_t)blockIdx.x * 2 + 1) * 128 + p] = s1 + red[1][p];
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
  // 128 positions x 8 chunk segments, combined in a fixed order (deterministic)
  __shared__ double seg[2][8][128];
  const int p = threadIdx.x & 127, sg = threadIdx.x >> 7, ch = chan_of_pos128(p);
  double s0 = 0.0, s1 = 0.0;
  if (MODE != 2) {
    const int per = (n_chunks + 7) / 8, c1 = min(n_chunks, (sg + 1) * per);
    for (int c = sg * per; c < c1; ++c) {
      s0 += part[((size_t)c * 2 + 0) * 128 + p];
      s1 += part[((size_t)c * 2 + 1) * 128 + p];
    }
    seg[0][sg][p] = s0;
    seg[1][sg][p] = s1;
    __syncthreads();
    s0 = s1 = 0.0;
    for (int k = 0; k < 8; ++k) { s0 += seg[0][k][p]; s1 += seg[1][k][p]; }
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
  hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(1), dim3(1024), 0, s, part, nc, Q, bn, eps, bw, mean_out, var_out);
  return hipGetLastError();
}
hipError_t launch_bn_running_stats(const float* bn, float eps, float* bw, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel<2>, dim3(1), dim3(1024), 0, s, nullptr, 0, 1, bn, eps, bw, nullptr, nullptr);
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
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(1), dim3(1024), 0, s, part, nc, Q, nullptr, 0.f, bw, dgamma, dbeta);
    hipLaunchKernelGGL(bn_backward_kernel<true>, dim3(grid), dim3(256), 0, s, g, z, bw, n, absmax_out);
  } else {
    hipLaunchKernelGGL(bn_backward_kernel<false>, dim3(grid), dim3(256), 0, s, g, z, bw, n, absmax_out);
  }
  return hipGetLastError();
}

}  // namespace dfn
