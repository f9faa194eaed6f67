// nerfh_grad_stages.hip — the HBM-bound stages of the render GRADIENT path (gfx950): compositing backward,
// per-ray reduction of the per-sample point gradients, and get_rays backward.  Together with
// nerfh_bwd.hip they are what autograd executes for loss.backward() through render(c2w = pose)
// (/root/reference/script/feature/direct_feature_matching.py:340-376).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_device.h"
#include "nerfh_kernels.h"

namespace dfn {

static inline int grid_for(size_t n, int block, int cap = 256 * 8) {
  const size_t g = (n + block - 1) / block;
  return int(g < 1 ? 1 : (g > size_t(cap) ? size_t(cap) : g));
}

// ------------------------------------------------------------------------------------------ compositing backward
// rgb = sum_i T_i (a_s,i c_s,i + a_t,i c_t,i),  T_i = prod_{j<i} exp(-delta_j (sigma_s,j + sigma_t,j))
// (raw2outputs_NeRFW, models/rendering.py:161-230, test_time: the returned rgb_map is the joint composite).
// With g = d L / d rgb and e_i = T_i g.(a_s c_s + a_t c_t):
//   d c_s,i = g T_i a_s,i                          d c_t,i = g T_i a_t,i
//   d sigma_s,i = delta_i [(1 - a_s,i) T_i g.c_s,i - S_i]      S_i = sum_{k>i} e_k  (sigma_i attenuates every later sample)
//   d sigma_t,i = delta_i [(1 - a_t,i) T_i g.c_t,i - S_i]      d beta_i = 0 (beta does not reach rgb)
// One wavefront per ray, SPL consecutive samples per lane.
template <int SPL>
__global__ __launch_bounds__(256) void composite_fine_backward_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                      const float* __restrict__ grad_rgb, size_t n_rays,
                                                                      int Nf, float* __restrict__ graw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < n_rays; ray += size_t(gridDim.x) * 4) {
    const float* rr = raw + ray * size_t(Nf) * 9;
    const float* zr = z + ray * size_t(Nf);
    float* gr = graw + ray * size_t(Nf) * 9;
    const float g0 = grad_rgb[ray * 3], g1 = grad_rgb[ray * 3 + 1], g2 = grad_rgb[ray * 3 + 2];
    float v[SPL][9], zz[SPL + 1];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      zz[k] = i < Nf ? zr[i] : 0.f;
#pragma unroll
      for (int c = 0; c < 9; ++c) v[k][c] = i < Nf ? rr[size_t(i) * 9 + c] : 0.f;
    }
    zz[SPL] = __shfl_down(zz[0], 1, 64);
    float a_s[SPL], a_t[SPL], om[SPL], dl[SPL];
    float pj = 1.f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      const bool ok = i < Nf;
      dl[k] = i + 1 < Nf ? sub_rn(zz[k + 1], zz[k]) : 1e2f;
      a_s[k] = ok ? sub_rn(1.f, expf(-mul_rn(dl[k], v[k][3]))) : 0.f;
      a_t[k] = ok ? sub_rn(1.f, expf(-mul_rn(dl[k], v[k][7]))) : 0.f;
      om[k] = ok ? expf(-mul_rn(dl[k], add_rn(v[k][3], v[k][7]))) : 1.f;   // 1 - alpha_joint
      pj = mul_rn(pj, om[k]);
    }
    float Tj = __shfl_up(wave_incl_prod(pj, lane), 1, 64);
    if (lane == 0) Tj = 1.f;
    float T[SPL], gcs[SPL], gct[SPL], e[SPL], esum = 0.f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      T[k] = Tj;
      gcs[k] = g0 * v[k][0] + g1 * v[k][1] + g2 * v[k][2];
      gct[k] = g0 * v[k][4] + g1 * v[k][5] + g2 * v[k][6];
      e[k] = Tj * (a_s[k] * gcs[k] + a_t[k] * gct[k]);
      esum += e[k];
      Tj = mul_rn(Tj, om[k]);
    }
    // S_i = sum over the samples strictly after i, added up from the far end (never `total - prefix`: that carries eps x total)
    float after[SPL];
    {
      float later = __shfl_down(wave_incl_suffix_sum(esum, lane), 1, 64);   // the lanes after this one
      if (lane == 63) later = 0.f;
#pragma unroll
      for (int k = SPL - 1; k >= 0; --k) { after[k] = later; later += e[k]; }
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      const float S = after[k];
      if (i < Nf) {
        float* o = gr + size_t(i) * 9;
        const float ws = T[k] * a_s[k], wt = T[k] * a_t[k];
        o[0] = g0 * ws; o[1] = g1 * ws; o[2] = g2 * ws;
        o[3] = dl[k] * ((1.f - a_s[k]) * T[k] * gcs[k] - S);
        o[4] = g0 * wt; o[5] = g1 * wt; o[6] = g2 * wt;
        o[7] = dl[k] * ((1.f - a_t[k]) * T[k] * gct[k] - S);
        o[8] = 0.f;
      }
    }
  }
}

hipError_t launch_composite_fine_backward(const float* raw, const float* z, const float* grad_rgb, size_t n_rays, int Nf,
                                          float* graw, hipStream_t stream) {
  if (!n_rays) return hipSuccess;
  const int spl = (Nf + 63) / 64;
  const dim3 grid(grid_for((n_rays + 3) / 4, 1, 256 * 16)), block(256);
#define DFN_COMPB(S) \
  hipLaunchKernelGGL(composite_fine_backward_kernel<S>, grid, block, 0, stream, raw, z, grad_rgb, n_rays, Nf, graw)
  if (spl <= 1) DFN_COMPB(1);
  else if (spl == 2) DFN_COMPB(2);
  else if (spl == 3) DFN_COMPB(3);
  else if (spl == 4) DFN_COMPB(4);
  else if (spl <= 6) DFN_COMPB(6);
  else if (spl <= 8) DFN_COMPB(8);
  else return hipErrorInvalidValue;
#undef DFN_COMPB
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ per-ray reduction
// pts = o + d z (models/rendering.py:292,305)  =>  d o = sum_i g_i,  d d = sum_i z_i g_i.
// viewdirs = d / |d| (rendering.py:366-371)    =>  d d += (gv - v (v.gv)) / |d|,  gv = sum_i gv_i.
__global__ __launch_bounds__(256) void ray_grad_reduce_kernel(const float* __restrict__ gpts, const float* __restrict__ z,
                                                              const float* __restrict__ rays_d, size_t n_rays, int Nf,
                                                              int derive_viewdirs, float* __restrict__ grad_o,
                                                              float* __restrict__ grad_d, float* __restrict__ grad_v, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < n_rays; ray += size_t(gridDim.x) * 4) {
    float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f}, sv[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < Nf; i += 64) {
      const float* q = gpts + (ray * size_t(Nf) + i) * 6;
      const float zi = z[ray * size_t(Nf) + i];
#pragma unroll
      for (int c = 0; c < 3; ++c) { so[c] += q[c]; sd[c] += zi * q[c]; sv[c] += q[3 + c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { so[c] = wave_sum(so[c]); sd[c] = wave_sum(sd[c]); sv[c] = wave_sum(sv[c]); }
    if (lane == 0) {
      if (derive_viewdirs) {
        const float d0 = rays_d[ray * 3], d1 = rays_d[ray * 3 + 1], d2 = rays_d[ray * 3 + 2];
        const float inv = 1.f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float v0 = d0 * inv, v1 = d1 * inv, v2 = d2 * inv;
        const float dot = v0 * sv[0] + v1 * sv[1] + v2 * sv[2];
        sd[0] += (sv[0] - v0 * dot) * inv;
        sd[1] += (sv[1] - v1 * dot) * inv;
        sd[2] += (sv[2] - v2 * dot) * inv;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        grad_o[ray * 3 + c] = so[c] + (accumulate ? grad_o[ray * 3 + c] : 0.f);
        grad_d[ray * 3 + c] = sd[c] + (accumulate ? grad_d[ray * 3 + c] : 0.f);
        if (grad_v) grad_v[ray * 3 + c] = sv[c];
      }
    }
  }
}

hipError_t launch_ray_grad_reduce(const float* gpts, const float* z, const float* rays_d, size_t n_rays, int Nf,
                                  int derive_viewdirs, float* grad_o, float* grad_d, float* grad_viewdirs,
                                  hipStream_t stream, int accumulate) {
  if (!n_rays) return hipSuccess;
  hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3(grid_for((n_rays + 3) / 4, 1, 256 * 16)), dim3(256), 0, stream, gpts, z,
                     rays_d, n_rays, Nf, derive_viewdirs, grad_o, grad_d, grad_viewdirs, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ get_rays backward
// rays_d[a] = sum_b cam[b] c2w[a][b], rays_o[a] = c2w[a][3] (models/ray_utils.py:5-15)
//   =>  d c2w[a][b] = sum_rays d rays_d[a] cam[b],  d c2w[a][3] = sum_rays d rays_o[a].
// One workgroup, fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void raygen_backward_kernel(int H, int W, float focal, const float* __restrict__ grad_o,
                                                               const float* __restrict__ grad_d, float* __restrict__ grad_c2w) {
  __shared__ float red[16][12];
  const size_t n = size_t(H) * W;
  grad_o += size_t(blockIdx.x) * n * 3;   // one workgroup per frame of a batch (dfn_raygen_frames_backward)
  grad_d += size_t(blockIdx.x) * n * 3;
  grad_c2w += size_t(blockIdx.x) * 12;
  const float hw = float(W) * .5f, hh = float(H) * .5f;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (size_t px = threadIdx.x; px < n; px += blockDim.x) {
    const int j = int(px / W), i = int(px - size_t(j) * W);
    const float cam[3] = {(float(i) - hw) / focal, -(float(j) - hh) / focal, -1.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float gd = grad_d[px * 3 + a];
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a * 4 + b] += gd * cam[b];
      acc[a * 4 + 3] += grad_o[px * 3 + a];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    float s = 0.f;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) s += red[w][threadIdx.x];
    grad_c2w[threadIdx.x] = s;
  }
}

hipError_t launch_raygen_backward(int H, int W, float focal, const float* grad_o, const float* grad_d, float* grad_c2w,
                                  hipStream_t stream, int frames) {
  if (frames < 1) return hipSuccess;
  hipLaunchKernelGGL(raygen_backward_kernel, dim3(frames), dim3(1024), 0, stream, H, W, focal, grad_o, grad_d, grad_c2w);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ bicubic resize backward
// Adjoint of bicubic_kernel (nn.Upsample(mode='bicubic'), align_corners=False, A = -0.75, border-replicated taps):
// g_out [UH,UW,C] -> g_in [H,W,C].  Gather form (deterministic): an input pixel collects, per axis, the weights of
// every tap that lands on it (several taps of one output pixel can land on a border pixel).
DFN_DEV float cubic_b1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
DFN_DEV float cubic_b2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
DFN_DEV float bicubic_axis_weight(float scale, int O, int n_in, int target) {
  const float f = scale * (float(O) + .5f) - .5f;
  const int i0 = int(floorf(f));
  const float t = f - float(i0);
  const float wt[4] = {cubic_b2(t + 1.f), cubic_b1(t), cubic_b1(1.f - t), cubic_b2(2.f - t)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) acc += min(max(i0 - 1 + a, 0), n_in - 1) == target ? wt[a] : 0.f;
  return acc;
}
// Gather form (deterministic): one 8-lane group per input element, the lanes share the rows of the output window that reaches it.
// NCHW: g_out is planar, [C, UH, UW] per frame (launch_bicubic's NCHW output).
template <bool NCHW>
__global__ __launch_bounds__(256) void bicubic_backward_kernel(const float* __restrict__ gout, int H, int W, int C, int UH, int UW,
                                                               float* __restrict__ gin) {
  const float sy = float(H) / float(UH), sx = float(W) / float(UW);
  const size_t n = size_t(H) * W * C;
  gout += size_t(blockIdx.y) * UH * UW * C;   // blockIdx.y = frame of a batch
  gin += size_t(blockIdx.y) * n;
  for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n * 8; t += size_t(gridDim.x) * blockDim.x) {
    const size_t i = t >> 3;
    const int sub = int(t & 7);
    const int c = int(i % C);
    const int x = int((i / C) % W), y = int(i / (size_t(C) * W));
    // output rows whose 4-tap window [floor(f)-1, floor(f)+2] can reach y (borders: everything beyond clamps onto them)
    const int Y0 = y == 0 ? 0 : max(0, int(floorf((float(y) - 2.5f) / sy)) - 1);
    const int Y1 = y == H - 1 ? UH - 1 : min(UH - 1, int(ceilf((float(y) + 2.5f) / sy)) + 1);
    const int X0 = x == 0 ? 0 : max(0, int(floorf((float(x) - 2.5f) / sx)) - 1);
    const int X1 = x == W - 1 ? UW - 1 : min(UW - 1, int(ceilf((float(x) + 2.5f) / sx)) + 1);
    float acc = 0.f;
    for (int Y = Y0 + sub; Y <= Y1; Y += 8) {
      const float wy = bicubic_axis_weight(sy, Y, H, y);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const float wx = bicubic_axis_weight(sx, X, W, x);
        if (wx != 0.f) row += wx * (NCHW ? gout[(size_t(c) * UH + Y) * UW + X] : gout[(size_t(Y) * UW + X) * C + c]);
      }
      acc += wy * row;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) gin[i] = acc;
  }
}
hipError_t launch_bicubic_backward(const float* gout, int H, int W, int C, int UH, int UW, float* gin, hipStream_t stream, int frames,
                                   bool nchw) {
  const size_t n = size_t(H) * W * C;
  if (!n || frames < 1) return hipSuccess;
  if (nchw) hipLaunchKernelGGL(bicubic_backward_kernel<true>, dim3(grid_for(n * 8, 256), frames), dim3(256), 0, stream, gout, H, W, C, UH, UW, gin);
  else hipLaunchKernelGGL(bicubic_backward_kernel<false>, dim3(grid_for(n * 8, 256), frames), dim3(256), 0, stream, gout, H, W, C, UH, UW, gin);
  return hipGetLastError();
}

}  // namespace dfn
