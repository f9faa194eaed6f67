// dfnet_loss.hip — the triplet loss of DFNet's training on the two feature stacks (gfx950), forward and backward.
// /root/reference/script/feature/misc.py:355-435: anchor f1, positive f2 [lvl, B, C, H, W]; negatives are the stacks
// rolled by one image along B; nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction='mean') takes the pairwise L2
// distance over the LAST axis (W), so one "row" = one (lvl, b, c, h) line of W floats:
//     loss = mean_rows max(||x - y + eps|| - ||x - z + eps|| + margin, 0)
// with (x, y, z) chosen by in-triplet hard-negative mining from four full-tensor MSEs.  In torch this is ~20 kernels
// and ten passes over 630 MB stacks ("HBM-bound reductions worth fusing"); here:
//   case pass      the four MSE sums in one read of both stacks (each element and its rolled neighbour), fp64 partials
//   forward pass   one wave per row: both distances, hinge, row statistics kept for the backward (8 bytes per row)
//   backward pass  element-wise, gather form (deterministic): an element collects its terms as x / y of its own row
//                  and as the negative z of the next image's row
// Stacks are addressed as base + lvl * level_stride + b * slab, so the two halves of one siamese [lvl, 2B, ...] tensor
// work in place, for the inputs and for the gradients.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"
#include "dfnet_kernels.h"

namespace dfn {

namespace {
struct Stack { const float* p; size_t level_stride; };
struct GStack { float* p; size_t level_stride; };

// roles per mining case (misc.py:424-433): x = anchor of the triplet, y = positive, z = negative = roll(zsrc)
//   case 0: (f1, f2, roll f2)   1: (f2, f1, roll f1)   2: (f1, f2, roll f1)   3: (f2, f1, roll f2)
__device__ __forceinline__ bool x_is_f1(int c) { return c == 0 || c == 2; }
__device__ __forceinline__ bool z_is_f1(int c) { return c == 1 || c == 2; }
}  // namespace

// ONE pass over both stacks for the whole forward: one wave per feature row reads its four vectors — f1[b], f2[b] and the rolled
// f1[b-1], f2[b-1] — once and leaves
//   row_stat[row][6] = d(f1, f2), d(f2, f1)  (the positive pair seen from either anchor: |x - y + eps| is not symmetric)
//                      d(f1, roll f2), d(f2, roll f1), d(f1, roll f1), d(f2, roll f2)   (anchor - negative of the four cases)
//   part[block][8] (fp64) = the four mining sums  sum (f1 - roll f2)^2, (f2 - roll f1)^2, (f1 - roll f1)^2, (f2 - roll f2)^2
//                           and the four cases' hinge sums  sum max(d_xy - d_xz + margin, 0)
// so that the case (a global argmin, misc.py:424-433) is picked AFTERWARDS by the finalize kernel from the same pass: the separate
// mining pass (a second read of both stacks, 0.22 ms per DFNet training step) is gone.
__global__ __launch_bounds__(256) void triplet_rows_kernel(Stack f1, Stack f2, int L, int B, int rows, int W, float margin, float eps,
                                                           float* __restrict__ row_stat, double* __restrict__ part) {
  __shared__ double red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slab = (size_t)rows * W, n_rows = (size_t)L * B * rows;
  double mse[4] = {0.0, 0.0, 0.0, 0.0}, hin[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < n_rows; row += (size_t)gridDim.x * 4) {
    const size_t r = row % rows, lb = row / rows;
    const int b = int(lb % B), l = int(lb / B), bm = b == 0 ? B - 1 : b - 1;
    const float* pa = f1.p + l * f1.level_stride + b * slab + r * W;
    const float* pp = f2.p + l * f2.level_stride + b * slab + r * W;
    const float* pan = f1.p + l * f1.level_stride + bm * slab + r * W;
    const float* ppn = f2.p + l * f2.level_stride + bm * slab + r * W;
    float sxy0 = 0.f, sxy1 = 0.f, sz0 = 0.f, sz1 = 0.f, sz2 = 0.f, sz3 = 0.f;
    for (int w = lane; w < W; w += 64) {
      const float a = pa[w], p = pp[w], an = pan[w], pn = ppn[w];
      const float d0 = a - pn, d1 = p - an, d2 = a - an, d3 = p - pn;
      mse[0] += (double)(d0 * d0); mse[1] += (double)(d1 * d1); mse[2] += (double)(d2 * d2); mse[3] += (double)(d3 * d3);
      const float exy0 = a - p + eps, exy1 = p - a + eps;
      const float e0 = d0 + eps, e1 = d1 + eps, e2 = d2 + eps, e3 = d3 + eps;   // (x - z) + eps, as torch's pairwise_distance
      sxy0 = fmaf(exy0, exy0, sxy0); sxy1 = fmaf(exy1, exy1, sxy1);
      sz0 = fmaf(e0, e0, sz0); sz1 = fmaf(e1, e1, sz1); sz2 = fmaf(e2, e2, sz2); sz3 = fmaf(e3, e3, sz3);
    }
    for (int o = 32; o > 0; o >>= 1) {
      sxy0 += __shfl_down(sxy0, o, 64); sxy1 += __shfl_down(sxy1, o, 64);
      sz0 += __shfl_down(sz0, o, 64); sz1 += __shfl_down(sz1, o, 64); sz2 += __shfl_down(sz2, o, 64); sz3 += __shfl_down(sz3, o, 64);
    }
    if (lane == 0) {
      const float dxy[2] = {sqrtf(sxy0), sqrtf(sxy1)}, dxz[4] = {sqrtf(sz0), sqrtf(sz1), sqrtf(sz2), sqrtf(sz3)};
      float* st = row_stat + 6 * row;
      st[0] = dxy[0]; st[1] = dxy[1]; st[2] = dxz[0]; st[3] = dxz[1]; st[4] = dxz[2]; st[5] = dxz[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hinge = dxy[x_is_f1(c) ? 0 : 1] - dxz[c] + margin;
        if (hinge > 0.f) hin[c] += (double)hinge;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = mse[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) { red[wave][k] = v; red[wave][4 + k] = hin[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 8) part[(size_t)blockIdx.x * 8 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// mode 0: naive (case 0); 1: two-case mining (misc.py:371-397); 2: four-case (misc.py:399-435).  torch.argmin: first minimum.
// One wave: lane l adds the blocks l, l + 64, ... in order, then a fixed shuffle tree (deterministic).
__global__ void triplet_finalize_kernel(const double* __restrict__ part, int n_blocks, int mode, double count, double n_rows, float margin,
                                        int* __restrict__ case_out, float* __restrict__ mse_out, float* __restrict__ margin_out,
                                        float* __restrict__ loss) {
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int c = threadIdx.x; c < n_blocks; c += 64)
    for (int k = 0; k < 8; ++k) s[k] += part[(size_t)c * 8 + k];
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 8; ++k) s[k] += __shfl_down(s[k], o, 64);
  if (threadIdx.x != 0) return;
  float m[4];
  for (int k = 0; k < 4; ++k) { m[k] = (float)(s[k] / count); if (mse_out) mse_out[k] = m[k]; }
  int best = 0;
  if (mode == 1) best = m[0] < m[1] ? 0 : 1;
  else if (mode == 2)
    for (int k = 1; k < 4; ++k) if (m[k] < m[best]) best = k;
  *case_out = best;
  *margin_out = margin;
  *loss = (float)(s[4 + best] / n_rows);
}

// g1 / g2: gradients w.r.t. f1 / f2 (every element written).  scale = grad_loss / n_rows.
__global__ __launch_bounds__(256) void triplet_backward_kernel(Stack f1, Stack f2, int L, int B, int rows, int W, float eps,
                                                               const int* __restrict__ case_in, const float* __restrict__ row_stat,
                                                               const float* __restrict__ margin_in, const float* __restrict__ grad_loss,
                                                               double n_rows, GStack g1, GStack g2) {
  const int c = *case_in;
  const float margin = *margin_in;
  const int ixy = x_is_f1(c) ? 0 : 1, ixz = 2 + c;   // this case's columns of row_stat (triplet_rows_kernel)
  const float scale = (float)((double)*grad_loss / n_rows);
  const uint32_t slab = (uint32_t)rows * W;
  const bool xf1 = x_is_f1(c), zf1 = z_is_f1(c);
  const Stack X = xf1 ? f1 : f2, Y = xf1 ? f2 : f1, Z = zf1 ? f1 : f2;
  const GStack GX = xf1 ? g1 : g2, GY = xf1 ? g2 : g1;
  const int b = blockIdx.y % B, l = blockIdx.y / B, bm = b == 0 ? B - 1 : b - 1, bp = b == B - 1 ? 0 : b + 1;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < slab; e += gridDim.x * blockDim.x) {
    const uint32_t r = e / (uint32_t)W;
    const size_t row = ((size_t)l * B + b) * rows + r, rowp = ((size_t)l * B + bp) * rows + r;
    const float dxy = row_stat[6 * row + ixy], dxz_raw = row_stat[6 * row + ixz];
    const float dxz = dxy - dxz_raw + margin >= 0.f ? dxz_raw : -1.f;   // clamp_min passes the gradient at hinge == 0, as torch
    const float xv = X.p[l * X.level_stride + (size_t)b * slab + e], yv = Y.p[l * Y.level_stride + (size_t)b * slab + e];
    float gx = 0.f, gy = 0.f;
    if (dxz >= 0.f) {   // hinge active on this row
      const float zv = Z.p[l * Z.level_stride + (size_t)bm * slab + e];
      const float ty = dxy > 0.f ? (xv - yv + eps) / dxy : 0.f, tz = dxz > 0.f ? (xv - zv + eps) / dxz : 0.f;
      gx = scale * (ty - tz);
      gy = -scale * ty;
    }
    // this element as the negative of image b+1's row: z_{b+1} = zsrc[b]
    const float dxzp_raw = row_stat[6 * rowp + ixz];
    const float dxzp = row_stat[6 * rowp + ixy] - dxzp_raw + margin >= 0.f ? dxzp_raw : -1.f;
    float gz = 0.f;
    if (dxzp > 0.f) {
      const float xp = X.p[l * X.level_stride + (size_t)bp * slab + e];
      const float zself = zf1 == xf1 ? xv : yv;   // Z stack is the same tensor as X (cases 2, 3) or as Y (cases 0, 1)
      gz = scale * (xp - zself + eps) / dxzp;
    }
    if (zf1 == xf1) gx += gz; else gy += gz;
    GX.p[l * GX.level_stride + (size_t)b * slab + e] = gx;
    GY.p[l * GY.level_stride + (size_t)b * slab + e] = gy;
  }
}

static inline int slab_chunks(size_t slab, int images) {   // ~64K elements per workgroup, <= 4096 workgroups in all
  size_t c = (slab + 65535) / 65536;
  const size_t cap = images > 0 ? (size_t)(4096 / images > 0 ? 4096 / images : 1) : 1;
  if (c > cap) c = cap;
  return int(c < 1 ? 1 : c);
}

hipError_t launch_triplet_forward(const float* f1, size_t ls1, const float* f2, size_t ls2, int L, int B, int rows, int W, float margin,
                                  float eps, int mode, double* part, int* case_out, float* mse_out, float* margin_out, float* row_stat,
                                  float* loss, hipStream_t s) {
  const Stack a{f1, ls1}, p{f2, ls2};
  const size_t slab = (size_t)rows * W, n = (size_t)L * B * slab, n_rows = (size_t)L * B * rows;
  const int grid = int((n_rows + 3) / 4 < 2048 ? (n_rows + 3) / 4 : 2048);
  hipLaunchKernelGGL(triplet_rows_kernel, dim3(grid), dim3(256), 0, s, a, p, L, B, rows, W, margin, eps, row_stat, part);
  hipLaunchKernelGGL(triplet_finalize_kernel, dim3(1), dim3(64), 0, s, part, grid, mode, (double)n, (double)n_rows, margin, case_out, mse_out,
                     margin_out, loss);
  return hipGetLastError();
}

hipError_t launch_triplet_finalize(const double* part, int n_blocks, int mode, double count, double n_rows, float margin, int* case_out,
                                   float* mse_out, float* margin_out, float* loss, hipStream_t s) {
  hipLaunchKernelGGL(triplet_finalize_kernel, dim3(1), dim3(64), 0, s, part, n_blocks, mode, count, n_rows, margin, case_out, mse_out, margin_out,
                     loss);
  return hipGetLastError();
}

hipError_t launch_triplet_backward(const float* f1, size_t ls1, const float* f2, size_t ls2, int L, int B, int rows, int W, float eps,
                                   const int* case_in, const float* row_stat, const float* margin_in, const float* grad_loss, float* g1,
                                   size_t gs1, float* g2, size_t gs2, hipStream_t s) {
  const Stack a{f1, ls1}, p{f2, ls2};
  const GStack ga{g1, gs1}, gp{g2, gs2};
  const size_t slab = (size_t)rows * W, n_rows = (size_t)L * B * rows;
  if (slab >= (1ull << 32)) return hipErrorInvalidValue;
  const int grid = slab_chunks(slab, L * B) * 16;   // 4K elements per workgroup pass
  hipLaunchKernelGGL(triplet_backward_kernel, dim3(grid, L * B), dim3(256), 0, s, a, p, L, B, rows, W, eps, case_in, row_stat, margin_in,
                     grad_loss, (double)n_rows, ga, gp);
  return hipGetLastError();
}

}  // namespace dfn

// ------------------------------------------------------------------------------------------ C ABI
namespace dfn {
// ------------------------------------------------------------------------------------------ cosine feature loss (DFNet_dm)
// feature/direct_feature_matching.py:114-136 feature_loss with per_channel = False — nn.CosineSimilarity(dim=1, eps=1e-6) on
// [C', H*W]: ONE cosine per feature row over its H*W pixels, loss = 1 - mean — applied per image to the selected pyramid levels
// (:352-358) and averaged over the batch: altogether 1 - the mean over every (level, image, channel) row.  torch runs it as
// index_select + permute/reshape copies + two norms, two clones, two divisions, a product, a sum and their backward over
// 157 MB per stack and level; here: one read of both stacks (row statistics), one element-wise pass for the gradient.
namespace {
constexpr int kCosSplit = 4;       // blocks per row
constexpr int kCosMaxLevels = 8;
struct CosArgs {
  const float* fr; const float* ft; float* g;
  size_t ls_r, ls_t, ls_g, HW;
  int lv[kCosMaxLevels];
  int n_levels, B, C;
};
__device__ __forceinline__ size_t cos_row_offset(const CosArgs& a, int row, size_t ls) {   // row = (li * B + b) * C + c
  const int bc = row % (a.B * a.C), li = row / (a.B * a.C);
  return size_t(a.lv[li]) * ls + size_t(bc) * a.HW;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
}  // namespace

// part[(row * kCosSplit + seg) * 3 + {0,1,2}] = sum fr ft, sum fr^2, sum ft^2 over the segment
__global__ __launch_bounds__(256) void cosine_stats_kernel(CosArgs a, float* __restrict__ part) {
  const int row = blockIdx.x, seg = blockIdx.y;
  const float* __restrict__ x = a.fr + cos_row_offset(a, row, a.ls_r);
  const float* __restrict__ y = a.ft + cos_row_offset(a, row, a.ls_t);
  const size_t per = (a.HW / 4 + kCosSplit - 1) / kCosSplit * 4;
  const size_t lo = size_t(seg) * per, hi = lo + per < a.HW ? lo + per : a.HW;
  float d = 0.f, nx = 0.f, ny = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec) {
    const size_t hi4 = lo + ((hi > lo ? hi - lo : 0) & ~size_t(3));
    for (size_t i = lo + size_t(threadIdx.x) * 4; i < hi4; i += 1024) {
      const float4 u = *reinterpret_cast<const float4*>(x + i), v = *reinterpret_cast<const float4*>(y + i);
      d += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
      nx += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
      ny += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (size_t i = hi4 + threadIdx.x; i < hi; i += 256) { d += x[i] * y[i]; nx += x[i] * x[i]; ny += y[i] * y[i]; }
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) { d += x[i] * y[i]; nx += x[i] * x[i]; ny += y[i] * y[i]; }
  }
  __shared__ float red[4][3];
  d = wave_sum(d); nx = wave_sum(nx); ny = wave_sum(ny);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = d; red[wave][1] = nx; red[wave][2] = ny; }
  __syncthreads();
  if (threadIdx.x < 3)
    part[(size_t(row) * kCosSplit + seg) * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// coef[row] = (ca, cb): d cos / d fr = ca ft + cb fr (torch: x / clamp_min(||x||, eps) per operand, then the dot product);
// loss = 1 - mean cos, fixed-order fp64 reduction.
__global__ __launch_bounds__(256) void cosine_finish_kernel(const float* __restrict__ part, int rows, float eps, float* __restrict__ coef,
                                                            float* __restrict__ loss) {
  double acc = 0.;
  for (int r = threadIdx.x; r < rows; r += 256) {
    float d = 0.f, nx = 0.f, ny = 0.f;
    for (int s = 0; s < kCosSplit; ++s) {
      d += part[(size_t(r) * kCosSplit + s) * 3];
      nx += part[(size_t(r) * kCosSplit + s) * 3 + 1];
      ny += part[(size_t(r) * kCosSplit + s) * 3 + 2];
    }
    const float nrx = sqrtf(nx), nry = sqrtf(ny);
    const float cx = fmaxf(nrx, eps), cy = fmaxf(nry, eps);
    const float c = d / (cx * cy);
    coef[2 * r] = 1.f / (cx * cy);
    coef[2 * r + 1] = nrx > eps ? -c / (cx * cx) : 0.f;
    acc += double(c);
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (int(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = float(1. - red[0] / double(rows));
}

// g[row, i] = -(grad_loss / rows) (ca ft_i + cb fr_i)
__global__ __launch_bounds__(256) void cosine_backward_kernel(CosArgs a, const float* __restrict__ coef, const float* __restrict__ grad_loss,
                                                              int rows) {
  const int row = blockIdx.x, seg = blockIdx.y;
  const float* __restrict__ x = a.fr + cos_row_offset(a, row, a.ls_r);
  const float* __restrict__ y = a.ft + cos_row_offset(a, row, a.ls_t);
  float* __restrict__ g = a.g + cos_row_offset(a, row, a.ls_g);
  const float s = -grad_loss[0] / float(rows);
  const float ca = s * coef[2 * row], cb = s * coef[2 * row + 1];
  const size_t per = (a.HW / 4 + kCosSplit - 1) / kCosSplit * 4;
  const size_t lo = size_t(seg) * per, hi = lo + per < a.HW ? lo + per : a.HW;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
  if (vec) {
    const size_t hi4 = lo + ((hi > lo ? hi - lo : 0) & ~size_t(3));
    for (size_t i = lo + size_t(threadIdx.x) * 4; i < hi4; i += 1024) {
      const float4 u = *reinterpret_cast<const float4*>(x + i), v = *reinterpret_cast<const float4*>(y + i);
      *reinterpret_cast<float4*>(g + i) = make_float4(ca * v.x + cb * u.x, ca * v.y + cb * u.y, ca * v.z + cb * u.z, ca * v.w + cb * u.w);
    }
    for (size_t i = hi4 + threadIdx.x; i < hi; i += 256) g[i] = ca * y[i] + cb * x[i];
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) g[i] = ca * y[i] + cb * x[i];
  }
}

}  // namespace dfn

using namespace dfn;

extern "C" size_t dfn_triplet_loss_state_bytes(int L, int B, int rows) {
  if (L < 1 || B < 1 || rows < 1) return 0;
  return triplet_state_bytes(L, B, rows);
}

namespace {
bool triplet_args_ok(const void* f1, const void* f2, int L, int B, int rows, int W, size_t ls1, size_t ls2) {
  const size_t need = (size_t)B * rows * W;
  return f1 && f2 && L >= 1 && B >= 1 && rows >= 1 && W >= 1 && (L == 1 || (ls1 >= need && ls2 >= need));
}
}  // namespace

extern "C" int dfn_triplet_loss_forward(const float* f1, size_t level_stride1, const float* f2, size_t level_stride2, int L, int B,
                                        int rows, int W, float margin, int mining, float* loss, void* state, size_t state_bytes,
                                        void* stream) {
  if (!triplet_args_ok(f1, f2, L, B, rows, W, level_stride1, level_stride2) || !loss || !state || mining < 0 || mining > 2)
    return set_error(DFN_ERR_ARG, "dfn_triplet_loss_forward: bad argument");
  if (state_bytes < dfn_triplet_loss_state_bytes(L, B, rows))
    return set_error(DFN_ERR_ARG, "dfn_triplet_loss_forward: state too small (%zu < %zu)", state_bytes, dfn_triplet_loss_state_bytes(L, B, rows));
  const TripletState t = carve_triplet(state, L, B, rows);
  hipError_t e = launch_triplet_forward(f1, level_stride1, f2, level_stride2, L, B, rows, W, margin, 1e-6f, mining, t.part, t.case_dev,
                                        mining ? t.mse : nullptr, t.margin, t.row_stat, loss, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_triplet_loss_forward: %s", hipGetErrorString(e));
  return DFN_OK;
}

extern "C" int dfn_triplet_loss_backward(const float* f1, size_t level_stride1, const float* f2, size_t level_stride2, int L, int B,
                                         int rows, int W, const float* grad_loss, const void* state, float* grad_f1,
                                         size_t grad_stride1, float* grad_f2, size_t grad_stride2, void* stream) {
  if (!triplet_args_ok(f1, f2, L, B, rows, W, level_stride1, level_stride2) || !grad_loss || !state ||
      !triplet_args_ok(grad_f1, grad_f2, L, B, rows, W, grad_stride1, grad_stride2))
    return set_error(DFN_ERR_ARG, "dfn_triplet_loss_backward: bad argument");
  const TripletState t = carve_triplet(const_cast<void*>(state), L, B, rows);
  hipError_t e = launch_triplet_backward(f1, level_stride1, f2, level_stride2, L, B, rows, W, 1e-6f, t.case_dev, t.row_stat, t.margin, grad_loss,
                                         grad_f1, grad_stride1, grad_f2, grad_stride2, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_triplet_loss_backward: %s", hipGetErrorString(e));
  return DFN_OK;
}

// ---- cosine feature loss: state = [rows * kCosSplit * 3 floats partial sums][rows * 2 floats coefficients]
extern "C" size_t dfn_feature_cosine_state_bytes(int n_levels, int B, int C) {
  if (n_levels < 1 || B < 1 || C < 1) return 0;
  return size_t(n_levels) * B * C * (kCosSplit * 3 + 2) * sizeof(float);
}
namespace {
int cos_args(CosArgs& a, const float* fr, size_t ls_r, const float* ft, size_t ls_t, const int* levels, int n_levels, int B, int C, size_t HW,
             const char* fn) {
  if (!fr || !ft || !levels || n_levels < 1 || n_levels > kCosMaxLevels || B < 1 || C < 1 || HW < 1)
    return set_error(DFN_ERR_ARG, "%s: bad argument (1 <= n_levels <= %d)", fn, kCosMaxLevels);
  a = CosArgs{};
  a.fr = fr; a.ft = ft; a.ls_r = ls_r; a.ls_t = ls_t; a.HW = HW; a.n_levels = n_levels; a.B = B; a.C = C;
  for (int i = 0; i < n_levels; ++i) {
    if (levels[i] < 0) return set_error(DFN_ERR_ARG, "%s: negative level", fn);
    a.lv[i] = levels[i];
  }
  return DFN_OK;
}
}  // namespace
extern "C" int dfn_feature_cosine_forward(const float* fr, size_t level_stride_r, const float* ft, size_t level_stride_t, const int* levels,
                                          int n_levels, int B, int C, size_t HW, float* loss, void* state, size_t state_bytes, void* stream) {
  CosArgs a;
  if (int rc = cos_args(a, fr, level_stride_r, ft, level_stride_t, levels, n_levels, B, C, HW, "dfn_feature_cosine_forward")) return rc;
  if (!loss || !state || state_bytes < dfn_feature_cosine_state_bytes(n_levels, B, C))
    return set_error(DFN_ERR_ARG, "dfn_feature_cosine_forward: null loss / state or state too small");
  const int rows = n_levels * B * C;
  float* part = static_cast<float*>(state);
  float* coef = part + size_t(rows) * kCosSplit * 3;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cosine_stats_kernel, dim3(rows, kCosSplit), dim3(256), 0, s, a, part);
  hipLaunchKernelGGL(cosine_finish_kernel, dim3(1), dim3(256), 0, s, part, rows, 1e-6f, coef, loss);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_feature_cosine_forward: %s", hipGetErrorString(e));
  return DFN_OK;
}
extern "C" int dfn_feature_cosine_backward(const float* fr, size_t level_stride_r, const float* ft, size_t level_stride_t, const int* levels,
                                           int n_levels, int B, int C, size_t HW, const float* grad_loss, const void* state, float* grad_fr,
                                           size_t grad_stride, void* stream) {
  CosArgs a;
  if (int rc = cos_args(a, fr, level_stride_r, ft, level_stride_t, levels, n_levels, B, C, HW, "dfn_feature_cosine_backward")) return rc;
  if (!grad_loss || !state || !grad_fr) return set_error(DFN_ERR_ARG, "dfn_feature_cosine_backward: null argument");
  a.g = grad_fr; a.ls_g = grad_stride;
  const int rows = n_levels * B * C;
  const float* coef = static_cast<const float*>(state) + size_t(rows) * kCosSplit * 3;
  hipLaunchKernelGGL(cosine_backward_kernel, dim3(rows, kCosSplit), dim3(256), 0, static_cast<hipStream_t>(stream), a, coef, grad_loss, rows);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_feature_cosine_backward: %s", hipGetErrorString(e));
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ DFNet_dm loss block
// feature/direct_feature_matching.py:359-370: photo_loss = mean((rgb - target)^2) over the [B,3,H,W] frames, pose_loss =
// mse_loss(pose_, pose) over [B,12], loss = w[0] pose_loss + w[1] photo_loss + w[2] feat_loss (combine_loss).  As torch ops this is
// three reductions, five scalar products / sums and ~15 autograd nodes of one tiny kernel each — 0.5 ms of an otherwise idle GPU
// per step (the host cannot enqueue them faster).  Here: one partial-sum pass + one finishing block forward, one element-wise pass
// backward.  Sums in fp64 over fixed chunks in a fixed order (deterministic).
namespace dfn {
namespace {
constexpr int kDmBlocks = 256;
__global__ __launch_bounds__(256) void dm_loss_partial_kernel(const float* __restrict__ rgb, const float* __restrict__ tgt, size_t n,
                                                             double* __restrict__ part) {
  __shared__ double red[4];
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double acc = 0.;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float d = rgb[i] - tgt[i];
    acc += double(d * d);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[0..3] = loss, photo_loss, pose_loss, feat_loss
__global__ __launch_bounds__(64) void dm_loss_finish_kernel(const double* __restrict__ part, int n_part, size_t n, const float* __restrict__ pose,
                                                           const float* __restrict__ pose_gt, int n_pose, const float* __restrict__ feat,
                                                           float w_pose, float w_photo, float w_feat, float* __restrict__ out) {
  double s = 0.;
  for (int i = threadIdx.x; i < n_part; i += 64) s += part[i];
  double p = 0.;
  for (int i = threadIdx.x; i < n_pose; i += 64) {
    const float d = pose[i] - pose_gt[i];
    p += double(d * d);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d, 64); p += __shfl_xor(p, d, 64); }
  if (threadIdx.x == 0) {
    const float photo = float(s / double(n)), pl = float(p / double(n_pose)), fl = feat ? feat[0] : 0.f;
    out[1] = photo; out[2] = pl; out[3] = fl;
    out[0] = w_pose * pl + w_photo * photo + w_feat * fl;   // left to right, as the reference's expression
  }
}
__global__ __launch_bounds__(256) void dm_loss_backward_kernel(const float* __restrict__ rgb, const float* __restrict__ tgt, size_t n,
                                                              const float* __restrict__ pose, const float* __restrict__ pose_gt, int n_pose,
                                                              float w_pose, float w_photo, float w_feat, const float* __restrict__ g,
                                                              float* __restrict__ grad_rgb, float* __restrict__ grad_pose,
                                                              float* __restrict__ grad_feat) {
  const float gl = g[0];
  const float cr = gl * w_photo * 2.f / float(n);
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    grad_rgb[i] = cr * (rgb[i] - tgt[i]);
  if (blockIdx.x == 0) {
    const float cp = gl * w_pose * 2.f / float(n_pose);
    for (int i = threadIdx.x; i < n_pose; i += blockDim.x) grad_pose[i] = cp * (pose[i] - pose_gt[i]);
    if (threadIdx.x == 0 && grad_feat) grad_feat[0] = gl * w_feat;
  }
}
}  // namespace
}  // namespace dfn

extern "C" size_t dfn_dm_loss_scratch_bytes(void) { return dfn::kDmBlocks * sizeof(double); }
extern "C" int dfn_dm_loss_forward(const float* rgb, const float* target, size_t n, const float* pose, const float* pose_gt, int n_pose,
                                   const float* feat_loss, float w_pose, float w_photo, float w_feat, float* out4, void* scratch,
                                   void* stream) {
  if (!rgb || !target || !n || !pose || !pose_gt || n_pose < 1 || !out4 || !scratch)
    return set_error(DFN_ERR_ARG, "dfn_dm_loss_forward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(scratch);
  hipLaunchKernelGGL(dfn::dm_loss_partial_kernel, dim3(dfn::kDmBlocks), dim3(256), 0, s, rgb, target, n, part);
  hipLaunchKernelGGL(dfn::dm_loss_finish_kernel, dim3(1), dim3(64), 0, s, part, dfn::kDmBlocks, n, pose, pose_gt, n_pose, feat_loss, w_pose,
                     w_photo, w_feat, out4);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_dm_loss_forward: %s", hipGetErrorString(e));
  return DFN_OK;
}
extern "C" int dfn_dm_loss_backward(const float* rgb, const float* target, size_t n, const float* pose, const float* pose_gt, int n_pose,
                                    float w_pose, float w_photo, float w_feat, const float* grad_loss, float* grad_rgb, float* grad_pose,
                                    float* grad_feat, void* stream) {
  if (!rgb || !target || !n || !pose || !pose_gt || n_pose < 1 || !grad_loss || !grad_rgb || !grad_pose)
    return set_error(DFN_ERR_ARG, "dfn_dm_loss_backward: bad argument");
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(dfn::dm_loss_backward_kernel, dim3(unsigned(blocks < 2048 ? blocks : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     rgb, target, n, pose, pose_gt, n_pose, w_pose, w_photo, w_feat, grad_loss, grad_rgb, grad_pose, grad_feat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_dm_loss_backward: %s", hipGetErrorString(e));
  return DFN_OK;
}
