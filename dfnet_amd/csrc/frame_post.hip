// frame_post.hip — render_path's per-frame back-end on the device (gfx950; SURVEY 8(f) N3, second half).
// /root/reference/script/models/rendering.py:423-452 turns every rendered frame into three 8-bit images and one number:
//   rgb8  = to8b(rgb)                to8b = (255 * np.clip(x, 0, 1)).astype(np.uint8): TRUNCATION (models/nerf.py:11)
//   disp8 = to8b(disp / np.max(disp))
//   gt8   = to8b(gt)
//   psnr  = -10 log10(mean((rgb - gt)^2))
// on the host, one frame at a time, after a device -> host copy of the fp32 frame.  Here a batch of frames is processed where it
// was rendered: pass 1 = per-frame maximum of disp (ordered-integer atomicMax), pass 2 = the three conversions + the squared
// error (fp64 block partials, one atomicAdd per block and frame).  What leaves the GPU is 7 bytes per pixel of uint8 instead of
// 16 of fp32, and 8 bytes per frame for the PSNR.  HBM-bound: 28 B in (rgb, disp twice, gt), 7 B out per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"

namespace dfn {

struct FrameStat { unsigned max_key; unsigned pad; double sse; };   // per frame, in the caller's scratch

// order-preserving map float -> unsigned (so that atomicMax on the key is max on the float; NaN sorts above +inf)
__device__ __forceinline__ unsigned float_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ uint8_t to8b(float x) {
  // 255 * clip(x, 0, 1) in fp32, then the C cast numpy's astype performs (NaN -> 0 as on x86-64 numpy builds)
  const float c = fminf(fmaxf(x, 0.f), 1.f);
  return (uint8_t)(int)(255.f * c);
}

__global__ __launch_bounds__(256) void disp_max_kernel(const float* __restrict__ disp, size_t hw, FrameStat* __restrict__ stat) {
  const float* d = disp + (size_t)blockIdx.y * hw;
  unsigned best = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned k = float_key(d[i]);
    best = k > best ? k : best;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = __shfl_xor(best, off, 64);
    best = o > best ? o : best;
  }
  __shared__ unsigned wbest[4];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = wbest[w] > best ? wbest[w] : best;
    atomicMax(&stat[blockIdx.y].max_key, best);
  }
}

__global__ __launch_bounds__(256) void frame_post_kernel(const float* __restrict__ rgb, const float* __restrict__ disp,
                                                         const float* __restrict__ gt, size_t gt_stride, size_t hw,
                                                         uint8_t* __restrict__ rgb8, uint8_t* __restrict__ disp8,
                                                         uint8_t* __restrict__ gt8, size_t gt8_stride, FrameStat* __restrict__ stat) {
  const size_t f = blockIdx.y;
  const float* r = rgb + f * hw * 3;
  const float* d = disp + f * hw;
  const float* g = gt ? gt + f * gt_stride : nullptr;
  uint8_t* r8 = rgb8 + f * hw * 3;
  uint8_t* d8 = disp8 + f * hw;
  uint8_t* g8 = (gt8 && g) ? gt8 + f * gt8_stride : nullptr;
  const float dmax = key_float(stat[f].max_key);
  double sse = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
    d8[i] = to8b(__fdiv_rn(d[i], dmax));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = r[3 * i + c];
      r8[3 * i + c] = to8b(v);
      if (g) {
        const float t = g[3 * i + c];
        if (g8 && (gt8_stride || f == 0)) g8[3 * i + c] = to8b(t);
        const float e = v - t;   // fp32 difference as numpy forms it, squared and summed in fp64
        sse += (double)e * (double)e;
      }
    }
  }
  if (!g) return;
  for (int off = 32; off >= 1; off >>= 1) sse += __shfl_xor(sse, off, 64);
  __shared__ double wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sse;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&stat[f].sse, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

__global__ void frame_post_finish_kernel(const FrameStat* __restrict__ stat, int n, double inv_count, float* __restrict__ mse,
                                         float* __restrict__ disp_max) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  if (mse) mse[f] = (float)(stat[f].sse * inv_count);
  if (disp_max) disp_max[f] = key_float(stat[f].max_key);
}

}  // namespace dfn

extern "C" size_t dfn_frame_post_scratch_bytes(int n_frames) { return (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(dfn::FrameStat); }

extern "C" int dfn_frame_post(const float* rgb, const float* disp, const float* gt, int gt_per_frame, int n_frames, int H, int W,
                              uint8_t* rgb8, uint8_t* disp8, uint8_t* gt8, float* mse, float* disp_max, void* scratch,
                              void* stream) {
  using namespace dfn;
  if (n_frames == 0) return DFN_OK;
  if (!rgb || !disp || !rgb8 || !disp8 || !scratch || n_frames < 0 || n_frames > 65535 || H < 1 || W < 1 || (mse && !gt))
    return set_error(DFN_ERR_ARG, "dfn_frame_post: bad argument (1 <= n_frames <= 65535; mse needs gt)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  FrameStat* stat = static_cast<FrameStat*>(scratch);
  if (hipMemsetAsync(stat, 0, dfn_frame_post_scratch_bytes(n_frames), s) != hipSuccess)
    return set_error(DFN_ERR_HIP, "dfn_frame_post: memset failed");
  const size_t hw = (size_t)H * W;
  const int bx = (int)((hw + 256 * 8 - 1) / (256 * 8) < 256 ? (hw + 256 * 8 - 1) / (256 * 8) : 256);   // >= 8 pixels per thread
  hipLaunchKernelGGL(disp_max_kernel, dim3(bx, n_frames), dim3(256), 0, s, disp, hw, stat);
  const size_t gstride = gt_per_frame ? hw * 3 : 0;
  hipLaunchKernelGGL(frame_post_kernel, dim3(bx, n_frames), dim3(256), 0, s, rgb, disp, gt, gstride, hw, rgb8, disp8, gt8, gstride, stat);
  if (mse || disp_max)
    hipLaunchKernelGGL(frame_post_finish_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, s, stat, n_frames, 1.0 / (double)(hw * 3), mse,
                       disp_max);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_frame_post: %s", hipGetErrorString(e));
  return DFN_OK;
}
