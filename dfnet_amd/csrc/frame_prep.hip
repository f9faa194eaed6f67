// frame_prep.hip — the dataset front-end's per-frame arithmetic on the device (gfx950).
// /root/reference/dataset_loaders/seven_scenes.py:324-352: a decoded 8-bit frame becomes
//   img  = cv2.resize(img / 255, (W, H), INTER_AREA)  -> ToTensor: fp32 [3, H, W] in [0, 1]
//   hist = round(histc(Y(img), bins, 0, 1) / sum * 100), Y = 0.299 R + 0.587 G + 0.114 B
// (the histogram is the index vector of NeRF-H's appearance / transient embeddings).  One kernel: a thread per output
// pixel averages its source box — INTER_AREA is the coverage-weighted mean; for an integer factor the plain box mean,
// taken here from the EXACT integer sum of the 8-bit samples — writes the three planes and votes into an LDS
// histogram; integer counts make the result independent of the order of the votes.  The last workgroup to finish
// turns the counts into rounded percentages (torch.round: half to even).  HBM-bound: 3 bytes in, 12 bytes out per
// output pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"

namespace dfn {

constexpr int kMaxHistBins = 64;

__global__ __launch_bounds__(256) void frame_prep_kernel(const uint8_t* __restrict__ src, int h, int w, int H, int W, int bins,
                                                         float* __restrict__ img, float* __restrict__ hist, unsigned* __restrict__ scratch) {
  __shared__ unsigned lh[kMaxHistBins];
  __shared__ bool last;
  for (int i = threadIdx.x; i < bins; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const int n = H * W;
  const bool integer = (h % H == 0) && (w % W == 0);
  const int fy = h / H, fx = w / W;
  const double sy = (double)h / H, sx = (double)w / W;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
    const int y = o / W, x = o - y * W;
    float c[3];
    if (integer) {
      unsigned s[3] = {0, 0, 0};
      for (int j = 0; j < fy; ++j) {
        const uint8_t* row = src + ((size_t)(y * fy + j) * w + (size_t)x * fx) * 3;
        for (int i = 0; i < fx; ++i) { s[0] += row[3 * i]; s[1] += row[3 * i + 1]; s[2] += row[3 * i + 2]; }
      }
      const float inv = 1.f / (255.f * float(fy * fx));
      for (int k = 0; k < 3; ++k) c[k] = float(s[k]) * inv;
    } else {   // fractional coverage of the source pixels by the output pixel's box
      const double y0 = y * sy, y1 = (y + 1) * sy, x0 = x * sx, x1 = (x + 1) * sx;
      double s[3] = {0.0, 0.0, 0.0};
      for (int j = (int)y0; j < h && j < y1; ++j) {
        const double wy = fmin(y1, j + 1.0) - fmax(y0, (double)j);
        for (int i = (int)x0; i < w && i < x1; ++i) {
          const double wt = wy * (fmin(x1, i + 1.0) - fmax(x0, (double)i));
          const uint8_t* p = src + ((size_t)j * w + i) * 3;
          s[0] += wt * p[0]; s[1] += wt * p[1]; s[2] += wt * p[2];
        }
      }
      for (int k = 0; k < 3; ++k) c[k] = float(s[k] / (255.0 * sy * sx));
    }
    for (int k = 0; k < 3; ++k) img[(size_t)k * n + o] = c[k];
    const float yl = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
    if (yl >= 0.f && yl <= 1.f) {   // torch.histc(min = 0, max = 1): the right edge belongs to the last bin
      int b = (int)(yl * (float)bins);
      if (b >= bins) b = bins - 1;
      atomicAdd(&lh[b], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += blockDim.x)
    if (lh[i]) atomicAdd(&scratch[i], lh[i]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&scratch[kMaxHistBins], 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int i = 0; i < bins; ++i) total += float(atomicAdd(&scratch[i], 0u));
    for (int i = 0; i < bins; ++i) hist[i] = rintf(float(atomicAdd(&scratch[i], 0u)) / total * 100.f);
  }
}

}  // namespace dfn

extern "C" size_t dfn_frame_prep_scratch_bytes(void) { return (dfn::kMaxHistBins + 1) * sizeof(unsigned); }

extern "C" int dfn_frame_prep(const uint8_t* rgb_hwc, int h, int w, int H, int W, int hist_bins, float* img_chw, float* hist,
                              void* scratch, void* stream) {
  using namespace dfn;
  if (!rgb_hwc || !img_chw || !hist || !scratch || h < 1 || w < 1 || H < 1 || W < 1 || H > h || W > w || hist_bins < 1 ||
      hist_bins > kMaxHistBins)
    return set_error(DFN_ERR_ARG, "dfn_frame_prep: bad argument (downscale only, 1 <= hist_bins <= %d)", kMaxHistBins);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, dfn_frame_prep_scratch_bytes(), s) != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_frame_prep: memset failed");
  const int n = H * W, grid = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(frame_prep_kernel, dim3(grid), dim3(256), 0, s, rgb_hwc, h, w, H, W, hist_bins, img_chw, hist,
                     static_cast<unsigned*>(scratch));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_frame_prep: %s", hipGetErrorString(e));
  return DFN_OK;
}
