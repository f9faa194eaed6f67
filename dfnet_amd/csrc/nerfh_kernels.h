// nerfh_kernels.h — internal launch interface between the C ABI (dfn_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dfn {

struct MlpArgs {
  const char* blob;        // packed MFMA fragments of the net (device)
  const uint32_t* tab;     // [n_units][2] = (byte offset, byte size) of each staging unit (device)
  int n_units;
  const float* rays_o;     // [n_rays,3]
  const float* rays_d;     // [n_rays,3]
  const float* z;          // fine: [n_rays, n_samples]; coarse: unused (linspace in-kernel)
  const float* ray_bias;   // fine: [n_rays, kRayBiasFloats]
  float* out;              // coarse: sigma [n_rays, n_samples]; fine: raw [n_rays, n_samples, 9]
  float* partial;          // fine, fused compositing: [n_rays * n_samples / 64][12] per-segment composites (raw unused)
  long long n_rays;
  int n_samples;
  float near, far;
  unsigned long long* timing;  // DFN_TIMING builds: per-wave cycle counters [total, dma wait, barrier, tile inputs]
  float in_scale;          // split-f16 only: weight scale x activation scale carried by the accumulators (else 1)
  int lindisp;             // coarse: depths linear in disparity instead of depth (rendering.py:272-273)
  int dma_waves;           // waves of a workgroup that issue the weight DMA (0 = the kernel geometry's default, launch_one)
  int* status;             // range guard (device int, may be null): bit 0 = an f16 activation overflowed to inf, bit 1 = a
                           // split-f16 hi half saturated (DFN_RANGE_*: dfn_nerfh_range_status)
  uint32_t* masks;         // fine, split-f16, raw output: also record the ReLU signs of every hidden unit for the gradient kernel's
                           // backward-only pass ([tiles][waves][kBwdMaskWords][64 lanes], as BwdArgs::masks); null otherwise
};

hipError_t launch_mlp(bool fine, int prec, int variant, const MlpArgs& a, int n_cu, hipStream_t stream, int width = 128);

// --- stages (nerfh_stages.hip)
// (frames > 1: c2w [frames,3,4], outputs [frames,H,W,3] — one launch for the frames of a mini-batch)
hipError_t launch_raygen(int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d,
                         float* viewdirs, hipStream_t stream, int frames = 1);
hipError_t launch_viewdirs(const float* rays_d, size_t n, float* viewdirs, hipStream_t stream);
hipError_t launch_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, size_t n, float* out_o,
                           float* out_d, hipStream_t stream);
hipError_t launch_posenc(const float* x, size_t n, int L, int mode, float* out, hipStream_t stream);

struct RayBiasWeights {       // device pointers, fp32
  const float* w_dir;         // [77][nout]  transposed dir_encoding.0.weight[:, W:W+77]
  const float* b_dir;         // [nout]
  const float* w_tr;          // [20][nout]  transposed transient_encoding.0.weight[:, W:W+20]
  const float* b_tr;          // [nout]
  const float* emb_a;         // [n_vocab, dim_a]
  const float* emb_t;         // [n_vocab, dim_t]
  int hist_bin, dim_a, dim_t, n_vocab;
  int nout;                   // outputs per table = netwidth / 2 (w_dir, w_tr are [*][nout]; the table row is 2 * nout floats)
};
hipError_t launch_ray_bias(const RayBiasWeights& w, const float* viewdirs, const float* hist,
                           size_t hist_rows, size_t n_rays, float* table, hipStream_t stream);

hipError_t launch_coarse_weights(const float* sigma, const float* z, size_t n, int N, float* weights,
                                 hipStream_t stream);
hipError_t launch_sample_pdf(const float* bins, const float* weights, size_t n, int nb, int Ni,
                             const float* u, float* out, hipStream_t stream);
hipError_t launch_sample_fine(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far,
                              float* z_fine, float* weights_coarse, float* z_samples, hipStream_t stream, int lindisp = 0);
hipError_t launch_composite_fine(const float* raw, const float* z, size_t n_rays, int Nf, float beta_min,
                                 int flags, float* rgb, float* disp, float* acc, float* depth,
                                 float* weights, float* beta, hipStream_t stream);

// Chains the per-64-sample segment composites written by the fused fine kernel: partial [n_rays, segs, 12].
hipError_t launch_composite_combine(const float* partial, size_t n_rays, int segs, float beta_min, int flags, float* rgb,
                                    float* disp, float* acc, hipStream_t stream);

// --- gradient path (nerfh_bwd.hip, nerfh_stages.hip)
struct BwdArgs {
  const char* blob;        // forward fine units followed by the backward (W^T) units
  const uint32_t* tab;
  int n_units;
  const float* rays_o;     // [n_rays,3]
  const float* rays_d;     // [n_rays,3]
  const float* viewdirs;   // [n_rays,3]
  const float* z;          // [n_rays, n_samples]
  const float* ray_bias;   // [n_rays, kRayBiasFloats]
  const float* graw;       // [n_rays, n_samples, 9]  d L / d raw
  float* gpts;             // [n_rays, n_samples, 6]  d L / d point (3), d L / d viewdir via this sample (3)
  long long n_rays;
  int n_samples;
  float in_scale;          // split-f16 forward units: scale carried by the accumulators (else 1)
  // Two-pass form (mode 1 = forward, writing raw and the ReLU masks; mode 2 = backward from them, no forward recompute):
  float* raw_out;          // mode 1: [n_rays, n_samples, 9]
  const float* raw_in;     // mode 2
  uint32_t* masks;         // [tiles][waves][kBwdMaskWords][64 lanes]: written by mode 1, read by mode 2
};
constexpr int kBwdMaskWords = 21;    // 8 trunk layers x 128 bits, dir_encoding 64, transient_encoding 4 x 64 — per point
constexpr int kBwdTilePoints = 256;  // points per workgroup tile of the split-f16 gradient kernel (8 waves x 32)
hipError_t launch_mlp_fine_backward(int prec, const BwdArgs& a, int n_cu, hipStream_t stream, int mode = 0);
// d L / d raw from d L / d rgb through the fine compositing (test-time, rgb only).
hipError_t launch_composite_fine_backward(const float* raw, const float* z, const float* grad_rgb, size_t n_rays, int Nf,
                                          float* graw, hipStream_t stream);
// Per-ray reduction of the per-sample gradients: d o = sum g, d d = sum z g (+ viewdir normalisation when
// `derive_viewdirs`), d viewdirs = sum gv (when grad_viewdirs != nullptr).  accumulate: add to what grad_o / grad_d hold.
hipError_t launch_ray_grad_reduce(const float* gpts, const float* z, const float* rays_d, size_t n_rays, int Nf,
                                  int derive_viewdirs, float* grad_o, float* grad_d, float* grad_viewdirs,
                                  hipStream_t stream, int accumulate = 0);
// get_rays backward: d c2w[3][4] from d rays_o / d rays_d of an H x W image.
hipError_t launch_raygen_backward(int H, int W, float focal, const float* grad_o, const float* grad_d, float* grad_c2w,
                                  hipStream_t stream, int frames = 1);
// adjoint of launch_bicubic: g_out [UH,UW,C] -> g_in [H,W,C].
// (frames > 1: a batch of frames per launch; nchw: the enlarged frames — g_out for the adjoint — are planar [C,UH,UW] per frame)
hipError_t launch_bicubic_backward(const float* gout, int H, int W, int C, int UH, int UW, float* gin, hipStream_t stream, int frames = 1,
                                   bool nchw = false);
hipError_t launch_bicubic(const float* in, int H, int W, int C, int UH, int UW, float* out, hipStream_t stream, int frames = 1,
                          bool nchw = false);

}  // namespace dfn
