/*
 * dfnet_hip.h — C ABI of libdfnet_hip.so: the MI355X (gfx950) implementation of DFNet's
 * NeRF-H volumetric-rendering hot path and DFNet feature-extractor forward.
 *
 * The reference (ActiveVisionLab/DFNet) has no FFI for this path — it sits behind plain
 * Python functions — so every entry point below names the reference function (file:line,
 * relative to /root/reference/script/) whose arithmetic it replaces.  INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - flat extern "C", opaque handles, int status (0 = DFN_OK, negative = error; text via
 *     dfn_last_error(), thread-local);
 *   - NO ownership transfer: every data buffer is a caller-allocated DEVICE pointer
 *     (fp32, contiguous, row-major) unless the parameter says "host";
 *   - the caller passes its hipStream_t as `void* stream`; functions enqueue and return
 *     (no implicit synchronisation) unless documented otherwise;
 *   - one handle per device; a handle's methods are not re-entrant, different handles are
 *     thread-safe.
 */
#ifndef DFNET_HIP_H
#define DFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DFN_OK = 0,
  DFN_ERR_ARG = -1,         /* bad argument (null pointer, size mismatch, unknown name) */
  DFN_ERR_HIP = -2,         /* a HIP runtime call failed */
  DFN_ERR_STATE = -3,       /* handle not committed / parameter missing */
  DFN_ERR_UNSUPPORTED = -4, /* configuration outside what the kernels implement */
  DFN_ERR_RANGE = -5        /* activations left the range of a narrow arithmetic mode (dfn_nerfh_range_status) */
};

/* Arithmetic of the MLP / conv contractions. */
enum {
  DFN_PREC_F16 = 0, /* f16 MFMA inputs, fp32 accumulate (v_mfma_f32_32x32x16_f16); PE via v_sin/v_cos */
  DFN_PREC_F32 = 1, /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32); PE via full-range sinf/cosf */
  DFN_PREC_F16X3 = 2 /* DFNet only: split-f16 — operands hi + lo in f16, three f16 MFMAs per product, fp32
                        activations: fp32-grade results at f16 MFMA rate (forward; its gradient convs run as F32) */
};

const char* dfn_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int dfn_abi_version(void);

/* ------------------------------------------------------------------ NeRF-H network handle
 * Replaces models/nerfw.py:220-354 (class NeRFW, coarse + fine), the two histogram
 * nn.Embedding tables (nerfw.py:385-391) and the Embedder (nerfw.py:98-133). */
typedef struct dfn_nerfh_s* dfn_nerfh_t;

typedef struct {
  int depth;          /* args.netdepth        (8)    */
  int width;          /* args.netwidth        (128, 256: register-resident kernels; other even widths: generic path) */
  int multires;       /* args.multires        (10)   */
  int multires_views; /* args.multires_views  (4)    */
  int hist_bin;       /* args.hist_bin        (10)   */
  int dim_a;          /* per-bin appearance dim (5): in_channels_a = hist_bin*dim_a */
  int dim_t;          /* per-bin transient dim  (2): in_channels_t = hist_bin*dim_t */
  int n_vocab;        /* args.N_vocab         (1000) */
} dfn_nerfh_desc;

int dfn_nerfh_create(const dfn_nerfh_desc* desc, dfn_nerfh_t* out);
int dfn_nerfh_destroy(dfn_nerfh_t h);
/* Set one parameter from a HOST fp32 buffer.  `name` is "<net>.<state_dict key>" with
 * <net> in {coarse, fine} (keys of nerfw.py:259-295, e.g. "fine.xyz_encoding_5.0.weight"),
 * or "embedding_a.weight" / "embedding_t.weight". */
int dfn_nerfh_set_param(dfn_nerfh_t h, const char* name, const float* host, size_t numel);
/* Pack all parameters into the MFMA fragment layouts and upload them (synchronous). */
int dfn_nerfh_commit(dfn_nerfh_t h);
/* Options of render_rays that the reference passes as keyword arguments (rendering.py:245-256), applied by every entry point
 * that takes this handle until changed.  DFN_RENDER_LINDISP: the coarse depths are linear in disparity, z = 1 / ((1 - t) / near +
 * t / far) (rendering.py:272-273; near must be > 0).  white_bkgd is not offered: in the reference it reaches the coarse compositor
 * in the output_transient slot (rendering.py:295) and raises TypeError at test time. */
/* DFN_RENDER_COARSE_F16 (not a reference keyword; a precision choice of this library): the COARSE network of the test-time render
 * runs with f16 MFMA inputs whatever `prec` the call names, the fine network in `prec`.  The coarse pass only places the importance
 * samples (z_samples are detached, rendering.py:302; its colour is never produced at test time), so with prec = DFN_PREC_F16X3 the pixel
 * is still composited from fp32-grade fine-network outputs. */
enum { DFN_RENDER_LINDISP = 1, DFN_RENDER_COARSE_F16 = 2 };
int dfn_nerfh_set_render_options(dfn_nerfh_t h, int flags);
/* Range guard of the narrow arithmetic modes (no counterpart in the reference, which computes in fp32).  DFN_PREC_F16 holds a
 * layer's outputs as f16 (|x| <= 65504), DFN_PREC_F16X3 as hi + lo f16 halves of 16 x (|x| < 4094; hidden layers of the render kernels: < 4094 max|w|, a few hundred to a few thousand for a NeRF-H checkpoint); a checkpoint whose hidden
 * activations leave that range would render clamped or non-finite frames.  Every MLP kernel launched through this handle keeps the
 * largest activation pattern it converts (one v_pk_max_u16 per converted register) and ORs a bit into a device flag when it
 * overflowed / saturated.  This call waits for `stream`, reads and clears the flag:
 *   flags != NULL: *flags = DFN_RANGE_* bits, returns DFN_OK;
 *   flags == NULL: returns DFN_ERR_RANGE (text in dfn_last_error) when a bit is set, DFN_OK otherwise.
 * The Python host checks it at the end of every render_path batch and before a checkpoint-driven CLI run reports a PSNR. */
enum { DFN_RANGE_F16_OVERFLOW = 1, DFN_RANGE_F16X3_SATURATED = 2 };
int dfn_nerfh_range_status(dfn_nerfh_t h, int* flags, void* stream);
/* The same read-and-clear ENQUEUED on `stream` without waiting: *host_flags (page-locked host memory — hipHostMalloc /
 * torch.Tensor.pin_memory — else the copy is synchronous) holds the DFN_RANGE_* bits once the stream has passed this point (record
 * an event behind the call).  For training loops (run_nerf.py:32-80), which must not drain the stream every step: the fused
 * training step (dfn_nerfh_train_backward) leaves ZEROS in every gradient tensor of a step whose operands left the split-f16
 * range — a skipped step, never clamped gradients — so the host may learn of it a step or two late, re-commit and carry on. */
int dfn_nerfh_range_status_async(dfn_nerfh_t h, int* host_flags, void* stream);

/* ------------------------------------------------------------------ stage-level entry points
 * (each is the production kernel of that stage; exposed so parity tests can check a stage
 *  against the oracle in isolation) */

/* models/ray_utils.py:5-15 get_rays + rendering.py:366-371 viewdirs.  c2w: device [3,4]
 * row-major (12 floats).  Outputs [H*W,3] each; viewdirs may be NULL. */
int dfn_raygen(int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d,
               float* viewdirs, void* stream);
/* The same for the B frames of a mini-batch in ONE launch (direct_feature_matching.py:340-348 renders the frames of its batch in a
 * loop): c2w [B,3,4] -> rays_o, rays_d (, viewdirs) [B,H,W,3]. */
int dfn_raygen_frames(int B, int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d, float* viewdirs, void* stream);
/* models/ray_utils.py:27-46 ndc_rays (render(ndc=True), rendering.py:374-376, calls it with near = 1): rays_o / rays_d [n,3] of an
 * H x W pinhole camera -> normalised device coordinates, out_o / out_d [n,3] (may alias the inputs). */
int dfn_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, size_t n, float* out_o,
                 float* out_d, void* stream);

/* models/nerfw.py:105-133 Embedder.embed.  x [n,3] -> out [n, 3+6L].  mode 0 = full-range
 * sinf/cosf, 1 = the fast v_sin/v_cos path the f16 MLP kernels use. */
int dfn_posenc(const float* x, size_t n, int L, int mode, float* out, void* stream);

/* Coarse test-time query: rendering.py:269-292 (z, pts) + nerfw.py:37-46,315-334 (PE + trunk
 * + static_sigma/Softplus).  sigma [n_rays, Nc]. */
int dfn_mlp_coarse(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                   size_t n_rays, int Nc, float near, float far, float* sigma, void* stream);

/* rendering.py:161-193 coarse alpha/weights.  sigma, z [n, N] -> weights [n, N]. */
int dfn_coarse_weights(const float* sigma, const float* z, size_t n, int N, float* weights,
                       void* stream);

/* rendering.py:24-65 sample_pdf.  bins [n, nb], weights [n, nb-1] -> out [n, Ni].  u NULL =
 * deterministic linspace(0,1,Ni); else u [n, Ni] are the uniform draws. */
int dfn_sample_pdf(const float* bins, const float* weights, size_t n, int nb, int Ni,
                   const float* u, float* out, void* stream);

/* Fused production sampler: rendering.py:295-304 — coarse weights of sigma over the
 * linspace z, z_mid, sample_pdf(det) on the interior weights, merge/sort with the coarse z.
 * z_fine [n_rays, Nc+Ni]; weights_coarse / z_samples optional (may be NULL). */
int dfn_sample_fine(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far,
                    float* z_fine, float* weights_coarse, float* z_samples, void* stream);
/* The same with the DFN_RENDER_* options (this stage takes no handle): DFN_RENDER_LINDISP = coarse depths linear in disparity. */
int dfn_sample_fine_opt(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far, int render_flags,
                        float* z_fine, float* weights_coarse, float* z_samples, void* stream);

/* Fine query: rendering.py:305-313 + nerfw.py:62-95,297-354.  hist [hist_rows, hist_bin]
 * (float-valued indices; hist_rows is 1 = one image for all rays, or n_rays); raw
 * [n_rays, Nf, 9]; `bias_ws` is scratch of dfn_fine_bias_bytes(n_rays) bytes. */
size_t dfn_fine_bias_bytes(size_t n_rays);
int dfn_mlp_fine(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                 const float* viewdirs, const float* hist, size_t hist_rows, size_t n_rays,
                 const float* z_fine, int Nf, float* raw, void* bias_ws, void* stream);

/* rendering.py:144-243 raw2outputs_NeRFW, typ="fine", output_transient=True.
 * flags: bit0 test_time, bit1 static_only, bit2 white_bkgd.  rgb [n,3], disp/acc [n];
 * depth, weights [n,Nf], beta optional (NULL to skip). */
enum { DFN_COMP_TEST_TIME = 1, DFN_COMP_STATIC_ONLY = 2, DFN_COMP_WHITE_BKGD = 4 };
int dfn_composite_fine(const float* raw, const float* z, size_t n_rays, int Nf, float beta_min,
                       int flags, float* rgb, float* disp, float* acc, float* depth,
                       float* weights, float* beta, void* stream);

/* ------------------------------------------------------------------ whole-path entry points */

/* Scratch needed by dfn_render_rays / dfn_render_image for up to n_rays rays. */
size_t dfn_render_workspace_bytes(size_t n_rays, int Nc, int Ni);

/* rendering.py:245-337 render_rays at test time (perturb=0, raw_noise_std=0, white_bkgd=False, test_time=True; lindisp as set
 * by dfn_nerfh_set_render_options) over caller-provided rays (rendering.py:361-362).
 * viewdirs may be NULL (computed as d/|d|, rendering.py:366-371).  raw (optional)
 * [n_rays, Nc+Ni, 9] is the `retraw` output. */
int dfn_render_rays(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                    const float* viewdirs, const float* hist, size_t hist_rows, size_t n_rays,
                    int Nc, int Ni, float near, float far, float* rgb, float* disp, float* acc,
                    float* raw, void* workspace, size_t workspace_bytes, void* stream);

/* rendering.py:353-400 render(c2w=...) at test time: get_rays + the above for a full H x W
 * image.  c2w device [3,4]; hist device [hist_bin]; rgb [H,W,3], disp/acc [H,W]. */
int dfn_render_image(dfn_nerfh_t h, int prec, const float* c2w, int H, int W, float focal,
                     float near, float far, int Nc, int Ni, const float* hist, float* rgb,
                     float* disp, float* acc, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------ gradient of the render
 * What loss.backward() runs through render(c2w = pose) in the DFNet_dm step
 * (feature/direct_feature_matching.py:340-376): test-time render, NeRF weights frozen, no gradient through
 * the importance sampler (z_samples.detach(), rendering.py:302) and hence none into the coarse net.
 * Only d L / d rgb is propagated (disp / acc are not differentiated); white_bkgd = False. */

/* raw2outputs_NeRFW backward, rgb only: grad_raw [n_rays, Nf, 9] = d L / d raw from grad_rgb [n_rays, 3]. */
int dfn_composite_fine_backward(const float* raw, const float* z, const float* grad_rgb, size_t n_rays,
                                int Nf, float* grad_raw, void* stream);
/* Fine-network input gradient: from grad_raw [n_rays, Nf, 9] to grad_pts [n_rays, Nf, 6] =
 * [d L / d sample point (3), d L / d viewdir through this sample (3)].  Other arguments as dfn_mlp_fine. */
int dfn_mlp_fine_backward(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                          const float* viewdirs, const float* hist, size_t hist_rows, size_t n_rays,
                          const float* z_fine, int Nf, const float* grad_raw, float* grad_pts,
                          void* bias_ws, void* stream);

/* The same gradient in two passes, for an autograd step that runs a forward anyway (split-f16 only).
 * dfn_mlp_fine_saving: dfn_mlp_fine in split-f16 that also records the ReLU sign of every hidden unit (masks:
 * dfn_mlp_fine_mask_bytes(n_rays * Nf) of device memory).  dfn_mlp_fine_backward_saved: d L / d (point, viewdir) per
 * sample from those masks and raw alone — no forward recompute (the head derivatives follow from the head outputs:
 * sigmoid' = y (1 - y), softplus' = 1 - exp(-y)).  Same outputs as dfn_mlp_fine / dfn_mlp_fine_backward. */
size_t dfn_mlp_fine_mask_bytes(size_t n_points);
int dfn_mlp_fine_saving(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                        const float* hist, size_t hist_rows, size_t n_rays, const float* z_fine, int Nf, float* raw,
                        void* masks, void* bias_ws, void* stream);
int dfn_mlp_fine_backward_saved(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                                const float* viewdirs, size_t n_rays, const float* z_fine, int Nf, const float* raw,
                                const void* masks, const float* grad_raw, float* grad_pts, void* stream);
/* Per-ray reduction of grad_pts [n_rays, Nf, 6] (pts = o + d z, rendering.py:292,305): grad_rays_o = sum g,
 * grad_rays_d = sum z g; with derive_viewdirs != 0 the view-direction part is folded into grad_rays_d through
 * viewdirs = d/|d| (rendering.py:366-371), otherwise grad_viewdirs (optional) receives sum gv. */
int dfn_ray_grad_reduce(const float* grad_pts, const float* z_fine, const float* rays_d, size_t n_rays, int Nf,
                        int derive_viewdirs, float* grad_rays_o, float* grad_rays_d, float* grad_viewdirs,
                        void* stream);
/* get_rays backward (ray_utils.py:5-15): grad_c2w [3,4] from grad_rays_o / grad_rays_d [H*W, 3]. */
int dfn_raygen_backward(int H, int W, float focal, const float* grad_rays_o, const float* grad_rays_d,
                        float* grad_c2w, void* stream);
/* ... and its batched form: grad_rays_o / grad_rays_d [B,H*W,3] -> grad_c2w [B,3,4]. */
int dfn_raygen_frames_backward(int B, int H, int W, float focal, const float* grad_rays_o, const float* grad_rays_d, float* grad_c2w,
                               void* stream);
/* Scratch needed by dfn_render_rays_backward / dfn_render_image_backward for up to n_rays rays. */
size_t dfn_render_backward_workspace_bytes(size_t n_rays, int Nc, int Ni);
/* d L / d rays_o, d L / d rays_d [n_rays, 3] of dfn_render_rays from grad_rgb [n_rays, 3] (recomputes the
 * forward).  viewdirs == NULL: they are d/|d| and the normalisation is differentiated into grad_rays_d
 * (render(rays=...), rendering.py:366-371); otherwise they are an independent input and grad_viewdirs
 * (optional) receives their gradient. */
int dfn_render_rays_backward(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d,
                             const float* viewdirs, const float* hist, size_t hist_rows, size_t n_rays,
                             int Nc, int Ni, float near, float far, const float* grad_rgb,
                             float* grad_rays_o, float* grad_rays_d, float* grad_viewdirs,
                             void* workspace, size_t workspace_bytes, void* stream);
/* d L / d c2w [3,4] (device) of dfn_render_image from grad_rgb [H, W, 3]: the above + get_rays backward
 * (ray_utils.py:5-15). */
int dfn_render_image_backward(dfn_nerfh_t h, int prec, const float* c2w, int H, int W, float focal,
                              float near, float far, int Nc, int Ni, const float* hist,
                              const float* grad_rgb, float* grad_c2w, void* workspace,
                              size_t workspace_bytes, void* stream);

/* nn.Upsample(size=(outH,outW), mode='bicubic') (align_corners=False) of an [H,W,C] device image: the
 * x`tinyscale` enlargement of a low-resolution render (feature/misc.py:230-237,
 * feature/direct_feature_matching.py:344-346).  out [outH,outW,C]. */
int dfn_upsample_bicubic(const float* in, int H, int W, int C, int outH, int outW, float* out, void* stream);

/* Adjoint of dfn_upsample_bicubic: grad_out [outH,outW,C] -> grad_in [H,W,C]. */
int dfn_upsample_bicubic_backward(const float* grad_out, int H, int W, int C, int outH, int outW, float* grad_in,
                                  void* stream);
/* Both for a batch of frames in one launch: in [B,H,W,C] -> out [B,outH,outW,C]; grad_out [B,outH,outW,C] -> grad_in [B,H,W,C].
 * out_nchw != 0: the enlarged frames (and grad_out) are planar [B,C,outH,outW] — the layout DFNet.forward takes, i.e. the
 * `.permute(0, 3, 1, 2)` of direct_feature_matching.py:346 folded into the store. */
int dfn_upsample_bicubic_frames(const float* in, int B, int H, int W, int C, int outH, int outW, int out_nchw, float* out, void* stream);
int dfn_upsample_bicubic_frames_backward(const float* grad_out, int B, int H, int W, int C, int outH, int outW, int out_nchw,
                                         float* grad_in, void* stream);

/* ------------------------------------------------------------------ DFNet feature extractor
 * Replaces feature/dfnet.py:74-172 (class DFNet / DFNet_s: VGG16 `features` stack, AdaptLayers,
 * UpsamplingBilinear2d, GAP + fc_pose).  n_taps = 3 (DFNet: conv1_2, conv3_3, conv5_3) or 1 (DFNet_s). */
typedef struct dfn_dfnet_s* dfn_dfnet_t;
int dfn_dfnet_create(int n_taps, int feat_dim, dfn_dfnet_t* out);
int dfn_dfnet_destroy(dfn_dfnet_t h);
/* HOST fp32 buffer; `name` is a state_dict key of the reference module: "encoder.<k>.weight|bias",
 * "adaptation_layers.adapt_layer_<i>.<0|2>.weight|bias", "...adapt_layer_<i>.3.<weight|bias|running_mean|
 * running_var>", "fc_pose.weight|bias". */
int dfn_dfnet_set_param(dfn_dfnet_t h, const char* name, const float* host, size_t numel);
/* Fold BatchNorm (eval mode, eps 1e-5) into the 5x5 convs, pack all convs into MFMA fragments, upload. */
int dfn_dfnet_commit(dfn_dfnet_t h);
size_t dfn_dfnet_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W);
/* DFNet.forward (dfnet.py:109-172).  x device [B,3,H,W] in [0,1].  features (if return_feature):
 * siamese == 0 (isSingleStream): [n_taps, B, 128, upH, upW]; siamese != 0: two stacks
 * [2][n_taps, B/2, 128, upH, upW] (target half first).  pose (if return_pose): [B, feat_dim].
 * With return_pose == 0 the encoder stops after the last tap (dfnet.py:133-136). */
int dfn_dfnet_forward(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int return_feature,
                      int siamese, int return_pose, int upH, int upW, float* features, float* pose,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The feature pyramid restricted to the levels a caller reads: DFNet.forward(return_feature=True, return_pose=False)
 * followed by the reference's torch.index_select(features, 0, args.feature_matching_lvl)
 * (feature/direct_feature_matching.py:354-357) — bit t of level_mask = level t is computed.  `features` has the layout of
 * dfn_dfnet_forward; the planes of levels outside the mask are NOT written, the encoder stops after the deepest level asked
 * for and the adaptation branches of the others do not run.  The planes that are written hold the same bits as
 * dfn_dfnet_forward's. */
int dfn_dfnet_forward_levels(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese, int level_mask,
                             int upH, int upW, float* features, void* workspace, size_t workspace_bytes, void* stream);

/* Input gradient of DFNet.forward's feature maps — what loss.backward() computes for the rendered image in the
 * DFNet_dm step (feature/direct_feature_matching.py:350-376; the feature extractor's weights are frozen there).
 * x device [B,3,H,W]; grad_features device, single-stream layout [n_taps, B, 128, upH, upW]; bit t of level_mask
 * says level t carries gradient (the planes of other levels are not read); grad_x [B,3,H,W].  Recomputes the
 * forward up to the deepest requested tap. */
size_t dfn_dfnet_backward_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W);
int dfn_dfnet_backward_input(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int upH, int upW,
                             const float* grad_features, int level_mask, float* grad_x, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Parameter gradients of the pose-regression path (feature/dfnet.py:121-170 with return_pose=True): what
 * loss.backward() leaves in the pose regressor's parameters in the DFNet_dm step
 * (feature/direct_feature_matching.py:372-374).  x device [B,3,H,W]; grad_pose device [B, feat_dim]; `grads` is a
 * HOST array of n_grads = 2 * 13 + 2 DEVICE pointers in state_dict order: encoder conv weight [co,ci,3,3] and bias
 * [co] for each of the 13 convs, then fc_pose.weight [feat_dim,512] and fc_pose.bias.  prec: DFN_PREC_F16X3 or
 * DFN_PREC_F32 (the forward recompute; gradient arithmetic is fp32).  BatchNorm / adaptation layers are not on
 * this path.  Recomputes the forward. */
size_t dfn_dfnet_backward_params_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W);
int dfn_dfnet_backward_params(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W,
                              const float* grad_pose, float* const* grads, int n_grads, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- training DFNet itself (run_feature.py:166-230): both heads, BatchNorm not folded.
 *
 * dfn_dfnet_forward_train: DFNet.forward while the module is being trained.  Outputs as dfn_dfnet_forward with
 * return_feature = 1.  The 5x5 adaptation convs run unfolded, followed by the BatchNorm2d:
 *   bn_batch = 1  train() mode: statistics of this batch (all B images, i.e. both streams of a siamese batch,
 *                 dfnet.py:131-143); bn_stats [n_taps][2][128] receives batch mean and BIASED batch variance per level
 *                 (the caller moves running_mean / running_var: momentum 0.1, unbiased variance)
 *   bn_batch = 0  --freezeBN (BatchNorm modules in eval(), utils.py:30-39): running statistics; bn_stats may be NULL.
 *                 Same values as dfn_dfnet_forward, but from the weights dfn_dfnet_refresh_train_params_device keeps
 *                 current.
 * keep = 0: streaming forward, workspace of dfn_dfnet_workspace_bytes.  keep = 1: the workspace (sized by
 * dfn_dfnet_backward_params_workspace_bytes) keeps every activation for dfn_dfnet_backward_all_params(have_forward =
 * 1) on the same handle, which then skips its forward recompute; the caller must leave the workspace untouched in
 * between and commit / refresh no weights.  With keep = 1, features may be NULL: the pose path only, for
 * dfn_dfnet_backward_all_params(grad_features = NULL, have_forward = 1, n_grads = 2 * 13 + 2).  Precision F32 or
 * F16X3. */
int dfn_dfnet_forward_train(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese,
                            int return_pose, int bn_batch, int keep, int upH, int upW, float* features,
                            float* pose, float* bn_stats, void* workspace, size_t workspace_bytes, void* stream);

/* Parameter gradients of both heads: grad_pose [B, feat_dim] (may be NULL), grad_features single-stream layout
 * [n_taps, B, 128, upH, upW] with level_mask as in dfn_dfnet_backward_input.  The pointers of
 * dfn_dfnet_backward_params are followed, per pyramid level, by adapt_layer_<t>.0.weight [64,C,1,1], .0.bias,
 * .2.weight [128,64,5,5], .2.bias and — bn_batch != 0 only — .3.weight, .3.bias [128]:
 *   bn_batch = 0  BatchNorm frozen on its running statistics (--freezeBN); n_grads = 2 * 13 + 2 + 4 * n_taps
 *   bn_batch = 1  batch statistics, the backward of dfn_dfnet_forward_train;  n_grads = 2 * 13 + 2 + 6 * n_taps
 * Gradients of levels outside level_mask are left untouched (zero them beforehand).  Workspace:
 * dfn_dfnet_backward_params_workspace_bytes; have_forward = 1: it is the workspace a matching
 * dfn_dfnet_forward_train(keep = 1) filled (DFN_ERR_STATE otherwise) and no forward is recomputed. */
int dfn_dfnet_backward_all_params(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W,
                                  const float* grad_pose, const float* grad_features, int upH, int upW,
                                  int level_mask, int bn_batch, int have_forward, float* const* grads, int n_grads,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- dataset front-end, per frame (dataset_loaders/seven_scenes.py:324-352).
 *
 * rgb_hwc: the decoded 8-bit frame [h, w, 3] in device memory.  Outputs: img_chw fp32 [3, H, W] in [0, 1] =
 * cv2.resize(img / 255, (W, H), INTER_AREA) followed by ToTensor (H <= h, W <= w; coverage-weighted box mean — exact
 * integer block sums when h % H == 0 and w % W == 0), and hist fp32 [hist_bins] = round(histc(Y, hist_bins, 0, 1) /
 * sum * 100) with Y = 0.299 R + 0.587 G + 0.114 B of the resized image: NeRF-H's histogram index vector.  scratch:
 * dfn_frame_prep_scratch_bytes() of device memory (zeroed by the call). */
size_t dfn_frame_prep_scratch_bytes(void);
int dfn_frame_prep(const uint8_t* rgb_hwc, int h, int w, int H, int W, int hist_bins, float* img_chw, float* hist,
                   void* scratch, void* stream);

/* ---- render_path's per-frame back-end (models/rendering.py:423-452 with to8b of models/nerf.py:11).
 *
 * A batch of n_frames rendered frames, device fp32: rgb [n, H, W, 3], disp [n, H, W]; gt (optional) [n, H, W, 3] when
 * gt_per_frame != 0, ONE frame [H, W, 3] compared with every render otherwise (the reference's single_gt_img).  Outputs, device:
 *   rgb8  [n, H, W, 3] uint8 = (255 * clip(rgb, 0, 1)).astype(uint8)            (truncation, as numpy)
 *   disp8 [n, H, W]    uint8 = the same of disp / max(disp), the maximum per frame
 *   gt8   (optional)   uint8 of gt: [n, H, W, 3], or [H, W, 3] for a single ground-truth frame
 *   mse   (optional, needs gt) fp32 [n] = mean((rgb - gt)^2) per frame (fp32 differences, fp64 sum): PSNR = -10 log10(mse)
 *   disp_max (optional) fp32 [n]
 * scratch: dfn_frame_post_scratch_bytes(n_frames) of device memory (zeroed by the call). */
size_t dfn_frame_post_scratch_bytes(int n_frames);
int dfn_frame_post(const float* rgb, const float* disp, const float* gt, int gt_per_frame, int n_frames, int H, int W,
                   uint8_t* rgb8, uint8_t* disp8, uint8_t* gt8, float* mse, float* disp_max, void* scratch, void* stream);

/* ---- the triplet loss of DFNet's training on the two feature stacks (feature/misc.py:355-435), fused.
 *
 * f1 (anchor, the rendered stream in run_feature.py:155) and f2 (positive, the target stream): fp32 device stacks
 * [L][B][rows][W] with rows = C * H; level l of a stack starts at base + l * level_stride (floats), images inside a
 * level are contiguous — so the two halves of one siamese [L, 2B, C, H, W] tensor are addressed in place.  Negatives
 * are the stacks rolled by one image (torch.roll(f, 1, dims=1)).  nn.TripletMarginLoss(margin, p = 2, eps = 1e-6,
 * reduction = 'mean'): pairwise distance over the LAST axis W, mean over the L * B * rows rows.
 *   mining 0  triplet_loss                            (f1, f2, roll f2)
 *   mining 1  triplet_loss_hard_negative_mining       anchor swap by two full-tensor MSEs
 *   mining 2  triplet_loss_hard_negative_mining_plus  the closest of four pairs decides (first minimum, torch.argmin)
 * loss: device float[1].  state (dfn_triplet_loss_state_bytes, device, caller-owned) carries the chosen case, the four
 * MSEs (floats at byte offset 16) and two distances per row from the forward to the backward.
 * Backward: grad_loss device float[1]; grad_f1 / grad_f2 are fully written, addressed like the inputs. */
size_t dfn_triplet_loss_state_bytes(int L, int B, int rows);
int dfn_triplet_loss_forward(const float* f1, size_t level_stride1, const float* f2, size_t level_stride2, int L,
                             int B, int rows, int W, float margin, int mining, float* loss, void* state,
                             size_t state_bytes, void* stream);
int dfn_triplet_loss_backward(const float* f1, size_t level_stride1, const float* f2, size_t level_stride2, int L,
                              int B, int rows, int W, const float* grad_loss, const void* state, float* grad_f1,
                              size_t grad_stride1, float* grad_f2, size_t grad_stride2, void* stream);

/* ---- the same triplet loss WITHOUT the enlarged feature stacks ("reductions worth fusing into the upsample";
 * feature/dfnet.py:142-160 + feature/misc.py:355-435).  Bilinear enlargement (align_corners) is linear and separable, so a row's
 * distances over the W output columns are closed forms of the LOW-RESOLUTION adapted maps (dfnet_triplet_pyr.hip); the training
 * step then never writes the [L, B, 128, upH, upW] stacks, their gradients, the upsample or its adjoint.
 *
 * dfn_dfnet_forward_train_pyramid: dfn_dfnet_forward_train(siamese = 1, keep = 1) that keeps every level's adapted map (before
 * BatchNorm's affine, with the level's BatchNorm work block) in the workspace (dfn_dfnet_backward_params_workspace_bytes) and
 * enlarges nothing; pose [B, feat_dim] and bn_stats as there.
 * dfn_dfnet_triplet_pyramid_forward: the loss of mining mode `mining` between the two halves of that batch as enlarged to
 * [upH, upW]; f1_half = which half is misc.py's f1 / anchor (run_feature.py:154 passes (features_rgb, features_target): the SECOND
 * half of cat([target, rgb]) -> 1).  loss: device float[1]; state: dfn_dfnet_triplet_pyramid_state_bytes of device memory.
 * dfn_dfnet_backward_all_params_triplet: dfn_dfnet_backward_all_params(have_forward = 1) whose feature gradient is grad_loss
 * (device float[1] = d L / d triplet loss) times the gradient of that loss, taken from `state`; every level carries gradient
 * (n_grads as there for all levels).  Same handle, same workspace, no weight refresh in between (DFN_ERR_STATE otherwise).
 * feature_images (even, 2 <= feature_images <= B): the LEADING frames of the batch are the siamese pair [stream 0 | stream 1] that
 * goes through the adaptation layers, BatchNorm statistics and the loss; frames beyond them run the encoder and the pose head only —
 * run_feature.py:211-222's second forward `feat_model(rgb_perturb, False)` on the synthesised views joins the siamese forward as ONE
 * encoder pass (B = 3 x featurenet_batch_size, feature_images = 2 x it; the encoder has no batch-coupled layer, so every number is the
 * two-pass one).  feature_images < B on precision F16X3 only.  pose [B, feat_dim] holds all frames. */
int dfn_dfnet_forward_train_pyramid(dfn_dfnet_t h, int prec, const float* x, int B, int feature_images, int H, int W,
                                    int return_pose, int bn_batch, float* pose, float* bn_stats, void* workspace,
                                    size_t workspace_bytes, void* stream);
size_t dfn_dfnet_triplet_pyramid_state_bytes(dfn_dfnet_t h, int feature_images, int upH);
int dfn_dfnet_triplet_pyramid_forward(dfn_dfnet_t h, int prec, int B, int feature_images, int H, int W, int upH, int upW,
                                      int f1_half, float margin, int mining, float* loss, void* state, size_t state_bytes,
                                      void* workspace, size_t workspace_bytes, void* stream);
int dfn_dfnet_backward_all_params_triplet(dfn_dfnet_t h, int prec, const float* x, int B, int feature_images, int H, int W,
                                          const float* grad_pose, const float* grad_loss, const void* state, size_t state_bytes,
                                          int f1_half, int upH, int upW, int bn_batch, float* const* grads, int n_grads,
                                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- the cosine feature loss of the DFNet_dm step (feature/direct_feature_matching.py:114-136 feature_loss with per_channel =
 * False, applied per image at :352-358 and averaged over the batch), fused for the whole mini-batch.
 * fr (features of the rendered images, differentiated) and ft (features of the target images): fp32 device stacks
 * [L][B][C][HW]; level l starts at base + l * level_stride (floats).  levels: HOST array of the n_levels (<= 8) pyramid levels the
 * loss reads (args.feature_matching_lvl).  nn.CosineSimilarity(dim=1, eps=1e-6) on [C', H*W] is one cosine per feature ROW over
 * its H*W pixels; loss = 1 - mean over every (level, image, channel) row = mean_b feature_loss(f_r[b], f_t[b]).  loss: device
 * float[1]; state: dfn_feature_cosine_state_bytes() of device memory carrying the row statistics to the backward.
 * Backward: grad_loss device float[1]; grad_fr is addressed like fr (grad_stride) and written at the selected levels only. */
size_t dfn_feature_cosine_state_bytes(int n_levels, int B, int C);
int dfn_feature_cosine_forward(const float* fr, size_t level_stride_r, const float* ft, size_t level_stride_t, const int* levels,
                               int n_levels, int B, int C, size_t HW, float* loss, void* state, size_t state_bytes, void* stream);
int dfn_feature_cosine_backward(const float* fr, size_t level_stride_r, const float* ft, size_t level_stride_t, const int* levels,
                                int n_levels, int B, int C, size_t HW, const float* grad_loss, const void* state, float* grad_fr,
                                size_t grad_stride, void* stream);

/* The rest of the DFNet_dm loss block (feature/direct_feature_matching.py:359-370): photo_loss = mean((rgb - target)^2) over the n
 * elements of the [B,3,H,W] frames, pose_loss = mse_loss(pose, pose_gt) over n_pose = 12 B floats, loss = w_pose pose_loss +
 * w_photo photo_loss + w_feat feat_loss (args.combine_loss_w; feat_loss: the device scalar dfn_feature_cosine_forward wrote, NULL = 0).
 * out4 (device) = loss, photo_loss, pose_loss, feat_loss.  Backward: grad_rgb [n], grad_pose [n_pose], grad_feat [1] (may be NULL)
 * from the device scalar grad_loss.  Sums in fp64 over fixed chunks (deterministic).  scratch: dfn_dm_loss_scratch_bytes() device bytes. */
size_t dfn_dm_loss_scratch_bytes(void);
int dfn_dm_loss_forward(const float* rgb, const float* target, size_t n, const float* pose, const float* pose_gt, int n_pose,
                        const float* feat_loss, float w_pose, float w_photo, float w_feat, float* out4, void* scratch, void* stream);
int dfn_dm_loss_backward(const float* rgb, const float* target, size_t n, const float* pose, const float* pose_gt, int n_pose,
                         float w_pose, float w_photo, float w_feat, const float* grad_loss, float* grad_rgb, float* grad_pose,
                         float* grad_feat, void* stream);

/* The pose regressor's rotation re-orthogonalisation (feature/direct_feature_matching.py:85-92, feature/misc.py:68-72:
 * `u, s, v = torch.svd(pose[:, :3, :3]); pose[:, :3, :3] = u @ v^T`) and its adjoint, closed form: U V^T is the orthogonal polar
 * factor of the 3x3 block (scaled Newton iteration in fp64, one thread per pose) — no SVD / GEMM library call in the DFNet_dm step.
 * pose_in / pose_out / grad_out / grad_in: device [B,3,4] fp32 (row-major [B,12]); the translation column is copied / passed
 * through.  A singular rotation block (the factor is not unique) yields NaN, as torch.svd's gradient does. */
int dfn_pose_orthogonalize(const float* pose_in, int B, float* pose_out, void* stream);
int dfn_pose_orthogonalize_backward(const float* pose_in, const float* grad_out, int B, float* grad_in, void* stream);

/* Weight and bias gradient of ONE stride-1 "same" convolution from torch-shaped fp32 device tensors, for parity tests against
 * autograd of torch.nn.functional.conv2d — the kernel behind the conv parameter gradients of dfn_dfnet_backward_params /
 * dfn_dfnet_backward_all_params (what loss.backward() leaves in the Conv2d parameters of feature/dfnet.py:8-40, :57-62 under
 * feature/direct_feature_matching.py:372-374 and run_feature.py:166-230):
 *   dW[co][ci][ky][kx] = sum_{b,y,x} grad_out[b,co,y,x] * input[b,ci,y+ky-ks/2,x+kx-ks/2],   db[co] = sum grad_out[b,co,y,x]
 * grad_out [B,cout,H,W], input [B,cin,H,W], dW [cout,cin,ks,ks], db [cout] (may be NULL).  cout, cin multiples of 64, ks 1, 3 or 5.
 * Split-f16 products (fp32-grade): operands split once into hi | lo f16 planes, three f16 MFMAs per product, fp32 accumulation,
 * chunk partials summed in a fixed order (bit-identical reruns). */
size_t dfn_conv_wgrad_scratch_bytes(int B, int H, int W, int cout, int cin, int ks);
int dfn_conv_wgrad(const float* grad_out, const float* input, int B, int H, int W, int cout, int cin, int ks, float* dW,
                   float* db, void* scratch, size_t scratch_bytes, void* stream);

/* After an optimizer step of DFNet's own training: re-pack encoder, fc_pose, the adaptation convs (unfolded) and the
 * BatchNorm tensors from DEVICE tensors — 2 * 13 + 2 + 8 * n_taps pointers: those of dfn_dfnet_backward_params, then
 * per level .0.weight, .0.bias, .2.weight, .2.bias, .3.weight, .3.bias, .3.running_mean, .3.running_var.  The
 * BatchNorm-FOLDED inference weights used by dfn_dfnet_forward are
 * NOT touched: re-commit from the host (dfn_dfnet_set_param + dfn_dfnet_commit) before evaluating.
 * prec_mask: bit p set = renew the fragments of precision p (DFN_PREC_*); a training loop runs in one precision and
 * has no use for the other two.  Precisions left out are STALE afterwards: calls in them return DFN_ERR_STATE until a
 * re-pack with their bit or a commit. */
int dfn_dfnet_refresh_train_params_device(dfn_dfnet_t h, const float* const* params, int n_params, int prec_mask,
                                          void* stream);

/* After an optimizer step: re-pack the pose path's parameters (13 encoder convs + fc_pose, forward and data-gradient
 * fragments) from DEVICE fp32 master copies, without the host round trip of set_param + commit.  `params`: HOST
 * array of DEVICE pointers in the order of dfn_dfnet_backward_params.  Adaptation layers are untouched.  prec_mask as
 * for dfn_dfnet_refresh_train_params_device. */
int dfn_dfnet_refresh_pose_params_device(dfn_dfnet_t h, const float* const* params, int n_params, int prec_mask,
                                         void* stream);

/* ------------------------------------------------------------------ NeRF-H training path
 * One optimisation step of run_nerf.py:50-66 — render(**render_kwargs_train) -> NerfWLoss -> loss.backward() — as
 * three calls: dfn_nerfh_train_forward, dfn_nerfw_loss, dfn_nerfh_train_backward.  The parameters are read from the
 * caller's fp32 device tensors (torch's master weights, row-major [out, in]) and the gradients are written to the caller's
 * gradient tensors, so an optimizer step needs no host round trip.  Two implementations behind the same calls:
 *   DFN_TRAIN_FUSED (default at netwidth 128): both networks as register-resident chains on split-f16 MFMA products
 *     (fp32-grade), forward and data-gradient; only the layer inputs / pre-activation gradients the WEIGHT gradients need
 *     are stored (as MFMA operands) and streamed once through the weight-gradient kernel; the step's weights are
 *     re-packed on the device at the start of the forward (csrc/nerfh_fused_*.hip).  The coarse network's stored operands are
 *     hi | lo f16 planes (4 bytes per element), the fine network's ONE f16 plane (2 bytes: half the step's HBM traffic; its weight
 *     gradients are sums over ~3e5 points whose f16 rounding errors average out — within 3e-5 of two planes at random-init weights,
 *     3e-4 on trained-like weights, csrc/nerfh_fused_train.h);
 *   DFN_TRAIN_FUSED_SPLIT: the same with hi | lo planes for BOTH networks (round-4 layout, 4.67 GB per 1 536-ray step);
 *   DFN_TRAIN_EXACT (any even netwidth): layer by layer on exact-fp32 MFMA products over activations kept in the workspace.
 *
 * `params` / `grads`: HOST arrays of dfn_nerfh_train_param_count() DEVICE pointers in the order of
 * dfn_nerfh_train_param_name(i): "coarse.<key>" (24: xyz_encoding_1..8.0, xyz_encoding_final, dir_encoding.0,
 * static_sigma.0, static_rgb.0 — weight then bias), "fine.<key>" (38: the same + transient_encoding.0/2/4/6,
 * transient_sigma.0, transient_rgb.0, transient_beta.0), "embedding_a.weight", "embedding_t.weight"
 * (state_dict order of models/nerfw.py:259-295). */
int dfn_nerfh_train_param_count(void);
const char* dfn_nerfh_train_param_name(int i);
size_t dfn_nerfh_train_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni);
/* Selects the implementation of the training step for this handle (the workspace size above covers all of them).  A step whose
 * split-f16 operands left their range (weights that outgrew the committed scale) leaves ZEROS in every gradient tensor and raises
 * the range flag (dfn_nerfh_range_status / _async): a skipped step, never clamped gradients. */
enum { DFN_TRAIN_FUSED = 0, DFN_TRAIN_EXACT = 1, DFN_TRAIN_FUSED_SPLIT = 2 };
int dfn_nerfh_set_train_mode(dfn_nerfh_t h, int mode);
/* Host-only consistency check of the fused step's tables for a netwidth-128 geometry (no device work): every parameter element
 * the chain kernels read is packed exactly once per pass and every gradient element is written by exactly one accumulator of the
 * weight-gradient stream (tests/test_host_logic.py).  DFN_OK or DFN_ERR_STATE with the first inconsistency in dfn_last_error(). */
int dfn_nerfh_train_tables_selfcheck(const dfn_nerfh_desc* desc);

/* models/rendering.py:245-337 render_rays with test_time=False (white_bkgd=False; lindisp per the handle's options) on caller rays
 * (run_nerf.py:50).  The reference's three random draws are INPUTS: t_rand [n_rays, Nc] = torch.rand (stratified
 * jitter, rendering.py:277-285; NULL = perturb 0), noise [n_rays, Nc] = torch.randn (x raw_noise_std, coarse alpha,
 * rendering.py:173; NULL = none), u [n_rays, Ni] = torch.rand (sample_pdf, rendering.py:35; NULL = linspace).
 * Outputs: rgb [n,3], disp, acc [n], raw [n, Nc+Ni, 9] (`retraw`; transient_sigmas = raw[..., 7]) and the training
 * extras rgb0 [n,3], disp0, acc0, z_std, beta [n] (rendering.py:323-329).  The workspace keeps every activation for
 * dfn_nerfh_train_backward. */
int dfn_nerfh_train_forward(dfn_nerfh_t h, const float* const* params, const float* rays_o, const float* rays_d,
                            const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near, float far,
                            const float* t_rand, const float* noise, float raw_noise_std, const float* u, float* rgb,
                            float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                            float* beta, void* workspace, size_t workspace_bytes, void* stream);

/* models/losses.py:19-57 NerfWLoss(coef, lambda_u) on (rgb_fine, rgb_coarse, beta, transient_sigmas = raw[..., 7])
 * vs target [n,3].  loss5: DEVICE buffer of DFN_NERFW_LOSS_FLOATS floats, 16-byte aligned; [0..4] = c_l, f_l, b_l, s_l and the PSNR
 * of rgb (run_nerf.py:62-64), the rest is scratch of the reduction.
 * Gradients of sum(loss): g_rgb [n,3], g_rgb0 [n,3], g_beta [n]; d / d transient_sigma is the constant
 * coef * lambda_u / (n * Nf) per sample (pass it to dfn_nerfh_train_backward as g_tsigma). */
enum { DFN_NERFW_LOSS_FLOATS = 160 };
int dfn_nerfw_loss(const float* rgb, const float* rgb0, const float* beta, const float* raw, const float* target,
                   size_t n_rays, int Nf, float coef, float lambda_u, float* loss5, float* g_rgb, float* g_rgb0,
                   float* g_beta, void* stream);

/* loss.backward() of the step (run_nerf.py:65): from d L / d rgb, d L / d rgb0, d L / d beta and the constant d L / d
 * transient_sigma to the gradient of EVERY parameter of both networks and both embedding tables (z_samples.detach(),
 * rendering.py:302, cuts the sampler).  Must follow dfn_nerfh_train_forward on the same workspace, rays and draws;
 * `raw` is that call's output.  d L / d transient_sigma = g_tsigma (constant) + g_tsigma_dense [n, Nf] (may be NULL).
 * Gradients are overwritten (not accumulated). */
int dfn_nerfh_train_backward(dfn_nerfh_t h, const float* const* params, const float* hist, size_t hist_rows, size_t n_rays,
                             int Nc, int Ni, const float* noise, float raw_noise_std, const float* raw, const float* g_rgb,
                             const float* g_rgb0, const float* g_beta, float g_tsigma, const float* g_tsigma_dense,
                             float* const* grads, void* workspace, size_t workspace_bytes, void* stream);

/* d L / d rays of the TRAINING render (the reference's render(test_time=False) is differentiable w.r.t. its rays / pose under autograd,
 * models/rendering.py:245-337: pts = o + d z enter both networks, the view direction the two dir_encodings; z carries no gradient —
 * near / far bounds, z_samples.detach()): same output gradients as dfn_nerfh_train_backward, the data-gradient chains of BOTH
 * networks continued through the encodings' Jacobians and reduced per ray.  Layer by layer on the exact-fp32 products: the forward
 * must have run in DFN_TRAIN_EXACT mode on the same workspace (DFN_ERR_STATE otherwise).  grad_rays_o / grad_rays_d [n,3] are
 * overwritten; scratch: dfn_nerfh_train_backward_rays_scratch_bytes. */
size_t dfn_nerfh_train_backward_rays_scratch_bytes(size_t n_rays, int Nc, int Ni);
int dfn_nerfh_train_backward_rays(dfn_nerfh_t h, const float* const* params, const float* rays_o, const float* rays_d, const float* hist,
                                  size_t hist_rows, size_t n_rays, int Nc, int Ni, const float* noise, float raw_noise_std,
                                  const float* raw, const float* g_rgb, const float* g_rgb0, const float* g_beta, float g_tsigma,
                                  const float* g_tsigma_dense, float* grad_rays_o, float* grad_rays_d, void* workspace,
                                  size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* Test-time render_rays for ANY netwidth on the same layer-by-layer exact-fp32 path (the register-resident kernels
 * behind dfn_render_rays exist for netwidth 128 and 256): models/rendering.py:245-337 with test_time=True, from
 * the handle's committed parameters.  raw [n_rays, Nc+Ni, 9] is required (output and scratch). */
size_t dfn_nerfh_generic_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni);
int dfn_nerfh_generic_render_rays(dfn_nerfh_t h, const float* rays_o, const float* rays_d, const float* hist, size_t hist_rows,
                                  size_t n_rays, int Nc, int Ni, float near, float far, float* rgb, float* disp, float* acc,
                                  float* raw, void* workspace, size_t workspace_bytes, void* stream);
/* The gradient of that render for ANY even netwidth (what dfn_render_rays_backward is for netwidth 128): d L / d rays_o,
 * d L / d rays_d [n_rays, 3] from grad_rgb [n_rays, 3]; viewdirs == NULL: they are d/|d| and the normalisation is differentiated
 * into grad_rays_d, otherwise grad_viewdirs (optional) receives their gradient.  Exact fp32; recomputes the forward keeping the
 * fine activations (workspace: dfn_nerfh_generic_backward_workspace_bytes).  Replaces loss.backward() through
 * render(c2w = pose) (feature/direct_feature_matching.py:340-376) when --netwidth is not 128. */
size_t dfn_nerfh_generic_backward_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni);
int dfn_nerfh_generic_render_rays_backward(dfn_nerfh_t h, const float* rays_o, const float* rays_d, const float* viewdirs,
                                           const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near,
                                           float far, const float* grad_rgb, float* grad_rays_o, float* grad_rays_d,
                                           float* grad_viewdirs, void* workspace, size_t workspace_bytes, void* stream);

/* The three fp32-MFMA products of the training path, for parity tests (torch.nn.functional.linear and its autograd):
 *   y[p, n]  = act(sum_k x[p / x_row_div, k] w[n, wcol + k] + b[n])   act: 0 none, 1 ReLU, 2 Sigmoid, 3 Softplus
 *   dx[p, k] = sum_n g[p, n] w[n, wcol + k]   (+ dx if accumulate; zeroed where relu_src[p, k] <= 0 if relu_src)
 *   dw[n, wcol + k] = sum_p g[p, n] x[p / x_row_div, k];  db[n] = sum_p g[p, n] (db may be NULL) */
int dfn_linear_forward(const float* x, int ldx, int K, const float* w, int ldw, int wcol, const float* b, int N, int act,
                       float* y, int ldy, size_t n_points, int x_row_div, void* stream);
int dfn_linear_backward_input(const float* g, int ldg, int N, const float* w, int ldw, int wcol, int K, float* dx, int lddx,
                              int accumulate, const float* relu_src, int ld_relu, size_t n_points, void* stream);
size_t dfn_linear_backward_weight_scratch_bytes(int N, int K, size_t n_points);
int dfn_linear_backward_weight(const float* g, int ldg, int N, const float* x, int ldx, int K, int x_row_div, float* dw, int ldw,
                               int wcol, float* db, void* scratch, size_t n_points, void* stream);

/* optimizer.step() of the training loops (torch.optim.Adam in the reference: run_nerf.py:65, run_feature.py:65,
 * feature/direct_feature_matching.py:237) as one multi-tensor pass: for every tensor, element by element and in the order of
 * torch/optim/adam.py's multi-tensor step,
 *   g = grad (+ weight_decay * param);  exp_avg += (1 - beta1) (g - exp_avg);  exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g g;
 *   param += step_size * exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps)
 * with the per-tensor step_size = -lr / (1 - beta1^step) and bias_correction2_sqrt = sqrt(1 - beta2^step) computed by the caller
 * (in double, as torch does); the hyper-parameters are the doubles torch holds, rounded to fp32 once inside.  `tensors` is a HOST array of n_tensors descriptors of fp32 device tensors; nothing is retained. */
typedef struct dfn_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  size_t numel;
  float step_size;
  float bias_correction2_sqrt;
} dfn_adam_tensor;
int dfn_adam_step(const dfn_adam_tensor* tensors, int n_tensors, double beta1, double beta2, double eps, double weight_decay,
                  void* stream);

/* Timing aid for bench.py: average device time in ms of the `which` kernel of the render path
 * (DFN_PROF_*) over the launches since the last reset, measured with HIP events recorded on the
 * launch `stream` around each launch.  Enabled by dfn_profile_enable(1); costs a sync when read. */
enum {
  DFN_PROF_COARSE = 0,      /* nerfh_coarse_kernel */
  DFN_PROF_FINE = 1,        /* nerfh_fine_kernel */
  DFN_PROF_SAMPLE_FINE = 2, /* sample_fine_kernel (coarse weights + sample_pdf + merge) */
  DFN_PROF_RAY_BIAS = 3,    /* ray_bias_kernel */
  DFN_PROF_COMBINE = 4,     /* composite_combine_kernel (fused-compositing path) */
  DFN_PROF_COMPOSITE = 5,   /* composite_fine_kernel (raw path) */
  DFN_PROF_SLOTS = 6
};
int dfn_profile_enable(int on);
int dfn_profile_read(int which, double* avg_ms, int* launches);


#ifdef __cplusplus
}
#endif
#endif /* DFNET_HIP_H */
