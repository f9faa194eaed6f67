// nerfh_train.h — launch interface of the NeRF-H TRAINING path (nerfh_train.hip -> nerfh_train_api.hip).
//
// The optimisation step of the reference (script/run_nerf.py:32-80: render(**render_kwargs_train) -> NerfWLoss ->
// loss.backward()) needs what the test-time kernels never materialise: every layer's activation (for the weight
// gradients) and gradients with respect to every parameter.  The training path therefore runs the two networks
// layer by layer over point-major fp32 activations in HBM with three generic fp32-MFMA products
// (v_mfma_f32_32x32x2_f32; exact fp32, the reference's arithmetic):
//     forward   Y[p, n]  = act( sum_seg X_seg[p / div, k] W[n, col_seg + k] + b[n] )
//     backward  dX[p, k] = sum_n G[p, n] W[n, col + k]          (+ accumulate, x ReLU mask of the layer input)
//     wgrad     dW[n, col + k] = sum_p G[p, n] X[p / div, k],   db[n] = sum_p G[p, n]
// reading the torch parameter tensors IN PLACE (row-major [out, in], nothing is packed, so an optimizer step needs no
// re-pack).  `div` broadcasts a per-ray segment (direction encoding + appearance embedding, transient embedding) over
// the samples of a ray.  Widths are run-time values: any netwidth works here.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dfn {
namespace train {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SOFTPLUS = 3 };

struct Seg {        // one input segment of a Linear
  const float* x;   // [P / div, ld]
  int ld;           // row stride of x (floats)
  int K;            // features taken from x (columns 0..K)
  int div;          // row of x = point / div (1: per point; N_samples: per ray)
  int wcol;         // first weight column this segment multiplies
};

// Y = act(sum_seg X_seg W[:, wcol:wcol+K]^T + b); y[p * ldy + n], n < N.
hipError_t gemm_fwd(const Seg* segs, int nseg, const float* W, int ldw, const float* b, int N, int act, float* y, int ldy,
                    long long P, hipStream_t s);
// dX[p, k] (k < K) = sum_{n<N} G[p * ldg + n] W[n * ldw + wcol + k]; `accumulate`: add to what dx holds;
// mask_src != nullptr: zero where mask_src[p * ldmask + k] <= 0 (ReLU gate of the layer whose OUTPUT is this dX's tensor).
hipError_t gemm_bwd(const float* G, int ldg, int N, const float* W, int ldw, int wcol, int K, float* dx, int lddx,
                    int accumulate, const float* mask_src, int ldmask, long long P, hipStream_t s);
// dW[n * ldw + seg.wcol + k] = sum_p G[p * ldg + n] X[(p / div) * ld + k]; db[n] = sum_p G[p, n] when db != nullptr.
// Deterministic: fixed point chunks -> partial sums in `scratch` -> fixed-order reduction.
size_t gemm_wgrad_scratch_floats(int N, int K, long long P);
hipError_t gemm_wgrad(const float* G, int ldg, int N, const Seg& xseg, float* dW, int ldw, float* db, float* scratch,
                      long long P, hipStream_t s);
// Up to three small products (P below the tiled form's threshold) in ONE launch + one reduction launch; `scratch` holds the
// products' partials back to back: sum of gemm_wgrad_scratch_floats(N, K, P) floats.  Same arithmetic and summation order as gemm_wgrad.
struct WgradJob { const float* G; int ldg, N; Seg x; float* dW; int ldw; float* db; };
bool gemm_wgrad_multi_ok(const WgradJob* jobs, int n, long long P);   // every product takes the small (untiled) form at this P
hipError_t gemm_wgrad_multi(const WgradJob* jobs, int n, float* scratch, long long P, hipStream_t s);

// ---- stages
// z[r, i] = lower + (upper - lower) * t_rand (rendering.py:277-285); t_rand == nullptr: the plain linspace depths; lindisp: the
// depths are linear in disparity (rendering.py:272-273).
hipError_t stratified_z(const float* t_rand, size_t R, int Nc, float near, float far, float* z, hipStream_t s, int lindisp = 0);
// pe[p, 0..63) = positional encoding (L = 10) of o + d z; column 63 zero.  [P][64].
hipError_t posenc_points(const float* rays_o, const float* rays_d, const float* z, size_t R, int Ns, float* pe, hipStream_t s);
// per-ray inputs: dir_in[r][0..27) = pe_dir(viewdir) (L = 4), then a = embedding_a[hist] (hist_bin*dim_a) when emb_a != nullptr;
// t_in[r] = embedding_t[hist] (hist_bin*dim_t) when emb_t != nullptr.  Row strides ld_dir / ld_t; padding zeroed.
// viewdirs + both networks' per-ray input rows + stratified coarse depths + the step's range word, one launch (nerfh_train.hip)
struct TrainRayPrepArgs {
  const float* rays_d; size_t R;
  const float* hist; size_t hist_rows; const float* emb_a; const float* emb_t; int hist_bin, dim_a, dim_t, n_vocab;
  float* view; float* dir_c; int ld_dc; float* dir_f; int ld_df; float* t_in; int ld_t;
  const float* t_rand; int Nc; float near, far; int lindisp; float* z;
  int* range_word;
};
hipError_t train_ray_prep(const TrainRayPrepArgs& a, hipStream_t s);
hipError_t ray_inputs(const float* viewdirs, const float* hist, size_t hist_rows, const float* emb_a, const float* emb_t,
                      int hist_bin, int dim_a, int dim_t, int n_vocab, size_t R, float* dir_in, int ld_dir, float* t_in, int ld_t,
                      hipStream_t s);
// Coarse training composite + importance sampling + merge (rendering.py:295-304 with raw2outputs_NeRFW typ="coarse",
// test_time=False): raw_c [R,Nc,4] = (rgb, sigma), z_c [R,Nc]; noise (nullable) x noise_std added to sigma before the
// relu; u (nullable -> linspace) [R,Ni].  Outputs z_fine [R,Nc+Ni], rgb0 [R,3], disp0, acc0, z_std [R].
hipError_t sample_fine_train(const float* raw_c, const float* z_c, const float* noise, float noise_std, const float* u, size_t R,
                             int Nc, int Ni, float* z_fine, float* rgb0, float* disp0, float* acc0, float* z_std, hipStream_t s);
// d L / d (pre-activation coarse outputs) [R,Nc,4] from d L / d rgb0 [R,3].
hipError_t composite_coarse_backward(const float* raw_c, const float* z_c, const float* noise, float noise_std, const float* g_rgb0,
                                     size_t R, int Nc, float* gpre, hipStream_t s, float* zero0 = nullptr, size_t n0 = 0,
                                     float* zero1 = nullptr, size_t n1 = 0);
// d L / d (pre-activation fine outputs) [R,Nf,9] from d L / d rgb [R,3], d L / d beta [R] and a constant d L / d
// transient_sigma per sample (training compositing: joint rgb, beta = sum w_t beta_t + beta_min).
hipError_t composite_fine_backward_train(const float* raw, const float* z, const float* g_rgb, const float* g_beta, float g_tsigma,
                                         const float* g_ts /* nullable dense [R,Nf] */, size_t R, int Nf, float* gpre, hipStream_t s);
// out[r, 0..C) = sum_s g[(r * Ns + s) * ld + c]
hipError_t sum_over_samples(const float* g, int ld, int C, size_t R, int Ns, float* out, int ldo, hipStream_t s);
// embedding-table gradients: grad_emb[idx(hist[r, b]), j] += g_in[r * ld + off + b * dim + j] (atomic fp32 adds).
hipError_t embedding_scatter(const float* g_in, int ld, int off, const float* hist, size_t hist_rows, int hist_bin, int dim,
                             int n_vocab, size_t R, float* grad_emb, hipStream_t s);
// Input-gradient pieces of the generic-width render gradient (nerfh_train_api.hip: dfn_nerfh_generic_render_rays_backward).
// g [P,9] (d L / d raw, the network's OUTPUTS) -> d L / d pre-activation in place: x y (1 - y) for the Sigmoid heads (columns
// 0-2, 4-6), x (1 - exp(-y)) for the Softplus heads (3, 7, 8); y = raw [P,9].
hipError_t head_prime(const float* raw, float* g, size_t P, hipStream_t s);
// Jacobians of the two positional encodings (models/nerfw.py:105-133): g_pe [P,64] (63 used) at x = o + d z and g_dpe [P,ld_d]
// (27 used) at the ray's viewdir -> gpts [P,6] = d L / d point (3), d L / d viewdir through this sample (3).
hipError_t posenc_backward(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z, const float* g_pe,
                           const float* g_dpe, int ld_d, size_t R, int Ns, float* gpts, hipStream_t s);
constexpr int kNerfwLossFloats = 160;   // loss buffer: 5 results + scratch for the per-block partial sums
// NerfWLoss (models/losses.py:19-57) forward + gradient.  tsigma = raw + 7 with stride 9.  loss[0..4) = c_l, f_l, b_l,
// s_l; loss[4] = psnr of rgb (run_nerf.py:62-64).  Gradients of sum(loss) * 1: g_rgb [R,3], g_rgb0 [R,3], g_beta [R];
// the transient-sigma gradient is the constant coef * lambda_u / (R * Nf) (returned through g_tsigma_const, host float).
hipError_t nerfw_loss(const float* rgb, const float* rgb0, const float* beta, const float* raw, const float* target, size_t R,
                      int Nf, float coef, float lambda_u, float* loss5, float* g_rgb, float* g_rgb0, float* g_beta, hipStream_t s);

}  // namespace train
}  // namespace dfn
