// nerfh_bwd.hip — gradient of the fine NeRF-H network w.r.t. its INPUTS (sample position and view direction)
// on the CDNA4 matrix cores (gfx950 only).
//
// The DFNet_dm step back-propagates a feature/photometric loss through render(c2w = predicted pose) into the
// pose (/root/reference/script/feature/direct_feature_matching.py:340-376, loss.backward()).  The NeRF weights
// are frozen there and the coarse net receives no gradient (z_samples.detach(), models/rendering.py:302), so what
// autograd computes is d L / d (point, viewdir) of the FINE net per sample.  This kernel does that in one pass
// per tile of sample points, activations and their gradients in registers throughout:
//   1. forward, exactly as nerfh_fine_kernel (same packed weights, same per-ray bias table), recording the ReLU
//      sign of every hidden unit as one bit per lane-resident feature (21 registers per 32-point block);
//   2. head derivatives (Sigmoid / Softplus, models/nerfw.py:281-295) applied to the incoming d L / d raw;
//   3. backward: every Linear again as a transposed MFMA product, now with W^T as the A operand
//      (nerfh_layout.h: BwdLayerId) — the C fragment of one backward layer, masked, is the B operand of the next;
//   4. the positional-encoding Jacobians (models/nerfw.py:105-133) fold d pe_xyz / d pe_dir to 3 + 3 floats.
// Output per sample: [d L/d point (3), d L/d viewdir through this sample (3)]; the per-ray reduction
// (d o = sum, d d = sum z * ..., viewdir normalisation) is ray_grad_reduce_kernel in nerfh_stages.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_mlp_core.h"
#include "nerfh_mask.h"

namespace dfn {

// Head activations of the forward recompute: split-f16 = the fp32-grade hardware forms of the inference kernel (exp_hw / rcp_nr,
// 1.5e-7: nerfh_device.h), exact fp32 = libm.
template <class PF, bool FAST> DFN_DEV float bwd_sigmoid(float v) { return PF::kSplit ? sigmoid_hw(v) : act_sigmoid<FAST>(v); }
template <class PF, bool FAST> DFN_DEV float bwd_softplus(float v) { return PF::kSplit ? softplus_hw(v) : act_softplus<FAST>(v); }

template <class PF, class P> constexpr uint32_t bwd_stride() {
  return bwd_max_unit_bytes<PF>() > bwd_max_unit_bytes<P>() ? bwd_max_unit_bytes<PF>() : bwd_max_unit_bytes<P>();
}
template <class PF, class P> constexpr uint32_t bwd_lds_bytes() { return 3 * bwd_stride<PF, P>(); }

// A plain layer of the backward kernel: a new staging unit, no bias folding, no activation, no pipelining.
#define DFN_FLAYER(KC, MB, EXTRA, RAYBIAS, IN, OUT, RB) \
  layer<PF, UMBF, false, NB, KC, MB, false, EXTRA, RAYBIAS, true, -1, true, false>(st, smem, IN, OUT, head, RB, carry)
// ... whose outputs are rectified by the layer's own epilogue (relu_mask then only reads the signs)
#define DFN_FLAYER_R(KC, MB, EXTRA, RAYBIAS, IN, OUT, RB) \
  layer<PF, UMBF, false, NB, KC, MB, (PF::kSlotsPerChunk == 8), EXTRA, RAYBIAS, true, -1, true, false>(st, smem, IN, OUT, head, RB, carry)
// Backward layers: the split of M-block m - 1's gradient fragment is issued piecewise behind the MFMAs of M-block m (layer<..., PIPE>,
// SCALE_FIRST form: these kernels keep the 2^10 weight scale and the per-point factors in Stager::lane_mul); the ReLU gate stays a
// mask pass over the finished operand.  4.01 -> 3.65 ms for the DFNet_dm step's 3.7 M points, fewer spilled registers (124 -> 64 B).
#define DFN_BLAYER(KC, MB, EXTRA, RAYBIAS, IN, OUT, RB) \
  layer<P, UMB, (P::kSplit && NB == 1), NB, KC, MB, false, EXTRA, RAYBIAS, true, -1, true, false, true, true>(st, smem, IN, OUT, head, RB, carry)
// ... a layer whose output the renormalisation inspects right afterwards (renorm_factor(..., &st.rmax)): no per-piece range tracking
#ifndef DFN_BWD_NOTRACK
#define DFN_BWD_NOTRACK 1
#endif
#define DFN_BLAYER_NT(KC, MB, EXTRA, RAYBIAS, IN, OUT, RB) \
  layer<P, UMB, (P::kSplit && NB == 1), NB, KC, MB, false, EXTRA, RAYBIAS, true, -1, true, false, true, true, !(DFN_BWD_NOTRACK && P::kSplit)>(st, smem, IN, OUT, head, RB, carry)

// PF: arithmetic of the forward recompute, P: arithmetic of the backward chain.  Split-f16 for both is the default:
// activations are O(1), and the gradient vector of a point is carried with a per-point power-of-two scale that is
// re-centred every other layer (renorm_factor), so its hi/lo halves stay in f16's normal range.
// MODE 0: forward recompute + backward in one pass.  MODE 1: the forward only, writing raw and the ReLU sign masks (the
// tracked forward of an autograd step).  MODE 2: the backward only, from those masks and raw — the head derivatives follow
// from the head OUTPUTS (sigmoid' = y (1 - y), softplus' = 1 - exp(-y)), so no activation has to be kept or recomputed.
template <class PF, class P, bool FAST, int WAVES, int UMBF, int UMB, int NB, int MODE = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void nerfh_fine_backward_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PPT = WAVES * NB * 32;
  constexpr int HC = chunks_of<P>(64), QC = chunks_of<P>(32), PC = chunks_of<P>(32), SC = chunks_of<P>(16);
  constexpr int FHC = chunks_of<PF>(64), FQC = chunks_of<PF>(32), FPC = chunks_of<PF>(32), FSC = chunks_of<PF>(16);
  using F = typename FragOf<P>::type;
  using FF = typename FragOf<PF>::type;
  Stager st;
  st.blob = a.blob; st.tab = a.tab; st.n_units = a.n_units; st.u = 0;
  st.waves = WAVES;
  st.dma_waves = WAVES == 8 ? 4 : WAVES;   // 8-wave workgroups: the four older waves idle at the unit barriers anyway (as launch_one, nerfh_mlp.hip)
  st.rmax = 0;
  st.in_scale = a.in_scale;
  st.out_scale = 1.f / a.in_scale;
  st.lane_mul = 1.f;
  st.t_sync = st.t_wait = 0;
  st.trace = nullptr;
  st.n_trace = 0;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = st.lane & 31, h = st.lane >> 5;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  const long long n_tiles = (n_pts + PPT - 1) / PPT;
  long long tile = blockIdx.x;
  if (tile >= n_tiles) return;
  stage_prime(st, smem, bwd_stride<PF, P>());
  for (; tile < n_tiles; tile += gridDim.x) {
    st.more = tile + gridDim.x < n_tiles;
    float x[NB][3], g[NB][9];
    long long pt[NB];
    uint32_t ray_of[NB];
    const float* rb_dir[NB];
    const float* rb_tr[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      pt[nb] = tile * PPT + st.wave * (NB * 32) + nb * 32 + p;
      const uint32_t q = uint32_t(pt[nb] < n_pts ? pt[nb] : n_pts - 1);
      ray_of[nb] = q / uint32_t(a.n_samples);
      const float z = a.z[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[nb][c] = add_rn(a.rays_o[ray_of[nb] * 3 + c], mul_rn(a.rays_d[ray_of[nb] * 3 + c], z));
#pragma unroll
      for (int c = 0; c < 9; ++c) g[nb][c] = (MODE != 1 && h == 0 && pt[nb] < n_pts) ? a.graw[size_t(q) * 9 + c] : 0.f;
      rb_dir[nb] = a.ray_bias + size_t(ray_of[nb]) * kRayBiasFloats;
      rb_tr[nb] = rb_dir[nb] + kRayBiasFloats / 2;
    }
    const float* const norb[NB] = {};
    f32x16 head[NB], carry[NB];
    uint32_t mk[8][NB][2], md[NB][1], mt[4][NB][1];
    // split-f16 gradient chain: per-point power-of-two scale sp of the gradient vector (renorm_factor()); 1 otherwise
    static_assert(!P::kSplit || NB == 1, "the per-lane renormalisation factor is one float: one point block per wave");
    float sp[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      sp[nb] = 1.f;
      if constexpr (P::kSplit) {
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < 9; ++c) mx = fmaxf(mx, fabsf(g[nb][c]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx > 0.f) {
          int e;
          (void)frexpf(mx, &e);
          sp[nb] = ldexpf(1.f, e < -64 ? 64 : -e);     // max |d raw| * sp in [0.5, 1) (scale capped at 2^64)
        }
      }
    }
    st.lane_mul = 1.f;

    F dth[NB][SC], drgb[NB][SC];
    float dsig_true[NB];
    [[maybe_unused]] float o9[NB][9];
    static_assert(MODE == 0 || NB == 1, "the two-pass form keeps one point block per wave");
    uint32_t* mwords = MODE ? a.masks + ((size_t)(tile * WAVES + st.wave) * kBwdMaskWords) * 64 + st.lane : nullptr;
    // MODE 2 fetches each layer's sign mask from the saved forward right before the layer that produces the gradient it gates
    // (the load's latency hides behind that layer's MFMAs) instead of holding all 21 words in registers for the whole chain.
    auto fetch_mk = [&](int l) { if constexpr (MODE == 2) { mk[l][0][0] = mwords[(2 * l) * 64]; mk[l][0][1] = mwords[(2 * l + 1) * 64]; } };
    auto fetch_md = [&]() { if constexpr (MODE == 2) md[0][0] = mwords[16 * 64]; };
    auto fetch_mt = [&](int l) { if constexpr (MODE == 2) mt[l][0][0] = mwords[(17 + l) * 64]; };
    if constexpr (MODE == 2) {
      // head derivatives from the saved forward
      const uint32_t q = uint32_t(pt[0] < n_pts ? pt[0] : n_pts - 1);
      const float* rw = a.raw_in + size_t(q) * 9;
      const bool on = h == 0 && pt[0] < n_pts;
      clear<P>(drgb[0]);
      clear<P>(dth[0]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ys = on ? rw[c] : 0.f, yt = on ? rw[4 + c] : 0.f;
        set_slot<P>(drgb[0], c, g[0][c] * ys * (1.f - ys) * sp[0]);
        set_slot<P>(dth[0], c, g[0][4 + c] * yt * (1.f - yt) * sp[0]);
      }
      dsig_true[0] = on ? g[0][3] * -expm1f(-rw[3]) : 0.f;
      set_slot<P>(dth[0], 3, on ? g[0][7] * -expm1f(-rw[7]) * sp[0] : 0.f);
      set_slot<P>(dth[0], 4, on ? g[0][8] * -expm1f(-rw[8]) * sp[0] : 0.f);
    } else {
    // ------------------------------------------------------------------ forward (recording ReLU signs)
    FF hid[NB][FHC];
    {
      FF pe[NB][FPC], u[NB][FHC];
      posenc_xyz<PF, FAST, NB, FPC>(x, h, pe);
      DFN_FLAYER_R(FPC, 4, false, false, pe, u, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(u[nb], mk[0][nb]);
      if constexpr (MODE == 1) { mwords[0 * 64] = mk[0][0][0]; mwords[1 * 64] = mk[0][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, u, hid, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(hid[nb], mk[1][nb]);
      if constexpr (MODE == 1) { mwords[2 * 64] = mk[1][0][0]; mwords[3 * 64] = mk[1][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, hid, u, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(u[nb], mk[2][nb]);
      if constexpr (MODE == 1) { mwords[4 * 64] = mk[2][0][0]; mwords[5 * 64] = mk[2][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, u, hid, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(hid[nb], mk[3][nb]);
      if constexpr (MODE == 1) { mwords[6 * 64] = mk[3][0][0]; mwords[7 * 64] = mk[3][0][1]; }
      {
        FF cat[NB][FPC + FHC];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int i = 0; i < FPC; ++i) cat[nb][i] = pe[nb][i];
#pragma unroll
          for (int i = 0; i < FHC; ++i) cat[nb][FPC + i] = hid[nb][i];
        }
        DFN_FLAYER_R(FPC + FHC, 4, false, false, cat, u, norb);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(u[nb], mk[4][nb]);
      if constexpr (MODE == 1) { mwords[8 * 64] = mk[4][0][0]; mwords[9 * 64] = mk[4][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, u, hid, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(hid[nb], mk[5][nb]);
      if constexpr (MODE == 1) { mwords[10 * 64] = mk[5][0][0]; mwords[11 * 64] = mk[5][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, hid, u, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(u[nb], mk[6][nb]);
      if constexpr (MODE == 1) { mwords[12 * 64] = mk[6][0][0]; mwords[13 * 64] = mk[6][0][1]; }
      DFN_FLAYER_R(FHC, 4, false, false, u, hid, norb);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FHC>(hid[nb], mk[7][nb]);
      if constexpr (MODE == 1) { mwords[14 * 64] = mk[7][0][0]; mwords[15 * 64] = mk[7][0][1]; }
    }
    // heads; their pre-activation gradients seed the backward pass
    {
      FF fin[NB][FHC], dummy[NB][FSC];
      DFN_FLAYER(FHC, 4, true, false, hid, fin, norb);   // xyz_encoding_final + static_sigma (5th M-block)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        dsig_true[nb] = h == 0 ? g[nb][3] * bwd_sigmoid<PF, FAST>(head[nb][0]) : 0.f;  // softplus' = sigmoid
        if constexpr (MODE == 1) o9[nb][3] = bwd_softplus<PF, FAST>(head[nb][0]);
      }
      {
        FF de[NB][FQC];
        DFN_FLAYER_R(FHC, 2, false, true, fin, de, rb_dir);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FQC>(de[nb], md[nb]);
        if constexpr (MODE == 1) mwords[16 * 64] = md[0][0];
        DFN_FLAYER(FQC, 0, true, false, de, dummy, norb);  // static_rgb
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          clear<P>(drgb[nb]);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float y = bwd_sigmoid<PF, FAST>(head[nb][c]);
            if constexpr (MODE == 1) o9[nb][c] = y;
            set_slot<P>(drgb[nb], c, h == 0 ? g[nb][c] * y * (1.f - y) * sp[nb] : 0.f);
          }
        }
      }
      {
        FF t0[NB][FQC], t1[NB][FQC];
        DFN_FLAYER_R(FHC, 2, false, true, fin, t0, rb_tr);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FQC>(t0[nb], mt[0][nb]);
        if constexpr (MODE == 1) mwords[17 * 64] = mt[0][0][0];
        DFN_FLAYER_R(FQC, 2, false, false, t0, t1, norb);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FQC>(t1[nb], mt[1][nb]);
        if constexpr (MODE == 1) mwords[18 * 64] = mt[1][0][0];
        DFN_FLAYER_R(FQC, 2, false, false, t1, t0, norb);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FQC>(t0[nb], mt[2][nb]);
        if constexpr (MODE == 1) mwords[19 * 64] = mt[2][0][0];
        DFN_FLAYER_R(FQC, 2, false, false, t0, t1, norb);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) relu_mask<PF, FQC>(t1[nb], mt[3][nb]);
        if constexpr (MODE == 1) mwords[20 * 64] = mt[3][0][0];
        DFN_FLAYER(FQC, 0, true, false, t1, dummy, norb);  // transient heads: rows 0..2 rgb, 3 sigma, 8 beta (C reg 4)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          clear<P>(dth[nb]);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float y = bwd_sigmoid<PF, FAST>(head[nb][c]);
            if constexpr (MODE == 1) o9[nb][4 + c] = y;
            set_slot<P>(dth[nb], c, h == 0 ? g[nb][4 + c] * y * (1.f - y) * sp[nb] : 0.f);
          }
          if constexpr (MODE == 1) { o9[nb][7] = bwd_softplus<PF, FAST>(head[nb][3]); o9[nb][8] = bwd_softplus<PF, FAST>(head[nb][4]); }
          set_slot<P>(dth[nb], 3, h == 0 ? g[nb][7] * bwd_sigmoid<PF, FAST>(head[nb][3]) * sp[nb] : 0.f);
          set_slot<P>(dth[nb], 4, h == 0 ? g[nb][8] * bwd_sigmoid<PF, FAST>(head[nb][4]) * sp[nb] : 0.f);
        }
      }
    }

    }   // MODE != 2
    if constexpr (MODE == 1) {
      if (h == 0 && pt[0] < n_pts) {
        float* dst = a.raw_out + size_t(pt[0]) * 9;
#pragma unroll
        for (int c = 0; c < 9; ++c) dst[c] = o9[0][c];
      }
      continue;
    }
    // ------------------------------------------------------------------ backward
    float gx[NB][3], gv[NB][3];
    F gh[NB][HC];
    {
      F cat[NB][HC];
      {
        F g0[NB][QC], g1[NB][QC];
        fetch_mt(3);
        DFN_BLAYER(SC, 2, false, false, dth, g1, norb);   // BW_THEAD
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) apply_mask<P, QC>(g1[nb], mt[3][nb]);
        fetch_mt(2);
        DFN_BLAYER(QC, 2, false, false, g1, g0, norb);    // BW_TE3
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) apply_mask<P, QC>(g0[nb], mt[2][nb]);
        fetch_mt(1);
        DFN_BLAYER(QC, 2, false, false, g0, g1, norb);    // BW_TE2
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) apply_mask<P, QC>(g1[nb], mt[1][nb]);
        fetch_mt(0);
        DFN_BLAYER_NT(QC, 2, false, false, g1, g0, norb);    // BW_TE1
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) apply_mask<P, QC>(g0[nb], mt[0][nb]);
        fetch_md();
        DFN_BLAYER_NT(SC, 2, false, false, drgb, g1, norb);  // BW_RGB
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          apply_mask<P, QC>(g1[nb], md[nb]);
#pragma unroll
          for (int i = 0; i < QC; ++i) { cat[nb][i] = g0[nb][i]; cat[nb][QC + i] = g1[nb][i]; }
        }
      }
      F cat2[NB][HC + SC];
      {
        F dfin[NB][HC];
        if constexpr (P::kSplit) {
          st.lane_mul = renorm_factor<P, HC>(cat[0], sp[0], &st.rmax);   // the two branches share sp up to here
          // d sigma_s joins the chain after this layer (slot 64 of BW_FIN's input) at the scale chosen HERE: when the colour /
          // transient gradients of a point are orders of magnitude below its density gradient (near-duplicate samples: alpha ~ 0)
          // a factor chosen from them alone overflowed that slot's f16 halves (inf -> NaN for the whole ray).  Keep
          // |d sigma_s| * sp * 16 below 2^11.
          float ds = fabsf(dsig_true[0]) * sp[0] * kX3ActScale * st.lane_mul;
          ds = fmaxf(ds, __shfl_xor(ds, 32, 64));
          if (ds > 2048.f) {
            int e;
            (void)frexpf(ds, &e);                 // ds = f * 2^e
            st.lane_mul *= ldexpf(1.f, 11 - e);
          }
        }
        DFN_BLAYER_NT(HC, 4, true, false, cat, dfin, norb);  // BW_FINCAT: d final + (head) d pe_dir
        if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          // direction-encoding Jacobian: half h holds frequencies 2h, 2h+1 (pe_dir_feature())
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = a.viewdirs[ray_of[nb] * 3 + c];
            float acc = h == 0 ? head[nb][12 + c] : 0.f;
            float vh = 0.f, vl = 0.f;
            if (P::kSplit) rev_split(v, vh, vl);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float f = float(1 << k) * (h ? 4.f : 1.f);
              float sn, cs;
              if (P::kSplit) rev_sincos(vh, vl, f, sn, cs);
              else { sn = sinf(v * f); cs = cosf(v * f); }
              acc += f * (cs * head[nb][6 * k + c] - sn * head[nb][6 * k + 3 + c]);
            }
            gv[nb][c] = acc / sp[nb];   // back to the true scale
          }
#pragma unroll
          for (int i = 0; i < HC; ++i) cat2[nb][i] = dfin[nb][i];
#pragma unroll
          for (int i = 0; i < SC; ++i) clear_one<P>(cat2[nb][HC + i]);
          set_slot<P>(cat2[nb], 64, dsig_true[nb] * sp[nb]);   // slot 64 of half 0: d sigma_s pre-activation (h == 1 lanes hold 0)
        }
      }
      if constexpr (P::kSplit) st.lane_mul = renorm_factor<P, HC + SC>(cat2[0], sp[0], &st.rmax);   // incl. the d sigma_s slot
      fetch_mk(7);
      DFN_BLAYER_NT(HC + SC, 4, false, false, cat2, gh, norb);  // BW_FIN
      if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
    }
    F gh2[NB][HC];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh[nb], mk[7][nb]);
    if constexpr (P::kSplit) st.lane_mul = renorm_factor<P, HC>(gh[0], sp[0], &st.rmax);
    fetch_mk(6);
    DFN_BLAYER(HC, 4, false, false, gh, gh2, norb);          // BW_L8
    if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh2[nb], mk[6][nb]);
    fetch_mk(5);
    DFN_BLAYER_NT(HC, 4, false, false, gh2, gh, norb);          // BW_L7
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh[nb], mk[5][nb]);
    if constexpr (P::kSplit) st.lane_mul = renorm_factor<P, HC>(gh[0], sp[0], &st.rmax);
    fetch_mk(4);
    DFN_BLAYER(HC, 4, false, false, gh, gh2, norb);          // BW_L6
    if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh2[nb], mk[4][nb]);
    // positional-encoding Jacobian of d pe (slot order pe_xyz_feature()), accumulated into gx
    // The split-f16 kernel evaluates the Jacobian's sin / cos with the exact-fract hardware path (rev_sincos: 4e-7 absolute, far
    // inside the gradient tolerance): libm's full-range sinf / cosf expand into a large-argument reduction whose temporaries, on
    // top of the live gradient vectors, were what spilled (152 + 61 registers around the two Jacobians).  Exact fp32 keeps libm.
    constexpr bool JFAST = P::kSplit;
    auto pe_jacobian = [&](const F (&dpe)[NB][PC], bool first) {
      [[maybe_unused]] FF pe[NB][FPC];
      if constexpr (!JFAST) {   // exact fp32: the encoding itself, recomputed (libm sin / cos) instead of kept alive
        float x2[NB][3];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            x2[nb][c] = x[nb][c];
            asm volatile("" : "+v"(x2[nb][c]));
          }
        posenc_xyz<PF, FAST, NB, FPC>(x2, h, pe);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float xc = x[nb][c];
          float uh = 0.f, ul = 0.f;
          if (JFAST) rev_split(xc, uh, ul);
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const float f = float(1 << k) * (h ? 32.f : 1.f);
            float sn, cs;
            if constexpr (JFAST) rev_sincos(uh, ul, f, sn, cs);
            else { sn = get_slot<PF>(pe[nb], 6 * k + c); cs = get_slot<PF>(pe[nb], 6 * k + 3 + c); }
            acc += f * (cs * get_slot<P>(dpe[nb], 6 * k + c) - sn * get_slot<P>(dpe[nb], 6 * k + 3 + c));
          }
          gx[nb][c] = (first ? 0.f : gx[nb][c]) + acc / sp[nb];
        }
        // raw coordinates: half 0 slots 30, 31 = x, y; half 1 slot 30 = z
        const float r30 = get_slot<P>(dpe[nb], 30) / sp[nb], r31 = get_slot<P>(dpe[nb], 31) / sp[nb];
        gx[nb][0] += h == 0 ? r30 : 0.f;
        gx[nb][1] += h == 0 ? r31 : 0.f;
        gx[nb][2] += h == 1 ? r30 : 0.f;
      }
    };
    {
      // BW_L5 = d h4 (M-blocks 0..3) + d pe_xyz through the skip connection (M-blocks 4, 5).  Two layer calls on the same input;
      // as one 6-M-block layer its 96-register output on top of input and masks spilled.
      fetch_mk(3);
      if constexpr (UMB <= 2) {
        {   // the d pe M-blocks come FIRST in the blob (Packer::pack_bwd) and are folded to 3 floats while only the input is live
          F dpe[NB][PC];
          DFN_BLAYER(HC, 2, false, false, gh2, dpe, norb);   // BW_L5, M-blocks 4, 5
          pe_jacobian(dpe, true);
        }
        DFN_BLAYER_NT(HC, 4, false, false, gh2, gh, norb);      // BW_L5, M-blocks 0..3
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh[nb], mk[3][nb]);
      } else {
        F big[NB][HC + PC];
        DFN_BLAYER(HC, 6, false, false, gh2, big, norb);
        F dpe[NB][PC];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int i = 0; i < HC; ++i) gh[nb][i] = big[nb][i];
#pragma unroll
          for (int i = 0; i < PC; ++i) dpe[nb][i] = big[nb][HC + i];
          apply_mask<P, HC>(gh[nb], mk[3][nb]);
        }
        pe_jacobian(dpe, true);
      }
    }
    if constexpr (P::kSplit) st.lane_mul = renorm_factor<P, HC>(gh[0], sp[0], &st.rmax);
    fetch_mk(2);
    DFN_BLAYER(HC, 4, false, false, gh, gh2, norb);          // BW_L4
    if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh2[nb], mk[2][nb]);
    fetch_mk(1);
    DFN_BLAYER_NT(HC, 4, false, false, gh2, gh, norb);          // BW_L3
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh[nb], mk[1][nb]);
    if constexpr (P::kSplit) st.lane_mul = renorm_factor<P, HC>(gh[0], sp[0], &st.rmax);
    fetch_mk(0);
    DFN_BLAYER(HC, 4, false, false, gh, gh2, norb);          // BW_L2
    if constexpr (P::kSplit) { sp[0] *= st.lane_mul; st.lane_mul = 1.f; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) apply_mask<P, HC>(gh2[nb], mk[0][nb]);
    {
      F dpe[NB][PC];
      DFN_BLAYER(HC, 2, false, false, gh2, dpe, norb);       // BW_L1
      pe_jacobian(dpe, false);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float o6[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o6[c] = gx[nb][c] + __shfl_xor(gx[nb][c], 32, 64);
        o6[3 + c] = gv[nb][c] + __shfl_xor(gv[nb][c], 32, 64);
      }
      if (h == 0 && pt[nb] < n_pts) {
        float* dst = a.gpts + size_t(pt[nb]) * 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) dst[c] = o6[c];
      }
    }
  }
}

template <class PF, class P, bool FAST, int WAVES, int UMBF, int UMB, int NB, int MODE = 0>
static hipError_t launch_bwd_one(const BwdArgs& a, int n_cu, hipStream_t stream) {
  constexpr int PPT = WAVES * NB * 32;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  if (n_pts <= 0) return hipSuccess;
  if (n_pts >= (1LL << 31)) return hipErrorInvalidValue;
  const long long n_tiles = (n_pts + PPT - 1) / PPT;
  const int grid = int(n_tiles < n_cu ? n_tiles : n_cu);
  const uint32_t lds = bwd_lds_bytes<PF, P>();
  auto kern = nerfh_fine_backward_kernel<PF, P, FAST, WAVES, UMBF, UMB, NB, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_mlp_fine_backward(int prec, const BwdArgs& a, int n_cu, hipStream_t stream, int mode) {
  if (mode) {   // two-pass form: split-f16 only (exact gates are its point)
    if (prec != 2) return hipErrorInvalidValue;
    static_assert(kBwdTilePoints == 8 * 32, "tile geometry of the split-f16 gradient kernel");
    return mode == 1 ? launch_bwd_one<PrecX3, PrecX3, false, 8, bwd_fwd_unit_mb<PrecX3>(), bwd_unit_mb<PrecX3>(), 1, 1>(a, n_cu, stream)
                     : launch_bwd_one<PrecX3, PrecX3, false, 8, bwd_fwd_unit_mb<PrecX3>(), bwd_unit_mb<PrecX3>(), 1, 2>(a, n_cu, stream);
  }
  // plain f16 gradient arithmetic is not offered: ReLU gates flipped by f16 rounding put it at 3e-2 of autograd (30x the
  // contract); split-f16 runs at the f16 MFMA rate with fp32-grade gates (nerfh_api.hip rejects DFN_PREC_F16 here)
  if (prec == 0) return hipErrorInvalidValue;
  if (prec == 2) return launch_bwd_one<PrecX3, PrecX3, false, 8, bwd_fwd_unit_mb<PrecX3>(), bwd_unit_mb<PrecX3>(), 1>(a, n_cu, stream);   // split-f16 forward and gradient chain
  return launch_bwd_one<PrecF32, PrecF32, false, 4, 1, 1, 1>(a, n_cu, stream);
}

}  // namespace dfn
