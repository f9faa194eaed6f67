// nerfh_fused_chain.hip — the register-resident forward and data-gradient chains of the NeRF-H TRAINING step (gfx950 only).
//
// Replaces, for netwidth 128, the layer-by-layer products of nerfh_train.hip in the optimisation step of the reference
// (/root/reference/script/run_nerf.py:50-66: render(**render_kwargs_train) ... loss.backward(); models/nerfw.py:297-354
// NeRFW.forward in its training branches, :47-95 run_network_NeRFW).  Both kernels are the machinery of nerfh_mlp_core.h /
// nerfh_bwd.hip — weights as the MFMA A operand streamed L2 -> LDS, 32 points per wavefront carried through the whole network in
// registers, split-f16 products (fp32-grade) — extended by what the WEIGHT gradients need and nothing else:
//   forward : every layer input X_l is stored exactly as the B-operand registers it already is (hi | lo planes, one coalesced
//             16-byte store per lane and plane) + one ReLU sign bit per hidden unit;
//   backward: from the heads' PRE-activation gradients (nerfh_train.hip: composite_*_backward*), gates from the sign bits; every
//             pre-activation gradient G_l is stored the same way.  The 32 points of a wavefront share one power-of-two scale
//             (re-centred every other layer, recorded per stored array and wave-tile): the weight gradient is a SUM over points,
//             so a point far below its wave's largest gradient needs no precision of its own, and the stored halves stay in
//             f16's range whatever the gradient magnitude.
// The weight gradients are then a pure stream over (G_l, X_l): nerfh_fused_wgrad.hip.  Layouts: nerfh_fused_train.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_fused_train.h"
#include "nerfh_mask.h"
#include "nerfh_mlp_core.h"

namespace dfn {
namespace fused {

namespace {
using P = PrecX3;
using F = half8x2;
constexpr int NB = 1, WAVES = 8, UMB = 2;
constexpr int HC = 8, QC = 4, PC = 4, SC = 2;   // chunks of a 128- / 64- / encoding- / head-wide operand
constexpr uint32_t kFwdStride = unit_bytes<P>(96, UMB);   // largest forward unit: two M-blocks of layer 5 (96 slots)
constexpr uint32_t kBwdStride = unit_bytes<P>(80, UMB);   // largest backward unit: two M-blocks of final^T + sigma (80 slots)

// (the pipelined conversion of the render kernels — layer<..., PIPE> in its SCALE_FIRST form, which nerfh_bwd.hip's backward chain uses —
// measured +-0 here: these chains wait on their operand stores, not on the conversion)
#define TF_LAYER(KC, MB, RELU, EXTRA, RAYBIAS, IN, OUT, RB) \
  layer<P, UMB, false, NB, KC, MB, RELU, EXTRA, RAYBIAS, true, -1, true, false>(st, smem, IN, OUT, head, RB, carry)
#define TB_LAYER(KC, MB, IN, OUT) \
  layer<P, UMB, false, NB, KC, MB, false, false, false, true, -1, true, false, true>(st, smem, IN, OUT, head, norb, carry)

DFN_DEV void init_stager(Stager& st, const ChainArgs& a) {
  st.blob = a.blob; st.tab = a.tab; st.n_units = a.n_units; st.u = 0;
  st.waves = WAVES;
  st.dma_waves = 4;   // the four older waves issue the weight DMA (they idle at the unit barriers anyway): 2.49 -> 2.47 ms per step, three alternations
  st.rmax = 0;
  st.in_scale = a.in_scale;
  st.out_scale = 1.f / a.in_scale;
  st.lane_mul = 1.f;
  st.t_sync = st.t_wait = 0;
  st.trace = nullptr;
  st.n_trace = 0;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}

// One stored array of this wavefront's tile: chunk c -> [hi plane | lo plane], lane (p, h) at 16-byte slot 2 p + h.
template <int KC, int PL, int N>
DFN_DEV void store_array(char* base, const F (&v)[N]) {
#pragma unroll
  for (int c = 0; c < KC; ++c) {   // written once, read once by another kernel: non-temporal
    if constexpr (PL == 2) {
      __builtin_nontemporal_store(v[c].hi, reinterpret_cast<half8*>(base + c * 2048));
      __builtin_nontemporal_store(v[c].lo, reinterpret_cast<half8*>(base + c * 2048 + 1024));
    } else {   // one plane: hi is the TRUNCATED half of the value (store_hidden); hi + lo in f16 arithmetic rounds the exact sum to nearest
      const half8 r = v[c].hi + v[c].lo;
      __builtin_nontemporal_store(r, reinterpret_cast<half8*>(base + c * 1024));
    }
  }
}

DFN_DEV float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
// Power of two that brings the largest |hi| of the WAVE's operand `v` to [8, 16) (renorm_factor of nerfh_mask.h, wave-wide).
template <int C, int N>
DFN_DEV float renorm_wave(const F (&v)[N], float sp_now) {
  half8 m = __builtin_elementwise_abs(v[0].hi);
#pragma unroll
  for (int c = 1; c < C; ++c) m = __builtin_elementwise_max(m, __builtin_elementwise_abs(v[c].hi));
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)m[j]);
  mx = wave_max(mx);
  if (!(mx > 0.f)) return 1.f;
  int e;
  (void)frexpf(mx, &e);
  const float f = ldexpf(1.f, 4 - e);
  return sp_now * f <= kSpCap ? f : 1.f;
}
}  // namespace

size_t chain_wave_tiles(long long n_points) {
  return size_t((n_points + kTilePoints - 1) / kTilePoints) * WAVES;
}

// ------------------------------------------------------------------------------------------ forward
template <bool FINE, int PL>
__global__ __launch_bounds__(WAVES * 64, 1) void train_fwd_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RAWC = FINE ? 9 : 4;
  Stager st;
  init_stager(st, a);
  const int p = st.lane & 31, h = st.lane >> 5;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  const long long n_tiles = (n_pts + kTilePoints - 1) / kTilePoints;
  long long tile = blockIdx.x;
  if (tile >= n_tiles) return;
  stage_prime(st, smem, kFwdStride);
  const uint32_t lane_slot = uint32_t(2 * p + h) * 16u;
  for (; tile < n_tiles; tile += gridDim.x) {
    st.more = tile + gridDim.x < n_tiles;
    const long long pt = tile * kTilePoints + st.wave * 32 + p;
    const uint32_t q = uint32_t(pt < n_pts ? pt : n_pts - 1);
    const uint32_t ray = q / uint32_t(a.n_samples);
    float x[NB][3];
    {
      const float z = a.z[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[0][c] = add_rn(a.rays_o[ray * 3 + c], mul_rn(a.rays_d[ray * 3 + c], z));
    }
    const float* rb_dir[NB] = {a.ray_bias + size_t(ray) * kRayBiasFloats};
    const float* rb_tr[NB] = {rb_dir[0] + kRayBiasFloats / 2};
    const float* const norb[NB] = {};
    const size_t wt = size_t(tile) * WAVES + st.wave;
    uint32_t* mwords = a.masks + wt * (kMaskWords * 64) + st.lane;
    auto arr = [&](int id, int kc) { return a.arrays + a.arr_off[id] + wt * size_t(kc) * (1024 * PL) + lane_slot; };
    f32x16 head[NB], carry[NB];
    uint32_t m2[2], m1[1];
    float o[RAWC];
    F hid[NB][HC];
    {
      F pe[NB][PC], u[NB][HC];
      posenc_xyz<P, false, NB, PC>(x, h, pe);
      store_array<PC, PL>(arr(XA_PE, PC), pe[0]);
      TF_LAYER(PC, 4, true, false, false, pe, u, norb);
      relu_mask<P, HC>(u[0], m2); mwords[0 * 64] = m2[0]; mwords[1 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H1, HC), u[0]);
      TF_LAYER(HC, 4, true, false, false, u, hid, norb);
      relu_mask<P, HC>(hid[0], m2); mwords[2 * 64] = m2[0]; mwords[3 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H2, HC), hid[0]);
      TF_LAYER(HC, 4, true, false, false, hid, u, norb);
      relu_mask<P, HC>(u[0], m2); mwords[4 * 64] = m2[0]; mwords[5 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H3, HC), u[0]);
      TF_LAYER(HC, 4, true, false, false, u, hid, norb);
      relu_mask<P, HC>(hid[0], m2); mwords[6 * 64] = m2[0]; mwords[7 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H4, HC), hid[0]);
      {
        F cat[NB][PC + HC];   // torch.cat([input_xyz, xyz_], -1) (nerfw.py:328-330)
        {   // the encoding is recomputed here (bit-identical): cheaper than 32 registers live across four layers
          float x2[NB][3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            x2[0][c] = x[0][c];
            asm volatile("" : "+v"(x2[0][c]));   // opaque copy: the two encodings are not merged
          }
          posenc_xyz<P, false, NB, PC>(x2, h, pe);
        }
#pragma unroll
        for (int i = 0; i < PC; ++i) cat[0][i] = pe[0][i];
#pragma unroll
        for (int i = 0; i < HC; ++i) cat[0][PC + i] = hid[0][i];
        TF_LAYER(PC + HC, 4, true, false, false, cat, u, norb);
      }
      relu_mask<P, HC>(u[0], m2); mwords[8 * 64] = m2[0]; mwords[9 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H5, HC), u[0]);
      TF_LAYER(HC, 4, true, false, false, u, hid, norb);
      relu_mask<P, HC>(hid[0], m2); mwords[10 * 64] = m2[0]; mwords[11 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H6, HC), hid[0]);
      TF_LAYER(HC, 4, true, false, false, hid, u, norb);
      relu_mask<P, HC>(u[0], m2); mwords[12 * 64] = m2[0]; mwords[13 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H7, HC), u[0]);
      TF_LAYER(HC, 4, true, false, false, u, hid, norb);
      relu_mask<P, HC>(hid[0], m2); mwords[14 * 64] = m2[0]; mwords[15 * 64] = m2[1];
      store_array<HC, PL>(arr(XA_H8, HC), hid[0]);
    }
    {
      F fin[NB][HC], dummy[NB][SC];
      TF_LAYER(HC, 4, false, true, false, hid, fin, norb);   // xyz_encoding_final (no activation) + static_sigma (5th M-block, row 0)
      o[3] = softplus(head[0][0]);
      store_array<HC, PL>(arr(XA_FIN, HC), fin[0]);
      {
        F de[NB][QC];
        TF_LAYER(HC, 2, true, false, true, fin, de, rb_dir);  // dir_encoding: per-ray bias = b + W[:, 128:] [pe_dir (, a)]
        relu_mask<P, QC>(de[0], m1); mwords[16 * 64] = m1[0];
        store_array<QC, PL>(arr(XA_DE, QC), de[0]);
        TF_LAYER(QC, 0, false, true, false, de, dummy, norb); // static_rgb
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = sigmoid(head[0][c]);
      }
      if constexpr (FINE) {
        F t0[NB][QC], t1[NB][QC];
        TF_LAYER(HC, 2, true, false, true, fin, t0, rb_tr);   // transient_encoding.0: per-ray bias = b + W[:, 128:] t
        relu_mask<P, QC>(t0[0], m1); mwords[17 * 64] = m1[0];
        store_array<QC, PL>(arr(XA_T0, QC), t0[0]);
        TF_LAYER(QC, 2, true, false, false, t0, t1, norb);
        relu_mask<P, QC>(t1[0], m1); mwords[18 * 64] = m1[0];
        store_array<QC, PL>(arr(XA_T1, QC), t1[0]);
        TF_LAYER(QC, 2, true, false, false, t1, t0, norb);
        relu_mask<P, QC>(t0[0], m1); mwords[19 * 64] = m1[0];
        store_array<QC, PL>(arr(XA_T2, QC), t0[0]);
        TF_LAYER(QC, 2, true, false, false, t0, t1, norb);
        relu_mask<P, QC>(t1[0], m1); mwords[20 * 64] = m1[0];
        store_array<QC, PL>(arr(XA_T3, QC), t1[0]);
        TF_LAYER(QC, 0, false, true, false, t1, dummy, norb); // transient heads: rows 0..2 rgb, 3 sigma, 8 (C register 4) beta
#pragma unroll
        for (int c = 0; c < 3; ++c) o[4 + c] = sigmoid(head[0][c]);
        o[7] = softplus(head[0][3]);
        o[8] = softplus(head[0][4]);
      }
    }
    if (h == 0 && pt < n_pts) {
      float* dst = a.raw_out + size_t(pt) * RAWC;
#pragma unroll
      for (int c = 0; c < RAWC; ++c) dst[c] = o[c];
    }
  }
  range_report<P>(st.rmax, a.status);
}

// ------------------------------------------------------------------------------------------ backward
// The body takes its place in the persistent grid as arguments (block `blk` of `nblk`): train_bwd_chain_kernel is the grid itself,
// train_bwd_chain_pair_kernel runs two networks' chains as the two halves of ONE grid.
template <bool FINE, int PL>
DFN_DEV void train_bwd_chain_body(const ChainArgs& a, char* smem, const int blk, const int nblk) {
  constexpr int RAWC = FINE ? 9 : 4;
  Stager st;
  init_stager(st, a);
  const int p = st.lane & 31, h = st.lane >> 5;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  const long long n_tiles = (n_pts + kTilePoints - 1) / kTilePoints;
  const size_t n_wt = size_t(n_tiles) * WAVES;
  long long tile = blk;
  if (tile >= n_tiles) return;
  stage_prime(st, smem, kBwdStride);
  const uint32_t lane_slot = uint32_t(2 * p + h) * 16u;
  for (; tile < n_tiles; tile += nblk) {
    st.more = tile + nblk < n_tiles;
    const long long pt = tile * kTilePoints + st.wave * 32 + p;
    const bool on = h == 0 && pt < n_pts;
    float g[RAWC];
    {
      const float* src = a.gpre + size_t(pt < n_pts ? pt : n_pts - 1) * RAWC;
#pragma unroll
      for (int c = 0; c < RAWC; ++c) g[c] = on ? src[c] : 0.f;
    }
    const float* const norb[NB] = {};
    const size_t wt = size_t(tile) * WAVES + st.wave;
    const uint32_t* mwords = a.masks + wt * (kMaskWords * 64) + st.lane;
    auto arr = [&](int id, int kc) { return a.arrays + a.arr_off[id] + wt * size_t(kc) * (1024 * PL) + lane_slot; };
    // stored value = true gradient x 16 sp (x3_split's operand scale x the wave's running scale)
    auto note_scale = [&](int id, float sp_) { if (st.lane == 0) a.gscale[size_t(id) * n_wt + wt] = kX3ActScale * sp_; };
    f32x16 head[NB], carry[NB];
    // the wave's common scale: largest |seed| x sp in [0.5, 1)
    float sp = 1.f;
    {
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < RAWC; ++c) mx = fmaxf(mx, fabsf(g[c]));
      mx = wave_max(mx);
      if (mx > 0.f) {
        int e;
        (void)frexpf(mx, &e);
        sp = ldexpf(1.f, e < -64 ? 64 : -e);
      }
    }
    st.lane_mul = 1.f;
    uint32_t mk[2], m1[1];
    F cat2[NB][HC + SC];
    {
      F cat[NB][FINE ? HC : QC];
      F drgb[NB][SC];
      clear<P>(drgb[0]);
#pragma unroll
      for (int c = 0; c < 3; ++c) set_slot<P>(drgb[0], c, g[c] * sp);
      store_array<SC, PL>(arr(GA_DRGB, SC), drgb[0]);
      note_scale(GA_DRGB, sp);
      if constexpr (FINE) {
        F g0[NB][QC], g1[NB][QC];
        {
          F dth[NB][SC];
          clear<P>(dth[0]);
#pragma unroll
          for (int c = 0; c < 3; ++c) set_slot<P>(dth[0], c, g[4 + c] * sp);
          set_slot<P>(dth[0], 3, g[7] * sp);
          set_slot<P>(dth[0], 4, g[8] * sp);    // slot 4 of half 0 = row 8 = transient_beta
          store_array<SC, PL>(arr(GA_DTH, SC), dth[0]);
          note_scale(GA_DTH, sp);
          m1[0] = mwords[20 * 64];
          TB_LAYER(SC, 2, dth, g1);              // transient heads^T -> d t3
        }
        apply_mask<P, QC>(g1[0], m1);
        store_array<QC, PL>(arr(GA_T3, QC), g1[0]);
        note_scale(GA_T3, sp);
        m1[0] = mwords[19 * 64];
        TB_LAYER(QC, 2, g1, g0);                 // transient_encoding.6^T -> d t2
        apply_mask<P, QC>(g0[0], m1);
        store_array<QC, PL>(arr(GA_T2, QC), g0[0]);
        note_scale(GA_T2, sp);
        m1[0] = mwords[18 * 64];
        TB_LAYER(QC, 2, g0, g1);                 // transient_encoding.4^T -> d t1
        apply_mask<P, QC>(g1[0], m1);
        store_array<QC, PL>(arr(GA_T1, QC), g1[0]);
        note_scale(GA_T1, sp);
        m1[0] = mwords[17 * 64];
        TB_LAYER(QC, 2, g1, g0);                 // transient_encoding.2^T -> d t0
        apply_mask<P, QC>(g0[0], m1);
        m1[0] = mwords[16 * 64];
        TB_LAYER(SC, 2, drgb, g1);               // static_rgb^T -> d dir_h
        apply_mask<P, QC>(g1[0], m1);
#pragma unroll
        for (int i = 0; i < QC; ++i) { cat[0][i] = g0[0][i]; cat[0][QC + i] = g1[0][i]; }
        store_array<HC, PL>(arr(GA_CAT, HC), cat[0]);
      } else {
        m1[0] = mwords[16 * 64];
        TB_LAYER(SC, 2, drgb, cat);              // static_rgb^T -> d dir_h
        apply_mask<P, QC>(cat[0], m1);
        store_array<QC, PL>(arr(GA_CAT, QC), cat[0]);
      }
      note_scale(GA_CAT, sp);
      {
        F dfin[NB][HC];
        st.lane_mul = renorm_wave<(FINE ? HC : QC)>(cat[0], sp);
        // d sigma_s joins the chain after this layer (slot 64 of final^T's input) at the scale chosen HERE: keep it inside the f16
        // range of the operand even when the colour gradients of the wave are orders of magnitude below its density gradients
        float ds = wave_max(fabsf(g[3])) * sp * kX3ActScale * st.lane_mul;
        if (ds > 2048.f) {
          int e;
          (void)frexpf(ds, &e);
          st.lane_mul *= ldexpf(1.f, 11 - e);
        }
        if constexpr (FINE) TB_LAYER(HC, 4, cat, dfin);   // [transient_encoding.0 ; dir_encoding.0]^T on the `final` columns -> d final
        else TB_LAYER(QC, 4, cat, dfin);                  // dir_encoding.0^T on the `final` columns
        sp *= st.lane_mul;
        st.lane_mul = 1.f;
#pragma unroll
        for (int i = 0; i < HC; ++i) cat2[0][i] = dfin[0][i];
#pragma unroll
        for (int i = 0; i < SC; ++i) clear_one<P>(cat2[0][HC + i]);
        set_slot<P>(cat2[0], 64, g[3] * sp);     // slot 64 of half 0: d sigma_s (pre-activation); half 1 holds 0
      }
    }
    store_array<HC + SC, PL>(arr(GA_CAT2, HC + SC), cat2[0]);
    note_scale(GA_CAT2, sp);
    F gh[NB][HC], gh2[NB][HC];
    st.lane_mul = renorm_wave<HC + SC>(cat2[0], sp);
    mk[0] = mwords[14 * 64]; mk[1] = mwords[15 * 64];
    TB_LAYER(HC + SC, 4, cat2, gh);              // [xyz_encoding_final ; static_sigma]^T -> d h8
    sp *= st.lane_mul; st.lane_mul = 1.f;
    apply_mask<P, HC>(gh[0], mk);
    store_array<HC, PL>(arr(GA_L8, HC), gh[0]);
    note_scale(GA_L8, sp);
    st.lane_mul = renorm_wave<HC>(gh[0], sp);
    mk[0] = mwords[12 * 64]; mk[1] = mwords[13 * 64];
    TB_LAYER(HC, 4, gh, gh2);                    // xyz_encoding_8^T -> d h7
    sp *= st.lane_mul; st.lane_mul = 1.f;
    apply_mask<P, HC>(gh2[0], mk);
    store_array<HC, PL>(arr(GA_L7, HC), gh2[0]);
    note_scale(GA_L7, sp);
    mk[0] = mwords[10 * 64]; mk[1] = mwords[11 * 64];
    TB_LAYER(HC, 4, gh2, gh);                    // xyz_encoding_7^T -> d h6
    apply_mask<P, HC>(gh[0], mk);
    store_array<HC, PL>(arr(GA_L6, HC), gh[0]);
    note_scale(GA_L6, sp);
    st.lane_mul = renorm_wave<HC>(gh[0], sp);
    mk[0] = mwords[8 * 64]; mk[1] = mwords[9 * 64];
    TB_LAYER(HC, 4, gh, gh2);                    // xyz_encoding_6^T -> d h5
    sp *= st.lane_mul; st.lane_mul = 1.f;
    apply_mask<P, HC>(gh2[0], mk);
    store_array<HC, PL>(arr(GA_L5, HC), gh2[0]);
    note_scale(GA_L5, sp);
    mk[0] = mwords[6 * 64]; mk[1] = mwords[7 * 64];
    TB_LAYER(HC, 4, gh2, gh);                    // xyz_encoding_5^T, the h columns -> d h4 (the encoding columns need no gradient)
    apply_mask<P, HC>(gh[0], mk);
    store_array<HC, PL>(arr(GA_L4, HC), gh[0]);
    note_scale(GA_L4, sp);
    st.lane_mul = renorm_wave<HC>(gh[0], sp);
    mk[0] = mwords[4 * 64]; mk[1] = mwords[5 * 64];
    TB_LAYER(HC, 4, gh, gh2);                    // xyz_encoding_4^T -> d h3
    sp *= st.lane_mul; st.lane_mul = 1.f;
    apply_mask<P, HC>(gh2[0], mk);
    store_array<HC, PL>(arr(GA_L3, HC), gh2[0]);
    note_scale(GA_L3, sp);
    mk[0] = mwords[2 * 64]; mk[1] = mwords[3 * 64];
    TB_LAYER(HC, 4, gh2, gh);                    // xyz_encoding_3^T -> d h2
    apply_mask<P, HC>(gh[0], mk);
    store_array<HC, PL>(arr(GA_L2, HC), gh[0]);
    note_scale(GA_L2, sp);
    st.lane_mul = renorm_wave<HC>(gh[0], sp);
    mk[0] = mwords[0 * 64]; mk[1] = mwords[1 * 64];
    TB_LAYER(HC, 4, gh, gh2);                    // xyz_encoding_2^T -> d h1
    sp *= st.lane_mul; st.lane_mul = 1.f;
    apply_mask<P, HC>(gh2[0], mk);
    store_array<HC, PL>(arr(GA_L1, HC), gh2[0]);
    note_scale(GA_L1, sp);
  }
  range_report<P>(st.rmax, a.status);
}
template <bool FINE, int PL>
__global__ __launch_bounds__(WAVES * 64, 1) void train_bwd_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  train_bwd_chain_body<FINE, PL>(a, smem, int(blockIdx.x), int(gridDim.x));
}
// The fine and the coarse network's chains in ONE launch: blocks [0, n_fine) are the fine chain's persistent grid, the rest the
// coarse chain's.  The two are independent (the coarse loss alone reaches the coarse network: rendering.py:302 detaches the
// samples), every workgroup takes a CU's whole register file, and workgroups are dispatched in block order — so the coarse chain's
// workgroups start on the CUs the fine chain leaves first.  At the reference's batch (1 536 rays of 64 + 128 samples: 1 152 fine
// tiles = 4.5 per CU, 384 coarse tiles = 1.5 per CU) that fills the half-tile tails of both chains: 6.0 tile times instead of
// 5 + 2.  One stream, one launch: the same overlap from a second HIP stream gave the time back through slower launches of every
// other kernel of the step, and hipExtAnyOrderLaunch is ignored on this part (LABBOOK R6.17).
template <int PLF>
__global__ __launch_bounds__(WAVES * 64, 1) void train_bwd_chain_pair_kernel(ChainArgs fine, ChainArgs coarse, int n_fine) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (int(blockIdx.x) < n_fine) train_bwd_chain_body<true, PLF>(fine, smem, int(blockIdx.x), n_fine);
  else train_bwd_chain_body<false, 2>(coarse, smem, int(blockIdx.x) - n_fine, int(gridDim.x) - n_fine);
}

// ------------------------------------------------------------------------------------------ launch
template <class K>
static hipError_t launch_chain(K kern, bool& attr_done, uint32_t lds, const ChainArgs& a, int n_cu, hipStream_t s) {
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  if (n_pts <= 0) return hipSuccess;
  if (n_pts >= (1LL << 31)) return hipErrorInvalidValue;
  const long long n_tiles = (n_pts + kTilePoints - 1) / kTilePoints;
  const int grid = int(n_tiles < n_cu ? n_tiles : n_cu);
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_train_forward_chain(bool fine, int planes, const ChainArgs& a, int n_cu, hipStream_t s) {
  static bool done[3] = {false, false, false};
  if (!fine) return planes == 2 ? launch_chain(train_fwd_chain_kernel<false, 2>, done[0], 3 * kFwdStride, a, n_cu, s) : hipErrorInvalidValue;
  if (planes == 1) return launch_chain(train_fwd_chain_kernel<true, 1>, done[1], 3 * kFwdStride, a, n_cu, s);
  if (planes == 2) return launch_chain(train_fwd_chain_kernel<true, 2>, done[2], 3 * kFwdStride, a, n_cu, s);
  return hipErrorInvalidValue;
}
hipError_t launch_train_backward_chain(bool fine, int planes, const ChainArgs& a, int n_cu, hipStream_t s) {
  static bool done[3] = {false, false, false};
  if (!fine) return planes == 2 ? launch_chain(train_bwd_chain_kernel<false, 2>, done[0], 3 * kBwdStride, a, n_cu, s) : hipErrorInvalidValue;
  if (planes == 1) return launch_chain(train_bwd_chain_kernel<true, 1>, done[1], 3 * kBwdStride, a, n_cu, s);
  if (planes == 2) return launch_chain(train_bwd_chain_kernel<true, 2>, done[2], 3 * kBwdStride, a, n_cu, s);
  return hipErrorInvalidValue;
}

hipError_t launch_train_backward_chain_pair(int planes_fine, const ChainArgs& fine, const ChainArgs& coarse, int n_cu, hipStream_t s) {
  static bool done[2] = {false, false};
  const long long pf = (long long)fine.n_rays * fine.n_samples, pc = (long long)coarse.n_rays * coarse.n_samples;
  if (pf <= 0 || pc <= 0 || pf >= (1LL << 31) || (planes_fine != 1 && planes_fine != 2)) return hipErrorInvalidValue;
  const long long tf = (pf + kTilePoints - 1) / kTilePoints, tc = (pc + kTilePoints - 1) / kTilePoints;
  const int gf = int(tf < n_cu ? tf : n_cu), gc = int(tc < n_cu ? tc : n_cu);   // each chain's own persistent grid, as launch_chain
  const void* kern = planes_fine == 1 ? reinterpret_cast<const void*>(train_bwd_chain_pair_kernel<1>)
                                      : reinterpret_cast<const void*>(train_bwd_chain_pair_kernel<2>);
  bool& attr_done = done[planes_fine - 1];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(3 * kBwdStride));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (planes_fine == 1) hipLaunchKernelGGL(train_bwd_chain_pair_kernel<1>, dim3(gf + gc), dim3(WAVES * 64), 3 * kBwdStride, s, fine, coarse, gf);
  else hipLaunchKernelGGL(train_bwd_chain_pair_kernel<2>, dim3(gf + gc), dim3(WAVES * 64), 3 * kBwdStride, s, fine, coarse, gf);
  return hipGetLastError();
}

}  // namespace fused
}  // namespace dfn
