// nerfh_mlp.hip — the NeRF-H MLP on the CDNA4 matrix cores (gfx950 only).
//
// Persistent workgroups (variant 1: two 4-wave workgroups per CU, running unsynchronised so one's
// epilogue VALU overlaps the other's MFMAs; variant 0: one 8-wave workgroup).  Each wavefront owns NB blocks of 32 sample
// points and carries their activations through the WHOLE network in registers: a Linear layer
// is computed transposed (weights = MFMA A operand, activations = B operand, see
// nerfh_layout.h), so the fp32 C fragments of layer l, after bias/ReLU and conversion, are the
// B operand of layer l+1 — no LDS or HBM round trip for activations.  Positional encoding,
// the point o + d*z, Softplus/Sigmoid heads are fused in.  Only the weights move: every
// layer's pre-permuted MFMA A-fragments stream L2 -> LDS with direct-to-LDS DMA
// (global_load_lds_dwordx4) into a double buffer shared by the workgroup's waves, one barrier per staging unit.
//
// Replaces (reference, /root/reference/script/): models/rendering.py:269-292,305-313 (points),
// models/nerfw.py:15-95 (run_network_NeRFW), :105-133 (Embedder.embed), :297-354 (NeRFW.forward).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nerfh_device.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"
#include "mfma_frag.h"

#include "nerfh_mlp_core.h"
#include "nerfh_mask.h"

namespace dfn {

// Head activations per arithmetic mode: f16 = hardware transcendentals (1e-6), split-f16 = the fp32-grade hardware forms
// (exp_hw / rcp_nr, 1.5e-7: nerfh_device.h), exact fp32 = libm.
template <class P, bool FAST> DFN_DEV float head_softplus(float v) { return FAST ? softplus_fast(v) : (P::kSplit ? softplus_hw(v) : softplus(v)); }
template <class P, bool FAST> DFN_DEV float head_sigmoid(float v) { return FAST ? sigmoid_fast(v) : (P::kSplit ? sigmoid_hw(v) : sigmoid(v)); }
template <class P, bool FAST> DFN_DEV float head_exp(float v) { return FAST ? __expf(v) : (P::kSplit ? exp_hw(v) : expf(v)); }

// three staging buffers in the ring (a ring of four with the DMA issued BEFORE the mid-unit barrier by the early waves fits the
// split-f16 units and was measured: +0.4 %, not kept)
template <class P, int W> constexpr uint32_t ring_slots() { return 3; }
// staging ring + per-wave next-tile input slots (8 dwords x 64 lanes per 64 points: z, o, d, next z)
template <class P, int UMB, int WAVES, int NB, int W = kWidth> constexpr uint32_t lds_bytes() {
  return ring_slots<P, W>() * max_unit_bytes<P>(UMB, W) + WAVES * ((NB * 32 + 63) / 64) * 8 * 256;
}

// Workgroups per CU the register allocation is sized for.  netwidth 256 in split-f16 / exact fp32 holds 2 x 128 registers of
// activations per point block: four waves per workgroup, ONE workgroup per CU (512 registers per lane).
template <class P, int WAVES, int NB, int W> constexpr int mlp_min_blocks() {
  if (W > kWidth && (P::kSplit || P::kSlotsPerChunk == 1 || NB >= 2)) return 1;
  return WAVES * NB >= 12 ? (WAVES == 8 ? 2 : 1) : 2;
}

// ------------------------------------------------------------------------------------------
template <class P, bool FAST, int WAVES, int UMB, int NB, bool PIPE, int W = kWidth>
__global__ __launch_bounds__(WAVES * 64, (mlp_min_blocks<P, WAVES, NB, W>())) void nerfh_coarse_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PPT = WAVES * NB * 32;
  using F = typename FragOf<P>::type;
  Stager st;
  st.blob = a.blob; st.tab = a.tab; st.n_units = a.n_units; st.u = 0;
  st.waves = WAVES;
  st.dma_waves = a.dma_waves > 0 && a.dma_waves < WAVES ? a.dma_waves : WAVES;
  st.in_scale = a.in_scale;
  st.out_scale = 1.f / a.in_scale;
  st.lane_mul = 1.f;
  st.rmax = 0;
  st.t_sync = st.t_wait = 0;
  st.trace = nullptr;
  st.n_trace = 0;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DFN_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_pro = 0;
  if (a.timing && blockIdx.x == 7) st.trace = a.timing + 8192 * 4 + st.wave * 192;  // after the per-wave totals
#endif
  const int p = st.lane & 31, h = st.lane >> 5;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  const long long n_tiles = (n_pts + PPT - 1) / PPT;
  long long tile = blockIdx.x;
  if (tile >= n_tiles) return;
  stage_prime(st, smem, max_unit_bytes<P>(UMB, W), ring_slots<P, W>());
  for (; tile < n_tiles; tile += gridDim.x) {
    st.more = tile + gridDim.x < n_tiles;
    float x[NB][3];
    uint32_t pt[NB];   // launch_one() bounds a launch to < 2^31 points: 32-bit indices (64-bit ones cost 2 registers each and spilled)
    const uint32_t npts = uint32_t(n_pts);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      pt[nb] = uint32_t(tile) * uint32_t(PPT) + st.wave * (NB * 32) + nb * 32 + p;
      const uint32_t q = pt[nb] < npts ? pt[nb] : npts - 1;
      const uint32_t ray = q / uint32_t(a.n_samples);
      const int i = int(q - ray * uint32_t(a.n_samples));
      const float z = coarse_z_at(i, a.n_samples, a.near, a.far, a.lindisp != 0);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        x[nb][c] = add_rn(a.rays_o[ray * 3 + c], mul_rn(a.rays_d[ray * 3 + c], z));
    }
    constexpr bool CY = PIPE && P::kSlotsPerChunk == 8, MERGE = UMB >= 8;
    F hid[NB][chunks_of<P>(W / 2)];
    f32x16 carry[NB];
    trunk<P, UMB, PIPE, FAST, NB, W>(st, smem, x, hid, carry);
    f32x16 head[NB];
    F dummy[NB][chunks_of<P>(16)];
    const float* const norb[NB] = {};
    layer<P, UMB, PIPE, NB, chunks_of<P>(W / 2), 0, false, true, false, !MERGE, (CY ? 6 : -1), true, false>(st, smem, hid, dummy, head, norb, carry);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if (h == 0 && pt[nb] < npts) a.out[pt[nb]] = head_softplus<P, FAST>(head[nb][0]);
  }
  range_report<P>(st.rmax, a.status);
}

// MASKS: the saving forward of the gradient path (dfn_mlp_fine_saving): the same kernel, recording one ReLU sign bit per hidden unit
// for nerfh_fine_backward_kernel's backward-only pass (its own forward-only mode ran without the pipelined conversion: 3.64 ms for the
// DFNet_dm step's 3.7 M points against this kernel's 3.1).
template <class P, bool FAST, int WAVES, int UMB, int NB, bool PIPE, int W = kWidth, bool MASKS = false>
__global__ __launch_bounds__(WAVES * 64, (mlp_min_blocks<P, WAVES, NB, W>())) void nerfh_fine_kernel(MlpArgs a) {
  static_assert(!MASKS || (P::kSplit && NB == 1 && WAVES == 8 && W == kWidth), "the sign masks follow the gradient kernel's tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PPT = WAVES * NB * 32;
  constexpr int HC = chunks_of<P>(W / 2), QC = chunks_of<P>(W / 4);
  constexpr int MBW = W / 32, MBQ = W / 64;       // M-blocks of a W-wide / a W/2-wide layer
  constexpr uint32_t USTRIDE = max_unit_bytes<P>(UMB, W);
  constexpr int PF_ROUNDS = (NB * 32 + 63) / 64;  // 64-point rounds of the next-tile input prefetch
  using F = typename FragOf<P>::type;
  Stager st;
  st.blob = a.blob; st.tab = a.tab; st.n_units = a.n_units; st.u = 0;
  st.waves = WAVES;
  st.dma_waves = a.dma_waves > 0 && a.dma_waves < WAVES ? a.dma_waves : WAVES;
  st.in_scale = a.in_scale;
  st.out_scale = 1.f / a.in_scale;
  st.lane_mul = 1.f;
  st.rmax = 0;
  st.t_sync = st.t_wait = 0;
  st.trace = nullptr;
  st.n_trace = 0;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DFN_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_pro = 0;
  if (a.timing && blockIdx.x == 7) st.trace = a.timing + 8192 * 4 + st.wave * 192;  // after the per-wave totals
#endif
  const int p = st.lane & 31, h = st.lane >> 5;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  const long long n_tiles = (n_pts + PPT - 1) / PPT;
  long long tile = blockIdx.x;
  if (tile >= n_tiles) return;
  stage_prime(st, smem, USTRIDE, ring_slots<P, W>());
  // Tile inputs.  The first tile's are loaded normally; every later tile's are PREFETCHED during the
  // previous tile's small layers with direct-to-LDS loads and only
  // waited for at the end of that tile, so the HBM latency of z / o / d is off the critical path.
  float zin[NB], znext[NB], oin[NB][3], din[NB][3];
  // launch_one() guarantees n_pts < 2^31: point / ray indices are 32-bit (as 64-bit values and as per-ray table POINTERS held
  // across the trunk they cost 16 registers at NB = 2 and the f16 kernel spilled them: any scratch use costs this kernel clock)
  uint32_t pt[NB], ray_of[NB];
  const uint32_t npts = uint32_t(n_pts);
  auto tile_coords = [&](long long t) {
    uint32_t lp = st.lane;
    asm volatile("" : "+v"(lp));   // the lane's offset inside the tile is formed here (hoisted as a loop invariant, it was spilled)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      pt[nb] = uint32_t(t) * uint32_t(PPT) + st.wave * (NB * 32) + nb * 32 + (lp & 31);
      const uint32_t q = pt[nb] < npts ? pt[nb] : npts - 1;
      ray_of[nb] = q / uint32_t(a.n_samples);
    }
  };
  tile_coords(tile);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const uint32_t q = pt[nb] < npts ? pt[nb] : npts - 1;
    zin[nb] = a.z[q];
    znext[nb] = a.z[q + 1 < npts ? q + 1 : q];
#pragma unroll
    for (int c = 0; c < 3; ++c) { oin[nb][c] = a.rays_o[ray_of[nb] * 3 + c]; din[nb][c] = a.rays_d[ray_of[nb] * 3 + c]; }
  }
  for (; tile < n_tiles; tile += gridDim.x) {
    st.more = tile + gridDim.x < n_tiles;
    float x[NB][3];
    uint32_t pt_cur[NB], ray_cur[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      pt_cur[nb] = pt[nb];
      ray_cur[nb] = ray_of[nb];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[nb][c] = add_rn(oin[nb][c], mul_rn(din[nb][c], zin[nb]));
    }
    // per-ray table rows: the pointers are formed where they are used (opaque ray index: not hoisted above the trunk)
    auto ray_table = [&](const float* (&rb)[NB], int half_table) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint32_t r = ray_cur[nb];
        asm volatile("" : "+v"(r));
        rb[nb] = a.ray_bias + (size_t)r * ray_bias_floats(W) + half_table * (ray_bias_floats(W) / 2);
      }
    };
    const float* const norb[NB] = {};
    F hid[NB][HC];
#ifdef DFN_TIMING
    {
      const unsigned long long c0 = __builtin_amdgcn_s_memtime();
      float keep = 0.f;
      for (int nb = 0; nb < NB; ++nb) keep += x[nb][0] + x[nb][1] + x[nb][2];
      asm volatile("" ::"v"(keep));   // inputs have arrived
      t_pro += __builtin_amdgcn_s_memtime() - c0;
    }
#endif
    constexpr bool CY = PIPE && P::kSlotsPerChunk == 8, MERGE = UMB >= 8;
    f32x16 carry[NB];
    [[maybe_unused]] uint32_t* mwords = nullptr;
    if constexpr (MASKS) mwords = a.masks + (size_t(tile) * WAVES + st.wave) * (kBwdMaskWords * 64) + st.lane;
    // sign words: trunk layer l -> 2l, 2l + 1; dir_encoding 16; transient_encoding.{0,2,4,6} 17..20 (nerfh_bwd.hip reads them back)
    auto sign128 = [&](int l, F (&h)[HC]) {
      if constexpr (MASKS) {
        uint32_t m[2];
        relu_mask<P, HC>(h, m);
        mwords[(2 * l) * 64] = m[0];
        mwords[(2 * l + 1) * 64] = m[1];
      }
    };
    auto sign64 = [&](int word, F (&h)[QC]) {
      if constexpr (MASKS) {
        uint32_t m[1];
        relu_mask<P, QC>(h, m);
        mwords[word * 64] = m[0];
      }
    };
    if constexpr (MASKS) trunk<P, UMB, PIPE, FAST, NB, W>(st, smem, x, hid, carry, sign128);
    else trunk<P, UMB, PIPE, FAST, NB, W>(st, smem, x, hid, carry);
    // xyz_encoding_final (no activation) + static_sigma
    F fin[NB][HC];
    f32x16 head[NB];
    layer<P, UMB, PIPE, NB, HC, MBW, false, true, false, true, (CY ? 6 : -1), true, false>(st, smem, hid, fin, head, norb, carry);
    sign128(7, hid[0]);   // layer 8's output, completed inside the layer above
    float o[NB][9];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o[nb][3] = head_softplus<P, FAST>(head[nb][0]);
    // dir_encoding (per-ray bias = b + W[:,128:] [pe_dir, a]) -> static_rgb
    {
      F de[NB][QC], dummy[NB][chunks_of<P>(16)];
      const float* rb_dir[NB];
      ray_table(rb_dir, 0);
      layer<P, UMB, PIPE, NB, HC, MBQ, true, false, true, true, -1, true, CY>(st, smem, fin, de, head, rb_dir, carry);
      layer<P, UMB, PIPE, NB, QC, 0, false, true, false, !MERGE, (CY ? 2 : -1), true, false>(st, smem, de, dummy, head, norb, carry);
      sign64(16, de[0]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[nb][c] = head_sigmoid<P, FAST>(head[nb][c]);
    }
    // transient branch
    {
      F t0[NB][QC], t1[NB][QC], dummy[NB][chunks_of<P>(16)];
      const float* rb_tr[NB];
      ray_table(rb_tr, 1);
      layer<P, UMB, PIPE, NB, HC, MBQ, true, false, true, true, -1, true, CY>(st, smem, fin, t0, head, rb_tr, carry);
      if (st.more) {
        // Prefetch the next tile's inputs by LDS-DMA (no destination registers).  Issued AFTER this unit's mid_sync so
        // that nothing younger than a weight DMA is ever waited for before the tile's end:
        // lane l of round r fetches z, o, d and the next sample's z of the wave's point 64 r + l into this wave's LDS slot.
        const uint32_t base = uint32_t(tile + gridDim.x) * uint32_t(PPT) + st.wave * (NB * 32);
        char* slot = smem + ring_slots<P, W>() * USTRIDE + st.wave * (PF_ROUNDS * 8 * 256);
#pragma unroll
        for (int r = 0; r < PF_ROUNDS; ++r) {
          const uint32_t ptn = base + r * 64 + st.lane;
          const uint32_t q = ptn < npts ? ptn : npts - 1;
          const uint32_t ray = q / uint32_t(a.n_samples);
          lds_dma_b32(a.z + q, slot + (r * 8) * 256);
          lds_dma_b32(a.z + (q + 1 < npts ? q + 1 : q), slot + (r * 8 + 7) * 256);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            lds_dma_b32(a.rays_o + ray * 3 + c, slot + (r * 8 + 1 + c) * 256);
            lds_dma_b32(a.rays_d + ray * 3 + c, slot + (r * 8 + 4 + c) * 256);
          }
        }
      }
      layer<P, UMB, PIPE, NB, QC, MBQ, true, false, false, !MERGE, (CY ? 2 : -1), true, CY>(st, smem, t0, t1, head, norb, carry);
      sign64(17, t0[0]);
      layer<P, UMB, PIPE, NB, QC, MBQ, true, false, false, !MERGE, (CY ? 2 : -1), true, CY>(st, smem, t1, t0, head, norb, carry);
      sign64(18, t1[0]);
      layer<P, UMB, PIPE, NB, QC, MBQ, true, false, false, !MERGE, (CY ? 2 : -1), true, CY>(st, smem, t0, t1, head, norb, carry);
      sign64(19, t0[0]);
      layer<P, UMB, PIPE, NB, QC, 0, false, true, false, !MERGE, (CY ? 2 : -1), true, false>(st, smem, t1, dummy, head, norb, carry);
      sign64(20, t1[0]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[nb][4 + c] = head_sigmoid<P, FAST>(head[nb][c]);
        o[nb][7] = head_softplus<P, FAST>(head[nb][3]);
        o[nb][8] = head_softplus<P, FAST>(head[nb][4]);  // C register 4 of half 0 = row 8 = transient_beta
      }
    }
    if (a.partial) {
      // Fused compositing (n_samples % (32 NB) == 0): this wave holds 32 NB consecutive samples of ONE ray, sample
      // 32 nb + p on lane p of half 0.  Transmittance factorises over segments, so the wave composites its
      // segment locally (products P and sums relative to the segment start) and a tiny combine pass chains
      // the segments of a ray (nerfh_stages.hip: composite_combine_kernel).  `raw` never reaches HBM.
      const bool live = h == 0;
      float Pj = 1.f, Ps = 1.f;                    // running products through the previous blocks of the segment
      float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_dso = 0.f, s_dj = 0.f, s_beta = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint32_t smp = pt_cur[nb] - ray_cur[nb] * uint32_t(a.n_samples);
        const float delta = smp + 1 == uint32_t(a.n_samples) ? 1e2f : sub_rn(znext[nb], zin[nb]);
        const float sg_s = o[nb][3], sg_t = o[nb][7];
        const float as = live ? sub_rn(1.f, head_exp<P, FAST>(-mul_rn(delta, sg_s))) : 0.f;
        const float at = live ? sub_rn(1.f, head_exp<P, FAST>(-mul_rn(delta, sg_t))) : 0.f;
        const float aj = live ? sub_rn(1.f, head_exp<P, FAST>(-mul_rn(delta, add_rn(sg_s, sg_t)))) : 0.f;
        // scans over the block's 32 samples on the DPP crossbar (lanes 32..63 carry alpha = 0: the identity), five VALU
        // instructions each instead of six dependent ds_bpermute round trips (nerfh_device.h)
        const float ij = scan32_prod(1.f - aj), is = scan32_prod(1.f - as);
        const float ej = lane_shr1(ij, 1.f), es = lane_shr1(is, 1.f);   // exclusive products (lane 0: 1)
        const float Tj = Pj * ej, Ts = Ps * es;
        const float ws = as * Tj, wt = at * Tj, wj = aj * Tj;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_rgb[c] += live ? ws * o[nb][c] + wt * o[nb][4 + c] : 0.f;
        s_acc += wj;
        s_dso += as * Ts * zin[nb];
        s_dj += wj * zin[nb];
        s_beta += live ? wt * o[nb][8] : 0.f;
        Pj *= read_lane31(ij);   // lane 31 holds the block's full product
        Ps *= read_lane31(is);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] = read_lane31(scan32_sum(s_rgb[c]));   // lanes 32..63 contribute zeros
      s_acc = read_lane31(scan32_sum(s_acc)); s_dso = read_lane31(scan32_sum(s_dso));
      s_dj = read_lane31(scan32_sum(s_dj)); s_beta = read_lane31(scan32_sum(s_beta));
      if (st.lane == 0 && pt_cur[0] < npts) {
        f32x4* dst = reinterpret_cast<f32x4*>(a.partial + (size_t)(pt_cur[0] / uint32_t(NB * 32)) * 12);   // one segment per wave
        dst[0] = f32x4{s_rgb[0], s_rgb[1], s_rgb[2], s_acc};
        dst[1] = f32x4{s_dso, s_dj, s_beta, Pj};
        dst[2] = f32x4{Ps, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (h == 0 && pt_cur[nb] < npts) {
          float* dst = a.out + (size_t)pt_cur[nb] * 9;
#pragma unroll
          for (int c = 0; c < 9; ++c) dst[c] = o[nb][c];
        }
    }
    if (st.more) {  // pick up the prefetched inputs of the next tile (this wave's own LDS slot: no barrier needed)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      asm volatile("" ::: "memory");
      tile_coords(tile + gridDim.x);
      const char* slot = smem + ring_slots<P, W>() * USTRIDE + st.wave * (PF_ROUNDS * 8 * 256);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        int loc = nb * 32 + p;
        asm volatile("" : "+v"(loc));   // formed here: as a loop invariant it was hoisted to the prologue and spilled
        const int r = loc >> 6, l = loc & 63;
        const float* f = reinterpret_cast<const float*>(slot + r * 8 * 256) + l;
        zin[nb] = f[0];
        znext[nb] = f[7 * 64];
#pragma unroll
        for (int c = 0; c < 3; ++c) { oin[nb][c] = f[(1 + c) * 64]; din[nb][c] = f[(4 + c) * 64]; }
      }
    }
  }
  range_report<P>(st.rmax, a.status);
#ifdef DFN_TIMING
  if (a.timing && st.lane == 0) {
    unsigned long long* t = a.timing + (blockIdx.x * WAVES + st.wave) * 4;
    t[0] = __builtin_amdgcn_s_memtime() - t_begin;
    t[1] = st.t_wait;
    t[2] = st.t_sync;
    t[3] = t_pro;
  }
#endif
}

// ------------------------------------------------------------------------------------------
template <class P, bool FAST, int WAVES, int UMB, int NB, int WG_PER_CU, bool PIPE, int W = kWidth>
static hipError_t launch_one(bool fine, const MlpArgs& a, int n_cu, hipStream_t stream) {
  constexpr int PPT = WAVES * NB * 32;
  const long long n_pts = (long long)a.n_rays * a.n_samples;
  if (n_pts <= 0) return hipSuccess;
  if (n_pts >= (1LL << 31)) return hipErrorInvalidValue;  // kernels index points with 32 bits; callers chunk
  const long long n_tiles = (n_pts + PPT - 1) / PPT;
  const long long slots = (long long)n_cu * WG_PER_CU;  // resident workgroups
  const int grid = int(n_tiles < slots ? n_tiles : slots);
  const uint32_t lds = lds_bytes<P, UMB, WAVES, NB, W>();
  auto kern = fine ? nerfh_fine_kernel<P, FAST, WAVES, UMB, NB, PIPE, W> : nerfh_coarse_kernel<P, FAST, WAVES, UMB, NB, PIPE, W>;
  int slot = fine ? 1 : 0;
  if constexpr (P::kSplit && NB == 1 && WAVES == 8 && W == kWidth && PIPE) {
    if (fine && a.masks) { kern = nerfh_fine_kernel<P, FAST, WAVES, UMB, NB, PIPE, W, true>; slot = 2; }
  } else if (a.masks) return hipErrorInvalidValue;
  static bool attr_done[3] = {false, false, false};
  if (!attr_done[slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done[slot] = true;
  }
  MlpArgs b = a;
  if (b.dma_waves == 0) {  // DFN_DMA_WAVES=n: A/B aid
    static int env = -1;
    if (env < 0) { const char* e = getenv("DFN_DMA_WAVES"); env = e ? atoi(e) : 0; }
    b.dma_waves = env > 0 ? env : (WAVES == 8 ? 4 : WAVES);  // 8-wave workgroups: the four older waves idle a third of their time at the unit barriers
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, b);
  return hipGetLastError();
}

// variant 0: 8 waves x NB 2, 1 WG/CU, unit = layer, epilogue pipelined into the next M-block's MFMAs
// variant 1: 4 waves x NB 2, 2 WG/CU, unit = 2 M-blocks  (2 waves/SIMD from different workgroups)
// variant 2: 4 waves x NB 3, 1 WG/CU, unit = layer      (1 wave/SIMD, 512 registers, pipelined epilogue)
// variant 3: variant 0 without the pipelined epilogue (A/B reference)
// netwidth 256: the plain variants (one point block per wave, no pipelined epilogue); staging units of 2 M-blocks (f16) /
// 1 M-block (f32, split-f16) keep three buffers inside the 160 KB of LDS (a 256 x 256 f16 layer is 128 KB).
hipError_t launch_mlp(bool fine, int prec, int variant, const MlpArgs& a, int n_cu, hipStream_t stream, int width) {
  if (width == 256) {
    if (prec == 0) {
      return launch_one<PrecF16, true, 8, unit_mb_w256<PrecF16>(), 1, 1, false, 256>(fine, a, n_cu, stream);
    }
    if (prec == 2) return launch_one<PrecX3, false, 4, unit_mb_w256<PrecX3>(), 1, 1, false, 256>(fine, a, n_cu, stream);
    return launch_one<PrecF32, false, 4, unit_mb_w256<PrecF32>(), 1, 1, false, 256>(fine, a, n_cu, stream);
  }
  if (width != kWidth) return hipErrorInvalidValue;
  if (prec == 0) {
    if (variant == 0) return launch_one<PrecF16, true, 8, 8, 2, 1, true>(fine, a, n_cu, stream);
    if (variant == 1) return launch_one<PrecF16, true, 4, 2, 2, 1, true>(fine, a, n_cu, stream);
    if (variant == 2) return launch_one<PrecF16, true, 4, 8, 3, 1, true>(fine, a, n_cu, stream);
    return launch_one<PrecF16, true, 8, 8, 2, 1, false>(fine, a, n_cu, stream);
  }
  if (prec == 2) {  // split-f16: variant 3 = without the pipelined epilogue (A/B reference), every other variant with it
    if (variant == 3) return launch_one<PrecX3, false, 8, unit_mb<PrecX3>(0), 1, 1, false>(fine, a, n_cu, stream);
    return launch_one<PrecX3, false, 8, unit_mb<PrecX3>(0), 1, 1, true>(fine, a, n_cu, stream);
  }
  if (variant == 1) return launch_one<PrecF32, false, 4, 1, 1, 1, false>(fine, a, n_cu, stream);
  return launch_one<PrecF32, false, 8, 1, 1, 1, false>(fine, a, n_cu, stream);
}

}  // namespace dfn
