// nerfh_fused_wgrad.hip — weight gradients of the fused NeRF-H training step as a stream over the stored operand arrays, plus
// the small kernels around the chains: per-step weight packing, per-ray bias tables from the master weights, per-ray sums of a
// stored gradient array (gfx950 only).  Layouts and the scheme: nerfh_fused_train.h.
//
// Replaces the weight-gradient third of loss.backward() in /root/reference/script/run_nerf.py:64 for the Linear layers of
// models/nerfw.py:259-295: dW_l[n, k] = sum_p G_l[p, n] X_l[p, k], db_l[n] = sum_p G_l[p, n].
//
// wgrad_stream_kernel<PL>: one workgroup = (job, chunk of wave-tiles).  Per wave-tile the job's G and X chunks (PL planes of 1 KiB
// each: hi | lo as the chain kernels split them, or the one f16 plane of the fine network) are DMA-ed HBM -> LDS into a ring of 2-4 stages (global_load_lds_dwordx4 nt, no registers);
// the [point][feature] image is read back as [feature][point] MFMA operands by ds_read_b64_tr_b16 (a 16-lane group reads
// 4 points x 16 features = 128 contiguous bytes: conflict-free), three v_mfma_f32_32x32x16_f16 per product and 16-point half;
// the wave-tile's fp32 block is folded into the master accumulator at the tile's power-of-two scale.  The kernel is bound by
// the stream (4 bytes per stored element, read once): a 128 x 128 layer moves 32 KiB per 32 points against 120 MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_fused_train.h"
#include "nerfh_mlp_core.h"

namespace dfn {
namespace fused {

namespace {
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 4, MAXB = 5;   // two 4-wave workgroups per CU run unsynchronised: one's barrier / LDS waits under the other's MFMAs
// LDS image of a stage: chunk c at c * kLdsChunk.  The 128-byte skew puts the two chunks of a 32-feature pair on different bank
// halves: the two 16-lane groups of a transposed read then touch 2 x 128 distinct bytes of a 256-byte bank row (conflict-free).
template <int PL> constexpr uint32_t lds_chunk() { return 1024u * PL + 128u; }

template <int OFF>
DFN_DEV u32x2 ds_tr16(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// 32 features (a chunk pair at LDS address cb; cb already holds the lane's part) x 16 points (P0 = 0 / 16) -> MFMA operand
// (hi, lo): lane (feature = lane & 31, k-half = lane >> 5) receives points P0 + 8 (lane >> 5) + 0..7.
struct Frag { u32x2 h0, h1, l0, l1; };
template <int P0, int PL>
DFN_DEV Frag issue_frag(uint32_t cb) {
  Frag f;
  f.h0 = ds_tr16<P0 * 32>(cb);
  f.h1 = ds_tr16<P0 * 32 + 128>(cb);
  if constexpr (PL == 2) {
    f.l0 = ds_tr16<P0 * 32 + 1024>(cb);
    f.l1 = ds_tr16<P0 * 32 + 1024 + 128>(cb);
  } else {
    f.l0 = f.h0; f.l1 = f.h1;   // unused
  }
  return f;
}
DFN_DEV half8 join(u32x2 a, u32x2 b) { return __builtin_bit_cast(half8, u32x4{a[0], a[1], b[0], b[1]}); }

DFN_DEV void wait_vmcnt(int n) {   // s_waitcnt vmcnt(n), n wave-uniform (gfx9 encoding: vmcnt[3:0] in bits 3:0, lgkmcnt / expcnt left open)
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
    case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
    case 13: __builtin_amdgcn_s_waitcnt(0x0F7D); break;
    case 14: __builtin_amdgcn_s_waitcnt(0x0F7E); break;
    default: __builtin_amdgcn_s_waitcnt(0x0F7F); break;
  }
  asm volatile("" ::: "memory");
}
// The stored operands are read exactly once: non-temporal (measured on one box: 2.27 -> 2.15 ms for the backward pass; the stream
// alone 4.5 TB/s, the products alone about as long — the two overlap once two workgroups share a CU).
DFN_DEV void lds_dma_b128_nt(const void* gptr, const char* lds_dst) {
  const uint32_t off = (uint32_t)(size_t)DFN_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gptr), "s"(off) : "memory");
}
typedef const float __attribute__((address_space(4))) const_f32;
DFN_DEV float scalar_f32(const float* p) { return *reinterpret_cast<const_f32*>(reinterpret_cast<uint64_t>(p)); }
}  // namespace

namespace {
// Operands of one output block of one wave-tile: G chunk pair (A) and X chunk pair (B), both 16-point halves, hi | lo.
struct BlockOps { Frag a0, a1, b0, b1; };
template <int PL>
DFN_DEV void issue_block(BlockOps& o, uint32_t ga, uint32_t xa) {
  o.a0 = issue_frag<0, PL>(ga);
  o.a1 = issue_frag<16, PL>(ga);
  o.b0 = issue_frag<0, PL>(xa);
  o.b1 = issue_frag<16, PL>(xa);
}
template <int PL>
DFN_DEV void wait_block(BlockOps& o) {   // every LDS read issued so far has returned (the asm ties the registers to the wait)
  if constexpr (PL == 2)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.a0.h0), "+v"(o.a0.h1), "+v"(o.a0.l0), "+v"(o.a0.l1), "+v"(o.a1.h0), "+v"(o.a1.h1), "+v"(o.a1.l0), "+v"(o.a1.l1),
                   "+v"(o.b0.h0), "+v"(o.b0.h1), "+v"(o.b0.l0), "+v"(o.b0.l1), "+v"(o.b1.h0), "+v"(o.b1.h1), "+v"(o.b1.l0), "+v"(o.b1.l1)
                 :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.a0.h0), "+v"(o.a0.h1), "+v"(o.a1.h0), "+v"(o.a1.h1), "+v"(o.b0.h0), "+v"(o.b0.h1), "+v"(o.b1.h0), "+v"(o.b1.h1)
                 :: "memory");
}
// hi*hi + hi*lo + lo*hi (one plane: the single product) over the tile's 32 points, folded into the master accumulator at the tile's scale
template <int PL>
DFN_DEV void mma_block(const BlockOps& o, bool bias, float inv, f32x16& master) {
  const half8 ones = {1, 1, 1, 1, 1, 1, 1, 1}, zeros = {0, 0, 0, 0, 0, 0, 0, 0};
  const half8 ah0 = join(o.a0.h0, o.a0.h1), al0 = join(o.a0.l0, o.a0.l1), ah1 = join(o.a1.h0, o.a1.h1), al1 = join(o.a1.l0, o.a1.l1);
  const half8 bh0 = bias ? ones : join(o.b0.h0, o.b0.h1), bl0 = bias ? zeros : join(o.b0.l0, o.b0.l1);
  const half8 bh1 = bias ? ones : join(o.b1.h0, o.b1.h1), bl1 = bias ? zeros : join(o.b1.l0, o.b1.l1);
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc, 0, 0, 0);
  if constexpr (PL == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) master[r] = fmaf(acc[r], inv, master[r]);
}
}  // namespace

#ifndef DFN_FW_TPS1
#define DFN_FW_TPS1 2
#endif
// TPS wave-tiles per ring stage: with one plane a wave-tile is half the bytes behind the same barrier, counted wait and address
// arithmetic — two tiles per stage give the one-plane stream the per-iteration payload of the two-plane one.
template <int PL>
__global__ __launch_bounds__(WAVES * 64, 2) void wgrad_stream_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t kLdsChunk = lds_chunk<PL>();
  constexpr int TPS = PL == 1 ? DFN_FW_TPS1 : 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int j = 0;
  while (j + 1 < a.n_jobs && int(blockIdx.x) >= a.job[j + 1].first_wg) ++j;
  const WJob& jb = a.job[j];
  const int kcg = jb.kcg, kcx0 = jb.kcx0, kcx1 = jb.kcx1;
  const int kc_all = kcg + kcx0 + kcx1;
  const uint32_t tile_bytes = uint32_t(kc_all) * kLdsChunk, stage_bytes = TPS * tile_bytes;
  int D = int(kWgradLdsBytes / stage_bytes);
  D = D > 4 ? 4 : D;
  const int chunk = int(blockIdx.x) - jb.first_wg;
  const int wt0 = chunk * jb.wt_per_chunk;
  int n_tiles = a.n_wt - wt0;
  n_tiles = n_tiles > jb.wt_per_chunk ? jb.wt_per_chunk : n_tiles;
  if (n_tiles <= 0) return;
  const int n_it = (n_tiles + TPS - 1) / TPS;
  const int np = kc_all * PL;                      // 1 KiB pieces per wave-tile
  const int my_np = TPS * ((np - wave + WAVES - 1) / WAVES);   // ... this wave issues per stage
  auto issue = [&](int it) {
#pragma unroll
    for (int sub = 0; sub < TPS; ++sub) {
      int tl = it * TPS + sub;
      tl = tl < n_tiles ? tl : n_tiles - 1;        // (an odd tail re-reads the last tile: the counted waits stay uniform; it is not multiplied)
      const size_t wt = size_t(wt0 + tl);
      char* dst = smem + uint32_t(it % D) * stage_bytes + sub * tile_bytes;
      for (int i = wave; i < np; i += WAVES) {
        const char* src;
        if (i < PL * kcg) src = jb.g + ((wt * size_t(jb.g_stride) + jb.g_chunk0) * PL + i) * 1024;
        else if (i < PL * (kcg + kcx0)) src = jb.x0 + (wt * size_t(PL * kcx0) + (i - PL * kcg)) * 1024;
        else src = jb.x1 + (wt * size_t(PL * kcx1) + (i - PL * (kcg + kcx0))) * 1024;
        lds_dma_b128_nt(src + lane * 16, dst + (i / PL) * kLdsChunk + (i % PL) * 1024);
      }
    }
  };
  for (int it = 0; it < D - 1 && it < n_it; ++it) issue(it);

  // this wave's output blocks: a contiguous run of `per` blocks b = gb * (nb_x + 1) + xb; xb == nb_x: the bias column block
  // (B operand = ones)
  const int nbx1 = jb.nb_x + jb.has_bias, nblk = jb.nb_g * nbx1;
  const int per = (nblk + WAVES - 1) / WAVES;
  const int g = lane >> 4, t = lane & 15;
  const uint32_t lane_part = uint32_t(g & 1) * kLdsChunk + uint32_t(8 * (g >> 1) + (t >> 2)) * 32u + uint32_t(t & 3) * 8u;
  uint32_t goff[MAXB], xoff[MAXB];
  bool valid[MAXB], is_bias[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int b = wave * per + i;
    valid[i] = i < per && b < nblk;
    const int gb = b / nbx1, xb = b - gb * nbx1;
    is_bias[i] = xb == jb.nb_x;
    goff[i] = uint32_t(2 * gb) * kLdsChunk + lane_part;
    xoff[i] = is_bias[i] ? goff[i] : uint32_t(kcg + 2 * xb) * kLdsChunk + lane_part;   // X chunks follow the G chunks in a stage (x0 then x1)
  }
  f32x16 master[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) master[i][r] = 0.f;

  for (int it = 0; it < n_it; ++it) {
    int ahead = n_it - 1 - it;
    ahead = ahead > D - 2 ? D - 2 : ahead;
    wait_vmcnt(ahead * my_np);
    __builtin_amdgcn_s_barrier();   // stage `it` landed for every wave; every wave has left stage it - 1
    asm volatile("" ::: "memory");
    if (it + D - 1 < n_it) issue(it + D - 1);
#pragma unroll
    for (int sub = 0; sub < TPS; ++sub) {
      if (it * TPS + sub >= n_tiles) break;
      const float inv = 1.f / scalar_f32(jb.gscale + wt0 + it * TPS + sub);
      const uint32_t sb = uint32_t(size_t(DFN_LDS_PTR(smem))) + uint32_t(it % D) * stage_bytes + sub * tile_bytes;
      // block pipeline: while block i's MFMAs run, block i + 1's operands are read (transposed) into the other register set — the
      // eight waves leave the barrier together, so without this every wave reads, then every wave multiplies
      BlockOps S[2];
      if (valid[0]) issue_block<PL>(S[0], sb + goff[0], sb + xoff[0]);
#pragma unroll
      for (int i = 0; i < MAXB; ++i) {
        if (!valid[i]) break;
        wait_block<PL>(S[i & 1]);
        if (i + 1 < MAXB && valid[i + 1]) issue_block<PL>(S[(i + 1) & 1], sb + goff[i + 1], sb + xoff[i + 1]);
        mma_block<PL>(S[i & 1], is_bias[i], inv, master[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    if (!valid[i]) continue;
    float* dst = jb.partial + (size_t(chunk) * nblk + (wave * per + i)) * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64] = master[i][r];
  }
}

template <int PL>
static hipError_t launch_wgrad_stream_t(const WgradArgs& a, int total_wgs, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stream_kernel<PL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(kWgradLdsBytes));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(wgrad_stream_kernel<PL>, dim3(total_wgs), dim3(WAVES * 64), kWgradLdsBytes, s, a);
  return hipGetLastError();
}
hipError_t launch_wgrad_stream(const WgradArgs& a, int total_wgs, int planes, hipStream_t s) {
  if (total_wgs <= 0) return hipSuccess;
  if (planes == 1) return launch_wgrad_stream_t<1>(a, total_wgs, s);
  if (planes == 2) return launch_wgrad_stream_t<2>(a, total_wgs, s);
  return hipErrorInvalidValue;
}

// Fixed-order sum of a job's chunk partials, scattered into the gradient tensors: weight blocks carry the X operand scale (16),
// the bias column block does not.  One thread per block element; blockIdx -> (job, block, quarter).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceArgs a, int total_quarters) {
  int q = blockIdx.x;
  int j = 0;
  for (; j < a.n_jobs; ++j) {
    const int nq = a.job[j].nb_g * (a.job[j].nb_x + a.job[j].has_bias) * 4;
    if (q < nq) break;
    q -= nq;
  }
  if (j >= a.n_jobs) return;
  const RJob& jb = a.job[j];
  const int nbx1 = jb.nb_x + jb.has_bias, nblk = jb.nb_g * nbx1;
  const int b = q >> 2, e = (q & 3) * 256 + threadIdx.x;
  const int m = a.map[jb.map_off + b * 1024 + e];
  if (m < 0) return;
  const float* src = jb.partial + size_t(b) * 1024 + e;
  // four interleaved chains, then a fixed tree (deterministic): the chunk loads of a thread are in flight together
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t cstride = size_t(nblk) * 1024;
  int c = 0;
  for (; c + 4 <= jb.n_chunks; c += 4)
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] += src[size_t(c + k) * cstride];
  for (int k = 0; c < jb.n_chunks; ++c, ++k) s4[k] += src[size_t(c) * cstride];
  const float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  const bool bias = (b % nbx1) == jb.nb_x;
  a.grads[m >> 20][m & 0xfffff] = bias ? sum : sum * (1.f / kX3ActScale);
}

static_assert(sizeof(ReduceArgs) + 16 <= 4096, "kernel arguments of the reduction launch");
hipError_t launch_wgrad_reduce(const ReduceArgs& a, hipStream_t s) {
  int total = 0;
  for (int j = 0; j < a.n_jobs; ++j) total += a.job[j].nb_g * (a.job[j].nb_x + a.job[j].has_bias) * 4;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(total), dim3(256), 0, s, a, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ the step's range guard
__global__ __launch_bounds__(256) void grads_guard_kernel(GuardArgs a) {
  const int v = *a.step_flag;   // written by kernels that finished before this launch: uniform
  if (!v) return;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicOr(a.range_flag, v);
  float* g = a.grads[blockIdx.y];
  const uint32_t n = a.numel[blockIdx.y];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) g[i] = 0.f;
}
hipError_t launch_grads_guard(const GuardArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(grads_guard_kernel, dim3(8, a.n), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ per-step weight packing
// Every f16 fragment element of a chain blob is one master weight (or zero): hi = f16(w x wscale), lo = f16(w x wscale - hi), as
// the host packer of nerfh_api.hip writes them at commit; the bias fragments are fp32 x bscale.
__global__ __launch_bounds__(256) void pack_units4_kernel(PackArgs4 a) {
  const PackBlob& b = a.b[blockIdx.y];
  const int stride = gridDim.x * blockDim.x;
  bool over = false;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n_welem; i += stride) {
    const PackElem e = b.welem[i];
    const float v = e.src >= 0 ? a.params[e.src >> 20][e.src & 0xfffff] * b.wscale : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    over |= !(fabsf(v) < 65504.f);
    *reinterpret_cast<_Float16*>(b.blob + e.off) = hi;
    *reinterpret_cast<_Float16*>(b.blob + e.off + 1024) = lo;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n_belem; i += stride) {
    const PackElem e = b.belem[i];
    *reinterpret_cast<float*>(b.blob + e.off) = e.src >= 0 ? a.params[e.src >> 20][e.src & 0xfffff] * b.bscale : 0.f;
  }
  if (over && a.status) atomicOr(a.status, 2);   // a weight left the range of the split-f16 operand scale (DFN_RANGE_X3_SATURATED)
}
hipError_t launch_pack4(const PackArgs4& a, hipStream_t s) {
  int n = 0;
  for (const PackBlob& b : a.b) { n = b.n_welem > n ? b.n_welem : n; n = b.n_belem > n ? b.n_belem : n; }
  if (n <= 0) return hipSuccess;
  int grid = (n + 255) / 256;
  grid = grid > 1024 ? 1024 : grid;
  hipLaunchKernelGGL(pack_units4_kernel, dim3(grid, 4), dim3(256), 0, s, a);
  return hipGetLastError();
}
// ------------------------------------------------------------------------------------------ per-ray bias tables from the master weights
// What launch_ray_bias (nerfh_stages.hip) computes from the committed copies, here from the step's own parameters and the per-ray
// inputs the training path already forms (nerfh_train.hip: ray_inputs).  64 outputs per table (netwidth 128).
// One launch for both networks: blockIdx.y = 0 the coarse network's table, 1 the fine one's (two launches of ~8 and ~17 us in a row
// on a one-stream step: the shorter now runs beside the longer, and a launch less).
struct RayBiasTrainArgs {
  const float* w_dir; const float* b_dir; int ldw_dir; int kd; const float* dir_in; int ld_dir;
  const float* w_te; const float* b_te; int ldw_te; int nt; const float* t_in; int ld_t;
  float* table;
};
__global__ __launch_bounds__(128) void ray_bias_train_kernel(RayBiasTrainArgs a0, RayBiasTrainArgs a1, size_t R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const RayBiasTrainArgs& a = blockIdx.y ? a1 : a0;
  const float* __restrict__ w_dir = a.w_dir; const float* __restrict__ b_dir = a.b_dir; const int ldw_dir = a.ldw_dir, kd = a.kd;
  const float* __restrict__ dir_in = a.dir_in; const int ld_dir = a.ld_dir;
  const float* __restrict__ w_te = a.w_te; const float* __restrict__ b_te = a.b_te; const int ldw_te = a.ldw_te, nt = a.nt;
  const float* __restrict__ t_in = a.t_in; const int ld_t = a.ld_t;
  float* __restrict__ table = a.table;
  float* s_wd = sm;               // [kd][64]
  float* s_wt = sm + kd * 64;     // [nt][64]
  float* s_in = s_wt + (w_te ? nt * 64 : 0);   // [kd + nt] the ray's inputs: one coalesced load per ray instead of kd dependent broadcast loads
  // consecutive threads read consecutive columns of a weight row (coalesced); the transposition happens in the LDS store.  Eight loads
  // of a thread are in flight together: one load -> one store at a time, this staging (38 + 10 dependent round trips per block) was
  // most of the kernel's 23 us.
  auto stage = [&](const float* __restrict__ wsrc, int ldw, int k, float* dst) {
    const int n = k * 64, bd = blockDim.x;
    for (int base = threadIdx.x; base < n; base += bd * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * bd, f = i / k, jj = i - f * k;
        v[u] = i < n ? wsrc[size_t(f) * ldw + kWidth + jj] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * bd, f = i / k, jj = i - f * k;
        if (i < n) dst[jj * 64 + f] = v[u];
      }
    }
  };
  stage(w_dir, ldw_dir, kd, s_wd);
  if (w_te) stage(w_te, ldw_te, nt, s_wt);
  __syncthreads();
  const int tbl = threadIdx.x >> 6, f = threadIdx.x & 63;
  const int mb = f >> 5, row = f & 31, hh = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
  const int slot = ((tbl * 2 + mb) * 2 + hh) * 16 + r;
  const float bias = tbl == 0 ? b_dir[f] : (w_te ? b_te[f] : 0.f);
  for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
    for (int i = threadIdx.x; i < kd + (w_te ? nt : 0); i += blockDim.x) s_in[i] = i < kd ? dir_in[ray * ld_dir + i] : t_in[ray * ld_t + (i - kd)];
    __syncthreads();
    float acc = bias;
    if (tbl == 0) {
      for (int jj = 0; jj < kd; ++jj) acc = fmaf(s_wd[jj * 64 + f], s_in[jj], acc);
    } else if (w_te) {
      for (int jj = 0; jj < nt; ++jj) acc = fmaf(s_wt[jj * 64 + f], s_in[kd + jj], acc);
    }
    table[ray * kRayBiasFloats + slot] = acc;
    __syncthreads();   // (the next ray's inputs overwrite s_in)
  }
}
hipError_t launch_ray_bias_train_pair(const float* w_dir_c, const float* b_dir_c, int ldw_dir_c, int kd_c, const float* dir_in_c, int ld_dir_c,
                                      float* table_c, const float* w_dir, const float* b_dir, int ldw_dir, int kd, const float* dir_in,
                                      int ld_dir, const float* w_te, const float* b_te, int ldw_te, int nt, const float* t_in, int ld_t,
                                      float* table_f, size_t R, hipStream_t s) {
  if (!R) return hipSuccess;
  const size_t lds_c = (size_t(kd_c) * 64 + size_t(kd_c)) * sizeof(float);
  const size_t lds_f = (size_t(kd + (w_te ? nt : 0)) * 64 + size_t(kd + (w_te ? nt : 0))) * sizeof(float);
  const size_t lds = lds_c > lds_f ? lds_c : lds_f;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const int grid = int(R < 256 ? R : 256);   // per table one workgroup per CU: every workgroup stages its weights once
  const RayBiasTrainArgs a0{w_dir_c, b_dir_c, ldw_dir_c, kd_c, dir_in_c, ld_dir_c, nullptr, nullptr, 0, 0, nullptr, 0, table_c};
  const RayBiasTrainArgs a1{w_dir, b_dir, ldw_dir, kd, dir_in, ld_dir, w_te, b_te, ldw_te, nt, t_in, ld_t, table_f};
  hipLaunchKernelGGL(ray_bias_train_kernel, dim3(grid, 2), dim3(128), lds, s, a0, a1, R);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ per-ray sums of a stored gradient array
// The columns of dir_encoding.0 / transient_encoding.0 beyond `final` multiply per-RAY inputs (direction encoding, embeddings):
// their gradients need sum_samples G[p, :] per ray (models/nerfw.py:62-95).  One block per ray; fixed summation order.
template <int PL>
DFN_DEV void frag_ray_sum_body(const char* __restrict__ arr, int kc, const float* __restrict__ gscale, int Ns, float* __restrict__ out, int ldo,
                               float* red) {
  // a lane owns one 16-byte piece (8 slots of one half of one chunk) and walks the ray's samples 256 / (2 kc) at a time: 16-byte loads
  // of the hi and the lo plane instead of one 2-byte load per value
  const size_t ray = blockIdx.x;
  const int np = 2 * kc;                       // pieces per point
  const int groups = 256 / np;                 // samples in flight per pass
  const int piece = threadIdx.x % np, sg = threadIdx.x / np;
  const int c = piece >> 1, hh = piece & 1;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int smp = sg; smp < Ns; smp += groups) {
    const size_t pt = ray * Ns + smp;
    const size_t wt = pt >> 5;
    const char* p = arr + (wt * kc + c) * (1024 * PL) + (2 * (pt & 31) + hh) * 16;
    const half8 hi = *reinterpret_cast<const half8*>(p);
    half8 lo = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (PL == 2) lo = *reinterpret_cast<const half8*>(p + 1024);
    const float inv = 1.f / gscale[wt];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float(hi[j]) + float(lo[j])) * inv;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
  __syncthreads();
  if (sg == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = acc[j];
      for (int k = 1; k < groups; ++k) s += red[(k * np + piece) * 8 + j];   // fixed order: deterministic
      const int slot = 8 * c + j;
      out[ray * ldo + 64 * (slot >> 5) + hidden_feature(hh, slot & 31)] = s;
    }
  }
}
template <int PL>
__global__ __launch_bounds__(256) void frag_ray_sum_kernel(const char* __restrict__ arr, int kc, const float* __restrict__ gscale, int Ns,
                                                           float* __restrict__ out, int ldo) {
  __shared__ float red[256 * 8];
  frag_ray_sum_body<PL>(arr, kc, gscale, Ns, out, ldo, red);
}
// both networks' per-ray sums in one launch: blockIdx.y = 0 the fine network's array (PLF planes), 1 the coarse network's (two planes)
struct FragRaySumArgs { const char* arr; int kc; const float* gscale; int Ns; float* out; int ldo; };
template <int PLF>
__global__ __launch_bounds__(256) void frag_ray_sum_pair_kernel(FragRaySumArgs f, FragRaySumArgs c) {
  __shared__ float red[256 * 8];
  if (blockIdx.y == 0) frag_ray_sum_body<PLF>(f.arr, f.kc, f.gscale, f.Ns, f.out, f.ldo, red);
  else frag_ray_sum_body<2>(c.arr, c.kc, c.gscale, c.Ns, c.out, c.ldo, red);
}
hipError_t launch_frag_ray_sum_pair(const char* arr_f, int kc_f, int planes_f, const float* gscale_f, int Ns_f, float* out_f, int ldo_f,
                                    const char* arr_c, int kc_c, const float* gscale_c, int Ns_c, float* out_c, int ldo_c, size_t R,
                                    hipStream_t s) {
  if (!R) return hipSuccess;
  if ((kc_f != 4 && kc_f != 8) || (kc_c != 4 && kc_c != 8) || (planes_f != 1 && planes_f != 2)) return hipErrorInvalidValue;
  const FragRaySumArgs f{arr_f, kc_f, gscale_f, Ns_f, out_f, ldo_f}, c{arr_c, kc_c, gscale_c, Ns_c, out_c, ldo_c};
  if (planes_f == 1) hipLaunchKernelGGL(frag_ray_sum_pair_kernel<1>, dim3(unsigned(R), 2), dim3(256), 0, s, f, c);
  else hipLaunchKernelGGL(frag_ray_sum_pair_kernel<2>, dim3(unsigned(R), 2), dim3(256), 0, s, f, c);
  return hipGetLastError();
}
hipError_t launch_frag_ray_sum(const char* arr, int kc, int planes, const float* gscale, size_t R, int Ns, float* out, int ldo, hipStream_t s) {
  if (!R) return hipSuccess;
  if ((kc != 4 && kc != 8) || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
  if (planes == 1) hipLaunchKernelGGL(frag_ray_sum_kernel<1>, dim3(unsigned(R)), dim3(256), 0, s, arr, kc, gscale, Ns, out, ldo);
  else hipLaunchKernelGGL(frag_ray_sum_kernel<2>, dim3(unsigned(R)), dim3(256), 0, s, arr, kc, gscale, Ns, out, ldo);
  return hipGetLastError();
}

}  // namespace fused
}  // namespace dfn
