// nerfh_train.hip — kernels of the NeRF-H TRAINING path (gfx950): generic fp32-MFMA Linear forward / data gradient /
// weight gradient over point-major activations, and the training-mode stages around them (stratified depths,
// positional encoding, coarse composite + importance sampling with injected draws, compositing backward, NerfWLoss).
// See nerfh_train.h for the formulation.
//
// Replaces (reference, /root/reference/script/): run_nerf.py:50-66 (one optimisation step's forward + backward),
// models/rendering.py:245-337 with test_time=False (render_rays), :132-243 (raw2outputs_NeRFW, both typ),
// models/nerfw.py:47-95 (run_network_NeRFW training branches), :297-354 (NeRFW.forward), models/losses.py:19-57.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_frag.h"
#include "nerfh_device.h"
#include "nerfh_layout.h"
#include "nerfh_train.h"

namespace dfn {
namespace train {

static inline int grid_for(size_t n, int block, int cap = 256 * 16) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return int(g < size_t(cap) ? g : size_t(cap));
}

DFN_DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

DFN_DEV float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return sigmoid(v);
    case ACT_SOFTPLUS: return softplus(v);
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------ forward
// One wave: 32 points x NBLK*32 outputs.  MFMA A operand = activations (rows = points), B = W^T (columns = output
// features): lane (i = lane & 31, kh = lane >> 5) feeds A[i][kh] and B[kh][i]; the C fragment of a lane is output
// feature i for the 16 points mblock_row(kh, r), so a store instruction writes two 128-byte row segments.
// A lane loads 4 consecutive contraction elements per step (k0 + 4 kh .. + 3) from its activation row and from its
// weight row: the order of the contraction index is free as long as A and B agree.
struct FwdArgs {
  Seg seg[3];
  int nseg;
  const float* W;
  int ldw;
  const float* b;
  int N, act;
  float* y;
  int ldy;
  long long P;
};

template <int NBLK>
__global__ __launch_bounds__(256) void gemm_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const long long p0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (p0 >= a.P) return;
  const int n0 = blockIdx.y * NBLK * 32;
  const long long row = p0 + i < a.P ? p0 + i : a.P - 1;
  f32x16 acc[NBLK];
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) {
    const int n = n0 + nb * 32 + i;
    const float bv = (a.b && n < a.N) ? a.b[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = bv;
  }
  for (int s = 0; s < a.nseg; ++s) {
    const Seg sg = a.seg[s];
    const float* x = sg.x + (sg.div == 1 ? row : row / sg.div) * sg.ld;
    const float* w[NBLK];
    bool wok[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      const int n = n0 + nb * 32 + i;
      wok[nb] = n < a.N;
      w[nb] = a.W + (size_t)(wok[nb] ? n : 0) * a.ldw + sg.wcol;
    }
    for (int k0 = 0; k0 < sg.K; k0 += 8) {
      const int k = k0 + 4 * kh;
      float av[4], bv[NBLK][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = k + q < sg.K ? x[k + q] : 0.f;
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[nb][q] = (wok[nb] && k + q < sg.K) ? w[nb][k + q] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mfma32(av[q], bv[nb][q], acc[nb]);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) {
    const int n = n0 + nb * 32 + i;
    if (n >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long pt = p0 + mblock_row(kh, r);
      if (pt < a.P) a.y[pt * a.ldy + n] = apply_act(acc[nb][r], a.act);
    }
  }
}

// ------------------------------------------------------------------------------------------ LDS-staged form of both products
// C[p, m] = sum_c A[p, c] Bs[m, c]: the weight slab Bs (forward: rows = outputs, columns = the concatenated inputs; data
// gradient: rows = inputs k, columns = outputs n, i.e. W transposed while it is staged) is staged ONCE per workgroup into LDS
// and the workgroup is persistent over 128-point tiles, so the weights cost one L2 read per workgroup instead of one per wave
// and the B operand is a conflict-free ds_read_b128 (row stride = 4 mod 32 floats).  A lane's A operand is one 16-byte load of
// its activation row per 8 contraction elements.
struct LdsGemmArgs {
  Seg seg[3];          // forward: input segments; data gradient: seg[0] = {G, ldg, N, 1, 0}
  int nseg;
  const float* W;
  int ldw, wcol;       // data gradient: first weight column of the K window
  const float* b;
  int M, act;          // rows of the slab handled in total (forward: N outputs; data gradient: K inputs)
  float* y;
  int ldy;
  int accumulate;
  const float* mask;
  int ldmask;
  long long P;
  int KP;              // slab row stride (floats)
  int bwd;
};

DFN_DEV int seg_koff(const LdsGemmArgs& a, int s) {   // column of segment s inside the slab (segments padded to multiples of 8)
  int o = 0;
  for (int t = 0; t < s; ++t) o += (a.seg[t].K + 7) & ~7;
  return o;
}

// Epilogue.  The points are the MFMA's A operand: C = [point][channel], a lane owns ONE channel of 16 points.  Written straight from
// the accumulators that is 64 four-byte stores per lane and tile (and, for the data gradient, as many loads of the old value and of
// the ReLU mask): an ablation without the epilogue put it at 16 % (forward) to 41 % (data gradient) of these kernels.  Each wave
// therefore turns its fragment through a small LDS buffer, half a 32 x 32 block at a time, and moves whole 128-byte row segments
// 16 bytes per lane: a quarter of the memory instructions, every line touched once.
constexpr int kTurnStride = 36;                      // floats per row of the turn buffer (16-byte aligned rows)
constexpr int kTurnFloats = 16 * kTurnStride;        // per wave: 16 points x 32 channels
template <int NBLK>
__global__ __launch_bounds__(256, 2) void gemm_lds_kernel(LdsGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * NBLK * 32;
  const int KP = a.KP;
  float* turn = slab + NBLK * 32 * KP + wave * kTurnFloats;
  // ---- stage the slab: rows m0 .. m0 + NBLK*32, zero padded
  for (int e = threadIdx.x; e < NBLK * 32 * KP; e += 256) slab[e] = 0.f;
  __syncthreads();
  if (!a.bwd) {
    for (int s = 0; s < a.nseg; ++s) {
      const int K = a.seg[s].K, ko = seg_koff(a, s), wc = a.seg[s].wcol;
      // a wave per row, 16 rows in flight per thread: coalesced reads, one memory latency per 16 rows
      for (int k = threadIdx.x & 63; k < K; k += 64)
        for (int r0 = threadIdx.x >> 6; r0 < NBLK * 32; r0 += 64) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int r = r0 + 4 * u;
            v[u] = (r < NBLK * 32 && m0 + r < a.M) ? a.W[(size_t)(m0 + r) * a.ldw + wc + k] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int r = r0 + 4 * u;
            if (r < NBLK * 32) slab[r * KP + ko + k] = v[u];
          }
        }
    }
  } else {
    const int N = a.seg[0].K;   // contraction = the layer's outputs; slab[kcol][n] = W[n][wcol + m0 + kcol]
    // a wave per weight row (coalesced reads, transposed LDS writes), 16 rows in flight per thread
    for (int r = threadIdx.x & 63; r < NBLK * 32; r += 64)
      for (int n0 = threadIdx.x >> 6; n0 < N; n0 += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int n = n0 + 4 * u;
          v[u] = (n < N && m0 + r < a.M) ? a.W[(size_t)n * a.ldw + a.wcol + m0 + r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int n = n0 + 4 * u;
          if (n < N) slab[r * KP + n] = v[u];
        }
      }
  }
  __syncthreads();
  const long long n_tiles = (a.P + 127) / 128;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long p0 = (tile * 4 + wave) * 32;
    if (p0 >= a.P) continue;
    const long long row = p0 + i < a.P ? p0 + i : a.P - 1;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + nb * 32 + i;
        acc[nb][r] = (a.b && m < a.M) ? a.b[m] : 0.f;
      }
    // data gradient: the ReLU-mask pieces this lane applies in the epilogue are fetched NOW, behind the products (waited for one
    // turn pass at a time they cost a memory latency each: 16 per tile)
    const bool mvec = a.bwd && a.mask && ((reinterpret_cast<uintptr_t>(a.mask) & 15) == 0) && (a.ldmask & 3) == 0 && (a.M & 3) == 0;
    f32x4 mk[NBLK][2][2];
    if (mvec) {
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const long long pt = p0 + 16 * hf + 8 * u + (lane >> 3);
            const int chan = m0 + nb * 32 + (lane & 7) * 4;
            mk[nb][hf][u] = (pt < a.P && chan < a.M) ? *reinterpret_cast<const f32x4*>(a.mask + (size_t)pt * a.ldmask + chan) : f32x4{1.f, 1.f, 1.f, 1.f};
          }
    }
    for (int s = 0; s < a.nseg; ++s) {
      const Seg sg = a.seg[s];
      const float* x = sg.x + (sg.div == 1 ? row : row / sg.div) * sg.ld;
      const int ko = seg_koff(a, s);
      const bool vec = ((reinterpret_cast<uintptr_t>(sg.x) & 15) == 0) && (sg.ld & 3) == 0 && sg.ld >= ((sg.K + 3) & ~3);
      const float* bl = slab + i * KP + ko + 4 * kh;
      // The lane's whole share of up to 128 contraction elements is loaded up front (16 x 16 bytes in flight), then the MFMAs run
      // without a memory wait; the other wave of the SIMD covers the one latency per block.
      for (int kb0 = 0; kb0 < sg.K; kb0 += 128) {
        constexpr int STEPS = 16;
        f32x4 av[STEPS];
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
          const int k = kb0 + 8 * t + 4 * kh;
          if (vec && k + 3 < ((sg.K + 3) & ~3)) {
            av[t] = *reinterpret_cast<const f32x4*>(x + k);
            if (k + 3 >= sg.K) {   // the row's own padding may hold anything: zero what lies beyond K
#pragma unroll
              for (int q = 0; q < 4; ++q) if (k + q >= sg.K) av[t][q] = 0.f;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) av[t][q] = k + q < sg.K ? x[k + q] : 0.f;
          }
        }
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
          const int k0 = kb0 + 8 * t;
          if (k0 < sg.K) {
            f32x4 bv[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) bv[nb] = *reinterpret_cast<const f32x4*>(bl + nb * 32 * KP + k0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mfma32(av[t][q], bv[nb][q], acc[nb]);
          }
        }
      }
    }
    // ---- epilogue (see above): per 32-channel block and half (16 points), fragment -> turn buffer -> rows
    const bool yvec = ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) && (a.ldy & 3) == 0;
    const bool acc_old = a.bwd && a.accumulate;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      if (m0 + nb * 32 >= a.M) continue;                 // wave-uniform
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int r = 0; r < 8; ++r)                      // registers 8 hf .. 8 hf + 7 = rows 16 hf + (r & 3) + 8 (r >> 2) + 4 kh
          turn[((r & 3) + 8 * (r >> 2) + 4 * kh) * kTurnStride + i] = acc[nb][8 * hf + r];
        wave_sync();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = 8 * u + (lane >> 3), c4 = (lane & 7) * 4;
          const long long pt = p0 + 16 * hf + row;
          const int chan = m0 + nb * 32 + c4;
          if (pt < a.P && chan < a.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(turn + row * kTurnStride + c4);
            const bool whole = chan + 3 < a.M;
            float* yp = a.y + (size_t)pt * a.ldy + chan;
            if (acc_old) {
              f32x4 o = {0.f, 0.f, 0.f, 0.f};
              if (whole && yvec) o = *reinterpret_cast<const f32x4*>(yp);
              else
#pragma unroll
                for (int q = 0; q < 4; ++q) if (chan + q < a.M) o[q] = yp[q];
              v += o;
            }
            if (a.bwd && a.mask) {
              const float* mp = a.mask + (size_t)pt * a.ldmask + chan;
              f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
              if (mvec) m4 = mk[nb][hf][u];
              else
#pragma unroll
                for (int q = 0; q < 4; ++q) if (chan + q < a.M) m4[q] = mp[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = m4[q] > 0.f ? v[q] : 0.f;
            }
            if (!a.bwd && a.act != ACT_NONE) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], a.act);
            }
            if (whole && yvec) *reinterpret_cast<f32x4*>(yp) = v;
            else
#pragma unroll
              for (int q = 0; q < 4; ++q) if (chan + q < a.M) yp[q] = v[q];
          }
        }
        wave_sync();
      }
    }
  }
}

// Launch the LDS form if the slab fits; false = caller falls back to the streaming kernels above.
static bool launch_lds_gemm(LdsGemmArgs a, int Kc_total, hipError_t& err, hipStream_t s) {
  int KP = (Kc_total + 7) & ~7;
  while ((KP & 31) != 4) KP += 4;
  a.KP = KP;
  constexpr size_t kTurnBytes = 4 * kTurnFloats * 4;   // the four waves' epilogue turn buffers
  constexpr size_t kLdsBudget = 80 * 1024;              // two workgroups per CU
  int nblk = a.M > 64 ? 4 : (a.M > 32 ? 2 : 1);
  while (nblk > 1 && size_t(nblk) * 32 * KP * 4 + kTurnBytes > kLdsBudget) nblk >>= 1;
  const size_t lds = size_t(nblk) * 32 * KP * 4 + kTurnBytes;
  if (lds > 150 * 1024) return false;
  const long long tiles = (a.P + 127) / 128;
  const int gy = (a.M + nblk * 32 - 1) / (nblk * 32);
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t p;
    int dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  long long gx = (long long)cus * (lds > kLdsBudget ? 1 : 2) / gy;
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) {
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      if (err != hipSuccess) return;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, s, a);
    err = hipGetLastError();
  };
  if (nblk == 4) go(gemm_lds_kernel<4>);
  else if (nblk == 2) go(gemm_lds_kernel<2>);
  else go(gemm_lds_kernel<1>);
  return true;
}

hipError_t gemm_fwd(const Seg* segs, int nseg, const float* W, int ldw, const float* b, int N, int act, float* y, int ldy,
                    long long P, hipStream_t s) {
  if (P <= 0 || N <= 0) return hipSuccess;
  if (nseg < 1 || nseg > 3) return hipErrorInvalidValue;
  if (P >= 1024) {   // persistent LDS-staged form (the weight slab is staged once per workgroup)
    LdsGemmArgs l{};
    int kc = 0;
    for (int k = 0; k < nseg; ++k) { l.seg[k] = segs[k]; kc += (segs[k].K + 7) & ~7; }
    l.nseg = nseg; l.W = W; l.ldw = ldw; l.wcol = 0; l.b = b; l.M = N; l.act = act; l.y = y; l.ldy = ldy; l.P = P; l.bwd = 0;
    hipError_t e = hipSuccess;
    if (launch_lds_gemm(l, kc, e, s)) return e;
  }
  FwdArgs a{};
  for (int k = 0; k < nseg; ++k) a.seg[k] = segs[k];
  a.nseg = nseg; a.W = W; a.ldw = ldw; a.b = b; a.N = N; a.act = act; a.y = y; a.ldy = ldy; a.P = P;
  const unsigned gx = unsigned((P + 127) / 128);
  if (N <= 32) hipLaunchKernelGGL(gemm_fwd_kernel<1>, dim3(gx, 1), dim3(256), 0, s, a);
  else if (N <= 64) hipLaunchKernelGGL(gemm_fwd_kernel<2>, dim3(gx, 1), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gemm_fwd_kernel<4>, dim3(gx, (N + 127) / 128), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ data gradient
// dX = G W: A = G (rows = points, contraction over the layer's outputs n), B[kh][j] = W[n][wcol + k0 + j] — a
// coalesced read of the weight row, no transposed copy of W is needed.
struct BwdGemmArgs {
  const float* G;
  int ldg, N;
  const float* W;
  int ldw, wcol, K;
  float* dx;
  int lddx, accumulate;
  const float* mask;
  int ldmask;
  long long P;
};

template <int KBLK>
__global__ __launch_bounds__(256) void gemm_bwd_kernel(BwdGemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const long long p0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (p0 >= a.P) return;
  const int kb0 = blockIdx.y * KBLK * 32;
  const long long row = p0 + i < a.P ? p0 + i : a.P - 1;
  const float* g = a.G + row * a.ldg;
  f32x16 acc[KBLK];
#pragma unroll
  for (int kb = 0; kb < KBLK; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  for (int n0 = 0; n0 < a.N; n0 += 8) {
    const int n = n0 + 4 * kh;
    float av[4], bv[KBLK][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) av[q] = n + q < a.N ? g[n + q] : 0.f;
#pragma unroll
    for (int kb = 0; kb < KBLK; ++kb) {
      const int col = kb0 + kb * 32 + i;
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[kb][q] = (n + q < a.N && col < a.K) ? a.W[(size_t)(n + q) * a.ldw + a.wcol + col] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int kb = 0; kb < KBLK; ++kb) acc[kb] = mfma32(av[q], bv[kb][q], acc[kb]);
  }
#pragma unroll
  for (int kb = 0; kb < KBLK; ++kb) {
    const int col = kb0 + kb * 32 + i;
    if (col >= a.K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long pt = p0 + mblock_row(kh, r);
      if (pt >= a.P) continue;
      float v = acc[kb][r];
      if (a.accumulate) v += a.dx[pt * a.lddx + col];
      if (a.mask && !(a.mask[pt * a.ldmask + col] > 0.f)) v = 0.f;
      a.dx[pt * a.lddx + col] = v;
    }
  }
}

hipError_t gemm_bwd(const float* G, int ldg, int N, const float* W, int ldw, int wcol, int K, float* dx, int lddx,
                    int accumulate, const float* mask_src, int ldmask, long long P, hipStream_t s) {
  if (P <= 0 || K <= 0) return hipSuccess;
  if (P >= 1024) {
    LdsGemmArgs l{};
    l.seg[0] = Seg{G, ldg, N, 1, 0};
    l.nseg = 1; l.W = W; l.ldw = ldw; l.wcol = wcol; l.b = nullptr; l.M = K; l.act = 0; l.y = dx; l.ldy = lddx;
    l.accumulate = accumulate; l.mask = mask_src; l.ldmask = ldmask; l.P = P; l.bwd = 1;
    hipError_t e = hipSuccess;
    if (launch_lds_gemm(l, (N + 7) & ~7, e, s)) return e;
  }
  BwdGemmArgs a{G, ldg, N, W, ldw, wcol, K, dx, lddx, accumulate, mask_src, ldmask, P};
  const unsigned gx = unsigned((P + 127) / 128);
  if (K <= 32) hipLaunchKernelGGL(gemm_bwd_kernel<1>, dim3(gx, 1), dim3(256), 0, s, a);
  else if (K <= 64) hipLaunchKernelGGL(gemm_bwd_kernel<2>, dim3(gx, 1), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gemm_bwd_kernel<4>, dim3(gx, (K + 127) / 128), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW = G^T X over points: A[i][kh] = G[p + kh'][n0 + i], B[kh][j] = X[p + kh'][k0 + j] — both coalesced reads of
// point-major rows.  A workgroup owns one 32-row block of outputs x KBLK 32-column blocks of inputs for one chunk of
// points; its four waves take a quarter of the chunk each and are summed through LDS in fixed order; chunk partials are
// reduced in fixed order by wgrad_reduce_kernel (deterministic).
constexpr int kWgradK = 4;  // 32-column blocks per workgroup
constexpr int kWtPtsC = 32;  // points per staged tile of gemm_wgrad_tile_kernel
// Points per chunk (multiple of 32).  The launch is chunks x items workgroups of 4 waves, two resident per CU: the chunk count
// is chosen so that ONE wave of workgroups fills the chip (512 slots on 256 CUs) instead of leaving a half-empty second
// round, and stays <= 256 so that the partial sums remain a small fraction of the product.
static inline int wgrad_items(int N, int K) { return ((N + 31) / 32) * ((K + kWgradK * 32 - 1) / (kWgradK * 32)); }
static inline int wgrad_chunk(long long P, int items) {
  long long chunks = 512 / (items < 1 ? 1 : items);
  if (chunks < 1) chunks = 1;
  if (chunks > 256) chunks = 256;
  long long ch = ((P + chunks - 1) / chunks + 31) / 32 * 32;
  if (ch < 32) ch = 32;
  return int(ch);
}
// the tiled kernel: per-point inputs, at most 128 outputs, enough points to give every workgroup whole tiles
static inline bool wgrad_tiled(int N, const Seg& x, long long P) { return x.div == 1 && N <= 128 && N > 32 && P >= 64 * 1024; }
static inline int wgrad_tile_chunk(long long P, int K) {
  long long chunks = 512 / ((K + 127) / 128);      // two workgroups per CU over all column groups
  if (chunks < 1) chunks = 1;
  long long ch = ((P + chunks - 1) / chunks + kWtPtsC - 1) / kWtPtsC * kWtPtsC;
  return int(ch < kWtPtsC ? kWtPtsC : ch);
}
size_t gemm_wgrad_scratch_floats(int N, int K, long long P) {
  // an upper bound over the segment widths a layer is split into: <= 512 chunks of [N][K] (+ [N])
  const long long ch = wgrad_chunk(P, wgrad_items(N, K)), chunks = (P + ch - 1) / ch;
  const long long worst = chunks > 512 ? chunks : 512;
  return size_t(worst) * (size_t(N) * K + N);
}

struct WgradArgs {
  const float* G;
  int ldg, N;
  Seg x;
  float* part;    // [chunks][N][K]
  float* bpart;   // [chunks][N]
  long long P;
  int chunk, kgroups, items;
};

typedef float WgradRed[kWgradK * 16 * 64 + 64];
// (the body takes its place in the grid as arguments: gemm_wgrad_kernel is one product, gemm_wgrad_multi_kernel up to three small ones)
DFN_DEV void gemm_wgrad_body(const WgradArgs& a, const int bx, const int item, WgradRed* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int nblk = item / a.kgroups, kg = item - nblk * a.kgroups;
  const int n0 = nblk * 32, k0 = kg * kWgradK * 32;
  const long long c0 = (long long)bx * a.chunk;
  const long long c1 = c0 + a.chunk < a.P ? c0 + a.chunk : a.P;
  const int sub = a.chunk / 4;                      // multiple of 8
  const long long w0 = c0 + (long long)wave * sub;
  const long long w1 = w0 + sub < c1 ? w0 + sub : c1;
  const int K = a.x.K;
  f32x16 acc[kWgradK];
#pragma unroll
  for (int kb = 0; kb < kWgradK; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  float bsum = 0.f;
  const bool nok = n0 + i < a.N;
  bool cok[kWgradK];
#pragma unroll
  for (int kb = 0; kb < kWgradK; ++kb) cok[kb] = k0 + kb * 32 + i < K;
  const float* gp = a.G + (w0 + 4 * kh) * a.ldg + n0 + i;
  const float* xp = a.x.x + (w0 + 4 * kh) * a.x.ld + k0 + i;     // div == 1 fast path
  // Two register stages: the loads of points p + 8 .. p + 15 are in flight while the MFMAs of p .. p + 7 run (one HBM latency
  // per 8 points would otherwise sit in front of every 16 MFMAs).
  float av[2][4], bv[2][kWgradK][4];
  auto load_stage = [&](int st, long long p) {
    const bool whole = p + 8 <= w1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long pq = p + 4 * kh + q;
      const bool ok = whole || pq < w1;
      av[st][q] = (ok && nok) ? gp[(size_t)q * a.ldg] : 0.f;
      const float* xr = a.x.div == 1 ? xp + (size_t)q * a.x.ld : a.x.x + (pq / a.x.div) * a.x.ld + k0 + i;
#pragma unroll
      for (int kb = 0; kb < kWgradK; ++kb) bv[st][kb][q] = (ok && cok[kb]) ? xr[kb * 32] : 0.f;
    }
    gp += (size_t)8 * a.ldg;
    xp += (size_t)8 * a.x.ld;
  };
  auto mma_stage = [&](int st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bsum += av[st][q];
#pragma unroll
      for (int kb = 0; kb < kWgradK; ++kb) acc[kb] = mfma32(av[st][q], bv[st][kb][q], acc[kb]);
    }
  };
  if (w0 < w1) load_stage(0, w0);
  for (long long p = w0; p < w1; p += 16) {
    if (p + 8 < w1) load_stage(1, p + 8);
    mma_stage(0);
    if (p + 8 < w1) {
      if (p + 16 < w1) load_stage(0, p + 16);
      mma_stage(1);
    }
  }
  if (wave > 0) {
    float* r = red[wave - 1];
#pragma unroll
    for (int kb = 0; kb < kWgradK; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) r[(kb * 16 + q) * 64 + lane] = acc[kb][q];
    r[kWgradK * 16 * 64 + lane] = bsum;
  }
  __syncthreads();
  if (wave > 0) return;
  for (int w = 0; w < 3; ++w) {
    const float* r = red[w];
#pragma unroll
    for (int kb = 0; kb < kWgradK; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[kb][q] += r[(kb * 16 + q) * 64 + lane];
    bsum += r[kWgradK * 16 * 64 + lane];
  }
  float* part = a.part + (size_t)bx * a.N * K;
#pragma unroll
  for (int kb = 0; kb < kWgradK; ++kb) {
    const int col = k0 + kb * 32 + i;
    if (col >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + mblock_row(kh, r);
      if (n < a.N) part[(size_t)n * K + col] = acc[kb][r];
    }
  }
  if (kg == 0 && a.bpart) {
    const float tot = bsum + __shfl_xor(bsum, 32, 64);
    if (kh == 0 && nok) a.bpart[(size_t)bx * a.N + n0 + i] = tot;
  }
}
__global__ __launch_bounds__(256, 2) void gemm_wgrad_kernel(WgradArgs a) {
  __shared__ WgradRed red[3];
  gemm_wgrad_body(a, int(blockIdx.x), int(blockIdx.y), red);
}
// Up to three SMALL products in one launch (blockIdx.z = product; grid = the largest chunk / item counts, the others' spare blocks
// return): the per-ray tails of the fused NeRF-H step are three such products in a row, each with its reduction — six launches of
// 5-8 us on a one-stream step where a tiny launch costs ~4.7 us whatever it does.
struct WgradMulti { WgradArgs a[3]; int chunks[3]; int n; };
__global__ __launch_bounds__(256, 2) void gemm_wgrad_multi_kernel(WgradMulti m) {
  __shared__ WgradRed red[3];
  const int j = blockIdx.z;
  if (int(blockIdx.x) >= m.chunks[j] || int(blockIdx.y) >= m.a[j].items) return;
  gemm_wgrad_body(m.a[j], int(blockIdx.x), int(blockIdx.y), red);
}

// Tiled form of the weight-gradient product for N <= 128 outputs and per-point inputs (every Linear of the two networks): a
// workgroup owns a chunk of points and ALL outputs — wave w the 32 outputs n0 = 32 w, each against a 128-column group of X — so the
// X rows of a 32-point tile are staged once in LDS for the four waves (gemm_wgrad_kernel re-reads them per 32-output item: 2.5x the
// bytes of this form) and nothing is reduced across waves.  Tile t + 1 (X into registers, the wave's G strip) is fetched while
// tile t is multiplied; one barrier per tile = 64 MFMAs per wave.
// NBW = 32-output blocks per workgroup: 4 (N <= 128: a wave = one block x all four 32-column blocks of the group) or 2 (N <= 64: a
// wave = one block x two column blocks).
constexpr int kWtPts = 32, kWtStride = 132;   // staged tile [32 points][128 columns + 4]: 16-byte rows, conflict-free scalar reads
template <int NBW>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_tile_kernel(WgradArgs a) {
  constexpr int KBW = NBW;                    // column blocks per wave: 4 / (4 / NBW)
  __shared__ __attribute__((aligned(16))) float xt[2][kWtPts * kWtStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int k0 = blockIdx.y * 128, n0 = (wave % NBW) * 32, kbw = (wave / NBW) * KBW;   // first of the wave's column blocks
  const long long c0 = (long long)blockIdx.x * a.chunk;
  const long long c1 = c0 + a.chunk < a.P ? c0 + a.chunk : a.P;
  const int K = a.x.K;
  f32x16 acc[KBW];
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  float bsum = 0.f;
  const bool nok = n0 + i < a.N;
  // staging: thread t moves the 16-byte pieces e = t + 256 u (u < 4) of the tile: point e / 32, columns 4 (e % 32) ..
  const bool xvec = ((reinterpret_cast<uintptr_t>(a.x.x) & 15) == 0) && (a.x.ld & 3) == 0;
  auto load_x = [&](f32x4 (&xr)[4], long long pt0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + 256 * u, pt = e >> 5, k4 = (e & 31) * 4;
      const long long p = pt0 + pt;
      const float* src = a.x.x + (size_t)(p < c1 ? p : c1 - 1) * a.x.ld + k0 + k4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < c1) {
        if (xvec && ((k0 + k4) & 3) == 0 && k0 + k4 + 3 < K) v = *reinterpret_cast<const f32x4*>(src);
        else
#pragma unroll
          for (int q = 0; q < 4; ++q) if (k0 + k4 + q < K) v[q] = src[q];
      }
      xr[u] = v;
    }
  };
  auto store_x = [&](const f32x4 (&xr)[4], int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + 256 * u, pt = e >> 5, k4 = (e & 31) * 4;
      *reinterpret_cast<f32x4*>(&xt[buf][pt * kWtStride + k4]) = xr[u];
    }
  };
  auto load_g = [&](float (&g)[16], long long pt0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long long p = pt0 + 8 * (j >> 2) + 4 * kh + (j & 3);
      g[j] = (nok && p < c1) ? a.G[(size_t)p * a.ldg + n0 + i] : 0.f;
    }
  };
  f32x4 xr[4];
  float g[16], gn[16];
  load_x(xr, c0);
  load_g(g, c0);
  store_x(xr, 0);
  __syncthreads();
  int cur = 0;
  for (long long pt0 = c0; pt0 < c1; pt0 += kWtPts) {
    const bool more = pt0 + kWtPts < c1;
    if (more) { load_x(xr, pt0 + kWtPts); load_g(gn, pt0 + kWtPts); }
    const float* xb = xt[cur] + 4 * kh * kWtStride + kbw * 32 + i;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float bv[KBW][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) bv[kb][q] = xb[(8 * st + q) * kWtStride + kb * 32];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bsum += g[4 * st + q];
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) acc[kb] = mfma32(g[4 * st + q], bv[kb][q], acc[kb]);
      }
    }
    if (more) {
      store_x(xr, cur ^ 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) g[j] = gn[j];
    }
    __syncthreads();
    cur ^= 1;
  }
  float* part = a.part + ((size_t)blockIdx.x) * a.N * K;
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb) {
    const int col = k0 + (kbw + kb) * 32 + i;
    if (col >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + mblock_row(kh, r);
      if (n < a.N) part[(size_t)n * K + col] = acc[kb][r];
    }
  }
  if (blockIdx.y == 0 && kbw == 0 && a.bpart) {
    const float tot = bsum + __shfl_xor(bsum, 32, 64);
    if (kh == 0 && nok) a.bpart[(size_t)blockIdx.x * a.N + n0 + i] = tot;
  }
}

// dW[e] = sum over chunks in a FIXED order: a workgroup owns 64 consecutive elements; wave s of its eight sums the chunks
// [s C/8, (s+1) C/8) of them (256-byte coalesced rows, 8 loads in flight), then the eight sums are added in wave order.
DFN_DEV void wgrad_reduce_body(const float* __restrict__ part, const float* __restrict__ bpart, int chunks, int N, int K,
                               float* __restrict__ dW, int ldw, int wcol, float* __restrict__ db, float (*sums)[64]) {
  const size_t nw = size_t(N) * K, total = nw + (db ? N : 0);
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t e = size_t(blockIdx.x) * 64 + lane;
  const bool ok = e < total, isw = e < nw;
  const float* src = isw ? part + e : bpart + (ok ? e - nw : 0);
  const size_t stride = isw ? nw : size_t(N);
  const int per = (chunks + 7) / 8, c0 = sl * per, c1 = c0 + per < chunks ? c0 + per : chunks;
  float s = 0.f;
  if (ok) {
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[size_t(c + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < c1; ++c) s += src[size_t(c) * stride];
  }
  sums[sl][lane] = s;
  __syncthreads();
  if (sl == 0 && ok) {
    float t = sums[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += sums[w][lane];
    if (isw) {
      const size_t n = e / K, k = e - n * K;
      dW[n * ldw + wcol + k] = t;
    } else {
      db[e - nw] = t;
    }
  }
}
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int chunks,
                                                           int N, int K, float* __restrict__ dW, int ldw, int wcol,
                                                           float* __restrict__ db) {
  __shared__ float sums[8][64];
  wgrad_reduce_body(part, bpart, chunks, N, K, dW, ldw, wcol, db, sums);
}
struct ReduceMulti { const float* part[3]; const float* bpart[3]; int chunks[3], N[3], K[3], ldw[3], wcol[3]; float* dW[3]; float* db[3]; };
__global__ __launch_bounds__(512) void wgrad_reduce_multi_kernel(ReduceMulti m) {
  __shared__ float sums[8][64];
  const int j = blockIdx.y;
  wgrad_reduce_body(m.part[j], m.bpart[j], m.chunks[j], m.N[j], m.K[j], m.dW[j], m.ldw[j], m.wcol[j], m.db[j], sums);   // (spare blocks: e >= total)
}

bool gemm_wgrad_multi_ok(const WgradJob* jobs, int n, long long P) {
  if (n < 1 || n > 3 || P <= 0) return false;
  for (int j = 0; j < n; ++j)
    if (jobs[j].N <= 0 || jobs[j].x.K <= 0 || wgrad_tiled(jobs[j].N, jobs[j].x, P)) return false;
  return true;
}
static_assert(sizeof(WgradMulti) + 16 <= 4096, "kernel arguments");
hipError_t gemm_wgrad_multi(const WgradJob* jobs, int n, float* scratch, long long P, hipStream_t s) {
  if (n < 1 || n > 3 || P <= 0) return hipErrorInvalidValue;
  WgradMulti m{};
  ReduceMulti r{};
  m.n = n;
  int max_chunks = 0, max_items = 0;
  size_t max_elems = 0;
  for (int j = 0; j < 3; ++j) {
    const WgradJob& q = jobs[j < n ? j : 0];
    if (q.N <= 0 || q.x.K <= 0 || wgrad_tiled(q.N, q.x, P)) return hipErrorInvalidValue;    // small products only
    const int ch = wgrad_chunk(P, wgrad_items(q.N, q.x.K));
    const int chunks = int((P + ch - 1) / ch);
    WgradArgs& a = m.a[j];
    a.G = q.G; a.ldg = q.ldg; a.N = q.N; a.x = q.x; a.P = P; a.chunk = ch;
    a.part = scratch;
    a.bpart = q.db ? scratch + size_t(chunks) * q.N * q.x.K : nullptr;
    a.kgroups = (q.x.K + kWgradK * 32 - 1) / (kWgradK * 32);
    a.items = ((q.N + 31) / 32) * a.kgroups;
    m.chunks[j] = j < n ? chunks : 0;
    r.part[j] = a.part; r.bpart[j] = a.bpart; r.chunks[j] = chunks; r.N[j] = j < n ? q.N : 0; r.K[j] = q.x.K; r.ldw[j] = q.ldw; r.wcol[j] = q.x.wcol;
    r.dW[j] = q.dW; r.db[j] = q.db;
    if (j < n) {
      scratch += gemm_wgrad_scratch_floats(q.N, q.x.K, P);
      max_chunks = chunks > max_chunks ? chunks : max_chunks;
      max_items = a.items > max_items ? a.items : max_items;
      const size_t elems = size_t(q.N) * q.x.K + (q.db ? q.N : 0);
      max_elems = elems > max_elems ? elems : max_elems;
    }
  }
  hipLaunchKernelGGL(gemm_wgrad_multi_kernel, dim3(max_chunks, max_items, n), dim3(256), 0, s, m);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)((max_elems + 63) / 64), n), dim3(512), 0, s, r);
  return hipGetLastError();
}

hipError_t gemm_wgrad(const float* G, int ldg, int N, const Seg& xseg, float* dW, int ldw, float* db, float* scratch,
                      long long P, hipStream_t s) {
  if (N <= 0 || xseg.K <= 0) return hipSuccess;
  if (P <= 0) return hipErrorInvalidValue;
  const bool tiled = wgrad_tiled(N, xseg, P);
  const int ch = tiled ? wgrad_tile_chunk(P, xseg.K) : wgrad_chunk(P, wgrad_items(N, xseg.K));
  const int chunks = int((P + ch - 1) / ch);
  WgradArgs a{};
  a.G = G; a.ldg = ldg; a.N = N; a.x = xseg; a.P = P; a.chunk = ch;
  a.part = scratch;
  a.bpart = db ? scratch + size_t(chunks) * N * xseg.K : nullptr;
  a.kgroups = (xseg.K + kWgradK * 32 - 1) / (kWgradK * 32);
  a.items = ((N + 31) / 32) * a.kgroups;
  if (tiled && N > 64) hipLaunchKernelGGL(gemm_wgrad_tile_kernel<4>, dim3(chunks, (xseg.K + 127) / 128), dim3(256), 0, s, a);
  else if (tiled) hipLaunchKernelGGL(gemm_wgrad_tile_kernel<2>, dim3(chunks, (xseg.K + 127) / 128), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gemm_wgrad_kernel, dim3(chunks, a.items), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((size_t(N) * xseg.K + (db ? N : 0) + 63) / 64)), dim3(512), 0, s, a.part, a.bpart, chunks,
                     N, xseg.K, dW, ldw, xseg.wcol, db);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ depths / encodings
__global__ __launch_bounds__(256) void stratified_z_kernel(const float* __restrict__ t_rand, size_t R, int Nc, float near, float far,
                                                           float* __restrict__ z, int lindisp) {
  const size_t n = R * size_t(Nc);
  for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const int i = int(e % Nc);
    const bool ld = lindisp != 0;
    const float zi = coarse_z_at(i, Nc, near, far, ld);
    if (!t_rand) { z[e] = zi; continue; }
    const float zm = i > 0 ? coarse_z_at(i - 1, Nc, near, far, ld) : zi, zp = i + 1 < Nc ? coarse_z_at(i + 1, Nc, near, far, ld) : zi;
    const float lower = i > 0 ? mul_rn(.5f, add_rn(zi, zm)) : zi;          // cat([z[:1], mids])
    const float upper = i + 1 < Nc ? mul_rn(.5f, add_rn(zp, zi)) : zi;      // cat([mids, z[-1:]])
    z[e] = add_rn(lower, mul_rn(sub_rn(upper, lower), t_rand[e]));
  }
}
hipError_t stratified_z(const float* t_rand, size_t R, int Nc, float near, float far, float* z, hipStream_t s, int lindisp) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(stratified_z_kernel, dim3(grid_for(R * Nc, 256)), dim3(256), 0, s, t_rand, R, Nc, near, far, z, lindisp);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void posenc_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ z, size_t R, int Ns, float* __restrict__ pe) {
  // one thread per (point, coordinate-frequency pair): 64 threads per point, column c of the 63-wide encoding each
  const size_t n = R * size_t(Ns) * 64;
  for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const size_t pt = e >> 6;
    const int c = int(e & 63);
    const size_t ray = pt / Ns;
    float v = 0.f;
    if (c < 63) {
      const int coord = c < 3 ? c : (c - 3) % 3;
      const float x = add_rn(rays_o[ray * 3 + coord], mul_rn(rays_d[ray * 3 + coord], z[pt]));
      if (c < 3) v = x;
      else {
        const int k = (c - 3) / 6;
        const bool is_cos = ((c - 3) % 6) >= 3;
        const float arg = x * float(1 << k);
        v = is_cos ? cosf(arg) : sinf(arg);
      }
    }
    pe[e] = v;
  }
}
hipError_t posenc_points(const float* rays_o, const float* rays_d, const float* z, size_t R, int Ns, float* pe, hipStream_t s) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(posenc_points_kernel, dim3(grid_for(R * Ns * 64, 256, 256 * 32)), dim3(256), 0, s, rays_o, rays_d, z, R, Ns, pe);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void head_prime_kernel(const float* __restrict__ raw, float* __restrict__ g, size_t n) {
  for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const int c = int(e % 9);
    const float y = raw[e];
    const bool softplus_head = c == 3 || c >= 7;
    g[e] *= softplus_head ? -expm1f(-y) : y * (1.f - y);   // softplus' from the output without the cancellation of 1 - e^-y at small densities
  }
}
hipError_t head_prime(const float* raw, float* g, size_t P, hipStream_t s) {
  if (!P) return hipSuccess;
  hipLaunchKernelGGL(head_prime_kernel, dim3(grid_for(P * 9, 256, 4096)), dim3(256), 0, s, raw, g, P * 9);
  return hipGetLastError();
}

// d/dx of [x, sin(2^k x), cos(2^k x)]: g_x + sum_k 2^k (cos(2^k x) g_sin_k - sin(2^k x) g_cos_k); columns as posenc_points_kernel.
DFN_DEV float posenc_adjoint(const float* g, float x, int coord, int L) {
  float acc = g[coord];
  for (int k = 0; k < L; ++k) {
    const float f = float(1 << k), arg = x * f;
    acc += f * (cosf(arg) * g[3 + 6 * k + coord] - sinf(arg) * g[3 + 6 * k + 3 + coord]);
  }
  return acc;
}
__global__ __launch_bounds__(256) void posenc_backward_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                              const float* __restrict__ viewdirs, const float* __restrict__ z,
                                                              const float* __restrict__ g_pe, const float* __restrict__ g_dpe, int ld_d,
                                                              size_t R, int Ns, float* __restrict__ gpts) {
  const size_t n = R * size_t(Ns) * 6;      // one thread per (point, output column)
  for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const size_t pt = e / 6;
    const int c = int(e - pt * 6);
    const size_t ray = pt / Ns;
    float v;
    if (c < 3) {
      const float x = add_rn(rays_o[ray * 3 + c], mul_rn(rays_d[ray * 3 + c], z[pt]));
      v = posenc_adjoint(g_pe + pt * 64, x, c, 10);
    } else {
      v = posenc_adjoint(g_dpe + pt * ld_d, viewdirs[ray * 3 + (c - 3)], c - 3, 4);
    }
    gpts[e] = v;
  }
}
hipError_t posenc_backward(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z, const float* g_pe,
                           const float* g_dpe, int ld_d, size_t R, int Ns, float* gpts, hipStream_t s) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(posenc_backward_kernel, dim3(grid_for(R * Ns * 6, 256, 8192)), dim3(256), 0, s, rays_o, rays_d, viewdirs, z, g_pe,
                     g_dpe, ld_d, R, Ns, gpts);
  return hipGetLastError();
}

__global__ __launch_bounds__(128) void ray_inputs_kernel(const float* __restrict__ viewdirs, const float* __restrict__ hist,
                                                         size_t hist_rows, const float* __restrict__ emb_a,
                                                         const float* __restrict__ emb_t, int hist_bin, int dim_a, int dim_t,
                                                         int n_vocab, size_t R, float* __restrict__ dir_in, int ld_dir,
                                                         float* __restrict__ t_in, int ld_t) {
  const int na = emb_a ? hist_bin * dim_a : 0, nt = (emb_t && t_in) ? hist_bin * dim_t : 0;
  for (size_t ray = blockIdx.x; ray < R; ray += gridDim.x) {
    const float* hrow = hist ? hist + (hist_rows == 1 ? 0 : ray) * hist_bin : nullptr;
    for (int c = threadIdx.x; c < ld_dir; c += blockDim.x) {
      float v = 0.f;
      if (c < kChDir) {
        const int coord = c < 3 ? c : (c - 3) % 3;
        const float x = viewdirs[ray * 3 + coord];
        if (c < 3) v = x;
        else {
          const int k = (c - 3) / 6;
          const float arg = x * float(1 << k);
          v = ((c - 3) % 6) >= 3 ? cosf(arg) : sinf(arg);
        }
      } else if (c < kChDir + na) {
        const int j = c - kChDir;
        long long idx = (long long)hrow[j / dim_a];   // .long() truncation (nerfw.py:69)
        idx = idx < 0 ? 0 : (idx >= n_vocab ? n_vocab - 1 : idx);
        v = emb_a[idx * dim_a + j % dim_a];
      }
      dir_in[ray * ld_dir + c] = v;
    }
    if (t_in)
      for (int c = threadIdx.x; c < ld_t; c += blockDim.x) {
        float v = 0.f;
        if (c < nt) {
          long long idx = (long long)hrow[c / dim_t];
          idx = idx < 0 ? 0 : (idx >= n_vocab ? n_vocab - 1 : idx);
          v = emb_t[idx * dim_t + c % dim_t];
        }
        t_in[ray * ld_t + c] = v;
      }
  }
}
hipError_t ray_inputs(const float* viewdirs, const float* hist, size_t hist_rows, const float* emb_a, const float* emb_t,
                      int hist_bin, int dim_a, int dim_t, int n_vocab, size_t R, float* dir_in, int ld_dir, float* t_in, int ld_t,
                      hipStream_t s) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(ray_inputs_kernel, dim3(grid_for(R, 1)), dim3(128), 0, s, viewdirs, hist, hist_rows, emb_a, emb_t, hist_bin,
                     dim_a, dim_t, n_vocab, R, dir_in, ld_dir, t_in, ld_t);
  return hipGetLastError();
}

// The head of the fused training forward in ONE launch (it was five tiny ones in a row on a one-stream step, ~4.7 us each): per ray the
// view direction (viewdirs_kernel), the coarse and the fine network's per-ray input rows (ray_inputs_kernel twice), the stratified
// coarse depths (stratified_z_kernel) — the same functions on the same values, bit for bit — and, by one thread, the step's range
// word cleared (the hipMemsetAsync in front of the weight packing).
__global__ __launch_bounds__(128) void train_ray_prep_kernel(TrainRayPrepArgs a) {
  __shared__ float s_v[3];
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.range_word) *a.range_word = 0;
  const int na = a.hist_bin * a.dim_a, nt = a.hist_bin * a.dim_t;
  for (size_t ray = blockIdx.x; ray < a.R; ray += gridDim.x) {
    if (threadIdx.x == 0) {
      float vx, vy, vz;
      normalize3(a.rays_d[ray * 3], a.rays_d[ray * 3 + 1], a.rays_d[ray * 3 + 2], vx, vy, vz);
      s_v[0] = vx; s_v[1] = vy; s_v[2] = vz;
      a.view[ray * 3] = vx; a.view[ray * 3 + 1] = vy; a.view[ray * 3 + 2] = vz;
    }
    __syncthreads();
    const float* hrow = a.hist + (a.hist_rows == 1 ? 0 : ray) * a.hist_bin;
    auto dir_value = [&](int c) {     // column c < kChDir of the direction encoding (ray_inputs_kernel)
      const int coord = c < 3 ? c : (c - 3) % 3;
      const float x = s_v[coord];
      if (c < 3) return x;
      const int k = (c - 3) / 6;
      const float arg = x * float(1 << k);
      return ((c - 3) % 6) >= 3 ? cosf(arg) : sinf(arg);
    };
    for (int c = threadIdx.x; c < a.ld_dc; c += blockDim.x) a.dir_c[ray * a.ld_dc + c] = c < kChDir ? dir_value(c) : 0.f;
    for (int c = threadIdx.x; c < a.ld_df; c += blockDim.x) {
      float v = 0.f;
      if (c < kChDir) v = dir_value(c);
      else if (c < kChDir + na) {
        const int j = c - kChDir;
        long long idx = (long long)hrow[j / a.dim_a];   // .long() truncation (nerfw.py:69)
        idx = idx < 0 ? 0 : (idx >= a.n_vocab ? a.n_vocab - 1 : idx);
        v = a.emb_a[idx * a.dim_a + j % a.dim_a];
      }
      a.dir_f[ray * a.ld_df + c] = v;
    }
    for (int c = threadIdx.x; c < a.ld_t; c += blockDim.x) {
      float v = 0.f;
      if (c < nt) {
        long long idx = (long long)hrow[c / a.dim_t];
        idx = idx < 0 ? 0 : (idx >= a.n_vocab ? a.n_vocab - 1 : idx);
        v = a.emb_t[idx * a.dim_t + c % a.dim_t];
      }
      a.t_in[ray * a.ld_t + c] = v;
    }
    const bool ld = a.lindisp != 0;
    for (int i = threadIdx.x; i < a.Nc; i += blockDim.x) {   // stratified_z_kernel
      const size_t e = ray * size_t(a.Nc) + i;
      const float zi = coarse_z_at(i, a.Nc, a.near, a.far, ld);
      if (!a.t_rand) { a.z[e] = zi; continue; }
      const float zm = i > 0 ? coarse_z_at(i - 1, a.Nc, a.near, a.far, ld) : zi, zp = i + 1 < a.Nc ? coarse_z_at(i + 1, a.Nc, a.near, a.far, ld) : zi;
      const float lower = i > 0 ? mul_rn(.5f, add_rn(zi, zm)) : zi;
      const float upper = i + 1 < a.Nc ? mul_rn(.5f, add_rn(zp, zi)) : zi;
      a.z[e] = add_rn(lower, mul_rn(sub_rn(upper, lower), a.t_rand[e]));
    }
    __syncthreads();   // (the next ray's direction overwrites s_v)
  }
}
hipError_t train_ray_prep(const TrainRayPrepArgs& a, hipStream_t s) {
  if (!a.R) return hipSuccess;
  hipLaunchKernelGGL(train_ray_prep_kernel, dim3(grid_for(a.R, 1)), dim3(128), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ coarse composite + sampler (training)
// One wave per ray (as nerfh_stages.hip).  alpha_i = 1 - exp(-delta_i relu(sigma_i + noise_i)), w = alpha T,
// rgb0 = sum w c, depth = sum w z, disp0 = 1 / max(1e-10, depth / sum w) (rendering.py:168-193,231-243); then
// sample_pdf on the interior weights with the injected u (rendering.py:24-65), z_std, and the sort of cat([z, z_samples]).
DFN_DEV void ray_weights_from_alpha_input(const float* seff, const float* z, int N, float* w, int lane) {
  float carry = 1.f;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int i = c0 + lane;
    float alpha = 0.f;
    if (i < N) {
      const float delta = i + 1 < N ? sub_rn(z[i + 1], z[i]) : 1e2f;
      alpha = sub_rn(1.f, expf(-mul_rn(delta, fmaxf(seff[i], 0.f))));
    }
    const float incl = wave_incl_prod(sub_rn(1.f, alpha), lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    if (i < N) w[i] = mul_rn(alpha, mul_rn(carry, excl));
    carry = mul_rn(carry, __shfl(incl, 63, 64));
  }
}

__global__ __launch_bounds__(256) void sample_fine_train_kernel(const float* __restrict__ raw_c, const float* __restrict__ z_c,
                                                                const float* __restrict__ noise, float noise_std,
                                                                const float* __restrict__ u, size_t R, int Nc, int Ni,
                                                                float* __restrict__ z_fine, float* __restrict__ rgb0,
                                                                float* __restrict__ disp0, float* __restrict__ acc0,
                                                                float* __restrict__ z_std) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Nf = Nc + Ni;
  const int per = 4 * Nc + 2 * Nf + Ni;
  float* s_sig = sm + size_t(wave) * per;   // [Nc] sigma + noise
  float* s_w = s_sig + Nc;                  // [Nc]
  float* s_mid = s_w + Nc;                  // [Nc]
  float* s_cdf = s_mid + Nc;                // [Nc]
  float* s_all = s_cdf + Nc;                // [Nf] cat([z, z_samples])
  float* s_out = s_all + Nf;                // [Nf]
  float* s_u = s_out + Nf;                  // [Ni]
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < R; ray += size_t(gridDim.x) * 4) {
    for (int i = lane; i < Nc; i += 64) {
      const float sg = raw_c[(ray * Nc + i) * 4 + 3];
      s_sig[i] = noise ? add_rn(sg, mul_rn(noise[ray * Nc + i], noise_std)) : sg;
      s_all[i] = z_c[ray * Nc + i];
    }
    if (u) for (int i = lane; i < Ni; i += 64) s_u[i] = u[ray * Ni + i];
    wave_sync();
    ray_weights_from_alpha_input(s_sig, s_all, Nc, s_w, lane);
    for (int i = lane; i < Nc - 1; i += 64) s_mid[i] = mul_rn(.5f, add_rn(s_all[i + 1], s_all[i]));
    wave_sync();
    // coarse maps
    float c3[3] = {0.f, 0.f, 0.f}, sa = 0.f, sd = 0.f;
    for (int i = lane; i < Nc; i += 64) {
      const float w = s_w[i];
      const float* rc = raw_c + (ray * Nc + i) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c) c3[c] += mul_rn(w, rc[c]);
      sa += w;
      sd += mul_rn(w, s_all[i]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) c3[c] = wave_sum(c3[c]);
    sa = wave_sum(sa);
    sd = wave_sum(sd);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb0[ray * 3 + c] = c3[c];
      acc0[ray] = sa;
      disp0[ray] = 1.f / fmaxf(1e-10f, sd / sa);
    }
    // inverse-CDF sampling on the interior weights
    {
      const int nb = Nc - 1, nw = nb - 1;
      const float* wts = s_w + 1;
      float part = 0.f;
      for (int i = lane; i < nw; i += 64) part += add_rn(wts[i], 1e-5f);
      const float total = wave_sum(part);
      float carry = 0.f;
      if (lane == 0) s_cdf[0] = 0.f;
      for (int c0 = 0; c0 < nw; c0 += 64) {
        const int i = c0 + lane;
        const float pdf = i < nw ? add_rn(wts[i], 1e-5f) / total : 0.f;
        const float incl = wave_incl_sum(pdf, lane);
        if (i < nw) s_cdf[i + 1] = add_rn(carry, incl);
        carry = add_rn(carry, __shfl(incl, 63, 64));
      }
      wave_sync();
      for (int j = lane; j < Ni; j += 64) {
        const float uj = u ? s_u[j] : unit_linspace(j, Ni);
        int lo = 0, hi = nb;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_cdf[mid] <= uj) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        const float cb = s_cdf[below], ca = s_cdf[above];
        float den = sub_rn(ca, cb);
        if (den < 1e-5f) den = 1.f;
        const float t = sub_rn(uj, cb) / den;
        s_all[Nc + j] = add_rn(s_mid[below], mul_rn(t, sub_rn(s_mid[above], s_mid[below])));
      }
    }
    wave_sync();
    float* zs = s_all + Nc;
    {  // z_std = std(z_samples, unbiased=False) (rendering.py:327)
      float sum = 0.f;
      for (int i = lane; i < Ni; i += 64) sum += zs[i];
      const float mean = wave_sum(sum) / float(Ni);
      float sq = 0.f;
      for (int i = lane; i < Ni; i += 64) { const float dlt = zs[i] - mean; sq += dlt * dlt; }
      sq = wave_sum(sq);
      if (lane == 0) z_std[ray] = sqrtf(sq / float(Ni));
    }
    // sort the samples (random u: arbitrary order), then merge with the sorted coarse depths by rank.  A power-of-two count (the
    // reference's 128) takes a bitonic network — log2(Ni) (log2(Ni) + 1) / 2 = 28 wave-wide compare-exchange steps; any other count
    // the odd-even transposition below (up to Ni double steps).  The sorted sequence does not depend on the network.
    const bool pow2 = (Ni & (Ni - 1)) == 0;
    if (pow2) {
      for (int k = 2; k <= Ni; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = lane; t < (Ni >> 1); t += 64) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
            const float a = zs[i], b = zs[l];
            if ((a > b) == ((i & k) == 0)) { zs[i] = b; zs[l] = a; }
          }
          wave_sync();
        }
    }
    for (int guard = 0; guard < Ni && !pow2; ++guard) {
      bool swapped = false;
      for (int phase = 0; phase < 2; ++phase) {
        for (int k = 2 * lane + phase; k + 1 < Ni; k += 128) {
          const float lo = zs[k], hi = zs[k + 1];
          if (lo > hi) { zs[k] = hi; zs[k + 1] = lo; swapped = true; }
        }
        wave_sync();
      }
      if (!__any(swapped)) break;
    }
    for (int i = lane; i < Nf; i += 64) {
      const float v = s_all[i];
      int lo = 0, hi, rank;
      if (i < Nc) {
        hi = Ni;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zs[mid] < v) lo = mid + 1; else hi = mid; }
        rank = i + lo;
      } else {
        hi = Nc;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_all[mid] <= v) lo = mid + 1; else hi = mid; }
        rank = (i - Nc) + lo;
      }
      s_out[rank] = v;
    }
    wave_sync();
    for (int i = lane; i < Nf; i += 64) z_fine[ray * Nf + i] = s_out[i];
    wave_sync();
  }
}
hipError_t sample_fine_train(const float* raw_c, const float* z_c, const float* noise, float noise_std, const float* u, size_t R,
                             int Nc, int Ni, float* z_fine, float* rgb0, float* disp0, float* acc0, float* z_std, hipStream_t s) {
  if (!R) return hipSuccess;
  const int per = 4 * Nc + 2 * (Nc + Ni) + Ni;
  hipLaunchKernelGGL(sample_fine_train_kernel, dim3(grid_for((R + 3) / 4, 1)), dim3(256), size_t(4) * per * 4, s, raw_c, z_c, noise,
                     noise_std, u, R, Nc, Ni, z_fine, rgb0, disp0, acc0, z_std);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ compositing backward (training)
// Coarse: rgb0 = sum_i w_i c_i, w_i = alpha_i T_i, alpha_i = 1 - exp(-delta_i relu(s_i)), s_i = sigma_i + noise_i.
//   d c_i = g w_i;   d s_i = [s_i > 0] delta_i ((1 - alpha_i) T_i g.c_i - S_i),  S_i = sum_{k>i} w_k g.c_k.
// Outputs are PRE-activation gradients: x c (1 - c) (Sigmoid head), x (1 - exp(-sigma)) (Softplus head).
__global__ __launch_bounds__(256) void composite_coarse_backward_kernel(const float* __restrict__ raw_c, const float* __restrict__ z_c,
                                                                        const float* __restrict__ noise, float noise_std,
                                                                        const float* __restrict__ g_rgb0, size_t R, int Nc,
                                                                        float* __restrict__ gpre, float* __restrict__ zero0, size_t n0,
                                                                        float* __restrict__ zero1, size_t n1) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // optional: two small buffers zeroed on the way (the embedding gradients that the step's scatter kernels accumulate into, far
  // behind this kernel in the stream: two memset launches less at the head of the backward pass)
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n0; i += size_t(gridDim.x) * blockDim.x) zero0[i] = 0.f;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n1; i += size_t(gridDim.x) * blockDim.x) zero1[i] = 0.f;
  float* s_e = sm + size_t(wave) * 3 * Nc;   // e_i = w_i g.c_i
  float* s_T = s_e + Nc;
  float* s_al = s_T + Nc;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < R; ray += size_t(gridDim.x) * 4) {
    const float g0 = g_rgb0[ray * 3], g1 = g_rgb0[ray * 3 + 1], g2 = g_rgb0[ray * 3 + 2];
    float carry = 1.f;
    for (int c0 = 0; c0 < Nc; c0 += 64) {
      const int i = c0 + lane;
      float alpha = 0.f;
      if (i < Nc) {
        const float sg = raw_c[(ray * Nc + i) * 4 + 3];
        const float se = noise ? add_rn(sg, mul_rn(noise[ray * Nc + i], noise_std)) : sg;
        const float delta = i + 1 < Nc ? sub_rn(z_c[ray * Nc + i + 1], z_c[ray * Nc + i]) : 1e2f;
        alpha = sub_rn(1.f, expf(-mul_rn(delta, fmaxf(se, 0.f))));
      }
      const float incl = wave_incl_prod(sub_rn(1.f, alpha), lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      if (i < Nc) {
        const float T = mul_rn(carry, excl);
        const float* rc = raw_c + (ray * Nc + i) * 4;
        s_T[i] = T;
        s_al[i] = alpha;
        s_e[i] = alpha * T * (g0 * rc[0] + g1 * rc[1] + g2 * rc[2]);
      }
      carry = mul_rn(carry, __shfl(incl, 63, 64));
    }
    wave_sync();
    // suffix sums S_i = sum_{k>i} e_k, from the back in 64-sample blocks
    float tail = 0.f;
    for (int c0 = ((Nc - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int i = c0 + lane;
      const float e = i < Nc ? s_e[i] : 0.f;
      const float suf = wave_incl_suffix_sum(e, lane);   // from the far end: eps x |S_i|, not eps x the block total
      const float blk = __shfl(suf, 0, 64);
      float later = __shfl_down(suf, 1, 64);
      if (lane == 63) later = 0.f;
      const float S = tail + later;          // strictly after i
      if (i < Nc) {
        const float* rc = raw_c + (ray * Nc + i) * 4;
        const float sg = rc[3];
        const float se = noise ? add_rn(sg, mul_rn(noise[ray * Nc + i], noise_std)) : sg;
        const float delta = i + 1 < Nc ? sub_rn(z_c[ray * Nc + i + 1], z_c[ray * Nc + i]) : 1e2f;
        const float T = s_T[i], al = s_al[i], w = al * T;
        const float gc = g0 * rc[0] + g1 * rc[1] + g2 * rc[2];
        float* o = gpre + (ray * Nc + i) * 4;
        o[0] = g0 * w * rc[0] * (1.f - rc[0]);
        o[1] = g1 * w * rc[1] * (1.f - rc[1]);
        o[2] = g2 * w * rc[2] * (1.f - rc[2]);
        const float ds = se > 0.f ? delta * ((1.f - al) * T * gc - S) : 0.f;
        o[3] = ds * -expm1f(-sg);
      }
      tail += blk;
    }
    wave_sync();
  }
}
hipError_t composite_coarse_backward(const float* raw_c, const float* z_c, const float* noise, float noise_std, const float* g_rgb0,
                                     size_t R, int Nc, float* gpre, hipStream_t s, float* zero0, size_t n0, float* zero1, size_t n1) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(composite_coarse_backward_kernel, dim3(grid_for((R + 3) / 4, 1)), dim3(256), size_t(4) * 3 * Nc * 4, s, raw_c,
                     z_c, noise, noise_std, g_rgb0, R, Nc, gpre, zero0, n0, zero1, n1);
  return hipGetLastError();
}

// Fine (training compositing, rendering.py:168-209): rgb = sum T (a_s c_s + a_t c_t), beta = sum T a_t b_t + beta_min.
// With g = d L / d rgb, gb = d L / d beta and e_i = T_i (a_s g.c_s + a_t (g.c_t + gb b_t)):
//   d c_s = g T a_s,  d c_t = g T a_t,  d b_t = gb T a_t,
//   d sigma_s = delta ((1 - a_s) T g.c_s - S),  d sigma_t = delta ((1 - a_t) T (g.c_t + gb b_t) - S) + g_tsigma,  S_i = sum_{k>i} e_k.
__global__ __launch_bounds__(256) void composite_fine_backward_train_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                            const float* __restrict__ g_rgb,
                                                                            const float* __restrict__ g_beta, float g_tsigma,
                                                                            const float* __restrict__ g_ts, size_t R, int Nf,
                                                                            float* __restrict__ gpre) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* s_e = sm + size_t(wave) * 2 * Nf;
  float* s_T = s_e + Nf;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < R; ray += size_t(gridDim.x) * 4) {
    const float g0 = g_rgb[ray * 3], g1 = g_rgb[ray * 3 + 1], g2 = g_rgb[ray * 3 + 2], gb = g_beta[ray];
    const float* rr = raw + ray * size_t(Nf) * 9;
    const float* zr = z + ray * size_t(Nf);
    float carry = 1.f;
    for (int c0 = 0; c0 < Nf; c0 += 64) {
      const int i = c0 + lane;
      float om = 1.f, e = 0.f;
      if (i < Nf) {
        const float* v = rr + size_t(i) * 9;
        const float delta = i + 1 < Nf ? sub_rn(zr[i + 1], zr[i]) : 1e2f;
        const float a_s = sub_rn(1.f, expf(-mul_rn(delta, v[3]))), a_t = sub_rn(1.f, expf(-mul_rn(delta, v[7])));
        om = expf(-mul_rn(delta, add_rn(v[3], v[7])));
        e = a_s * (g0 * v[0] + g1 * v[1] + g2 * v[2]) + a_t * (g0 * v[4] + g1 * v[5] + g2 * v[6] + gb * v[8]);
      }
      const float incl = wave_incl_prod(om, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      if (i < Nf) {
        const float T = carry * excl;
        s_T[i] = T;
        s_e[i] = T * e;
      }
      carry *= __shfl(incl, 63, 64);
    }
    wave_sync();
    float tail = 0.f;
    for (int c0 = ((Nf - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int i = c0 + lane;
      const float e = i < Nf ? s_e[i] : 0.f;
      const float suf = wave_incl_suffix_sum(e, lane);
      const float blk = __shfl(suf, 0, 64);
      float later = __shfl_down(suf, 1, 64);
      if (lane == 63) later = 0.f;
      const float S = tail + later;
      if (i < Nf) {
        const float* v = rr + size_t(i) * 9;
        const float delta = i + 1 < Nf ? sub_rn(zr[i + 1], zr[i]) : 1e2f;
        const float a_s = sub_rn(1.f, expf(-mul_rn(delta, v[3]))), a_t = sub_rn(1.f, expf(-mul_rn(delta, v[7])));
        const float T = s_T[i], ws = T * a_s, wt = T * a_t;
        const float gcs = g0 * v[0] + g1 * v[1] + g2 * v[2], gct = g0 * v[4] + g1 * v[5] + g2 * v[6] + gb * v[8];
        float* o = gpre + (ray * size_t(Nf) + i) * 9;
        o[0] = g0 * ws * v[0] * (1.f - v[0]);
        o[1] = g1 * ws * v[1] * (1.f - v[1]);
        o[2] = g2 * ws * v[2] * (1.f - v[2]);
        o[3] = delta * ((1.f - a_s) * T * gcs - S) * -expm1f(-v[3]);
        o[4] = g0 * wt * v[4] * (1.f - v[4]);
        o[5] = g1 * wt * v[5] * (1.f - v[5]);
        o[6] = g2 * wt * v[6] * (1.f - v[6]);
        o[7] = (delta * ((1.f - a_t) * T * gct - S) + g_tsigma + (g_ts ? g_ts[ray * size_t(Nf) + i] : 0.f)) * -expm1f(-v[7]);
        o[8] = gb * wt * -expm1f(-v[8]);
      }
      tail += blk;
    }
    wave_sync();
  }
}
hipError_t composite_fine_backward_train(const float* raw, const float* z, const float* g_rgb, const float* g_beta, float g_tsigma,
                                         const float* g_ts, size_t R, int Nf, float* gpre, hipStream_t s) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(composite_fine_backward_train_kernel, dim3(grid_for((R + 3) / 4, 1)), dim3(256), size_t(4) * 2 * Nf * 4, s, raw, z,
                     g_rgb, g_beta, g_tsigma, g_ts, R, Nf, gpre);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ small reductions
__global__ __launch_bounds__(256) void sum_over_samples_kernel(const float* __restrict__ g, int ld, int C, size_t R, int Ns,
                                                               float* __restrict__ out, int ldo) {
  // a block per (ray, 64-column group): lane = column (coalesced 256-byte row reads), the four waves take every fourth sample
  // and are added in fixed order through LDS (deterministic)
  __shared__ float red[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cgroups = (C + 63) / 64;
  for (size_t blk = blockIdx.x; blk < R * cgroups; blk += gridDim.x) {
    const size_t ray = blk / cgroups;
    const int c = int(blk - ray * cgroups) * 64 + lane;
    float s = 0.f;
    if (c < C) {
      const float* q = g + ray * size_t(Ns) * ld + c;
      for (int k = wave; k < Ns; k += 4) s += q[size_t(k) * ld];
    }
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) out[ray * ldo + c] = ((s + red[0][lane]) + red[1][lane]) + red[2][lane];
    __syncthreads();
  }
}
hipError_t sum_over_samples(const float* g, int ld, int C, size_t R, int Ns, float* out, int ldo, hipStream_t s) {
  if (!R) return hipSuccess;
  hipLaunchKernelGGL(sum_over_samples_kernel, dim3(grid_for(R * ((C + 63) / 64), 1)), dim3(256), 0, s, g, ld, C, R, Ns, out, ldo);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void embedding_scatter_kernel(const float* __restrict__ g_in, int ld, int off,
                                                                const float* __restrict__ hist, size_t hist_rows, int hist_bin,
                                                                int dim, int n_vocab, size_t R, float* __restrict__ grad_emb) {
  // One histogram for all rays (the usual case: the rays of a step come from one image): every ray adds into the same
  // hist_bin x dim entries, so a block first sums its rays per entry in LDS and issues ONE global atomic per entry
  // (1536 rays x 50 values on 50 addresses were 90 us of serialised atomics).  Per-ray histograms: direct atomics.
  __shared__ float acc[1024];
  const int per = hist_bin * dim;
  if (hist_rows == 1 && per <= 1024) {
    const size_t r0 = size_t(blockIdx.x) * 64, r1 = r0 + 64 < R ? r0 + 64 : R;
    // all 256 threads: thread (entry j, ray phase q) sums every nq-th ray of the block's 64 (was: `per` threads walking all 64 with one
    // dependent load each), the phases meet in LDS
    const int nq = per <= 256 ? 256 / per : 1;
    __shared__ float part[1024];
    for (int idx = threadIdx.x; idx < per * nq; idx += blockDim.x) {
      const int j = idx % per, q = idx / per;
      float s = 0.f;
      for (size_t r = r0 + q; r < r1; r += nq) s += g_in[r * ld + off + j];
      part[idx] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < per; j += blockDim.x) {
      float s = 0.f;
      for (int q = 0; q < nq; ++q) s += part[q * per + j];   // fixed order
      acc[j] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < per; j += blockDim.x) {
      long long idx = (long long)hist[j / dim];
      idx = idx < 0 ? 0 : (idx >= n_vocab ? n_vocab - 1 : idx);
      atomicAdd(grad_emb + idx * dim + j % dim, acc[j]);
    }
    return;
  }
  const size_t n = R * size_t(per);
  for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const size_t ray = e / per;
    const int j = int(e - ray * per);
    long long idx = (long long)hist[(hist_rows == 1 ? 0 : ray) * hist_bin + j / dim];
    idx = idx < 0 ? 0 : (idx >= n_vocab ? n_vocab - 1 : idx);
    atomicAdd(grad_emb + idx * dim + j % dim, g_in[ray * ld + off + j]);
  }
}
hipError_t embedding_scatter(const float* g_in, int ld, int off, const float* hist, size_t hist_rows, int hist_bin, int dim,
                             int n_vocab, size_t R, float* grad_emb, hipStream_t s) {
  if (!R) return hipSuccess;
  const bool shared = hist_rows == 1 && hist_bin * dim <= 1024;
  const int grid = shared ? int((R + 63) / 64) : grid_for(R * hist_bin * dim, 256);
  hipLaunchKernelGGL(embedding_scatter_kernel, dim3(grid), dim3(256), 0, s, g_in, ld, off, hist, hist_rows, hist_bin, dim, n_vocab, R,
                     grad_emb);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ NerfWLoss
// c_l = 0.5 mean((rgb0 - t)^2), f_l = mean((rgb - t)^2 / (2 beta^2)), b_l = 3 + mean(log beta), s_l = lambda_u mean(sigma_t),
// all x coef (losses.py:43-57); psnr = -10 log10(mean((rgb - t)^2)).  One workgroup; fp64 accumulators.
__global__ __launch_bounds__(1024) void nerfw_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb0,
                                                          const float* __restrict__ beta, const float* __restrict__ raw,
                                                          const float* __restrict__ target, size_t R, int Nf, float coef,
                                                          float lambda_u, float* __restrict__ loss5, float* __restrict__ g_rgb,
                                                          float* __restrict__ g_rgb0, float* __restrict__ g_beta,
                                                          double* __restrict__ part /* [gridDim.x] sigma_t partial sums */) {
  __shared__ double red[5][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // every block: its slice of sum(transient sigma) -> part[block]
  {
    double a = 0;
    const size_t n = R * size_t(Nf);
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) a += raw[e * 9 + 7];
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 0) red[0][wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
      for (int w = 0; w < nw; ++w) t += red[0][w];
      part[blockIdx.x] = t;
    }
    __syncthreads();
  }
  if (blockIdx.x != 0) return;
  // block 0: the per-ray terms and their gradient seeds (the sigma partials are folded in by nerfw_loss_finish_kernel)
  double acc[4] = {0, 0, 0, 0};   // c, f, log beta, mse
  const double inv3R = 1.0 / (3.0 * double(R));
  for (size_t r = threadIdx.x; r < R; r += blockDim.x) {
    const float b = beta[r];
    float gb = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float t = target[r * 3 + c];
      const float d0 = rgb0[r * 3 + c] - t, d1 = rgb[r * 3 + c] - t;
      acc[0] += 0.5 * double(d0) * d0;
      acc[1] += double(d1) * d1 / (2.0 * double(b) * b);
      acc[3] += double(d1) * d1;
      g_rgb0[r * 3 + c] = coef * d0 * float(inv3R);
      g_rgb[r * 3 + c] = coef * d1 / (b * b) * float(inv3R);
      gb -= d1 * d1 / (b * b * b);
    }
    acc[2] += log(double(b));
    g_beta[r] = coef * (gb * float(inv3R) + 1.f / (b * float(R)));
  }
  for (int k = 0; k < 4; ++k) {
    double v = acc[k];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k)
      for (int w = 0; w < nw; ++w) t[k] += red[k][w];
    loss5[0] = float(coef * t[0] * inv3R);
    loss5[1] = float(coef * t[1] * inv3R);
    loss5[2] = float(coef * (3.0 + t[2] / double(R)));
    loss5[4] = float(-10.0 * log10(t[3] * inv3R));
  }
}
__global__ void nerfw_loss_finish_kernel(const double* __restrict__ part, int nparts, size_t R, int Nf, float coef, float lambda_u,
                                         float* __restrict__ loss5) {
  double t = 0;
  for (int i = 0; i < nparts; ++i) t += part[i];   // fixed order
  loss5[3] = float(coef * lambda_u * t / (double(R) * Nf));
}
hipError_t nerfw_loss(const float* rgb, const float* rgb0, const float* beta, const float* raw, const float* target, size_t R,
                      int Nf, float coef, float lambda_u, float* loss5, float* g_rgb, float* g_rgb0, float* g_beta, hipStream_t s) {
  if (!R) return hipErrorInvalidValue;
  // the per-block sigma partials live in the caller's loss buffer behind the five results (kNerfwLossFloats floats, 16-byte aligned)
  constexpr int kBlocks = 64;
  static_assert(8 + 2 * kBlocks <= kNerfwLossFloats, "loss buffer too small for the partial sums");
  double* part = reinterpret_cast<double*>(loss5 + 8);
  hipLaunchKernelGGL(nerfw_loss_kernel, dim3(kBlocks), dim3(1024), 0, s, rgb, rgb0, beta, raw, target, R, Nf, coef, lambda_u, loss5, g_rgb,
                     g_rgb0, g_beta, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nerfw_loss_finish_kernel, dim3(1), dim3(1), 0, s, part, kBlocks, R, Nf, coef, lambda_u, loss5);
  return hipGetLastError();
}

}  // namespace train
}  // namespace dfn
