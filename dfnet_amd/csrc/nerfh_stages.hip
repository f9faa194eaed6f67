// nerfh_stages.hip — the HBM-bound stages around the NeRF-H MLP (gfx950 only): ray generation,
// the per-ray folded bias of the two layers that take per-ray/per-image constants,
// coarse alpha/weights + inverse-CDF importance sampling + sort, and the static+transient
// alpha compositing.  One wavefront (64 lanes) owns one ray in the sampling / compositing
// kernels: scans along the ray are wave-level shuffles, per-ray arrays live in LDS.
//
// Replaces (reference, /root/reference/script/): models/ray_utils.py:5-15,
// models/rendering.py:24-65 (sample_pdf), :132-243 (raw2outputs_NeRFW), :295-304, :364-389;
// models/nerfw.py:69-81 (embedding lookup) and the dir/appearance/transient columns of
// dir_encoding / transient_encoding (nerfw.py:336-346).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_device.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"

namespace dfn {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

static inline int grid_for(size_t n, int block, int cap = 256 * 8) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return int(g < size_t(cap) ? g : size_t(cap));
}

// ------------------------------------------------------------------------------------------ raygen
__global__ __launch_bounds__(256) void raygen_kernel(int H, int W, float focal, const float* __restrict__ c2w,
                                                     float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                     float* __restrict__ viewdirs) {
  const size_t n = size_t(H) * W;
  // blockIdx.y = frame of a batch (dfn_raygen_frames): its own pose, its own slice of the outputs
  c2w += size_t(blockIdx.y) * 12;
  rays_o += size_t(blockIdx.y) * n * 3;
  rays_d += size_t(blockIdx.y) * n * 3;
  if (viewdirs) viewdirs += size_t(blockIdx.y) * n * 3;
  float R[3][3], t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b) R[a][b] = c2w[a * 4 + b];
    t[a] = c2w[a * 4 + 3];
  }
  const float hw = float(W) * .5f, hh = float(H) * .5f;
  for (size_t px = blockIdx.x * size_t(blockDim.x) + threadIdx.x; px < n; px += size_t(gridDim.x) * blockDim.x) {
    const int j = int(px / W), i = int(px - size_t(j) * W);
    const float dx = (float(i) - hw) / focal;
    const float dy = -(float(j) - hh) / focal;
    float d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)  // sum over the camera axis, left to right, no FMA (ray_utils.py:12)
      d[a] = add_rn(add_rn(mul_rn(dx, R[a][0]), mul_rn(dy, R[a][1])), mul_rn(-1.f, R[a][2]));
    float vd[3];
    normalize3(d[0], d[1], d[2], vd[0], vd[1], vd[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rays_o[px * 3 + a] = t[a];
      rays_d[px * 3 + a] = d[a];
      if (viewdirs) viewdirs[px * 3 + a] = vd[a];
    }
  }
}

hipError_t launch_raygen(int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d,
                         float* viewdirs, hipStream_t stream, int frames) {
  const size_t n = size_t(H) * W;
  if (!n || frames < 1) return hipSuccess;
  hipLaunchKernelGGL(raygen_kernel, dim3(grid_for(n, 256), frames), dim3(256), 0, stream, H, W, focal, c2w, rays_o, rays_d, viewdirs);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void viewdirs_kernel(const float* __restrict__ d, size_t n, float* __restrict__ v) {
  for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < n; r += size_t(gridDim.x) * blockDim.x) {
    normalize3(d[r * 3], d[r * 3 + 1], d[r * 3 + 2], v[r * 3], v[r * 3 + 1], v[r * 3 + 2]);
  }
}
hipError_t launch_viewdirs(const float* rays_d, size_t n, float* viewdirs, hipStream_t stream) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(viewdirs_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, rays_d, n, viewdirs);
  return hipGetLastError();
}

// Normalised device coordinates of forward-facing rays (models/ray_utils.py:27-46), operation by operation in the reference's
// order: origins to the near plane, then o' = (sx ox/oz, sy oy/oz, 1 + 2 near/oz), d' = (sx (dx/dz - ox/oz), sy (dy/dz - oy/oz),
// -2 near/oz) with sx = -1 / (W / (2 focal)), sy = -1 / (H / (2 focal)) (Python doubles there, rounded to fp32 at the multiply).
__global__ __launch_bounds__(256) void ndc_rays_kernel(float sx, float sy, float near, const float* __restrict__ ro,
                                                       const float* __restrict__ rd, size_t n, float* __restrict__ oo,
                                                       float* __restrict__ od) {
  for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < n; r += size_t(gridDim.x) * blockDim.x) {
    const float dx = rd[r * 3], dy = rd[r * 3 + 1], dz = rd[r * 3 + 2];
    const float t = __fdiv_rn(-add_rn(near, ro[r * 3 + 2]), dz);
    const float ox = add_rn(ro[r * 3], mul_rn(t, dx)), oy = add_rn(ro[r * 3 + 1], mul_rn(t, dy)), oz = add_rn(ro[r * 3 + 2], mul_rn(t, dz));
    const float two_n = mul_rn(2.f, near);
    oo[r * 3] = __fdiv_rn(mul_rn(sx, ox), oz);
    oo[r * 3 + 1] = __fdiv_rn(mul_rn(sy, oy), oz);
    oo[r * 3 + 2] = add_rn(1.f, __fdiv_rn(two_n, oz));
    od[r * 3] = mul_rn(sx, sub_rn(__fdiv_rn(dx, dz), __fdiv_rn(ox, oz)));
    od[r * 3 + 1] = mul_rn(sy, sub_rn(__fdiv_rn(dy, dz), __fdiv_rn(oy, oz)));
    od[r * 3 + 2] = __fdiv_rn(-two_n, oz);
  }
}
hipError_t launch_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, size_t n, float* out_o,
                           float* out_d, hipStream_t stream) {
  if (!n) return hipSuccess;
  const float sx = float(-1. / (double(W) / (2. * double(focal)))), sy = float(-1. / (double(H) / (2. * double(focal))));
  hipLaunchKernelGGL(ndc_rays_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, sx, sy, near, rays_o, rays_d, n, out_o, out_d);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ posenc (test entry)
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x, size_t n, int L, int mode,
                                                     float* __restrict__ out) {
  const int C = 3 + 6 * L;
  for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < n; r += size_t(gridDim.x) * blockDim.x) {
    float* o = out + r * C;
    for (int c = 0; c < 3; ++c) {
      const float xc = x[r * 3 + c];
      o[c] = xc;
      float uh, ul;
      rev_split(xc, uh, ul);
      float s5[5], c5[5];
      for (int k = 0; k < L; ++k) {
        const float f = float(1u << k);
        float s, cs;
        if (mode == 1) {  // exactly what the f16 MLP kernels do: groups of five octaves from one sin/cos
          if (k % 5 == 0) rev_sincos_octaves5(uh, ul, f, s5, c5);
          s = s5[k % 5];
          cs = c5[k % 5];
        } else {
          s = sinf(xc * f);
          cs = cosf(xc * f);
        }
        o[3 + 6 * k + c] = s;
        o[3 + 6 * k + 3 + c] = cs;
      }
    }
  }
}
hipError_t launch_posenc(const float* x, size_t n, int L, int mode, float* out, hipStream_t stream) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, x, n, L, mode, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ per-ray bias
// table[ray][0] = b_dir + W_dir[:, W:] . [pe_dir(viewdir) (27), a (hist_bin*dim_a)]     (nout = netwidth / 2 outputs)
// table[ray][1] = b_tr  + W_tr [:, W:] . t (hist_bin*dim_t)
// stored in C-fragment order [tbl][mb][h][r] so the fine kernel's accumulators load it directly.
__global__ __launch_bounds__(256) void ray_bias_kernel(RayBiasWeights w, const float* __restrict__ viewdirs,
                                                       const float* __restrict__ hist, size_t hist_rows,
                                                       size_t n_rays, float* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int na = w.hist_bin * w.dim_a, nt = w.hist_bin * w.dim_t;
  const int kd = kChDir + na;  // rows of w_dir
  const int NO = w.nout, nmb = NO / 32;
  float* s_wd = sm;                  // [kd][NO]
  float* s_wt = s_wd + kd * NO;      // [nt][NO]
  float* s_in = s_wt + nt * NO;      // [4 rays][kd + nt]
  float* s_base = s_in + 4 * (kd + nt);  // [2][NO] image-constant part (shared histogram)
  for (int i = threadIdx.x; i < kd * NO; i += blockDim.x) s_wd[i] = w.w_dir[i];
  for (int i = threadIdx.x; i < nt * NO; i += blockDim.x) s_wt[i] = w.w_tr[i];
  const int stride_in = kd + nt;
  const bool shared = hist_rows == 1;
  auto gather = [&](const float* hrow, float* dst, int i) {  // embedding lookups: a (na) then t (nt)
    const bool is_a = i < na;
    const int j = is_a ? i : i - na;
    const int dim = is_a ? w.dim_a : w.dim_t;
    long long idx = (long long)hrow[j / dim];  // .long() truncation (nerfw.py:69)
    idx = idx < 0 ? 0 : (idx >= w.n_vocab ? w.n_vocab - 1 : idx);
    dst[kChDir + i] = (is_a ? w.emb_a : w.emb_t)[idx * dim + (j % dim)];
  };
  if (shared) {  // one histogram for every ray: fold bias + appearance / transient columns once per block
    for (int i = threadIdx.x; i < na + nt; i += blockDim.x) gather(hist, s_in, i);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * NO; e += blockDim.x) {
      const int tbl = e / NO, f = e - tbl * NO;
      float acc = tbl ? w.b_tr[f] : w.b_dir[f];
      if (tbl == 0) for (int j = kChDir; j < kd; ++j) acc = fmaf(s_wd[j * NO + f], s_in[j], acc);
      else for (int j = 0; j < nt; ++j) acc = fmaf(s_wt[j * NO + f], s_in[kd + j], acc);
      s_base[e] = acc;
    }
  }
  const int rpb = shared ? 4 : 2;                         // rays per block iteration
  const int sub = shared ? threadIdx.x >> 6 : threadIdx.x >> 7;
  const int o = shared ? threadIdx.x & 63 : threadIdx.x & 127;
  float* my_in = s_in + sub * stride_in;
  __syncthreads();   // weights (and the shared-histogram base) staged
  // the next ray's view direction is fetched while this ray is processed: a ray is a handful of dependent LDS round trips, and the
  // HBM latency of its 12 input bytes at the head of every iteration was most of the kernel's time
  float v_next = 0.f;
  if (o < 6 * kLdir && size_t(blockIdx.x) * rpb + sub < n_rays) v_next = viewdirs[(size_t(blockIdx.x) * rpb + sub) * 3 + o % 3];
  for (size_t base = size_t(blockIdx.x) * rpb; base < n_rays; base += size_t(gridDim.x) * rpb) {
    // shared histogram: a ray is ONE wave's work (its inputs live in the wave's own LDS slice), so the waves of a block run free of
    // each other; per-ray histograms: two waves share a ray's inputs and meet at the block barrier
    if (shared) wave_sync(); else __syncthreads();
    const size_t ray = base + sub;
    const bool ok = ray < n_rays;
    if (ok) {
      if (o < 6 * kLdir) {  // pe_dir (27): [v, sin(2^k v), cos(2^k v)] — one lane per (coordinate, octave, sin | cos)
        const int coord = o % 3, k = (o / 3) % kLdir, is_cos = o / (3 * kLdir);
        const float v = v_next, arg = v * float(1 << k);
        const size_t rn = ray + size_t(gridDim.x) * rpb;
        if (rn < n_rays) v_next = viewdirs[rn * 3 + coord];
        if (o < 3) my_in[o] = v;
        my_in[3 + 6 * k + 3 * is_cos + coord] = is_cos ? cosf(arg) : sinf(arg);
      }
      if (!shared) {
        const float* hrow = hist + ray * w.hist_bin;
        for (int i = o; i < na + nt; i += 128) gather(hrow, my_in, i);
      }
    }
    if (shared) wave_sync(); else __syncthreads();
    if (ok) {
      auto put = [&](int tbl, int f, float v) {
        const int mb = f >> 5, row = f & 31;
        const int h = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
        table[ray * size_t(2 * NO) + ((tbl * nmb + mb) * 2 + h) * 16 + r] = v;
      };
      if (shared) {
        for (int f = o; f < NO; f += 64) {
          float acc = s_base[f];
          for (int j = 0; j < kChDir; ++j) acc = fmaf(s_wd[j * NO + f], my_in[j], acc);
          put(0, f, acc);
          put(1, f, s_base[NO + f]);
        }
      } else {
        for (int e = o; e < 2 * NO; e += 128) {
          const int tbl = e / NO, f = e - tbl * NO;
          float acc;
          if (tbl == 0) {
            acc = w.b_dir[f];
            for (int j = 0; j < kd; ++j) acc = fmaf(s_wd[j * NO + f], my_in[j], acc);
          } else {
            acc = w.b_tr[f];
            for (int j = 0; j < nt; ++j) acc = fmaf(s_wt[j * NO + f], my_in[kd + j], acc);
          }
          put(tbl, f, acc);
        }
      }
    }
  }
}

hipError_t launch_ray_bias(const RayBiasWeights& w, const float* viewdirs, const float* hist,
                           size_t hist_rows, size_t n_rays, float* table, hipStream_t stream) {
  if (!n_rays) return hipSuccess;
  const int na = w.hist_bin * w.dim_a, nt = w.hist_bin * w.dim_t;
  const size_t lds = size_t((kChDir + na) * w.nout + nt * w.nout + 4 * (kChDir + na + nt) + 2 * w.nout) * 4;
  const size_t rpb = hist_rows == 1 ? 4 : 2;
  static bool attr_done = false;
  if (lds > 64 * 1024 && !attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ray_bias_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(ray_bias_kernel, dim3(grid_for((n_rays + rpb - 1) / rpb, 1)), dim3(256), lds, stream, w, viewdirs,
                     hist, hist_rows, n_rays, table);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ per-ray wave helpers
// alpha_i = 1 - exp(-delta_i * relu(sigma_i)), w_i = alpha_i * prod_{j<i} (1 - alpha_j), for one ray
// held as sig[0..N) / z[0..N) in LDS (rendering.py:161-193).  Writes w[0..N).
DFN_DEV void ray_coarse_weights(const float* sig, const float* z, int N, float* w, int lane) {
  float carry = 1.f;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int i = c0 + lane;
    float alpha = 0.f;
    if (i < N) {
      const float delta = i + 1 < N ? sub_rn(z[i + 1], z[i]) : 1e2f;
      alpha = sub_rn(1.f, expf(-mul_rn(delta, fmaxf(sig[i], 0.f))));
    }
    const float incl = wave_incl_prod(sub_rn(1.f, alpha), lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    if (i < N) w[i] = mul_rn(alpha, mul_rn(carry, excl));
    carry = mul_rn(carry, __shfl(incl, 63, 64));
  }
}

// Inverse-CDF sampling of one ray (rendering.py:24-65).  bins[0..nb), wts[0..nb-1) in LDS;
// cdf[0..nb) scratch in LDS; writes out[0..Ni).  u == nullptr -> linspace(0,1,Ni).
DFN_DEV void ray_sample_pdf(const float* bins, const float* wts, int nb, float* cdf, const float* u, int Ni,
                            float* out, int lane) {
  const int nw = nb - 1;
  float part = 0.f;
  for (int i = lane; i < nw; i += 64) part += add_rn(wts[i], 1e-5f);
  const float total = wave_sum(part);
  float carry = 0.f;
  if (lane == 0) cdf[0] = 0.f;
  for (int c0 = 0; c0 < nw; c0 += 64) {
    const int i = c0 + lane;
    const float pdf = i < nw ? add_rn(wts[i], 1e-5f) / total : 0.f;
    const float incl = wave_incl_sum(pdf, lane);
    if (i < nw) cdf[i + 1] = add_rn(carry, incl);
    carry = add_rn(carry, __shfl(incl, 63, 64));
  }
  wave_sync();  // cdf visible to the whole wave
  for (int j = lane; j < Ni; j += 64) {
    const float uj = u ? u[j] : unit_linspace(j, Ni);
    int lo = 0, hi = nb;  // first index with cdf > u  (searchsorted right=True)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < nb - 1 ? lo : nb - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    float den = sub_rn(c1, c0);
    if (den < 1e-5f) den = 1.f;
    const float t = sub_rn(uj, c0) / den;
    out[j] = add_rn(bins[below], mul_rn(t, sub_rn(bins[above], bins[below])));
  }
}

// ------------------------------------------------------------------------------------------ standalone stage kernels
__global__ __launch_bounds__(256) void coarse_weights_kernel(const float* __restrict__ sigma, const float* __restrict__ z,
                                                             size_t n, int N, float* __restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* s_sig = sm + size_t(wave) * 3 * N;
  float* s_z = s_sig + N;
  float* s_w = s_z + N;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < n; ray += size_t(gridDim.x) * 4) {
    for (int i = lane; i < N; i += 64) { s_sig[i] = sigma[ray * N + i]; s_z[i] = z[ray * N + i]; }
    wave_sync();
    ray_coarse_weights(s_sig, s_z, N, s_w, lane);
    wave_sync();
    for (int i = lane; i < N; i += 64) weights[ray * N + i] = s_w[i];
    wave_sync();
  }
}
hipError_t launch_coarse_weights(const float* sigma, const float* z, size_t n, int N, float* weights,
                                 hipStream_t stream) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(coarse_weights_kernel, dim3(grid_for((n + 3) / 4, 1)), dim3(256), size_t(4) * 3 * N * 4, stream,
                     sigma, z, n, N, weights);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ wts,
                                                         size_t n, int nb, int Ni, const float* __restrict__ u,
                                                         float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = 3 * nb + 2 * Ni;
  float* s_bins = sm + size_t(wave) * per;
  float* s_w = s_bins + nb;
  float* s_cdf = s_w + nb;
  float* s_u = s_cdf + nb;
  float* s_out = s_u + Ni;
  for (size_t row = size_t(blockIdx.x) * 4 + wave; row < n; row += size_t(gridDim.x) * 4) {
    for (int i = lane; i < nb; i += 64) s_bins[i] = bins[row * nb + i];
    for (int i = lane; i < nb - 1; i += 64) s_w[i] = wts[row * (nb - 1) + i];
    if (u) for (int i = lane; i < Ni; i += 64) s_u[i] = u[row * Ni + i];
    wave_sync();
    ray_sample_pdf(s_bins, s_w, nb, s_cdf, u ? s_u : nullptr, Ni, s_out, lane);
    wave_sync();
    for (int i = lane; i < Ni; i += 64) out[row * Ni + i] = s_out[i];
    wave_sync();
  }
}
hipError_t launch_sample_pdf(const float* bins, const float* weights, size_t n, int nb, int Ni, const float* u,
                             float* out, hipStream_t stream) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3(grid_for((n + 3) / 4, 1)), dim3(256), size_t(4) * (3 * nb + 2 * Ni) * 4,
                     stream, bins, weights, n, nb, Ni, u, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ fused sampler
// sigma [R,Nc] -> z_fine [R,Nc+Ni]: coarse weights over the linspace depths, z_mid bins,
// sample_pdf(det) on the interior weights, then an exact sort of cat([z, z_samples]) (fix-up passes +
// rank merge of two sorted lists; ties: coarse depth first, so the result is always a permutation).
// __launch_bounds__'s second argument is waves per SIMD: 8 caps the kernel at 64 registers and costs 140 bytes of scratch per lane
// (-Rpass-analysis), and is still the fastest setting — a ray is ~50 dependent LDS round trips, so resident waves count for more than
// the spills: 61 440 rays in 71.9 us at 8, 78.0 at 6 (76 B of scratch), 78.5 at 4 (none); same box, three alternations each
// (tools/build_variant.sh + tools/gpu_ab_libs.sh "python tools/gpu_stage_timing.py").
#ifndef DFN_SF_WAVES
#define DFN_SF_WAVES 8
#endif
__global__ __launch_bounds__(256, DFN_SF_WAVES) void sample_fine_kernel(const float* __restrict__ sigma, size_t n_rays, int Nc, int Ni,
                                                          float near, float far, float* __restrict__ z_fine,
                                                          float* __restrict__ weights_out, float* __restrict__ zs_out, int lindisp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Nf = Nc + Ni;
  const int NfP = (Nf + 3) & ~3;
  const int per = (4 * Nc + NfP + Nf + 3) & ~3;
  float* s_sig = sm + size_t(wave) * per;   // [Nc]
  float* s_w = s_sig + Nc;                  // [Nc]
  float* s_mid = s_w + Nc;                  // [Nc]  (Nc-1 used)
  float* s_cdf = s_mid + Nc;                // [Nc]
  float* s_all = s_cdf + Nc;                // [NfP]  cat([z, z_samples]), 16-byte aligned
  float* s_out = s_all + NfP;               // [Nf]
  const size_t ray0 = size_t(blockIdx.x) * 4 + wave, stride = size_t(gridDim.x) * 4;
  // the next ray's first 64 densities are fetched while this ray is processed (a ray is ~50 dependent LDS round trips: the load
  // latency at its head was a tenth of it)
  float sig_next = (ray0 < n_rays && lane < Nc) ? sigma[ray0 * Nc + lane] : 0.f;
  for (size_t ray = ray0; ray < n_rays; ray += stride) {
    if (lane < Nc) s_sig[lane] = sig_next;
    for (int i = lane; i < Nc; i += 64) {
      if (i >= 64) s_sig[i] = sigma[ray * Nc + i];
      s_all[i] = coarse_z_at(i, Nc, near, far, lindisp != 0);
    }
    if (ray + stride < n_rays && lane < Nc) sig_next = sigma[(ray + stride) * Nc + lane];
    for (int i = Nf + lane; i < NfP; i += 64) s_all[i] = __builtin_inff();
    wave_sync();
    ray_coarse_weights(s_sig, s_all, Nc, s_w, lane);
    for (int i = lane; i < Nc - 1; i += 64) s_mid[i] = mul_rn(.5f, add_rn(s_all[i + 1], s_all[i]));
    wave_sync();
    ray_sample_pdf(s_mid, s_w + 1, Nc - 1, s_cdf, nullptr, Ni, s_all + Nc, lane);
    wave_sync();
    if (weights_out) for (int i = lane; i < Nc; i += 64) weights_out[ray * Nc + i] = s_w[i];
    if (zs_out) for (int i = lane; i < Ni; i += 64) zs_out[ray * Ni + i] = s_all[Nc + i];
    // Exact sort of cat([z, z_samples]): z is sorted by construction; the inverse-CDF samples are
    // sorted up to last-ulp inversions at bin boundaries, so a couple of odd-even transposition
    // passes (repeated until a full pass swaps nothing = sorted) fix them, then the two sorted lists
    // are merged by rank: rank(z_i) = i + #{samples < z_i}, rank(s_j) = j + #{z <= s_j}.
    float* zs = s_all + Nc;
    bool unsorted = false;                             // the usual case needs no pass at all: one look at every neighbour pair
    for (int k = lane; k + 1 < Ni; k += 64) unsorted |= zs[k] > zs[k + 1];
    for (int guard = 0; __any(unsorted) && guard < Ni; ++guard) {
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
    // Ranks without searching.  The coarse depths are a uniform grid: cnt_j = #{i : z_i <= s_j} is the sample's grid cell (computed,
    // then made exact against the stored depths), rank(s_j) = j + cnt_j.  A coarse depth is preceded by the samples below it:
    // s_j < z_i  <=>  cnt_j <= i, so rank(z_i) = i + (inclusive prefix over c <= i of the histogram of cnt), one wave scan.
    int* hist = reinterpret_cast<int*>(s_sig);       // [Nc + 1] over s_sig | s_w (both dead: the weights were written above)
    for (int i = lane; i <= Nc; i += 64) hist[i] = 0;
    wave_sync();
    // (lindisp: the grid is uniform in 1/z)
    const float g0 = lindisp ? 1.f / near : near, g1 = lindisp ? 1.f / far : far;
    const float cell = float(Nc - 1) / (g1 - g0);
    for (int j = lane; j < Ni; j += 64) {
      const float v = zs[j];
      int c = int(floorf(((lindisp ? 1.f / v : v) - g0) * cell)) + 1;
      c = c < 0 ? 0 : (c > Nc ? Nc : c);
      while (c < Nc && s_all[c] <= v) ++c;           // exact against the stored depths (the estimate is within a cell)
      while (c > 0 && s_all[c - 1] > v) --c;
      s_out[j + c] = v;
      atomicAdd(&hist[c], 1);
    }
    wave_sync();
    int before = 0;                                   // samples below the chunk's first coarse depth
    for (int c0 = 0; c0 < Nc; c0 += 64) {
      const int i = c0 + lane;
      int incl = i < Nc ? hist[i] : 0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      if (i < Nc) s_out[i + before + incl] = s_all[i];
      before += __shfl(incl, 63, 64);
    }
    wave_sync();
    for (int i = lane; i < Nf; i += 64) z_fine[ray * Nf + i] = s_out[i];
    wave_sync();
  }
}
hipError_t launch_sample_fine(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far,
                              float* z_fine, float* weights_coarse, float* z_samples, hipStream_t stream, int lindisp) {
  if (!n_rays) return hipSuccess;
  const int Nf = Nc + Ni, NfP = (Nf + 3) & ~3;
  int per = 4 * Nc + NfP + Nf;
  per = (per + 3) & ~3;
  hipLaunchKernelGGL(sample_fine_kernel, dim3(grid_for((n_rays + 3) / 4, 1)), dim3(256), size_t(4) * per * 4, stream,
                     sigma, n_rays, Nc, Ni, near, far, z_fine, weights_coarse, z_samples, lindisp);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ compositing
// One wave per ray, SPL consecutive samples per lane (rendering.py:144-243, fine branch).
template <int SPL>
__global__ __launch_bounds__(256) void composite_fine_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                             size_t n_rays, int Nf, float beta_min, int flags,
                                                             float* __restrict__ rgb, float* __restrict__ disp,
                                                             float* __restrict__ acc, float* __restrict__ depth,
                                                             float* __restrict__ weights, float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) float craw[];   // per wave: one ray's raw [Nf][9] (when staged)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool static_depth = (flags & 1) && (flags & 2);  // test_time && static_only
  const bool white = flags & 4;
  // A lane owns SPL consecutive samples = 9 SPL floats at a stride of 36 SPL bytes across lanes: read straight from HBM that is 27
  // four-byte loads per lane, each touching 64 different lines (2.4 TB/s).  The ray's 9 Nf floats are therefore staged through LDS
  // with 16-byte coalesced loads (a lane stride of 9 SPL floats is odd for SPL = 1, 3: conflict-free reads).
  const bool staged = (flags & 8) != 0;   // the launcher's decision: 16-byte alignment and the staged rays fit the default dynamic LDS
  float* sr = craw + size_t(wave) * Nf * 9;
  for (size_t ray = size_t(blockIdx.x) * 4 + wave; ray < n_rays; ray += size_t(gridDim.x) * 4) {
    const float* rr = raw + ray * size_t(Nf) * 9;
    const float* zr = z + ray * size_t(Nf);
    float v[SPL][9], zz[SPL + 1];
    if (staged) {
      const f32x4_t* src = reinterpret_cast<const f32x4_t*>(rr);
      f32x4_t* dst = reinterpret_cast<f32x4_t*>(sr);
      for (int i = lane; i < Nf * 9 / 4; i += 64) dst[i] = src[i];
      wave_sync();
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      zz[k] = i < Nf ? zr[i] : 0.f;
#pragma unroll
      for (int c = 0; c < 9; ++c) v[k][c] = i < Nf ? (staged ? sr[i * 9 + c] : rr[size_t(i) * 9 + c]) : 0.f;
    }
    if (staged) wave_sync();   // the next ray's staging overwrites the buffer
    zz[SPL] = __shfl_down(zz[0], 1, 64);
    float a_s[SPL], a_t[SPL], a_j[SPL];
    float pj = 1.f, ps = 1.f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      const float delta = i + 1 < Nf ? sub_rn(zz[k + 1], zz[k]) : 1e2f;
      const bool ok = i < Nf;
      a_s[k] = ok ? sub_rn(1.f, expf(-mul_rn(delta, v[k][3]))) : 0.f;
      a_t[k] = ok ? sub_rn(1.f, expf(-mul_rn(delta, v[k][7]))) : 0.f;
      a_j[k] = ok ? sub_rn(1.f, expf(-mul_rn(delta, add_rn(v[k][3], v[k][7])))) : 0.f;
      pj = mul_rn(pj, sub_rn(1.f, a_j[k]));
      ps = mul_rn(ps, sub_rn(1.f, a_s[k]));
    }
    float Tj = __shfl_up(wave_incl_prod(pj, lane), 1, 64);
    float Ts = __shfl_up(wave_incl_prod(ps, lane), 1, 64);
    if (lane == 0) { Tj = 1.f; Ts = 1.f; }
    float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_depth = 0.f, s_beta = 0.f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int i = lane * SPL + k;
      const float ws = mul_rn(a_s[k], Tj), wt = mul_rn(a_t[k], Tj), wj = mul_rn(a_j[k], Tj);
#pragma unroll
      for (int c = 0; c < 3; ++c) s_rgb[c] += mul_rn(ws, v[k][c]) + mul_rn(wt, v[k][4 + c]);
      s_acc += wj;
      s_beta += mul_rn(wt, v[k][8]);
      s_depth += static_depth ? mul_rn(mul_rn(a_s[k], Ts), zz[k]) : mul_rn(wj, zz[k]);
      if (weights && i < Nf) weights[ray * size_t(Nf) + i] = wj;
      Tj = mul_rn(Tj, sub_rn(1.f, a_j[k]));
      Ts = mul_rn(Ts, sub_rn(1.f, a_s[k]));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
    s_acc = wave_sum(s_acc);
    s_depth = wave_sum(s_depth);
    s_beta = wave_sum(s_beta);
    if (lane == 0) {
      const float bg = white ? 1.f - s_acc : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[ray * 3 + c] = s_rgb[c] + bg;
      disp[ray] = 1.f / fmaxf(1e-10f, s_depth / s_acc);
      acc[ray] = s_acc;
      if (depth) depth[ray] = s_depth;
      if (beta) beta[ray] = s_beta + beta_min;
    }
  }
}

hipError_t launch_composite_fine(const float* raw, const float* z, size_t n_rays, int Nf, float beta_min, int flags,
                                 float* rgb, float* disp, float* acc, float* depth, float* weights, float* beta,
                                 hipStream_t stream) {
  if (!n_rays) return hipSuccess;
  const int spl = (Nf + 63) / 64;
  const dim3 grid(grid_for((n_rays + 3) / 4, 1, 256 * 16)), block(256);
  size_t lds = size_t(4) * Nf * 9 * sizeof(float);   // the four waves' staged rays
  if ((Nf & 3) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 && lds <= 64 * 1024) flags |= 8;
  else lds = 0;
#define DFN_COMP(S)                                                                                                \
  hipLaunchKernelGGL(composite_fine_kernel<S>, grid, block, lds, stream, raw, z, n_rays, Nf, beta_min, flags, rgb, \
                     disp, acc, depth, weights, beta)
  if (spl <= 1) DFN_COMP(1);
  else if (spl == 2) DFN_COMP(2);
  else if (spl == 3) DFN_COMP(3);
  else if (spl == 4) DFN_COMP(4);
  else if (spl <= 6) DFN_COMP(6);
  else if (spl <= 8) DFN_COMP(8);
  else return hipErrorInvalidValue;
#undef DFN_COMP
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ segment combine
// Fused path: the fine MLP kernel composites each 64-sample segment locally; a ray's S segments chain as
// total += T_prefix * local_sum, T_prefix *= P_segment (rendering.py:176-230 factorised over segments).
__global__ __launch_bounds__(256) void composite_combine_kernel(const float* __restrict__ partial, size_t n_rays, int segs,
                                                                float beta_min, int flags, float* __restrict__ rgb,
                                                                float* __restrict__ disp, float* __restrict__ acc) {
  const bool static_depth = (flags & 1) && (flags & 2);
  for (size_t ray = blockIdx.x * size_t(blockDim.x) + threadIdx.x; ray < n_rays; ray += size_t(gridDim.x) * blockDim.x) {
    float Tj = 1.f, Ts = 1.f, r = 0.f, g = 0.f, b = 0.f, a = 0.f, d = 0.f;
    for (int s = 0; s < segs; ++s) {
      const float4* q = reinterpret_cast<const float4*>(partial + (ray * segs + s) * 12);
      const float4 u = q[0], v = q[1], w = q[2];
      r += Tj * u.x; g += Tj * u.y; b += Tj * u.z; a += Tj * u.w;
      d += static_depth ? Ts * v.x : Tj * v.y;
      Tj *= v.w;
      Ts *= w.x;
    }
    const float bg = (flags & 4) ? 1.f - a : 0.f;
    rgb[ray * 3] = r + bg; rgb[ray * 3 + 1] = g + bg; rgb[ray * 3 + 2] = b + bg;
    disp[ray] = 1.f / fmaxf(1e-10f, d / a);
    acc[ray] = a;
  }
}
hipError_t launch_composite_combine(const float* partial, size_t n_rays, int segs, float beta_min, int flags, float* rgb,
                                    float* disp, float* acc, hipStream_t stream) {
  if (!n_rays) return hipSuccess;
  hipLaunchKernelGGL(composite_combine_kernel, dim3(grid_for(n_rays, 256)), dim3(256), 0, stream, partial, n_rays, segs,
                     beta_min, flags, rgb, disp, acc);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ bicubic resize
// nn.Upsample(size, mode='bicubic') (align_corners=False, A = -0.75, border-replicated taps) on an
// [H, W, C] image -> [UH, UW, C]: the x4 enlargement of a quarter-resolution render
// (/root/reference/script/feature/misc.py:230-237, direct_feature_matching.py:344-346).
DFN_DEV float cubic_w1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }       // |x| <= 1
DFN_DEV float cubic_w2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }  // 1 < |x| < 2

// NCHW: the output is written planar, [C, UH, UW] per frame (what DFNet.forward takes: the permute(0, 3, 1, 2) of
// direct_feature_matching.py:346 folded into the store) instead of [UH, UW, C].
template <bool NCHW>
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ in, int H, int W, int C, int UH, int UW,
                                                      float* __restrict__ out) {
  const float sy = float(H) / float(UH), sx = float(W) / float(UW);
  const size_t n = size_t(UH) * UW * C;
  in += size_t(blockIdx.y) * H * W * C;   // blockIdx.y = frame of a batch
  out += size_t(blockIdx.y) * n;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    int c, X, Y;
    if (NCHW) { X = int(i % UW); Y = int((i / UW) % UH); c = int(i / (size_t(UW) * UH)); }
    else { c = int(i % C); X = int((i / C) % UW); Y = int(i / (size_t(C) * UW)); }
    const float fy = sy * (float(Y) + .5f) - .5f, fx = sx * (float(X) + .5f) - .5f;
    const int iy = int(floorf(fy)), ix = int(floorf(fx));
    const float ty = fy - float(iy), tx = fx - float(ix);
    const float wy[4] = {cubic_w2(ty + 1.f), cubic_w1(ty), cubic_w1(1.f - ty), cubic_w2(2.f - ty)};
    const float wx[4] = {cubic_w2(tx + 1.f), cubic_w1(tx), cubic_w1(1.f - tx), cubic_w2(2.f - tx)};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), H - 1);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), W - 1);
        row += wx[b] * in[(size_t(yy) * W + xx) * C + c];
      }
      acc += wy[a] * row;
    }
    out[i] = acc;
  }
}
hipError_t launch_bicubic(const float* in, int H, int W, int C, int UH, int UW, float* out, hipStream_t stream, int frames, bool nchw) {
  const size_t n = size_t(UH) * UW * C;
  if (!n || frames < 1) return hipSuccess;
  if (nchw) hipLaunchKernelGGL(bicubic_kernel<true>, dim3(grid_for(n, 256), frames), dim3(256), 0, stream, in, H, W, C, UH, UW, out);
  else hipLaunchKernelGGL(bicubic_kernel<false>, dim3(grid_for(n, 256), frames), dim3(256), 0, stream, in, H, W, C, UH, UW, out);
  return hipGetLastError();
}

}  // namespace dfn
