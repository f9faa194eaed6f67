// optim_adam.hip — optimizer.step() of the three training loops as ONE multi-tensor launch (gfx950).
//
// The reference steps torch.optim.Adam after every loss.backward(): /root/reference/script/run_nerf.py:65 (NeRF-H: 66 tensors,
// 0.34 M parameters), script/run_feature.py:65 and script/feature/direct_feature_matching.py (DFNet: 46 tensors, 15.4 M parameters).
// torch's multi-tensor implementation is eight to ten launches, each a pass over every tensor (lerp, mul, addcmul, sqrt, div, add,
// addcdiv): 66-70 us of a 1.8 ms NeRF-H step although the arithmetic is 0.3 M elements, 0.26 ms of a 9.6 ms DFNet step.  Here a
// step is one pass: every element is read (param, grad, exp_avg, exp_avg_sq) and written (param, exp_avg, exp_avg_sq) once.
//
// Arithmetic, element by element, in the order of torch/optim/adam.py (_multi_tensor_adam, the non-capturable branch):
//     g   = grad (+ weight_decay * param)
//     m   = m + (1 - beta1) * (g - m)                          (lerp_)
//     v   = v * beta2 + (1 - beta2) * g * g                    (mul_, addcmul_)
//     den = sqrt(v) / bias_correction2_sqrt + eps              (sqrt, div_, add_)
//     p   = p + step_size * (m / den),  step_size = -lr / bias_correction1      (addcdiv_)
// bias corrections and step_size are per TENSOR (each tensor carries its own step count in torch's state) and computed by the
// host in double, as torch does.  fp32 tensors only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"

namespace dfn {
namespace {
constexpr int kAdamMaxJobs = 72;          // tensors per launch: 72 x 48 B + header < the 4 KB kernel-argument limit
constexpr int kAdamBlockElems = 4096;     // elements per workgroup: 256 threads x 4 float4
struct AdamJob {
  float* p; const float* g; float* m; float* v;
  uint32_t n;            // elements
  uint32_t first_block;  // first workgroup of this tensor in the launch
  float step_size;       // -lr / bias_correction1
  float bc2_sqrt;        // sqrt(bias_correction2)
};
struct AdamArgs {
  AdamJob job[kAdamMaxJobs];
  int n_jobs;
  float one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay;
};
static_assert(sizeof(AdamArgs) <= 4096 - 64, "kernel arguments");

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamArgs& a, float step_size, float bc2_sqrt) {
  if (a.weight_decay != 0.f) g = g + a.weight_decay * p;
  m = m + a.one_minus_beta1 * (g - m);
  v = v * a.beta2 + a.one_minus_beta2 * g * g;
  const float den = sqrtf(v) / bc2_sqrt + a.eps;
  p = p + step_size * (m / den);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamArgs a) {
  // the tensor of this workgroup: binary search over first_block (wave-uniform: scalar loads from the argument segment)
  int lo = 0, hi = a.n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamJob& j = a.job[lo];
  const uint32_t base = (blockIdx.x - j.first_block) * uint32_t(kAdamBlockElems);
  const float ss = j.step_size, bs = j.bc2_sqrt;
  const bool vec = (((uintptr_t)j.p | (uintptr_t)j.g | (uintptr_t)j.m | (uintptr_t)j.v) & 15) == 0;
  if (vec && base + kAdamBlockElems <= j.n) {       // a whole block of an aligned tensor: four float4 per thread, all loads in flight
    float4 p[4], g[4], m[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = base + (k * 256 + threadIdx.x) * 4;
      p[k] = *reinterpret_cast<const float4*>(j.p + i);
      g[k] = *reinterpret_cast<const float4*>(j.g + i);
      m[k] = *reinterpret_cast<const float4*>(j.m + i);
      v[k] = *reinterpret_cast<const float4*>(j.v + i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam_elem(p[k].x, g[k].x, m[k].x, v[k].x, a, ss, bs);
      adam_elem(p[k].y, g[k].y, m[k].y, v[k].y, a, ss, bs);
      adam_elem(p[k].z, g[k].z, m[k].z, v[k].z, a, ss, bs);
      adam_elem(p[k].w, g[k].w, m[k].w, v[k].w, a, ss, bs);
      const uint32_t i = base + (k * 256 + threadIdx.x) * 4;
      *reinterpret_cast<float4*>(j.p + i) = p[k];
      *reinterpret_cast<float4*>(j.m + i) = m[k];
      *reinterpret_cast<float4*>(j.v + i) = v[k];
    }
    return;
  }
  for (uint32_t i = base + threadIdx.x; i < base + kAdamBlockElems && i < j.n; i += 256) {
    float p = j.p[i], m = j.m[i], v = j.v[i];
    adam_elem(p, j.g[i], m, v, a, ss, bs);
    j.p[i] = p; j.m[i] = m; j.v[i] = v;
  }
}
}  // namespace
}  // namespace dfn

extern "C" int dfn_adam_step(const dfn_adam_tensor* tensors, int n_tensors, double beta1, double beta2, double eps, double weight_decay,
                             void* stream) {
  using namespace dfn;
  if (n_tensors < 0 || (n_tensors && !tensors)) return set_error(DFN_ERR_ARG, "dfn_adam_step: bad argument");
  for (int i = 0; i < n_tensors; ++i) {
    const dfn_adam_tensor& t = tensors[i];
    if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return set_error(DFN_ERR_ARG, "dfn_adam_step: tensor %d has a null pointer", i);
    if (t.numel >= (size_t(1) << 32)) return set_error(DFN_ERR_ARG, "dfn_adam_step: tensor %d has %zu elements (2^32 at most)", i, t.numel);
    if (!(t.bias_correction2_sqrt > 0.f)) return set_error(DFN_ERR_ARG, "dfn_adam_step: tensor %d: bias_correction2_sqrt must be positive", i);
  }
  AdamArgs a{};
  // the hyper-parameters arrive as the doubles torch holds them in: 1 - beta is formed in double and rounded once, as the scalar
  // arguments of torch's kernels are (1 - float(0.999) is 1.3e-5 off 0.001)
  a.one_minus_beta1 = float(1.0 - beta1);
  a.beta2 = float(beta2);
  a.one_minus_beta2 = float(1.0 - beta2);
  a.eps = float(eps);
  a.weight_decay = float(weight_decay);
  int i = 0;
  while (i < n_tensors) {
    a.n_jobs = 0;
    uint64_t blocks = 0;
    for (; i < n_tensors && a.n_jobs < kAdamMaxJobs; ++i) {
      const dfn_adam_tensor& t = tensors[i];
      if (!t.numel) continue;
      const uint64_t nb = (t.numel + kAdamBlockElems - 1) / kAdamBlockElems;
      if (blocks + nb >= (uint64_t(1) << 31)) break;      // (never with real models: the grid of one launch)
      AdamJob& j = a.job[a.n_jobs++];
      j.p = t.param; j.g = t.grad; j.m = t.exp_avg; j.v = t.exp_avg_sq;
      j.n = uint32_t(t.numel);
      j.first_block = uint32_t(blocks);
      j.step_size = t.step_size;
      j.bc2_sqrt = t.bias_correction2_sqrt;
      blocks += nb;
    }
    if (!a.n_jobs) continue;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(uint32_t(blocks)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_adam_step: %s", hipGetErrorString(e));
  }
  return DFN_OK;
}
