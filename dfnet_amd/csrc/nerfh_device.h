// nerfh_device.h — small device helpers shared by the MLP and the sampling/compositing kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dfn {

#define DFN_DEV __device__ __forceinline__

// ---- uncontracted fp32 arithmetic ---------------------------------------------------------------
// hipcc contracts a*b+c into an FMA by default and HIP's __fmul_rn/__fadd_rn do not stop it; these do
// (the pragma removes the `contract` flag from the instruction itself, which survives inlining).
// Used wherever the reference's op-by-op rounding is reproduced (depths, rays, sampling, compositing).
DFN_DEV float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
DFN_DEV float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
DFN_DEV float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// ---- linspace / coarse depths ---------------------------------------------------------------
// torch.linspace(0, 1, n)[i]: torch fills symmetrically from both ends, and its upper half
// `end - step*(n-1-i)` is compiled to ONE fused multiply-add (verified against torch CPU for
// n = 5..192: the fused form matches bit for bit, the unfused one does not).
DFN_DEV float unit_linspace(int i, int n) {
  const float step = n > 1 ? 1.f / float(n - 1) : 0.f;
  return i < n / 2 ? mul_rn(step, float(i)) : fmaf(-step, float(n - 1 - i), 1.f);
}
// z = near*(1-t) + far*t without FMA contraction; lindisp: z = 1/((1/near)(1-t) + (1/far) t), linear in disparity
// (reference: models/rendering.py:269-273).
DFN_DEV float coarse_z_at(int i, int n, float near, float far, bool lindisp = false) {
  const float t = unit_linspace(i, n);
  if (lindisp)
    return __fdiv_rn(1.f, add_rn(mul_rn(__fdiv_rn(1.f, near), sub_rn(1.f, t)), mul_rn(__fdiv_rn(1.f, far), t)));
  return add_rn(mul_rn(near, sub_rn(1.f, t)), mul_rn(far, t));
}

// ---- view direction -----------------------------------------------------------------------------
// v = d / |d| with every operation correctly rounded and uncontracted, so that every kernel that
// derives a view direction from the same d produces the same bits (rendering.py:366-371).
DFN_DEV void normalize3(float x, float y, float z, float& vx, float& vy, float& vz) {
  const float n = __fsqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
  vx = __fdiv_rn(x, n);
  vy = __fdiv_rn(y, n);
  vz = __fdiv_rn(z, n);
}

// ---- positional-encoding trig -----------------------------------------------------------------
// FAST path: u = x/(2 pi) as an unevaluated sum uh + ul (two-term product), so that
// frac(2^k u) keeps full fp32 accuracy up to 2^9 |x|; then the hardware sin/cos, which take
// their argument in revolutions.
constexpr float kInv2PiHi = 0.15915494f;     // fl32(1/(2 pi)) = 0x1.45f306p-3
constexpr float kInv2PiLo = 6.4206382e-09f;  // 1/(2 pi) - fl32(1/(2 pi))

DFN_DEV void rev_split(float x, float& uh, float& ul) {
  uh = x * kInv2PiHi;
  ul = fmaf(x, kInv2PiHi, -uh) + x * kInv2PiLo;
}
// sin/cos of 2*pi*f*(uh+ul) for f a power of two.
DFN_DEV void rev_sincos(float uh, float ul, float f, float& s, float& c) {
  const float t = __builtin_amdgcn_fractf(uh * f) + ul * f;
  s = __builtin_amdgcn_sinf(t);
  c = __builtin_amdgcn_cosf(t);
}
// Five octaves f0*2^k, k = 0..4, of one coordinate: hardware sin/cos at f0, then four double-angle
// steps (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a).  Each step doubles the inherited phase error:
// measured max abs error 7e-6 at the top octave vs fp64 (4e-7 at the base) -- 30x below the f16
// rounding of the encoding itself.  6 transcendentals per point and half instead of 30.
DFN_DEV void rev_sincos_octaves5(float uh, float ul, float f0, float (&s)[5], float (&c)[5]) {
  rev_sincos(uh, ul, f0, s[0], c[0]);
#pragma unroll
  for (int k = 1; k < 5; ++k) {
    const float t = s[k - 1] * c[k - 1];
    s[k] = t + t;
    c[k] = fmaf(-2.f * s[k - 1], s[k - 1], 1.f);
  }
}

// Wave-level ordering of LDS traffic: DS operations of one wavefront execute in issue order, so
// all that is needed between a lane's LDS store and another lane's load is to stop the
// compiler from reordering them.
DFN_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- activations -----------------------------------------------------------------------------
DFN_DEV float softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }  // nn.Softplus(beta=1, threshold=20)
DFN_DEV float sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
// Hardware-transcendental forms for the f16 path (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1e-6 relative;
// softplus loses only the sub-6e-8 tail of log(1+e^x) for very negative x): two orders of magnitude
// below the f16 rounding of the layer inputs, a fifth of the instructions.
DFN_DEV float softplus_fast(float v) { return v > 15.f ? v : __logf(1.f + __expf(v)); }
DFN_DEV float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
// fp32-grade forms on the hardware transcendentals (split-f16 kernels): e^x = 2^(t + r) with x log2(e) = t + r held as an
// unevaluated sum (fma remainder + the low word of log2 e) so that v_exp_f32's 1-ulp result is not degraded by the rounding of
// the product (libm's expf: ~35 instructions; this: 6); 1 / x = v_rcp_f32 + one Newton step.  Max relative error vs fp64
// ~1.5e-7 (tests/test_gpu_nerfh.py: raw parity of the split-f16 path unchanged at 1e-6).
DFN_DEV float exp_hw(float x) {
  const float t = x * 1.44269502f;
  const float r = fmaf(x, 1.44269502f, -t) + x * 1.92596299e-8f;
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e * r, 0.693147182f, e);
}
DFN_DEV float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return r * fmaf(-x, r, 2.f);
}
DFN_DEV float softplus_hw(float v) { return v > 20.f ? v : log1pf(exp_hw(v)); }
// (the exponent is clamped: beyond e^80 the sum is inf, and rcp_nr(inf) = 0 x fma(-inf, 0, 2) = NaN where sigmoid is 0 — a single
// saturated colour logit would poison its pixel; with the clamp a logit below -80 gives 1 / (1 + e^80) = 1.8e-35)
DFN_DEV float sigmoid_hw(float v) { return rcp_nr(1.f + exp_hw(fminf(-v, 80.f))); }
template <bool FAST> DFN_DEV float act_softplus(float v) { return FAST ? softplus_fast(v) : softplus(v); }
template <bool FAST> DFN_DEV float act_sigmoid(float v) { return FAST ? sigmoid_fast(v) : sigmoid(v); }

// ---- wavefront (64-lane) scans and reductions ---------------------------------------------------
DFN_DEV float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d, 64);
    if (lane >= d) v *= t;
  }
  return v;
}
DFN_DEV float wave_incl_sum(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
// Inclusive SUFFIX sum (lanes >= lane), accumulated from the far end: the error of S_i is eps x |S_i|, not eps x the wave's total as
// with `total - prefix` — the compositing gradients subtract S_i from a term of its own size (measured: d sigma 3.7e-4 -> round-off
// against float64 autograd, tools/gpu_debug_raygrad.py).
DFN_DEV float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_down(v, d, 64);
    if (lane + d < 64) v += t;
  }
  return v;
}
// ---- 32-lane scans on the DPP crossbar (gfx9 row_shr / row_bcast:15 / wave_shr): five VALU instructions per scan instead of
// six ds_bpermute round trips (~100 cycles each, dependent).  Lanes 0..31 hold the data (one point block of an MLP wave);
// lanes 32..63 must carry the identity.  The scan order differs from wave_incl_* (Hillis-Steele inside 16-lane rows, then one
// cross-row step): results agree to fp32 round-off, not bit for bit.
template <int CTRL, int ROW_MASK>
DFN_DEV float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
DFN_DEV float scan32_prod(float v) {   // inclusive prefix product over lanes 0..31
  v *= dpp_move<0x111, 0xf>(1.f, v);
  v *= dpp_move<0x112, 0xf>(1.f, v);
  v *= dpp_move<0x114, 0xf>(1.f, v);
  v *= dpp_move<0x118, 0xf>(1.f, v);
  v *= dpp_move<0x142, 0x2>(1.f, v);   // row_bcast:15: lane 15 into every lane of row 1
  return v;
}
DFN_DEV float scan32_sum(float v) {    // inclusive prefix sum over lanes 0..31 (lane 31 = the total)
  v += dpp_move<0x111, 0xf>(0.f, v);
  v += dpp_move<0x112, 0xf>(0.f, v);
  v += dpp_move<0x114, 0xf>(0.f, v);
  v += dpp_move<0x118, 0xf>(0.f, v);
  v += dpp_move<0x142, 0x2>(0.f, v);
  return v;
}
DFN_DEV float lane_shr1(float v, float first) { return dpp_move<0x138, 0xf>(first, v); }   // wave_shr:1: lane l <- lane l-1, lane 0 <- first
DFN_DEV float read_lane31(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31)); }

DFN_DEV float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

}  // namespace dfn
