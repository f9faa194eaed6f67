// mfma_frag.h — precision traits -> MFMA operand types and the matching 32x32 instruction.
#pragma once
#include <hip/hip_runtime.h>

#include "nerfh_layout.h"

namespace dfn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class P> struct FragOf;
template <> struct FragOf<PrecF16> { using type = half8; using elem = _Float16; };
template <> struct FragOf<PrecF32> { using type = float; using elem = float; };
struct half8x2 { half8 hi, lo; };
template <> struct FragOf<PrecX3> { using type = half8x2; using elem = float; };

// D = A(32 x k) * B(k x 32) + C; A/B lanes: row/col = lane&31, k-half = lane>>5.
template <class P>
__device__ __forceinline__ f32x16 mfma(typename FragOf<P>::type a, typename FragOf<P>::type b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma<PrecF16>(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma<PrecF32>(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <>
__device__ __forceinline__ f32x16 mfma<PrecX3>(half8x2 a, half8x2 b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
}
// x (true scale) -> (hi, lo) halves of x * kX3ActScale.  No saturation here (it costs 6 % of the split-f16 render):
// the MLP's activations are O(1..100), hi overflows only beyond |x| = 4 094 (the pipelined render kernels narrow their accumulators
// before scaling them: 4 094 x the network's largest |weight| there — nerfh_mlp_core.h: X3Piece; guarded either way).
__device__ __forceinline__ void x3_split(float x, _Float16& hi, _Float16& lo) {
  const float xs = x * kX3ActScale;
  hi = (_Float16)xs;
  lo = (_Float16)(xs - (float)hi);
}

#define DFN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

}  // namespace dfn
