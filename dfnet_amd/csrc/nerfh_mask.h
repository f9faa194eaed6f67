// nerfh_mask.h — helpers of the gradient chains (nerfh_bwd.hip: input gradient; nerfh_fused_chain.hip: training): ReLU sign masks
// as bit fields over the lane-resident B-operand registers, operand clearing, and the power-of-two renormalisation of a
// split-f16 gradient vector.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerfh_mlp_core.h"

namespace dfn {

template <class P, int N>
DFN_DEV float get_slot(const typename FragOf<P>::type (&arr)[N], int s) {
  if constexpr (P::kSplit) return ((float)arr[s >> 3].hi[s & 7] + (float)arr[s >> 3].lo[s & 7]) * (1.f / kX3ActScale);
  else if constexpr (P::kSlotsPerChunk == 8) return (float)arr[s >> 3][s & 7];
  else return arr[s];
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Sign mask of the first C chunks of v: bit = (v > 0).  f16 / split-f16: v is ALREADY rectified (the producing layer applied ReLU).
// Bit layout (split-f16 and f16): within a 32-bit word (4 chunks), the even slot of pair w of chunk cc sits at bit 4 cc + w and
// the odd slot at bit 16 + 4 cc + w — so that a PAIR of elements is handled by a few 32-bit integer instructions: after
// the ReLU the hi half of a split value is >= 0 and non-zero exactly when the value is positive (values below f16's smallest
// subnormal / 16 = 3.7e-9 count as zero).  fp32: bit e of word e / 32.
template <class P, int C, int N>
DFN_DEV void relu_mask(typename FragOf<P>::type (&v)[N], uint32_t (&m)[(C * P::kSlotsPerChunk + 31) / 32]) {
  constexpr int S = P::kSlotsPerChunk;
#pragma unroll
  for (int w = 0; w < (C * S + 31) / 32; ++w) m[w] = 0u;
  if constexpr (S == 8) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      u32x4 hw;
      if constexpr (P::kSplit) hw = __builtin_bit_cast(u32x4, v[c].hi);
      else hw = __builtin_bit_cast(u32x4, v[c]);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        // 1 per non-zero half: |h| clamped to 1 as an unsigned 16-bit integer (sign bit cleared first: -0 counts as zero), two
        // instructions + the shift-or instead of five in plain 32-bit arithmetic.  Inline asm on purpose: the same thing written
        // with clang's u16x2 vector types miscompiled (wrong gates, caught by test_mlp_fine_backward_vs_autograd).
        uint32_t nz;
        asm("v_and_b32 %0, 0x7fff7fff, %1\n\tv_pk_min_u16 %0, %0, %2" : "=v"(nz) : "v"(hw[w]), "s"(0x00010001u));
        m[c >> 2] |= nz << (4 * (c & 3) + w);
      }
    }
  } else {   // exact fp32: the layer leaves the pre-activation, rectified here
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const bool pos = v[c] > 0.f;
      m[c >> 5] |= (pos ? 1u : 0u) << (c & 31);
      v[c] = pos ? v[c] : 0.f;
    }
  }
}
// d pre-activation = d activation where the unit was active (torch: relu'(0) = 0).
template <class P, int C, int N>
DFN_DEV void apply_mask(typename FragOf<P>::type (&v)[N], const uint32_t (&m)[(C * P::kSlotsPerChunk + 31) / 32]) {
  constexpr int S = P::kSlotsPerChunk;
  if constexpr (S == 8) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      u32x4 keep;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
#ifdef DFN_MASK_MUL
        const uint32_t t = (m[c >> 2] >> (4 * (c & 3) + w)) & 0x00010001u;
        keep[w] = t * 0xFFFFu;   // 0xFFFF per active half
#else
        // 0xFFFF per active half in TWO packed 16-bit shifts: the pair's two bits sit at bit k of either half of the word (relu_mask's
        // layout), so a left shift by 15 - k puts each into its half's sign bit and an arithmetic right shift by 15 spreads it —
        // instead of shift, and, multiply.  (op_sel_hi:[0,1]: the scalar shift count serves both halves.)
        uint32_t t;
        asm("v_pk_lshlrev_b16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "s"(15 - (4 * (c & 3) + w)), "v"(m[c >> 2]));
        asm("v_pk_ashrrev_i16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "s"(15), "v"(t));
        keep[w] = t;
#endif
      }
      if constexpr (P::kSplit) {
        v[c].hi = __builtin_bit_cast(half8, __builtin_bit_cast(u32x4, v[c].hi) & keep);
        v[c].lo = __builtin_bit_cast(half8, __builtin_bit_cast(u32x4, v[c].lo) & keep);
      } else {
        v[c] = __builtin_bit_cast(half8, __builtin_bit_cast(u32x4, v[c]) & keep);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = ((m[c >> 5] >> (c & 31)) & 1u) ? v[c] : 0.f;
  }
}
template <class P>
DFN_DEV void clear_one(typename FragOf<P>::type& v) {
  if constexpr (P::kSplit) { v.hi = half8{0, 0, 0, 0, 0, 0, 0, 0}; v.lo = half8{0, 0, 0, 0, 0, 0, 0, 0}; }
  else if constexpr (P::kSlotsPerChunk == 8) v = half8{0, 0, 0, 0, 0, 0, 0, 0};
  else v = 0.f;
}
template <class P, int N>
DFN_DEV void clear(typename FragOf<P>::type (&v)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c) {
    if constexpr (P::kSplit) { v[c].hi = half8{0, 0, 0, 0, 0, 0, 0, 0}; v[c].lo = half8{0, 0, 0, 0, 0, 0, 0, 0}; }
    else if constexpr (P::kSlotsPerChunk == 8) v[c] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    else v[c] = 0.f;
  }
}

// Split-f16 gradient chain: a point's gradient vector is carried as sp * g with a per-point power of two sp (both
// halves of a point's lanes hold the same sp); every backward layer is linear, so sp rides through the MFMAs and the
// masks untouched and is divided out of the final 3 + 3 numbers.  Magnitudes drift from layer to layer, so sp is
// re-centred now and then: this returns the power of two that brings the largest |hi| of `v` to [8, 16) (= sp * g in
// [0.5, 1) after the x16 operand scale); it is applied to the NEXT layer's outputs (Stager::lane_mul), one layer lagged.
// `sp_now`: the scale the vector already carries.  The total is capped at 2^96: a point whose incoming gradient is ~1e-10 (a
// near-duplicate sample, delta z ~ 1e-8) and shrinks further along the chain would otherwise push sp past fp32's range, and
// inf * (d sigma) = NaN poisoned its ray's — and through the pose reduction its frame's — gradient.  Beyond the cap the halves are
// allowed to underflow: the true value is below 2^-96 of the gradient scale.
constexpr float kSpCap = 7.9228163e28f;   // 2^96
// rmax (optional): the range guard's running maximum — the operand's largest |hi| is taken here anyway, so the layer that produced `v`
// need not track its conversions (layer<..., TRACK = false>): a saturated hi half (65504) raises the guard's pattern.
template <class P, int C, int N>
DFN_DEV float renorm_factor(const typename FragOf<P>::type (&v)[N], float sp_now, uint32_t* rmax = nullptr) {
  if constexpr (!P::kSplit) return 1.f;
  else {
    half8 m = __builtin_elementwise_abs(v[0].hi);
#pragma unroll
    for (int c = 1; c < C; ++c) m = __builtin_elementwise_max(m, __builtin_elementwise_abs(v[c].hi));
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)m[j]);
    if (rmax && !(mx < 65504.f)) *rmax |= 0x7bff7bffu;   // (also catches a NaN)
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (!(mx > 0.f)) return 1.f;
    int e;
    (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
    const float f = ldexpf(1.f, 4 - e);
    return sp_now * f <= kSpCap ? f : 1.f;
  }
}

}  // namespace dfn
