// nerfh_mlp_core.h — device building blocks of the register-resident NeRF-H MLP, shared by the forward kernels
// (nerfh_mlp.hip) and the input-gradient kernel (nerfh_bwd.hip): weight staging (Stager / begin_unit), the
// generic MFMA layer, the positional encoding and the 8-layer trunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_frag.h"
#include "nerfh_device.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"

#ifndef DFN_PF
#define DFN_PF 3  // f16 A fragments in flight per wave
#endif

namespace dfn {

// chunks (B-operand registers groups) per 32 produced features / per n slots
template <class P> constexpr int chunks_of(int slots) { return slots / P::kSlotsPerChunk; }

template <class P, int KC>
DFN_DEV void set_slot(typename FragOf<P>::type (&arr)[KC], int s, float v) {
  if constexpr (P::kSplit) {
    _Float16 hi, lo;
    x3_split(v, hi, lo);
    arr[s >> 3].hi[s & 7] = hi;
    arr[s >> 3].lo[s & 7] = lo;
  } else if constexpr (P::kSlotsPerChunk == 8) arr[s >> 3][s & 7] = (_Float16)v;
  else arr[s] = v;
}

// ------------------------------------------------------------------------------------------
// Weight staging: the packed blob is a sequence of units (offset, bytes) in execution order.
struct Stager {
  const char* blob;
  const uint32_t* tab;
  int n_units;
  int u;               // unit that the NEXT open_unit() returns
  uint32_t slot;       // ring slot holding (or receiving) unit u
  uint32_t ring;       // staging buffers in the ring (3)
  uint32_t ustride;    // bytes per ring slot
  int lane, wave, waves;
  int dma_waves;       // the first dma_waves waves issue the weight DMA (the older wave of each SIMD idles at the unit barriers anyway)
  uint32_t ubase, uoff;      // LDS offset of the open unit / bytes of it consumed by the layers so far
  uint32_t pf_off, pf_size;  // table entry of the unit the NEXT mid_sync() will start streaming (prefetched)
  unsigned long long t_sync, t_wait, t_last;  // DFN_TIMING: cycles in unit waits / barrier
  unsigned long long* trace;
  int n_trace;
  bool more;           // another tile follows this one (wave-uniform)
  float in_scale, out_scale;  // split-f16: accumulators carry in_scale x the true value (MlpArgs), out_scale = 1 / in_scale
  float lane_mul;             // split-f16: extra per-LANE power-of-two factor on the next layer's outputs (gradient renormalisation)
  uint32_t rmax;              // range guard: running unsigned maximum of the packed f16 activations (hi halves for split-f16) this
                              // lane has produced, sign bits cleared: >= 0x7c00 = an inf, 0x7bff = the saturation value
};

// Range guard of the narrow arithmetic modes.  A ReLU output is a non-negative f16, so its bit pattern orders like the value: one
// v_pk_max_u16 per converted register keeps the largest pattern seen; layers without ReLU clear the sign bits first.  Checked
// once per kernel (range_report): f16 conversions overflow to inf (0x7c00), the split-f16 hi halves are truncating conversions
// that saturate at 65504 (0x7bff).
DFN_DEV uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
DFN_DEV void range_track(uint32_t& rmax, uint32_t packed, bool nonneg) {
  rmax = pk_max_u16(rmax, nonneg ? packed : (packed & 0x7fff7fffu));
}
template <class P>
DFN_DEV void range_report(uint32_t rmax, int* status) {
  if constexpr (P::kSlotsPerChunk == 8) {
    const uint32_t thr = P::kSplit ? 0x7bffu : 0x7c00u;
    if (status && ((rmax & 0xffffu) >= thr || (rmax >> 16) >= thr)) atomicOr(status, P::kSplit ? 2 : 1);
  }
}

// Direct-to-LDS DMA, issued as inline asm ON PURPOSE.  With the builtin, LLVM cannot tell which LDS bytes a DMA
// writes and makes EVERY later ds_read wait for vmcnt(0) (SIInsertWaitcnts, LDS-DMA aliasing): the first A-fragment
// read after a unit's DMA was issued then stalls for the whole L2 round trip, every unit (measured: 20 % of the
// kernel).  Here the landing is awaited explicitly (mid_sync: counted vmcnt + barrier) before any wave reads it.
// The compiler's own vmcnt bookkeeping stays safe: memory returns in order and it can only under-count loads that
// are in flight, so its waits are at worst longer than needed.  wave-contiguous: lane l writes 16 (4) bytes at
// lds_off + 16 l (4 l).
DFN_DEV void lds_dma_b128(const void* gptr, const char* lds_dst) {
  const uint32_t off = (uint32_t)(size_t)DFN_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(off) : "memory");
}
DFN_DEV void lds_dma_b32(const void* gptr, const char* lds_dst) {
  const uint32_t off = (uint32_t)(size_t)DFN_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(off) : "memory");
}

DFN_DEV void stage_issue_at(const Stager& st, char* smem, uint32_t off, uint32_t size, uint32_t lds_off) {
  const char* src = st.blob + off + st.lane * 16;
  if (st.wave >= st.dma_waves) return;
  for (uint32_t p = st.wave * kPiece; p < size; p += st.dma_waves * kPiece) lds_dma_b128(src + p, smem + lds_off + p);
}
// The unit table is read through the CONSTANT address space so that the loads are scalar (s_load, lgkmcnt): as
// vector loads they would sit in the vmcnt queue behind the DMA and every wait for them would wait for the DMA too.
typedef const uint32_t __attribute__((address_space(4))) const_u32;
DFN_DEV uint32_t tab_entry(const Stager& st, int i) {
  return reinterpret_cast<const_u32*>(reinterpret_cast<uint64_t>(st.tab))[i];
}
DFN_DEV void stage_issue(const Stager& st, char* smem, int unit, uint32_t lds_off) {
  stage_issue_at(st, smem, tab_entry(st, 2 * unit), tab_entry(st, 2 * unit + 1), lds_off);
}

// Three staging buffers rotate: while unit u is computed out of `lds_nn`... (see below) the DMA of unit u+1 is
// landing and unit u+2's is issued.  Synchronisation happens in the MIDDLE of a unit, not at its boundary:
//   open_unit()  no wait at all: the unit it returns was made visible by the mid_sync() of the previous unit;
//   mid_sync()   called once while a unit is being computed: wait for this wave's share of the NEXT unit (issued
//                a whole unit ago), barrier (every share landed; every wave has left the unit before this one),
//                then start the DMA of the unit after next into the buffer the previous unit occupied.
// So a wave flows from one layer into the next without stopping, and the barrier falls where both waves of a
// SIMD still have MFMAs queued on either side of it.
// What the ablations say about the remaining cost of streaming (fine kernel, one run): no barrier +4 %, no DMA
// +21 %, same DMA instructions with a quarter of the bytes +19 %, half the pieces +16 %, every piece from one
// cached KiB +2 %; staggering the two waves of a SIMD, or staging through VGPRs + ds_write_b128 instead of the
// DMA: no gain / -2.5 %.  The cost follows the BYTES landing in LDS, not instructions, waits or L2; a third of
// it is shader cycles, two thirds is clock (the traffic is paid in power: DESIGN.md section 3.1).
DFN_DEV void stage_prime(Stager& st, char* smem, uint32_t unit_stride, uint32_t ring = 3) {
  st.slot = 0; st.ring = ring; st.ustride = unit_stride;
  st.u = 0;
  stage_issue(st, smem, 0, 0);
  stage_issue(st, smem, st.n_units > 1 ? 1 : 0, unit_stride);
  const int n2 = st.n_units > 2 ? 2 : 0;
  st.pf_off = tab_entry(st, 2 * n2);
  st.pf_size = tab_entry(st, 2 * n2 + 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("" ::: "memory");
  __syncthreads();
}

// Returns the LDS byte offset of unit st.u (already visible) and advances the rotation.
DFN_DEV uint32_t open_unit(Stager& st) {
  const uint32_t cur = st.slot * st.ustride;
  st.slot = st.slot + 1 == st.ring ? 0u : st.slot + 1;   // now: the slot of the unit after the one just opened (landing / landed)
  int nxt = st.u + 1;
  if (nxt == st.n_units) nxt = 0;
  st.u = nxt;
  return cur;
}

// Once per unit, between two of its MFMA chunks.  After open_unit(): st.u = the unit after the open one.
DFN_DEV void mid_sync(Stager& st, char* smem) {
#ifdef DFN_TIMING
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
  // this wave's share of the next unit has landed
  // (the builtin, not asm: LLVM's waitcnt pass then knows its own older loads — the per-ray bias — have landed and
  // does not re-wait for them later with a count that the untracked DMA loads would turn into a wait for the DMA)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("" ::: "memory");
#ifdef DFN_TIMING
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
  // unit after next: open unit index is st.u - 1, so this is st.u + 1 (possibly in the next tile); it goes to the slot after
  // the next unit's, which held the unit BEFORE the open one: free once every wave is inside the open unit, i.e. after the barrier
  int n2 = st.u + 1;
  bool next_tile = st.u == 0;   // the open unit is the tile's last: both st.u and st.u + 1 belong to the next tile
  if (n2 >= st.n_units) { n2 -= st.n_units; next_tile = true; }
  const uint32_t dst = (st.slot + 1 == st.ring ? 0u : st.slot + 1) * st.ustride;
  __builtin_amdgcn_s_barrier();   // every share landed; every wave is inside the open unit
  asm volatile("" ::: "memory");
#ifdef DFN_TIMING
  const unsigned long long c2 = __builtin_amdgcn_s_memtime();
  st.t_wait += c1 - c0;
  st.t_sync += c2 - c1;
  if (st.trace && st.n_trace < 96 && st.lane == 0) {  // timeline of the first units: (enter, after barrier)
    st.trace[2 * st.n_trace] = c0;
    st.trace[2 * st.n_trace + 1] = c2;
  }
  ++st.n_trace;
#endif
  if (!next_tile || st.more) stage_issue_at(st, smem, st.pf_off, st.pf_size, dst);
  {  // fetch the table entry needed by the next call now, so its scalar-load latency is off the critical path
    int n3 = n2 + 1;
    if (n3 >= st.n_units) n3 -= st.n_units;
    st.pf_off = tab_entry(st, 2 * n3);
    st.pf_size = tab_entry(st, 2 * n3 + 1);
  }
}

// ------------------------------------------------------------------------------------------
// One 32-row M-block: acc[nb] += W_mb * Bin[nb] over KC chunks.  `wb` = LDS byte offset of the
// block's first A fragment, already including lane * kLaneBytes.
template <class P>
DFN_DEV typename FragOf<P>::type load_afrag(const char* p);

template <class P, int NB, int KC>
DFN_DEV void mblock_mma(const char* smem, uint32_t wb, const typename FragOf<P>::type (&Bin)[NB][KC],
                        f32x16 (&acc)[NB]) {
  using F = typename FragOf<P>::type;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const F a = load_afrag<P>(smem + wb + kc * 64 * P::kLaneBytes);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma<P>(a, Bin[nb][kc], acc[nb]);
  }
}

template <class P>
DFN_DEV typename FragOf<P>::type load_afrag(const char* p) {
  if constexpr (P::kSplit) {
    half8x2 f;
    f.hi = *reinterpret_cast<const half8*>(p);
    f.lo = *reinterpret_cast<const half8*>(p + 1024);
    return f;
  } else {
    return *reinterpret_cast<const typename FragOf<P>::type*>(p);
  }
}

DFN_DEV f32x16 load16(const float* p) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; r[8 + i] = c[i]; r[12 + i] = d[i]; }
  return r;
}

// C fragment -> B-operand registers of the next layer (ReLU optional).
template <class P, bool RELU, int OC>
DFN_DEV void store_hidden(const f32x16& acc, typename FragOf<P>::type (&out)[OC], int mb, float oscale, uint32_t& rmax) {
  if constexpr (P::kSplit) {
    // two values per conversion (v_cvt_pkrtz_f16_f32): hi is the TRUNCATED f16 of the scaled value — any f16 within an ulp will do,
    // the lo half takes the exact remainder — so a pair costs 2 packed converts + 2 mixed-precision subtractions
    typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
    const float os = oscale * kX3ActScale;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        float x0 = acc[8 * c + j] * os, x1 = acc[8 * c + j + 1] * os;      // true values x the operand scale (one power-of-two product)
        if (RELU) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
        const pk2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        // remainders x - hi in fp32 with the f16 halves read in place (v_fma_mix_f32; hipcc converts and subtracts otherwise)
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x1));
        const pk2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        range_track(rmax, hb, RELU);
        out[2 * mb + c].hi[j] = (_Float16)h[0]; out[2 * mb + c].hi[j + 1] = (_Float16)h[1];
        out[2 * mb + c].lo[j] = (_Float16)l[0]; out[2 * mb + c].lo[j + 1] = (_Float16)l[1];
      }
  } else if constexpr (P::kSlotsPerChunk == 8) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)acc[8 * c + j];
      if (RELU) {
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        v = __builtin_elementwise_max(v, zero);
      }
      {   // range guard: one packed maximum over the four registers, then one update of the running maximum
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w = __builtin_bit_cast(u32x4, v);
        if (!RELU) w &= 0x7fff7fffu;
        rmax = pk_max_u16(rmax, pk_max_u16(pk_max_u16(w[0], w[1]), pk_max_u16(w[2], w[3])));
      }
      out[2 * mb + c] = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * mb + r] = RELU ? fmaxf(acc[r], 0.f) : acc[r];
  }
}

template <class P, bool RELU, int OC>
DFN_DEV void store_hidden(const f32x16& acc, typename FragOf<P>::type (&out)[OC], int mb, float oscale = 1.f) {
  uint32_t unused = 0;
  store_hidden<P, RELU>(acc, out, mb, oscale, unused);
}

// One eighth of store_hidden: output register pair i (0..7) of an M-block's C fragment.  The two empty
// asm statements anchor the conversion at this point of the instruction stream: without them LLVM treats the
// pure arithmetic as freely movable and sinks it out from between the MFMAs it is meant to hide behind.
template <class P, bool RELU, int OC>
DFN_DEV void store_hidden_piece(const f32x16& acc, typename FragOf<P>::type (&out)[OC], int mb, int i, float oscale, uint32_t& rmax) {
  if constexpr (P::kSplit) {
    // split-f16: piece i = two consecutive results -> one register of the hi plane + one of the lo plane (the arithmetic of
    // store_hidden: scale, ReLU, truncating hi pair, exact remainders by v_fma_mix_f32, lo pair), as ONE volatile asm block so
    // that its eight VALU instructions stay between the MFMAs they hide behind.  `oscale` already holds out_scale x kX3ActScale.
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const int c = i >> 2, j = (i & 3) * 2;
    uint32_t hb, lb;
    float t0, t1;
    if (RELU)
      asm volatile("v_mul_f32 %2, %4, %6\n\tv_mul_f32 %3, %5, %6\n\tv_max_f32 %2, %2, 0\n\tv_max_f32 %3, %3, 0\n\t"
                   "v_cvt_pkrtz_f16_f32 %0, %2, %3\n\t"
                   "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                   "v_cvt_pkrtz_f16_f32 %1, %2, %3"
                   : "=&v"(hb), "=&v"(lb), "=&v"(t0), "=&v"(t1) : "v"(acc[8 * c + j]), "v"(acc[8 * c + j + 1]), "v"(oscale));
    else
      asm volatile("v_mul_f32 %2, %4, %6\n\tv_mul_f32 %3, %5, %6\n\t"
                   "v_cvt_pkrtz_f16_f32 %0, %2, %3\n\t"
                   "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                   "v_cvt_pkrtz_f16_f32 %1, %2, %3"
                   : "=&v"(hb), "=&v"(lb), "=&v"(t0), "=&v"(t1) : "v"(acc[8 * c + j]), "v"(acc[8 * c + j + 1]), "v"(oscale));
    if (RELU) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(rmax) : "v"(hb));   // range guard (range_track), pinned with the piece
    else range_track(rmax, hb, false);
    const half2v hv = __builtin_bit_cast(half2v, hb), lv = __builtin_bit_cast(half2v, lb);
    out[2 * mb + c].hi[j] = hv[0]; out[2 * mb + c].hi[j + 1] = hv[1];
    out[2 * mb + c].lo[j] = lv[0]; out[2 * mb + c].lo[j + 1] = lv[1];
  } else if constexpr (P::kSlotsPerChunk == 8) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const int c = i >> 2, j = i & 3;
    // volatile asm = fixed position in the instruction stream (between the MFMAs it hides behind), plain
    // register reads of the finished accumulators, no copies.  The accumulators read here were written by
    // MFMAs at least three MFMA issues earlier, which covers the XDL-write -> VALU-read wait states.
    uint32_t bits;
    if (RELU)
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=v"(bits) : "v"(acc[8 * c + 2 * j]), "v"(acc[8 * c + 2 * j + 1]));
    else
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(bits) : "v"(acc[8 * c + 2 * j]), "v"(acc[8 * c + 2 * j + 1]));
    if (RELU) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(rmax) : "v"(bits));   // range guard (range_track), pinned with the piece
    else range_track(rmax, bits, false);
    const half2v v = __builtin_bit_cast(half2v, bits);
    out[2 * mb + c][2 * j] = v[0];
    out[2 * mb + c][2 * j + 1] = v[1];
  } else {
    float x0 = acc[2 * i], x1 = acc[2 * i + 1];
    asm volatile("" : "+v"(x0), "+v"(x1));
    x0 = RELU ? fmaxf(x0, 0.f) : x0;
    x1 = RELU ? fmaxf(x1, 0.f) : x1;
    asm volatile("" : "+v"(x0), "+v"(x1));
    out[16 * mb + 2 * i] = x0;
    out[16 * mb + 2 * i + 1] = x1;
  }
}

// The split-f16 conversion piece in three parts, so that a K-chunk can place them BETWEEN its three dependent MFMAs (A after the
// first, B after the second, C after the third) instead of as one block behind them: the parts of store_hidden_piece<PrecX3>.
struct X3Piece {
  uint32_t hb, hs, lb;
  float t0, t1;
  // A: ReLU on the UNSCALED accumulators (they carry in_scale x the value: the packer keeps in_scale x |activation| inside f16).
  template <bool RELU>
  DFN_DEV void A(const f32x16& acc, int i) {
    const int c = i >> 2, j = (i & 3) * 2;
    if (RELU) asm volatile("v_max_f32 %0, %2, 0\n\tv_max_f32 %1, %3, 0" : "=&v"(t0), "=&v"(t1) : "v"(acc[8 * c + j]), "v"(acc[8 * c + j + 1]));
    else asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(t0), "=&v"(t1) : "v"(acc[8 * c + j]), "v"(acc[8 * c + j + 1]));
  }
  // B: truncated hi pair of the unscaled values, scaled as a PACKED f16 product (a power of two: exact), then the two remainders
  // t x s - hi' in one mixed-precision FMA each — the accumulator -> operand scale costs one instruction per PAIR instead of two.
  DFN_DEV void B(float s, uint32_t s2) {
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\t"
                 "v_pk_mul_f16 %1, %0, %5\n\t"
                 "v_fma_mix_f32 %2, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mix_f32 %3, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                 : "=&v"(hb), "=&v"(hs), "+v"(t0), "+v"(t1) : "v"(s), "v"(s2));
  }
  template <bool RELU, int OC>
  DFN_DEV void C(half8x2 (&out)[OC], int mb, int i, uint32_t& rmax) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const int c = i >> 2, j = (i & 3) * 2;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(lb) : "v"(t0), "v"(t1));
    // range guard on the UNSCALED hi pair: that conversion is the one that saturates
    if (RELU) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(rmax) : "v"(hb));
    else range_track(rmax, hb, false);
    const half2v hv = __builtin_bit_cast(half2v, hs), lv = __builtin_bit_cast(half2v, lb);
    out[2 * mb + c].hi[j] = hv[0]; out[2 * mb + c].hi[j + 1] = hv[1];
    out[2 * mb + c].lo[j] = lv[0]; out[2 * mb + c].lo[j + 1] = lv[1];
  }
};

// A layer whose MB output M-blocks feed the next layer, plus (EXTRA) one trailing head M-block
// whose raw accumulators go back to the caller.  Weights arrive in staging units of UMB M-blocks;
// NEWUNIT = false continues inside the unit opened by the previous layer (small layers are packed
// several to a unit: no barrier, no DMA wait between them).
//
// Software pipeline: inside a staging unit the A fragments form one flat stream t = (m-block, chunk);
// fragment t+PF is fetched from LDS before the MFMAs of fragment t issue, so the ~100+-cycle LDS
// latency hides under PF*NB MFMAs instead of stalling every chunk.  The bias is the C operand of
// each accumulator's FIRST MFMA (no register copies) and is fetched one M-block ahead; RAYBIAS
// layers seed their accumulators from the per-ray table.
//
// PIPE: the f32 -> f16 (+ReLU) conversion of M-block m-1 is issued piecewise between the MFMAs of
// M-block m.  CIN >= 0: the previous layer left its LAST M-block unconverted in `carry`; it lands in
// chunks CIN, CIN+1 of Bin and is converted during this layer's chunks 0..CIN-1 (before they are read).
// COUT: leave this layer's last M-block in `carry` for the next layer instead of converting it here.
// NOBIAS: the unit's bias fragments are all zero (backward layers): no bias reads, the first MFMA starts from a zero C operand.
// SCALE_FIRST (split-f16, PIPE): the pipelined conversion in its scale-in-fp32-first form (store_hidden_piece, a block of pieces behind
// each chunk's MFMAs) instead of the three-part X3Piece form, which converts the UNSCALED accumulators and so needs the render
// kernels' unit weight scale — the gradient kernels keep the 2^10 scale and per-point renormalisation factors (Stager::lane_mul).
// TRACK = false: the range guard does not watch this layer's conversions (the caller checks the finished operand itself: the gradient
// kernel's renormalisation already takes the maximum |hi| of it, renorm_factor).
template <class P, int UMB, bool PIPE, int NB, int KC, int MB, bool RELU, bool EXTRA, bool RAYBIAS, bool NEWUNIT,
          int CIN, bool CIN_RELU, bool COUT, bool NOBIAS = false, bool SCALE_FIRST = false, bool TRACK = true>
DFN_DEV void layer(Stager& st, char* smem, typename FragOf<P>::type (&Bin)[NB][KC],
                   typename FragOf<P>::type (&Bout)[NB][(MB ? MB : 1) * chunks_of<P>(16)],
                   f32x16 (&head)[NB], const float* const (&raybias)[NB], f32x16 (&carry)[NB]) {
  using F = typename FragOf<P>::type;
  constexpr int TOT = MB + (EXTRA ? 1 : 0);
  constexpr int PF = P::kSplit ? 2 : (P::kSlotsPerChunk == 8 ? DFN_PF : 4);  // fragments in flight
  constexpr uint32_t FB = 64 * P::kLaneBytes;         // bytes of one A fragment
  constexpr int PPK = (8 * NB + KC - 1) / KC;         // conversion pieces (8 per point block) interleaved per chunk
  constexpr int PPKI = CIN > 0 ? (8 * NB + CIN - 1) / CIN : 0;  // same for the carried-in M-block
  static_assert(CIN < 0 || (PIPE && P::kSlotsPerChunk == 8 && CIN > 0 && CIN + 1 < KC + 1), "carry-in needs the pipelined f16 path");
  static_assert(!COUT || (PIPE && MB >= 1 && !EXTRA), "carry-out needs a regular last M-block");
  static_assert(NEWUNIT || UMB >= TOT, "a layer that continues a unit must fit in it");
  const int h = st.lane >> 5;
  uint32_t rmax_sink = 0;
  uint32_t& rmax_ = TRACK ? st.rmax : rmax_sink;
  const float pscale = P::kSplit ? st.out_scale * st.lane_mul * kX3ActScale : 1.f;  // split-f16 pieces: accumulator -> operand scale
  uint32_t pscale2 = 0;   // ... the same factor as a packed f16 pair (X3Piece::B)
  if constexpr (P::kSplit) asm("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(pscale2) : "v"(pscale));
  constexpr bool RB_ALL = RAYBIAS && NB <= 2;  // fetch all per-ray seeds at entry (latency behind the barrier)
  f32x16 rb[RB_ALL ? TOT : 1][NB];
  if (RB_ALL) {
#pragma unroll
    for (int mb = 0; mb < TOT; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        rb[mb][nb] = load16(raybias[nb] + (mb * 2 + h) * 16);
        if constexpr (P::kSplit) rb[mb][nb] *= st.in_scale;   // accumulators carry in_scale x the true value
      }
  }
  f32x16 pend[PIPE ? NB : 1];  // accumulators of M-block mb-1 while M-block mb runs
#pragma unroll
  for (int u0 = 0; u0 < TOT; u0 += UMB) {
    const int nmb = (TOT - u0) < UMB ? (TOT - u0) : UMB;  // M-blocks in this unit (compile-time after unrolling)
    const int nt = nmb * KC;
    uint32_t ub;
    if (NEWUNIT) { ub = open_unit(st); st.ubase = ub; st.uoff = 0; }
    else ub = st.ubase + st.uoff;
    st.uoff += nmb * KC * FB + nmb * 128;
    // split-f16 fragment = a hi plane (64 lanes x 16 B) followed by a lo plane: both reads stay lane-linear 16-byte
    const char* wl = smem + ub + st.lane * (P::kSplit ? 16 : P::kLaneBytes);
    const char* bl = smem + ub + nmb * KC * FB + h * 64;
    F a[PF];
#define DFN_AFRAG(t) load_afrag<P>(wl + (t) * FB)
#pragma unroll
    for (int t = 0; t < PF; ++t)
      if (t < nt) a[t] = DFN_AFRAG(t);
    f32x16 bias = {};
    if (!RAYBIAS && !NOBIAS) bias = load16(reinterpret_cast<const float*>(bl));
#pragma unroll
    for (int lm = 0; lm < UMB; ++lm) {
      if (lm < nmb) {
        const int mb = u0 + lm;
        f32x16 acc[NB];
        f32x16 bias_next = bias;
        if (RAYBIAS) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = RB_ALL ? rb[RB_ALL ? mb : 0][nb] : load16(raybias[nb] + (mb * 2 + h) * 16);
            if constexpr (P::kSplit && !RB_ALL) acc[nb] *= st.in_scale;
          }
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int t = lm * KC + kc;
          if (NEWUNIT && lm == 0 && kc == (KC > 1 ? KC / 2 : 0)) mid_sync(st, smem);
          const F cur = a[t % PF];
          if (t + PF < nt) a[t % PF] = DFN_AFRAG(t + PF);
          if (kc == (PIPE ? KC / 2 : 0) && !RAYBIAS && !NOBIAS && lm + 1 < nmb) bias_next = load16(reinterpret_cast<const float*>(bl + (lm + 1) * 128));
          __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this chunk's MFMAs (hipcc otherwise sinks it to its use)
          if constexpr (P::kSplit && PIPE && NB == 1 && !SCALE_FIRST) {
            // split-f16: the conversion pieces of this chunk go BETWEEN its three dependent MFMAs, a third each (the block form
            // behind them measured 46.5 against 42.7 cycles per MFMA in tools/ubench/x3loop.hip)
            constexpr int PPKI_ = (CIN > 0) ? (8 * NB + CIN - 1) / CIN : 1, PPK_ = (8 * NB + KC - 1) / KC;
            constexpr int NPMAX = PPKI_ > PPK_ ? PPKI_ : PPK_;
            const bool from_carry = CIN > 0 && mb == 0 && kc < CIN, from_pend = mb >= 1;
            const int np = from_carry ? PPKI_ : (from_pend ? PPK_ : 0), base = kc * np;
            X3Piece pc[NPMAX];
            // kc == 0: the accumulators the pieces read (pend / carry) were written by the MFMA issued just before this chunk; an
            // XDL result is not interlocked against VALU reads from inline asm, so the first chunk keeps the block form -- all three
            // parts behind its third MFMA, >= 3 MFMA issues after the producer (read one issue later, the last correction product
            // of some results was missing: 1e-6 instead of 2.4e-7 against exact fp32).
            const bool late = kc == 0;
            f32x16 c0 = (kc == 0) ? (RAYBIAS ? acc[0] : bias) : acc[0];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, Bin[0][kc].hi, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#define DFN_X3_PARTS_A \
            _Pragma("unroll") for (int q = 0; q < NPMAX; ++q) \
              if (q < np && base + q < 8) { \
                if (from_carry) pc[q].template A<CIN_RELU>(carry[0], base + q); \
                else pc[q].template A<RELU>(pend[0], base + q); \
              }
#define DFN_X3_PARTS_B \
            _Pragma("unroll") for (int q = 0; q < NPMAX; ++q) \
              if (q < np && base + q < 8) pc[q].B(pscale, pscale2);
            if (!late) { DFN_X3_PARTS_A }
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, Bin[0][kc].lo, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!late) { DFN_X3_PARTS_B }
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.lo, Bin[0][kc].hi, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (late) { DFN_X3_PARTS_A DFN_X3_PARTS_B }
#undef DFN_X3_PARTS_A
#undef DFN_X3_PARTS_B
#pragma unroll
            for (int q = 0; q < NPMAX; ++q)
              if (q < np && base + q < 8) {
                if (from_carry) pc[q].template C<CIN_RELU>(Bin[0], CIN / 2, base + q, rmax_);
                else pc[q].template C<RELU>(Bout[0], mb - 1, base + q, rmax_);
              }
            acc[0] = c0;
            if (from_carry && kc == CIN - 1) asm volatile("s_nop 3");
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if (kc == 0) acc[nb] = mfma<P>(cur, Bin[nb][0], RAYBIAS ? acc[nb] : bias);
            else acc[nb] = mfma<P>(cur, Bin[nb][kc], acc[nb]);
          }
          if (CIN > 0 && mb == 0 && kc < CIN) {  // the previous layer's last M-block -> chunks CIN, CIN+1 of Bin
#pragma unroll
            for (int q = 0; q < PPKI; ++q) {
              const int piece = kc * PPKI + q;
              if (piece < 8 * NB) store_hidden_piece<P, CIN_RELU>(carry[piece >> 3], Bin[piece >> 3], CIN / 2, piece & 7, pscale, rmax_);
            }
            if (kc == CIN - 1) asm volatile("s_nop 3");  // VALU-written B operand is read by the very next MFMA
          }
          if (PIPE && mb >= 1) {  // previous M-block's conversion, block by block (frees pend[0] first)
#pragma unroll
            for (int q = 0; q < PPK; ++q) {
              const int piece = kc * PPK + q;
              if (piece < 8 * NB) store_hidden_piece<P, RELU>(pend[piece >> 3], Bout[piece >> 3], mb - 1, piece & 7, pscale, rmax_);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bias = bias_next;
        if (mb < MB) {
          if (PIPE) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) pend[nb] = acc[nb];
          } else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) store_hidden<P, RELU>(acc[nb], Bout[nb], mb, st.out_scale * st.lane_mul, rmax_);
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {  // (the M-block before a head is converted during the head's MFMAs)
            head[nb] = acc[nb];
            if constexpr (P::kSplit) head[nb] *= st.out_scale * st.lane_mul;
          }
        }
      }
    }
  }
  if (PIPE && MB >= 1 && !EXTRA) {  // the last M-block has no successor inside this layer
    if (COUT) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) carry[nb] = pend[nb];
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) store_hidden<P, RELU>(pend[nb], Bout[nb], MB - 1, st.out_scale * st.lane_mul, rmax_);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Positional encoding of a point into the layer-1 B operand (slot map: pe_xyz_feature()).
// FAST: x/(2pi) in two-term extended precision, exact fract, then v_sin_f32 / v_cos_f32 (which
// take revolutions), five octaves per coordinate-half from one evaluation by double-angle steps.  Otherwise: split-f16 takes
// every octave from the same exact-fract hardware path (4e-7), exact fp32 the full-range sinf/cosf of the exact product
// x * 2^k, bit-for-bit the reference's sin(x * freq) up to libm rounding.
template <class P, bool FAST, int NB, int PC>
DFN_DEV void posenc_xyz(const float (&x)[NB][3], int h, typename FragOf<P>::type (&pe)[NB][PC]) {
  const float base = h ? 32.f : 1.f;  // half h owns frequencies 2^(5h) .. 2^(5h+4)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xc = x[nb][c];
      if (FAST) {
        float uh, ul, sn[5], cs[5];
        rev_split(xc, uh, ul);
        rev_sincos_octaves5(uh, ul, base, sn, cs);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          set_slot<P>(pe[nb], 6 * k + c, sn[k]);
          set_slot<P>(pe[nb], 6 * k + 3 + c, cs[k]);
        }
      } else if constexpr (P::kSplit) {
        // split-f16: every octave from the exact-fract hardware path (x / 2pi in two-term extended precision, exact v_fract,
        // v_sin / v_cos in revolutions; max abs error 4.1e-7 vs fp64 = fp32 round-off of the encoding).  libm's full-range
        // sinf / cosf expand to ~60 instructions each (large-argument reduction): 60 of them per point were HALF of this
        // kernel's instruction stream (7 000 of 16 900 per tile, next to 1 080 MFMAs).
        float uh, ul;
        rev_split(xc, uh, ul);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          float sn, cs;
          rev_sincos(uh, ul, base * float(1 << k), sn, cs);
          set_slot<P>(pe[nb], 6 * k + c, sn);
          set_slot<P>(pe[nb], 6 * k + 3 + c, cs);
        }
      } else {
        const float xb = xc * base;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const float a = xb * float(1 << k);
          set_slot<P>(pe[nb], 6 * k + c, sinf(a));
          set_slot<P>(pe[nb], 6 * k + 3 + c, cosf(a));
        }
      }
    }
    set_slot<P>(pe[nb], 30, h ? x[nb][2] : x[nb][0]);
    set_slot<P>(pe[nb], 31, h ? 0.f : x[nb][1]);
  }
}

// ------------------------------------------------------------------------------------------
// The 8-layer trunk (xyz_encoding_1..8, skip concat [pe, h] before layer 5).
// `hook(l, h)` (optional) is called once per trunk layer l = 0..6 with that layer's COMPLETE output operand (point block 0) — one
// layer late, because a pipelined layer hands its last M-block to the next layer unconverted; layer 7's output (`out`) completes in
// the caller's next layer.  The saving forward of the gradient path records the ReLU signs there (nerfh_mlp.hip).
struct NoTrunkHook {
  template <class A> DFN_DEV void operator()(int, A&) const {}
};
template <class P, int UMB, bool PIPE, bool FAST, int NB, int W = kWidth, class Hook = NoTrunkHook>
DFN_DEV void trunk(Stager& st, char* smem, const float (&x)[NB][3],
                   typename FragOf<P>::type (&out)[NB][chunks_of<P>(W / 2)], f32x16 (&carry)[NB], Hook hook = Hook()) {
  using F = typename FragOf<P>::type;
  constexpr int PC = chunks_of<P>(32), HC = chunks_of<P>(W / 2), MBW = W / 32;
  constexpr bool CY = PIPE && P::kSlotsPerChunk == 8;  // hand a layer's last M-block to the next layer unconverted
  constexpr int CI = CY ? HC - 2 : -1;                 // ... where it lands in the last two chunks of the W-wide input
  static_assert(!PIPE || W == kWidth, "the pipelined epilogue is tuned for netwidth 128");
  const int h = st.lane >> 5;
  f32x16 nohead[NB];
  const float* const norb[NB] = {};
  F pe[NB][PC];
  posenc_xyz<P, FAST, NB, PC>(x, h, pe);
  F a[NB][HC], b[NB][HC];
  layer<P, UMB, PIPE, NB, PC, MBW, true, false, false, true, -1, true, CY>(st, smem, pe, a, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  hook(0, a[0]);
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, b, a, nohead, norb, carry);
  hook(1, b[0]);
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  hook(2, a[0]);
  {
    F cat[NB][PC + HC];
    if constexpr (P::kSlotsPerChunk == 8 && (!PIPE || P::kSplit)) {  // recompute: cheaper than 32 VGPRs live across 4 layers
      float x2[NB][3];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          x2[nb][c] = x[nb][c];
          asm volatile("" : "+v"(x2[nb][c]));  // opaque copy: stops the compiler from CSE-ing the two encodings
        }
      posenc_xyz<P, FAST, NB, PC>(x2, h, pe);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int i = 0; i < PC; ++i) cat[nb][i] = pe[nb][i];
#pragma unroll
      for (int i = 0; i < HC; ++i) cat[nb][PC + i] = b[nb][i];
    }
    layer<P, l5_unit_mb_p<P>(UMB), PIPE, NB, PC + HC, MBW, true, false, false, true, (CY ? PC + HC - 2 : -1), true, CY>(st, smem, cat, a, nohead, norb, carry);
    hook(3, reinterpret_cast<F(&)[HC]>(cat[0][PC]));   // layer 4's output: the h part of the skip concat, completed inside layer 5
  }
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  hook(4, a[0]);
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, b, a, nohead, norb, carry);
  hook(5, b[0]);
  layer<P, UMB, PIPE, NB, HC, MBW, true, false, false, true, CI, true, CY>(st, smem, a, out, nohead, norb, carry);  // out's last two chunks stay in `carry`
  hook(6, a[0]);
}

}  // namespace dfn
