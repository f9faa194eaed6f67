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
  if constexpr (P::kSlotsPerChunk == 8) arr[s >> 3][s & 7] = (_Float16)v;
  else arr[s] = v;
}

// ------------------------------------------------------------------------------------------
// Weight staging: the packed blob is a sequence of units (offset, bytes) in execution order.
struct Stager {
  const char* blob;
  const uint32_t* tab;
  int n_units;
  int u;               // unit that the NEXT begin_unit() makes readable
  uint32_t lds_cur;    // LDS byte offset holding (or receiving) unit u
  uint32_t lds_nxt;
  int lane, wave, waves;
  uint32_t ubase, uoff;      // LDS offset of the open unit / bytes of it consumed by the layers so far
  uint32_t pf_off, pf_size;  // table entry of the unit the NEXT begin_unit() will start streaming (prefetched)
  unsigned long long t_sync, t_wait, t_last;  // DFN_TIMING: cycles in unit waits / barrier
  unsigned long long* trace;
  int n_trace;
  int younger_loads;   // global loads issued AFTER this unit's DMA that may stay in flight across the next begin_unit()
  int skew;            // s_sleep units (64 clk) the second wave of each SIMD waits after every unit barrier
  bool more;           // another tile follows this one (wave-uniform)
};

DFN_DEV void stage_issue_at(const Stager& st, char* smem, uint32_t off, uint32_t size, uint32_t lds_off) {
  const char* src = st.blob + off + st.lane * 16;
  for (uint32_t p = st.wave * kPiece; p < size; p += st.waves * kPiece)
    __builtin_amdgcn_global_load_lds((const void*)(src + p), DFN_LDS_PTR(smem + lds_off + p), 16, 0, 0);
}
DFN_DEV void stage_issue(const Stager& st, char* smem, int unit, uint32_t lds_off) {
  stage_issue_at(st, smem, st.tab[2 * unit], st.tab[2 * unit + 1], lds_off);
}

// Make unit st.u readable and start streaming the following one into the other buffer.
// Returns the LDS byte offset of the readable unit.
DFN_DEV uint32_t begin_unit(Stager& st, char* smem) {
#ifdef DFN_ABL_NOSYNC  // ablation: no DMA, no barrier
  return st.lds_cur;
#endif
#ifdef DFN_TIMING
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
  // this wave's share of unit u has landed (vector memory returns in order: loads issued after the DMA,
  // i.e. the next tile's input prefetch, may remain outstanding)
  if (st.younger_loads == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (st.younger_loads == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st.younger_loads = 0;
#ifdef DFN_TIMING
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();                                  // everyone's share landed; everyone left unit u-1
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
  int nxt = st.u + 1;
  const bool wrap = nxt == st.n_units;
  if (wrap) nxt = 0;
  if (!wrap || st.more) stage_issue_at(st, smem, st.pf_off, st.pf_size, st.lds_nxt);
  {  // fetch the table entry needed by the next call now, so its scalar-load latency is off the critical path
    int nn = nxt + 1;
    if (nn == st.n_units) nn = 0;
    st.pf_off = st.tab[2 * nn];
    st.pf_size = st.tab[2 * nn + 1];
  }
  // De-phase the two waves that share a SIMD (waves w and w+4 of an 8-wave workgroup): after the
  // barrier they would otherwise run MFMA phases and epilogue (VALU) phases in lockstep and never
  // overlap one's VALU with the other's MFMAs.
  if (st.skew > 0 && st.wave >= 4) {
    for (int i = 0; i < st.skew; ++i) __builtin_amdgcn_s_sleep(1);
  }
  const uint32_t cur = st.lds_cur;
  st.lds_cur = st.lds_nxt;
  st.lds_nxt = cur;
  st.u = nxt;
  return cur;
}

// ------------------------------------------------------------------------------------------
// One 32-row M-block: acc[nb] += W_mb * Bin[nb] over KC chunks.  `wb` = LDS byte offset of the
// block's first A fragment, already including lane * kLaneBytes.
template <class P, int NB, int KC>
DFN_DEV void mblock_mma(const char* smem, uint32_t wb, const typename FragOf<P>::type (&Bin)[NB][KC],
                        f32x16 (&acc)[NB]) {
  using F = typename FragOf<P>::type;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const F a = *reinterpret_cast<const F*>(smem + wb + kc * 64 * P::kLaneBytes);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma<P>(a, Bin[nb][kc], acc[nb]);
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
DFN_DEV void store_hidden(const f32x16& acc, typename FragOf<P>::type (&out)[OC], int mb) {
#ifdef DFN_ABL_NOEPI  // ablation: no conversion/ReLU work (results are garbage, timing only)
  asm volatile("" ::"v"(acc));
  return;
#endif
  if constexpr (P::kSlotsPerChunk == 8) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)acc[8 * c + j];
      if (RELU) {
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        v = __builtin_elementwise_max(v, zero);
      }
      out[2 * mb + c] = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * mb + r] = RELU ? fmaxf(acc[r], 0.f) : acc[r];
  }
}

// One eighth of store_hidden: output register pair i (0..7) of an M-block's C fragment.  The two empty
// asm statements anchor the conversion at this point of the instruction stream: without them LLVM treats the
// pure arithmetic as freely movable and sinks it out from between the MFMAs it is meant to hide behind.
template <class P, bool RELU, int OC>
DFN_DEV void store_hidden_piece(const f32x16& acc, typename FragOf<P>::type (&out)[OC], int mb, int i) {
  if constexpr (P::kSlotsPerChunk == 8) {
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
template <class P, int UMB, bool PIPE, int NB, int KC, int MB, bool RELU, bool EXTRA, bool RAYBIAS, bool NEWUNIT,
          int CIN, bool CIN_RELU, bool COUT>
DFN_DEV void layer(Stager& st, char* smem, typename FragOf<P>::type (&Bin)[NB][KC],
                   typename FragOf<P>::type (&Bout)[NB][(MB ? MB : 1) * chunks_of<P>(16)],
                   f32x16 (&head)[NB], const float* const (&raybias)[NB], f32x16 (&carry)[NB]) {
  using F = typename FragOf<P>::type;
  constexpr int TOT = MB + (EXTRA ? 1 : 0);
  constexpr int PF = P::kSlotsPerChunk == 8 ? DFN_PF : 4;  // fragments in flight
  constexpr uint32_t FB = 64 * P::kLaneBytes;         // bytes of one A fragment
  constexpr int PPK = (8 * NB + KC - 1) / KC;         // conversion pieces (8 per point block) interleaved per chunk
  constexpr int PPKI = CIN > 0 ? (8 * NB + CIN - 1) / CIN : 0;  // same for the carried-in M-block
  static_assert(CIN < 0 || (PIPE && P::kSlotsPerChunk == 8 && CIN > 0 && CIN + 1 < KC + 1), "carry-in needs the pipelined f16 path");
  static_assert(!COUT || (PIPE && MB >= 1 && !EXTRA), "carry-out needs a regular last M-block");
  static_assert(NEWUNIT || UMB >= TOT, "a layer that continues a unit must fit in it");
  const int h = st.lane >> 5;
  constexpr bool RB_ALL = RAYBIAS && NB <= 2;  // fetch all per-ray seeds at entry (latency behind the barrier)
  f32x16 rb[RB_ALL ? TOT : 1][NB];
  if (RB_ALL) {
#pragma unroll
    for (int mb = 0; mb < TOT; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) rb[mb][nb] = load16(raybias[nb] + (mb * 2 + h) * 16);
  }
  f32x16 pend[PIPE ? NB : 1];  // accumulators of M-block mb-1 while M-block mb runs
#pragma unroll
  for (int u0 = 0; u0 < TOT; u0 += UMB) {
    const int nmb = (TOT - u0) < UMB ? (TOT - u0) : UMB;  // M-blocks in this unit (compile-time after unrolling)
    const int nt = nmb * KC;
    uint32_t ub;
    if (NEWUNIT) { ub = begin_unit(st, smem); st.ubase = ub; st.uoff = 0; }
    else ub = st.ubase + st.uoff;
    st.uoff += nmb * KC * FB + nmb * 128;
    const char* wl = smem + ub + st.lane * P::kLaneBytes;
    const char* bl = smem + ub + nmb * KC * FB + h * 64;
    F a[PF];
#ifdef DFN_ABL_NOLDS
#define DFN_AFRAG(t) a0_abl
    const F a0_abl = *reinterpret_cast<const F*>(wl);
#else
#define DFN_AFRAG(t) (*reinterpret_cast<const F*>(wl + (t) * FB))
#endif
#pragma unroll
    for (int t = 0; t < PF; ++t)
      if (t < nt) a[t] = DFN_AFRAG(t);
    f32x16 bias = {};
    if (!RAYBIAS) bias = load16(reinterpret_cast<const float*>(bl));
#pragma unroll
    for (int lm = 0; lm < UMB; ++lm) {
      if (lm < nmb) {
        const int mb = u0 + lm;
        f32x16 acc[NB];
        f32x16 bias_next = bias;
        if (RAYBIAS) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = RB_ALL ? rb[RB_ALL ? mb : 0][nb] : load16(raybias[nb] + (mb * 2 + h) * 16);
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int t = lm * KC + kc;
          const F cur = a[t % PF];
          if (t + PF < nt) a[t % PF] = DFN_AFRAG(t + PF);
          if (kc == (PIPE ? KC / 2 : 0) && !RAYBIAS && lm + 1 < nmb) bias_next = load16(reinterpret_cast<const float*>(bl + (lm + 1) * 128));
          __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this chunk's MFMAs (hipcc otherwise sinks it to its use)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if (kc == 0) acc[nb] = mfma<P>(cur, Bin[nb][0], RAYBIAS ? acc[nb] : bias);
            else acc[nb] = mfma<P>(cur, Bin[nb][kc], acc[nb]);
          }
          if (CIN > 0 && mb == 0 && kc < CIN) {  // the previous layer's last M-block -> chunks CIN, CIN+1 of Bin
#pragma unroll
            for (int q = 0; q < PPKI; ++q) {
              const int piece = kc * PPKI + q;
              if (piece < 8 * NB) store_hidden_piece<P, CIN_RELU>(carry[piece >> 3], Bin[piece >> 3], CIN / 2, piece & 7);
            }
            if (kc == CIN - 1) asm volatile("s_nop 3");  // VALU-written B operand is read by the very next MFMA
          }
          if (PIPE && mb >= 1) {  // previous M-block's conversion, block by block (frees pend[0] first)
#pragma unroll
            for (int q = 0; q < PPK; ++q) {
              const int piece = kc * PPK + q;
              if (piece < 8 * NB) store_hidden_piece<P, RELU>(pend[piece >> 3], Bout[piece >> 3], mb - 1, piece & 7);
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
            for (int nb = 0; nb < NB; ++nb) store_hidden<P, RELU>(acc[nb], Bout[nb], mb);
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) head[nb] = acc[nb];  // (the M-block before a head is converted during the head's MFMAs)
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
      for (int nb = 0; nb < NB; ++nb) store_hidden<P, RELU>(pend[nb], Bout[nb], MB - 1);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Positional encoding of a point into the layer-1 B operand (slot map: pe_xyz_feature()).
// FAST: x/(2pi) in two-term extended precision, exact fract, then v_sin_f32 / v_cos_f32 (which
// take revolutions).  Otherwise full-range sinf/cosf of the exact product x * 2^k, bit-for-bit
// the reference's sin(x * freq) up to libm rounding.
template <class P, bool FAST, int NB, int PC>
DFN_DEV void posenc_xyz(const float (&x)[NB][3], int h, typename FragOf<P>::type (&pe)[NB][PC]) {
#ifdef DFN_ABL_NOPE  // ablation: no trig (timing only)
  for (int nb = 0; nb < NB; ++nb)
    for (int s = 0; s < 32; ++s) set_slot<P>(pe[nb], s, x[nb][s % 3]);
  return;
#endif
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
template <class P, int UMB, bool PIPE, bool FAST, int NB>
DFN_DEV void trunk(Stager& st, char* smem, const float (&x)[NB][3],
                   typename FragOf<P>::type (&out)[NB][chunks_of<P>(64)], f32x16 (&carry)[NB]) {
  using F = typename FragOf<P>::type;
  constexpr int PC = chunks_of<P>(32), HC = chunks_of<P>(64);
  constexpr bool CY = PIPE && P::kSlotsPerChunk == 8;  // hand a layer's last M-block to the next layer unconverted
  constexpr int CI = CY ? 6 : -1;                      // ... where it lands in chunks 6, 7 of the 128-wide input
  const int h = st.lane >> 5;
  f32x16 nohead[NB];
  const float* const norb[NB] = {};
  F pe[NB][PC];
  posenc_xyz<P, FAST, NB, PC>(x, h, pe);
  F a[NB][HC], b[NB][HC];
  layer<P, UMB, PIPE, NB, PC, 4, true, false, false, true, -1, true, CY>(st, smem, pe, a, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, b, a, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  {
    F cat[NB][PC + HC];
    if constexpr (P::kSlotsPerChunk == 8 && !PIPE) {  // recompute: cheaper than 32 VGPRs live across 4 layers
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
    layer<P, UMB, PIPE, NB, PC + HC, 4, true, false, false, true, (CY ? PC + 6 : -1), true, CY>(st, smem, cat, a, nohead, norb, carry);
  }
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, a, b, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, b, a, nohead, norb, carry);
  layer<P, UMB, PIPE, NB, HC, 4, true, false, false, true, CI, true, CY>(st, smem, a, out, nohead, norb, carry);  // out's chunks 6, 7 stay in `carry`
}

}  // namespace dfn
