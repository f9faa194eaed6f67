// dfnet_wgrad_s.hip — weight (and bias) gradients of DFNet's convolutions as a stream over PRE-SPLIT operands (gfx950 only).
//
// What loss.backward() leaves in the conv parameters of the networks DFNet_dm / run_feature.py train
// (/root/reference/script/feature/direct_feature_matching.py:372-374, script/run_feature.py:166-230; the forward being differentiated
// is feature/dfnet.py:121-136, :57-62):
//     dW[co][ci][ky][kx] = sum over (b, y, x) of g[b, y, x, co] * in[b, y + ky - R, x + kx - R, ci],      db[co] = sum g[b, y, x, co]
// Both operands arrive already split into the two f16 halves the matrix cores multiply (hi = f16(s x), lo = f16(s x - hi)), in the
// row-planar split storage of dfnet_conv.hip (split_piece): the forward convolutions store their activations that way and the
// gate kernel of the backward chain stores the gated gradient that way, ONCE per value.  The old kernel (dfnet_wgrad.hip:
// conv_wgrad_x3_kernel) loaded fp32 and re-split both operands in registers for every (output block, input block) pair and kernel
// row: 21.6 vector instructions per MFMA, 0.11 of the f16 MFMA peak.  Here a product costs no vector arithmetic at all:
//   * a workgroup owns a 64 x 64 channel tile (2 x 2 blocks of 32) and a strip of the image: columns [x0, x0 + 16 NKX), rows
//     [ya, yb) of one frame.  Rows stream HBM/L2 -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, no registers) into two rings:
//     gradient rows and input rows (+ KS - 1 halo columns), [row][block, plane][pixel][K-chunk][lane half][8 slots] f16 — 64 bytes per pixel;
//   * the contraction runs over PIXELS (16 per v_mfma_f32_32x32x16_f16): the [pixel][channel] LDS image is read back as
//     [channel][pixel] MFMA operands by ds_read_b64_tr_b16; a kernel tap (ky, kx) is an address offset (ring row, 64 bytes per pixel);
//   * eight waves = four (output block, input block) pairs x two row phases (even / odd rows of the strip); a wave keeps the KS x TY
//     tap accumulators of its pair and, for the first input-block tile, the bias column (B operand = ones).  The two phases are added
//     through LDS at the end, so a workgroup leaves ONE partial tile per pair: 256 workgroups = 37.7 MB of partials per 3x3 layer,
//     summed in a fixed order by the finalize kernel (deterministic), which also un-permutes into the state_dict layout.
// Bound: MFMA (3 per product) with 9.7 bytes of DMA per clock and CU at full MFMA rate; see DESIGN.md section 6b.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfnet_kernels.h"
#include "mfma_frag.h"

namespace dfn {

namespace {
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define WGS_DEV __device__ __forceinline__

template <int OFF>
WGS_DEV u32x2 ds_tr16(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
WGS_DEV half8 join(u32x2 a, u32x2 b) { return __builtin_bit_cast(half8, u32x4{a[0], a[1], b[0], b[1]}); }
// one operand fragment: 32 channels x 16 pixels, hi and lo planes (PLANE bytes apart), two transposed reads (2 x 4 pixels) each
struct Frag { u32x2 h0, h1, l0, l1; };
template <int OFF, int PLANE>
WGS_DEV void issue_frag(Frag& f, uint32_t addr) {
  f.h0 = ds_tr16<OFF>(addr);
  f.h1 = ds_tr16<OFF + 256>(addr);
  f.l0 = ds_tr16<OFF + PLANE>(addr);
  f.l1 = ds_tr16<OFF + PLANE + 256>(addr);
}
// every LDS read issued before the last N has returned (LDS returns in order); the asm ties the fragment's registers to the wait
template <int N>
WGS_DEV void wait_frag(Frag& f) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.h0), "+v"(f.h1), "+v"(f.l0), "+v"(f.l1) : "n"(N) : "memory");
}
WGS_DEV void lds_dma_b128(const void* gptr, uint32_t lds_off) {   // 64 lanes x 16 bytes -> LDS [lds_off, lds_off + 1024)
  const uint32_t off = __builtin_amdgcn_readfirstlane(lds_off);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(off) : "memory");
}
WGS_DEV void wait_vmcnt_upto(int n) {   // s_waitcnt vmcnt(n), n wave-uniform (gfx9 encoding: vmcnt = bits [3:0] and [15:14])
  switch (n) {
#define WGS_VMC(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
    WGS_VMC(0) WGS_VMC(1) WGS_VMC(2) WGS_VMC(3) WGS_VMC(4) WGS_VMC(5) WGS_VMC(6) WGS_VMC(7) WGS_VMC(8) WGS_VMC(9) WGS_VMC(10)
    WGS_VMC(11) WGS_VMC(12) WGS_VMC(13) WGS_VMC(14) WGS_VMC(15) WGS_VMC(16) WGS_VMC(17) WGS_VMC(18) WGS_VMC(19) WGS_VMC(20)
    WGS_VMC(21) WGS_VMC(22) WGS_VMC(23) WGS_VMC(24) WGS_VMC(25) WGS_VMC(26) WGS_VMC(27) WGS_VMC(28) WGS_VMC(29) WGS_VMC(30)
    WGS_VMC(31) WGS_VMC(32)
#undef WGS_VMC
    default: __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  asm volatile("" ::: "memory");
}
// MFMA operand index i (row of A / column of B: lane & 31) -> stored position inside a 32-channel block (16 h + slot): the transposed
// read hands lane (K-chunk kc = i >> 4, feature f = i & 15) the channel of lane half h = f >> 3, slot 8 kc + (f & 7).
WGS_DEV int pos_of_operand_index(int i) { return 16 * ((i >> 3) & 1) + 8 * (i >> 4) + (i & 7); }
__device__ __host__ inline int chan_of_pos_s(int e) { return 4 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2); }
}  // namespace

struct WgradSArgs {
  const char* g;       // gated gradient, split row-planar [B][H][mblks][2][4][W] x 16 bytes
  const char* in;      // the conv's input, split row-planar [B][H][nblks][2][4][W] x 16 bytes
  const char* zeros;   // >= 16 bytes of zeros (DMA source of padding pixels / rows)
  float* part;         // [chunk][mblk][nblk][KS*KS][32 co position][32 ci position]
  float* part_b;       // [chunk][mblk][32 co position], written by the tiles of input block 0 (may be null: no bias gradient)
  int B, H, W, mblks, nblks;
  int n_xs, n_yc, rch;   // strips per image row, row chunks per image, rows per chunk
  int fb;                // frames per chunk (a workgroup walks fb frames of its strip; > 1 only when the grid would exceed the chip)
  int ky0;               // first kernel row of this launch (TY rows per launch)
  int n_wgs;             // logical workgroups (the 1-D grid is padded to a multiple of 8 for the XCD-aware order)
};

template <int KS, int TY, int NKX, int PD>
struct WgsGeom {
  static constexpr int R = KS / 2, TAPS = TY * KS;
  static constexpr int TC = 16 * NKX, IC = TC + KS - 1;
  static constexpr int IN_IMG = IC * 64, G_IMG = TC * 64;                  // one (block, plane) image of a row
  static constexpr int IN_ROW = (4 * IN_IMG + 1023) & ~1023, G_ROW = 4 * G_IMG;
  static constexpr int NPI = IN_ROW / 1024, NPG = G_ROW / 1024;            // 1 KiB DMA pieces per row
  static constexpr int PPI = (NPI + 7) / 8, PPG = (NPG + 7) / 8;           // ... per wave
  static constexpr int NIN = TY + 1 + 2 * PD, NG = 2 + 2 * PD;             // ring slots (rows)
  static constexpr int LDS = NIN * IN_ROW + NG * G_ROW;
  static constexpr int STAGE_DMAS = 2 * (PPI + PPG), PRE_DMAS = (TY - 1) * PPI;
  static constexpr int RED_TAPS = 3;                                       // taps per round of the final two-phase reduction
  static_assert(LDS <= 160 * 1024, "rings do not fit in LDS");
  static_assert(LDS >= 4 * RED_TAPS * 4096, "reduction buffer");
  static_assert(PD * STAGE_DMAS + PRE_DMAS <= 32, "vmcnt range");
};

template <int KS, int TY, int NKX, int PD>
__global__ __launch_bounds__(512, 2) void conv_wgrad_s_kernel(WgradSArgs a) {
  using G = WgsGeom<KS, TY, NKX, PD>;
  constexpr int R = G::R, TAPS = G::TAPS, TC = G::TC, IC = G::IC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)DFN_LDS_PTR(smem);
  const uint32_t lds_in = lds0, lds_g = lds0 + G::NIN * G::IN_ROW;

  // XCD-aware order: logical workgroup L = chunk * pairs + pair; the pairs of one strip read the same rows and sit on one XCD's L2
  const int n8 = (a.n_wgs + 7) / 8;
  const int L = (blockIdx.x & 7) * n8 + (blockIdx.x >> 3);
  if (L >= a.n_wgs) return;
  const int npn = a.nblks / 2, pairs = (a.mblks / 2) * npn;
  const int chunk = L / pairs, pair = L - chunk * pairs;
  const int mblk0 = (pair / npn) * 2, nblk0 = (pair % npn) * 2;
  const int yc = chunk % a.n_yc, xs = (chunk / a.n_yc) % a.n_xs, bg = chunk / (a.n_yc * a.n_xs);
  const int ya = yc * a.rch, yb = min(ya + a.rch, a.H), x0 = xs * TC;
  const int nsteps = (yb - ya + 1) / 2;
  const int b_lo = bg * a.fb, b_hi = min(b_lo + a.fb, a.B);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, mb_l = (wave >> 1) & 1, nb_l = wave & 1;
  const bool with_bias = a.part_b && nblk0 == 0 && nb_l == 0;   // wave-uniform

  // ---- DMA sources of this lane (fixed for the strip): byte offset inside a (frame, row) of the split tensors, or "zeros"
  uint32_t in_off[G::PPI], g_off[G::PPG];
  unsigned in_ok = 0, g_ok = 0;
#pragma unroll
  for (int i = 0; i < G::PPI; ++i) {
    const int p = min(wave + 8 * i, G::NPI - 1);
    const int slot = p * 64 + lane;
    const int img = slot / (IC * 4), rem = slot - img * (IC * 4);
    const int px = rem >> 2, kc = (rem >> 1) & 1, h = rem & 1, x = x0 - R + px;
    const bool ok = img < 4 && x >= 0 && x < a.W;
    in_off[i] = ok ? (uint32_t)((((nblk0 + (img >> 1)) * 2 + kc) * 4 + (img & 1) * 2 + h) * a.W + x) * 16u : 0u;
    in_ok |= unsigned(ok) << i;
  }
#pragma unroll
  for (int i = 0; i < G::PPG; ++i) {
    const int p = min(wave + 8 * i, G::NPG - 1);
    const int slot = p * 64 + lane;
    const int img = slot / (TC * 4), rem = slot - img * (TC * 4);
    const int px = rem >> 2, kc = (rem >> 1) & 1, h = rem & 1, x = x0 + px;
    const bool ok = x < a.W;
    g_off[i] = ok ? (uint32_t)((((mblk0 + (img >> 1)) * 2 + kc) * 4 + (img & 1) * 2 + h) * a.W + x) * 16u : 0u;
    g_ok |= unsigned(ok) << i;
  }
  const size_t in_row_bytes = (size_t)a.nblks * 8 * a.W * 16, g_row_bytes = (size_t)a.mblks * 8 * a.W * 16;
  const char* in_frame = a.in + (size_t)b_lo * a.H * in_row_bytes;
  const char* g_frame = a.g + (size_t)b_lo * a.H * g_row_bytes;
  // ring slots: input row r -> (r - rin0) mod NIN with rin0 = the first row of step 0's window; gradient row y -> (y - ya) mod NG
  const int rin0 = ya + a.ky0 - R;
  auto dma_in_piece = [&](int r, int i) {                   // piece i (compile-time after unrolling) of input row r
    const bool row_ok = r >= 0 && r < a.H;                  // rows outside the image: zeros (their products vanish)
    const char* base = in_frame + (size_t)(row_ok ? r : 0) * in_row_bytes;
    const uint32_t dst = lds_in + uint32_t((r - rin0) % G::NIN) * G::IN_ROW;
#pragma unroll
    for (int k = 0; k < G::PPI; ++k)
      if (k == i) lds_dma_b128((row_ok && ((in_ok >> k) & 1)) ? base + in_off[k] : a.zeros, dst + min(wave + 8 * k, G::NPI - 1) * 1024);
  };
  auto dma_g_piece = [&](int y, int i) {
    const bool row_ok = y < yb;                             // the odd phase's row past the strip: zeros
    const char* base = g_frame + (size_t)(row_ok ? y : 0) * g_row_bytes;
    const uint32_t dst = lds_g + uint32_t((y - ya) % G::NG) * G::G_ROW;
#pragma unroll
    for (int k = 0; k < G::PPG; ++k)
      if (k == i) lds_dma_b128((row_ok && ((g_ok >> k) & 1)) ? base + g_off[k] : a.zeros, dst + min(wave + 8 * k, G::NPG - 1) * 1024);
  };
  auto dma_in_row = [&](int r) {
#pragma unroll
    for (int i = 0; i < G::PPI; ++i) dma_in_piece(r, i);
  };
  // stage s = what step s needs beyond step s - 1: the last two rows of its input window, its two gradient rows; piece n of
  // STAGE_DMAS: [input row A][input row B][gradient row A][gradient row B]
  auto issue_stage_piece = [&](int s, int n) {
    const int y = ya + 2 * s;
    if (n < G::PPI) dma_in_piece(y + a.ky0 - R + TY - 1, n);
    else if (n < 2 * G::PPI) dma_in_piece(y + a.ky0 - R + TY, n - G::PPI);
    else if (n < 2 * G::PPI + G::PPG) dma_g_piece(y, n - 2 * G::PPI);
    else dma_g_piece(y + 1, n - 2 * G::PPI - G::PPG);
  };
  auto issue_stage = [&](int s) {
#pragma unroll
    for (int n = 0; n < G::STAGE_DMAS; ++n) issue_stage_piece(s, n);
  };
  f32x16 acc[TAPS], acc_b;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_b[r] = 0.f;

  // transposed-read address of this lane inside a (block, plane) image: 16-lane group q = lane >> 4 reads K-chunk q & 1 of pixels
  // 8 (q >> 1) + 0..3 (second read: + 4 pixels = 256 bytes); lane t of the group fetches pixel t >> 2, 8-byte piece t & 3
  const int q4 = lane >> 4, t16 = lane & 15;
  const uint32_t lane_part = uint32_t(8 * (q4 >> 1) + (t16 >> 2)) * 64u + uint32_t(q4 & 1) * 32u + uint32_t(t16 & 3) * 8u;
  const uint32_t a_lane = lds_g + uint32_t(mb_l * 2) * G::G_IMG + lane_part;
  const uint32_t b_lane = lds_in + uint32_t(nb_l * 2) * G::IN_IMG + lane_part;
  const half8 ones = {1, 1, 1, 1, 1, 1, 1, 1};

  for (int b = b_lo; b < b_hi; ++b, in_frame += (size_t)a.H * in_row_bytes, g_frame += (size_t)a.H * g_row_bytes) {
  if (b > b_lo) __builtin_amdgcn_s_barrier();               // every wave has left the previous frame's last rows
#pragma unroll
  for (int t = 0; t < TY - 1; ++t) dma_in_row(rin0 + t);    // the first TY - 1 rows of step 0's window
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < nsteps) issue_stage(s);
  for (int s = 0; s < nsteps; ++s) {
    // stage s landed (this wave's pieces), then everyone's; stages s + 1 .. s + PD - 1 may still be in flight
    int ahead = nsteps - 1 - s;
    ahead = ahead > PD - 1 ? PD - 1 : ahead;
    wait_vmcnt_upto(ahead * G::STAGE_DMAS);
    __builtin_amdgcn_s_barrier();          // ... and every wave has left step s - 1: its rows' ring slots are free
    asm volatile("" ::: "memory");
    const bool more = s + PD < nsteps;     // a later stage is issued during this step
    const int y = ya + 2 * s + grp;        // this phase's gradient row
    if (y < yb) {
      const uint32_t ga = a_lane + uint32_t((y - ya) % G::NG) * G::G_ROW;
      Frag af[NKX];
#pragma unroll
      for (int ks = 0; ks < NKX; ++ks) {
        // (immediate offsets need compile-time values: the k-step offset is added to the base register)
        issue_frag<0, G::G_IMG>(af[ks], ga + uint32_t(ks) * 1024u);
      }
      uint32_t brow[TY];
#pragma unroll
      for (int ty = 0; ty < TY; ++ty) brow[ty] = b_lane + uint32_t((y + a.ky0 - R + ty - rin0) % G::NIN) * G::IN_ROW;
      // units u = (ty, ks, kx), software-pipelined: the reads of units u + 1 and u + 2 are in flight while unit u is multiplied;
      // the DMA pieces of the stage issued during this step go out one at a time between the units' MFMA groups (an LDS-DMA
      // instruction stalls its wave 60-185 cycles); measured equal to a burst behind the barrier, kept for the shorter ramp after it
      constexpr int NU = TY * NKX * KS, DEPTH = NU > 2 ? 2 : NU - 1;
      constexpr int DMA_EVERY = G::STAGE_DMAS >= NU ? 1 : NU / G::STAGE_DMAS;
      auto unit_addr = [&](int u) { return brow[u / (NKX * KS)] + uint32_t(((u / KS) % NKX) * 16 + u % KS) * 64u; };
      Frag bf[3];
#pragma unroll
      for (int u = 0; u < DEPTH && u < NU; ++u) issue_frag<0, G::IN_IMG>(bf[u], unit_addr(u));
      half8 ah[NKX], al[NKX];
      int next_dma = 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int ty = u / (NKX * KS), ks = (u / KS) % NKX, kx = u % KS;
        if (u + DEPTH < NU) issue_frag<0, G::IN_IMG>(bf[(u + DEPTH) % 3], unit_addr(u + DEPTH));
        const int newer = (NU - 1 - u) < DEPTH ? (NU - 1 - u) : DEPTH;   // fragments issued after unit u's
        if (newer == 2) wait_frag<8>(bf[u % 3]);
        else if (newer == 1) wait_frag<4>(bf[u % 3]);
        else wait_frag<0>(bf[u % 3]);
        if (u == 0) {   // the A fragments were issued before every B fragment: they have returned as well
#pragma unroll
          for (int k2 = 0; k2 < NKX; ++k2) {
            asm volatile("" : "+v"(af[k2].h0), "+v"(af[k2].h1), "+v"(af[k2].l0), "+v"(af[k2].l1));
            ah[k2] = join(af[k2].h0, af[k2].h1);
            al[k2] = join(af[k2].l0, af[k2].l1);
          }
        }
        const half8 bh = join(bf[u % 3].h0, bf[u % 3].h1), bl = join(bf[u % 3].l0, bf[u % 3].l1);
        const int t = ty * KS + kx;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc[t], 0, 0, 0);
        if (with_bias && ty == 0 && kx == 0) {
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], ones, acc_b, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], ones, acc_b, 0, 0, 0);
        }
        if (more && u % DMA_EVERY == 0 && u / DMA_EVERY < G::STAGE_DMAS) {
#pragma unroll
          for (int n = 0; n < G::STAGE_DMAS; ++n) if (n == u / DMA_EVERY) issue_stage_piece(s + PD, n);
          next_dma = u / DMA_EVERY + 1;
        }
      }
      if (more) {   // what the unit loop did not reach
#pragma unroll
        for (int n = 0; n < G::STAGE_DMAS; ++n) if (n >= next_dma) issue_stage_piece(s + PD, n);
      }
    } else if (more) {
      issue_stage(s + PD);
    }
  }
  }   // frames

  // ---- the odd phase hands its accumulators to the even one through LDS (the rings are dead), RED_TAPS taps per round
  constexpr int NT = TAPS + 1;           // + the bias tile
  float* red = reinterpret_cast<float*>(smem) + (size_t)(wave & 3) * G::RED_TAPS * 1024;
  const int col = lane & 31, hh = lane >> 5;
  const int e_n = pos_of_operand_index(col);
  float* dst = a.part + ((((size_t)chunk * a.mblks + mblk0 + mb_l) * a.nblks + nblk0 + nb_l) * (KS * KS) + a.ky0 * KS) * 1024;
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += G::RED_TAPS) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) {
#pragma unroll
      for (int t = t0; t < t0 + G::RED_TAPS && t < NT; ++t) {
        const f32x16& v = t < TAPS ? acc[t < TAPS ? t : 0] : acc_b;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((t - t0) * 16 + r) * 64 + lane] = v[r];
      }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int t = t0; t < t0 + G::RED_TAPS && t < NT; ++t) {
        if (t < TAPS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[t < TAPS ? t : 0][r] + red[((t - t0) * 16 + r) * 64 + lane];
            const int e_m = pos_of_operand_index((r & 3) + 8 * (r >> 2) + 4 * hh);
            dst[(t * 32 + e_m) * 32 + e_n] = v;
          }
        } else if (with_bias && col == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc_b[r] + red[((t - t0) * 16 + r) * 64 + lane];
            const int e_m = pos_of_operand_index((r & 3) + 8 * (r >> 2) + 4 * hh);
            a.part_b[((size_t)chunk * a.mblks + mblk0 + mb_l) * 32 + e_m] = v;
          }
        }
      }
    }
  }
}

// Sum the chunks in a fixed order (eight interleaved chains, then a fixed tree: deterministic) and write dW[co][ci][ky][kx] /
// db[co] in the state_dict layout; the operands were pre-scaled by powers of two (g: gscale[0], activations: kConvActScale).
__global__ __launch_bounds__(256) void wgrad_s_finalize_kernel(const float* __restrict__ part, const float* __restrict__ part_b, int mblks,
                                                               int nblks, int T, int n_chunks, int cout, int cin, float* __restrict__ dW,
                                                               float* __restrict__ db, const float* __restrict__ gscale) {
  const size_t n = (size_t)mblks * nblks * T * 1024, nb = part_b && db ? (size_t)mblks * 32 : 0;
  const float inv_g = gscale ? gscale[1] : 1.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n + nb; i += (size_t)gridDim.x * blockDim.x) {
    const bool bias = i >= n;
    const float* src = bias ? part_b + (i - n) : part + i;
    const size_t stride = bias ? nb : n;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int ch = 0;
    for (; ch + 8 <= n_chunks; ch += 8)
#pragma unroll
      for (int k = 0; k < 8; ++k) s8[k] += src[(size_t)(ch + k) * stride];
    for (int k = 0; ch < n_chunks; ++ch, ++k) s8[k] += src[(size_t)ch * stride];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    if (bias) {
      const size_t j = i - n;
      const int co = 32 * int(j >> 5) + chan_of_pos_s(int(j & 31));
      if (co < cout) db[co] = s * inv_g;
    } else {
      const int cj = int(i & 31), ci_ = int((i >> 5) & 31);
      size_t r = i >> 10;
      const int t = int(r % T); r /= T;
      const int nblk = int(r % nblks), mblk = int(r / nblks);
      const int co = 32 * mblk + chan_of_pos_s(ci_), ci = 32 * nblk + chan_of_pos_s(cj);
      if (co < cout && ci < cin) dW[((size_t)co * cin + ci) * T + t] = s * (inv_g * (1.f / kConvActScale));
    }
  }
}

// fp32 blocked [B*H*W][nblk][32] -> the split row-planar storage (dfnet_conv.hip: split_piece), scale = scale_dev[0] or `scale`.
// One thread = 8 consecutive positions of a pixel's block (lane half h, K-chunk kc): a 16-byte hi piece and a 16-byte lo piece.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int B, int H, int W, int nblk, float scale,
                                                         const float* __restrict__ scale_dev, char* __restrict__ out) {
  const float sc = scale_dev ? scale_dev[0] : scale;
  const size_t rows = (size_t)B * H, n = rows * nblk * 4 * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = int(i % W);
    size_t r = i / W;
    const int hk = int(r & 3); r >>= 2;               // hk = kc * 2 + h
    const int blk = int(r % nblk);
    const size_t row = r / nblk;
    const int kc = hk >> 1, h = hk & 1;
    const float* src = x + ((row * W + xx) * nblk + blk) * 32 + 16 * h + 8 * kc;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xs = (k < 4 ? v0[k] : v1[k - 4]) * sc;
      hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);
      lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);
    }
    char* d = out + ((((row * nblk + blk) * 2 + kc) * 4 + h) * W + xx) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + 2 * (size_t)W * 16) = lo;
  }
}

hipError_t launch_split_rows(const float* x, int B, int H, int W, int nblk, float scale, const float* scale_dev, void* out, hipStream_t s) {
  const size_t n = (size_t)B * H * nblk * 4 * W;
  const int blocks = int((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(split_rows_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, x, B, H, W, nblk, scale, scale_dev, static_cast<char*>(out));
  return hipGetLastError();
}

// fp32 NCHW [B][C][H][W] (torch layout) -> the split row-planar storage; C a multiple of 32.  One thread = one 16-byte piece pair.
__global__ __launch_bounds__(256) void nchw_to_split_kernel(const float* __restrict__ x, int B, int C, int H, int W, float scale,
                                                            const float* __restrict__ scale_dev, char* __restrict__ out) {
  const float sc = scale_dev ? scale_dev[0] : scale;
  const int nblk = C / 32;
  const size_t rows = (size_t)B * H, n = rows * nblk * 4 * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = int(i % W);
    size_t r = i / W;
    const int hk = int(r & 3); r >>= 2;
    const int blk = int(r % nblk);
    const size_t row = r / nblk;
    const int kc = hk >> 1, h = hk & 1, y = int(row % H);
    const size_t b = row / H;
    half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = 32 * blk + chan_of_pos_s(16 * h + 8 * kc + k);
      const float xs = x[((b * C + c) * H + y) * W + xx] * sc;
      hi[k] = (_Float16)fminf(fmaxf(xs, -65000.f), 65000.f);
      lo[k] = (_Float16)fminf(fmaxf(xs - (float)hi[k], -65000.f), 65000.f);
    }
    char* d = out + ((((row * nblk + blk) * 2 + kc) * 4 + h) * W + xx) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + 2 * (size_t)W * 16) = lo;
  }
}
hipError_t launch_nchw_to_split(const float* x, int B, int C, int H, int W, float scale, const float* scale_dev, void* out, hipStream_t s) {
  if (C % 32) return hipErrorInvalidValue;
  const size_t n = (size_t)B * H * (C / 32) * 4 * W;
  const int blocks = int((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(nchw_to_split_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, x, B, C, H, W, scale, scale_dev, static_cast<char*>(out));
  return hipGetLastError();
}

namespace {
template <int KS, int TY, int NKX, int PD>
hipError_t launch_wgs_t(WgradSArgs a, int n_launch_rows, hipStream_t s, int ky0_first = 0) {
  using G = WgsGeom<KS, TY, NKX, PD>;
  auto kern = conv_wgrad_s_kernel<KS, TY, NKX, PD>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  for (int ky = 0; ky < n_launch_rows; ++ky) {
    a.ky0 = ky0_first + ky * TY;
    if (a.ky0 > 0) a.part_b = nullptr;   // the bias column rides with the first kernel row only
    hipLaunchKernelGGL(kern, dim3((a.n_wgs + 7) / 8 * 8), dim3(512), G::LDS, s, a);
  }
  return hipGetLastError();
}
}  // namespace

namespace {
struct WgsPlan { int nkx, n_xs, n_yc, rch, fb, chunks, pairs; };
// Strip / chunk geometry of one layer: depends on the shape only (the partial sums, hence the results, are reproducible).
WgsPlan plan_wgs(int ks, int B, int H, int W, int cout, int cin) {
  WgsPlan p{};
  p.pairs = (cout / 64) * (cin / 64);
  // strip width: 32 columns, or 48 where that pads the row less (W = 40: one 48-column strip instead of two of 32)
  auto padded = [&](int tc) { return (W + tc - 1) / tc * tc; };
  p.nkx = (ks != 5 && padded(48) < padded(32)) ? 3 : 2;
  const int tc = 16 * p.nkx;
  p.n_xs = (W + tc - 1) / tc;
  // one workgroup per CU (86-126 KB of LDS, eight waves): frames per workgroup and row chunks so that the grid fills the 256 CUs
  // without a second, ragged round
  p.fb = 1;
  while (p.pairs * ((B + p.fb - 1) / p.fb) * p.n_xs > 256 && p.fb < B) ++p.fb;
  const int n_bg = (B + p.fb - 1) / p.fb;
  const int base = p.pairs * n_bg * p.n_xs;
  int n_yc = base >= 256 ? 1 : 256 / base;
  if (n_yc > (H + 3) / 4) n_yc = (H + 3) / 4;                      // at least ~4 rows per chunk
  if (n_yc < 1) n_yc = 1;
  p.rch = (H + n_yc - 1) / n_yc;
  p.rch += p.rch & 1;                                               // whole row pairs: the odd phase idles only at the image end
  p.n_yc = (H + p.rch - 1) / p.rch;
  p.chunks = n_bg * p.n_xs * p.n_yc;
  return p;
}
}  // namespace

// scratch the split weight gradient of one layer needs: floats of chunk partials (weights) and of bias partials
void conv_wgrad_split_scratch(int ks, int B, int H, int W, int cout, int cin, size_t* part_floats, size_t* part_b_floats) {
  const WgsPlan p = plan_wgs(ks, B, H, W, cout, cin);
  *part_floats = (size_t)p.chunks * (cout / 32) * (cin / 32) * ks * ks * 1024;
  *part_b_floats = (size_t)p.chunks * cout;
}

// dW [cout][cin][ks][ks] (and db [cout] when db != null) from the split tensors g (gradient w.r.t. the conv's pre-activation, scaled by
// gscale[0]) and in (the conv's input, scaled by kConvActScale).  part / part_b: scratch (conv_wgrad_split_scratch).
hipError_t launch_conv_wgrad_split(int ks, const void* g, const void* in, const void* zeros, int B, int H, int W, int cout, int cin,
                                   float* part, size_t part_floats, float* part_b, size_t part_b_floats, float* dW, float* db,
                                   const float* gscale, hipStream_t s) {
  const int mblks = cout / 32, nblks = cin / 32, T = ks * ks;
  if (cout % 64 || cin % 64 || (ks != 1 && ks != 3 && ks != 5) || !zeros) return hipErrorInvalidValue;
  const WgsPlan p = plan_wgs(ks, B, H, W, cout, cin);
  if ((size_t)p.chunks * mblks * nblks * T * 1024 > part_floats) return hipErrorInvalidValue;
  if (db && (size_t)p.chunks * mblks * 32 > part_b_floats) return hipErrorInvalidValue;
  if ((size_t)(mblks > nblks ? mblks : nblks) * 8 * W * 16 >= (1ull << 31)) return hipErrorInvalidValue;   // 32-bit in-row offsets
  WgradSArgs a{};
  a.g = static_cast<const char*>(g); a.in = static_cast<const char*>(in); a.zeros = static_cast<const char*>(zeros);
  a.part = part; a.part_b = db ? part_b : nullptr;
  a.B = B; a.H = H; a.W = W; a.mblks = mblks; a.nblks = nblks;
  a.n_xs = p.n_xs; a.n_yc = p.n_yc; a.rch = p.rch; a.fb = p.fb;
  a.n_wgs = p.chunks * p.pairs;
  hipError_t e;
  if (ks == 3) e = p.nkx == 3 ? launch_wgs_t<3, 3, 3, 1>(a, 1, s) : launch_wgs_t<3, 3, 2, 2>(a, 1, s);
  else if (ks == 1) e = p.nkx == 3 ? launch_wgs_t<1, 1, 3, 2>(a, 1, s) : launch_wgs_t<1, 1, 2, 2>(a, 1, s);
  else {   // 5x5: kernel rows (0, 1), (2, 3), (4) — ten tap accumulators per wave, the gradient rows streamed three times instead of five
    e = launch_wgs_t<5, 2, 2, 2>(a, 2, s);
    if (e == hipSuccess) e = launch_wgs_t<5, 1, 2, 2>(a, 1, s, 4);
  }
  if (e != hipSuccess) return e;
  const size_t n = (size_t)mblks * nblks * T * 1024 + (db ? (size_t)mblks * 32 : 0);
  hipLaunchKernelGGL(wgrad_s_finalize_kernel, dim3(int((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, part, a.part_b, mblks,
                     nblks, T, p.chunks, cout, cin, dW, db, gscale);
  return hipGetLastError();
}

}  // namespace dfn
