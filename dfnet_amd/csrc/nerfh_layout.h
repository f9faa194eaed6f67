// nerfh_layout.h — the NeRF-H MLP as a sequence of MFMA "layers", shared by the host-side
// weight packer (nerfh_pack.cpp) and the device kernels (nerfh_mlp.hip).
//
// Formulation: every Linear is computed TRANSPOSED on the matrix cores,
//     H_out^T[feature, point] = W[feature, k] * H_in^T[k, point]
// with the weights as the MFMA A operand (M = 32 output features per M-block) and the
// activations of 32 points as the B operand (N = 32 points).  The 32x32 C/D fragment of lane
// (p = lane&31, h = lane>>5) holds, for point p, output features
//     row(h, r) = (r&3) + 8*(r>>2) + 4*h,  r = 0..15
// of the M-block.  Because the order of the contraction index is free, the C registers of
// one layer ARE the B operand of the next once the next layer's weight columns are
// permuted to match — activations never leave registers between layers.
//
// "Slot" s of half h: the s-th contraction element held by the lanes of half h.  A layer with
// K (padded) inputs has K/2 slots per half.  f16 path: 8 slots per MFMA K-chunk
// (v_mfma_f32_32x32x16_f16), f32 path: 1 slot per chunk (v_mfma_f32_32x32x2_f32).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define DFN_HD __host__ __device__
#else
#define DFN_HD
#endif

namespace dfn {

constexpr int kWidth = 128;      // netwidth the kernels are specialised for
constexpr int kLxyz = 10;        // multires
constexpr int kLdir = 4;         // multires_views
constexpr int kChXyz = 63, kChDir = 27;

enum LayerId {
  LY_L1 = 0, LY_L2, LY_L3, LY_L4, LY_L5, LY_L6, LY_L7, LY_L8,
  LY_FIN,    // xyz_encoding_final (4 M-blocks, no activation) + static_sigma (5th M-block, row 0)
  LY_DIR,    // dir_encoding.0 on the `final` columns; pe_dir/appearance columns fold into a per-ray bias
  LY_RGB,    // static_rgb.0 (rows 0..2)
  LY_TE0,    // transient_encoding.0 on the `final` columns; transient-embedding columns -> per-ray bias
  LY_TE1, LY_TE2, LY_TE3,  // transient_encoding.{2,4,6}
  LY_THEAD,  // rows 0..2 transient_rgb, row 3 transient_sigma, row 8 transient_beta (all on half 0)
  LY_SIG,    // coarse net: static_sigma alone (row 0)
  LY_COUNT
};

struct LayerShape { int slots; int mb; };  // slots per half (= K/2), number of 32-row M-blocks

// W = netwidth: 128 (every kernel variant) or 256 (the plain one-M-block-unit variants, nerfh_mlp.hip).
DFN_HD constexpr LayerShape layer_shape(int id, int W = kWidth) {
  return id == LY_L1 ? LayerShape{32, W / 32}
       : id == LY_L5 ? LayerShape{32 + W / 2, W / 32}
       : id <= LY_L8 ? LayerShape{W / 2, W / 32}
       : id == LY_FIN ? LayerShape{W / 2, W / 32 + 1}
       : id == LY_DIR ? LayerShape{W / 2, W / 64}
       : id == LY_RGB ? LayerShape{W / 4, 1}
       : id == LY_TE0 ? LayerShape{W / 2, W / 64}
       : id <= LY_TE3 ? LayerShape{W / 4, W / 64}
       : id == LY_THEAD ? LayerShape{W / 4, 1}
       : LayerShape{W / 2, 1};  // LY_SIG
}

// The layer sequences the kernels execute, in order.
constexpr int kCoarseSeq[] = {LY_L1, LY_L2, LY_L3, LY_L4, LY_L5, LY_L6, LY_L7, LY_L8, LY_SIG};
constexpr int kFineSeq[] = {LY_L1, LY_L2, LY_L3, LY_L4, LY_L5, LY_L6, LY_L7, LY_L8, LY_FIN,
                            LY_DIR, LY_RGB, LY_TE0, LY_TE1, LY_TE2, LY_TE3, LY_THEAD};
// Staging-unit group of each layer when a unit may hold whole layers (unit_mb >= 8): small layers share a unit.
constexpr int kCoarseGroup[] = {0, 1, 2, 3, 4, 5, 6, 7, 7};
constexpr int kFineGroup[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 10, 10, 10, 10, 10};
constexpr int kCoarseLayers = sizeof(kCoarseSeq) / sizeof(int);
constexpr int kFineLayers = sizeof(kFineSeq) / sizeof(int);

// Feature index (within a 128- or 64-wide hidden vector) held in slot s of half h when the
// vector was produced by M-blocks of a previous layer.
DFN_HD constexpr int hidden_feature(int h, int s) {
  return 32 * (s >> 4) + 4 * h + (s & 3) + 8 * ((s & 15) >> 2);
}
// Row of an M-block held in C register r of half h.
DFN_HD constexpr int mblock_row(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Slot -> column of the 63-wide xyz positional encoding (-1 = zero padding).  Half h computes
// frequencies 5h..5h+4: slots 6k'+c (c<3: sin of coord c, c>=3: cos of coord c-3), then the
// raw coordinates in slots 30, 31 (half 0: x, y; half 1: z, pad).
DFN_HD constexpr int pe_xyz_feature(int h, int s) {
  return s < 30 ? 3 + 6 * (5 * h + s / 6) + (s % 6) : (s == 30 ? (h ? 2 : 0) : (h ? -1 : 1));
}

// Per-precision fragment geometry.
struct PrecF16 { static constexpr int kSlotsPerChunk = 8, kLaneBytes = 16; static constexpr bool kSplit = false; };
struct PrecF32 { static constexpr int kSlotsPerChunk = 1, kLaneBytes = 4; static constexpr bool kSplit = false; };
// Split-f16: every operand is hi + lo in f16 (hi = f16(x), lo = f16(x - hi)) and a product is accumulated in fp32
// as hi*hi + hi*lo + lo*hi — three v_mfma_f32_32x32x16_f16 instead of eight v_mfma_f32_32x32x2_f32 per 16
// contraction elements, fp32-grade results (the dropped lo*lo term is 2^-22 relative).  A fragment = 8 hi + 8 lo
// halves per lane.  Operands are pre-scaled by powers of two (activations x kX3ActScale, weights x 2^s per
// network) so the lo parts stay normal f16 numbers; MlpArgs::{in,out}_scale carry the product and its inverse.
struct PrecX3 { static constexpr int kSlotsPerChunk = 8, kLaneBytes = 32; static constexpr bool kSplit = true; };
constexpr float kX3ActScale = 16.f;

// Kernel variants (nerfh_mlp.hip): workgroup width and staging granularity (M-blocks per unit).
//   variant 0: 8 waves per workgroup, 1 workgroup per CU, a unit = a whole layer (f16) / one M-block (f32)
//   variant 1: 4 waves per workgroup, 2 workgroups per CU, a unit = 2 M-blocks (f16) / one M-block (f32)
//   variant 3: variant 0's geometry without the pipelined epilogue (A/B reference)
//   variant 2: 4 waves per workgroup x 3 point blocks, 1 workgroup per CU (1 wave per SIMD, 512 VGPRs), unit as variant 0
constexpr int kVariants = 4;
template <class P> DFN_HD constexpr int unit_mb(int variant) {
  return P::kSlotsPerChunk == 1 ? 1 : (P::kSplit ? 2 : (variant == 1 ? 2 : 8));
}
DFN_HD constexpr int variant_waves(int variant) { return variant == 0 ? 8 : 4; }
// netwidth 256: M-blocks per staging unit
template <class P> DFN_HD constexpr int unit_mb_w256() { return (P::kSlotsPerChunk == 1 || P::kSplit) ? 1 : 2; }

constexpr uint32_t kPiece = 1024;  // staging granule: one wave-wide 16-byte LDS-DMA

DFN_HD constexpr uint32_t align_piece(uint32_t b) { return (b + kPiece - 1) / kPiece * kPiece; }

// Bytes of one staging unit holding `nmb` M-blocks of a layer with `slots` slots per half.
template <class P>
DFN_HD constexpr uint32_t unit_bytes(int slots, int nmb) {
  return align_piece(uint32_t(nmb) * (slots / P::kSlotsPerChunk) * 64 * P::kLaneBytes + uint32_t(nmb) * 2 * 16 * 4);
}
// Largest unit of either network when at most `umb` M-blocks go into one unit (sizes the two LDS
// staging buffers): the widest layers are L5 (96 slots, 4 M-blocks) and FIN (64 slots, 5 M-blocks).
// In the merged layout (umb >= 8) layer 5 (48 fragments) is split into two units of 2 M-blocks so that THREE staging
// buffers fit the 160 KB of LDS; the largest unit is then the merged transient group (44 fragments + 9 bias blocks).
DFN_HD constexpr int l5_unit_mb(int umb) { return umb >= 8 ? 2 : umb; }
// ... per precision: split-f16 with two-M-block units keeps layer 5 (96 slots: 25 KB per M-block) in one-M-block units so that
// three staging buffers + the input-prefetch slots stay inside the LDS.
template <class P> DFN_HD constexpr int l5_unit_mb_p(int umb) { return (P::kSplit && umb == 2) ? 1 : l5_unit_mb(umb); }
template <class P>
DFN_HD constexpr uint32_t max_unit_bytes(int umb, int W = kWidth) {
  const int m5 = l5_unit_mb_p<P>(umb), mbw = W / 32;
  const uint32_t a = unit_bytes<P>(32 + W / 2, m5 < mbw ? m5 : mbw), b = unit_bytes<P>(W / 2, umb < mbw + 1 ? umb : mbw + 1);
  const uint32_t g = umb >= 8 ? align_piece(44u * 64 * P::kLaneBytes + 9u * 128) : 0;
  return a > b ? (a > g ? a : g) : (b > g ? b : g);
}

// ---- input-gradient kernel (nerfh_bwd.hip) ---------------------------------------------------------------
// The backward pass of a Linear is the same transposed product with W^T as the A operand: the rows of an
// M-block are INPUT features of the forward layer and the contraction slots are its OUTPUT features, in the
// C-fragment order the previous backward layer leaves them in (hidden_feature()).
enum BwdLayerId {
  BW_THEAD = 0,  // [d rgb_t(3), d sigma_t, d beta] -> d t3 (64)
  BW_TE3, BW_TE2, BW_TE1,   // transient_encoding.{6,4,2}^T
  BW_RGB,        // d rgb_s(3) -> d dir_h (64)
  BW_FINCAT,     // [d t0_pre (64) ; d dir_pre (64)] -> d final (128) + 5th M-block: d pe_dir (27, pe_dir_feature())
  BW_FIN,        // [d final (128) ; d sigma_s_pre] -> d h8 (128)
  BW_L8, BW_L7, BW_L6,
  BW_L5,         // d h5_pre -> d h4 (128) + M-blocks 4,5: d pe_xyz in positional-encoding slot order
  BW_L4, BW_L3, BW_L2,
  BW_L1,         // d h1_pre -> d pe_xyz (M-blocks 0,1)
  BW_COUNT
};
DFN_HD constexpr LayerShape bwd_layer_shape(int id) {
  return id == BW_THEAD || id == BW_RGB ? LayerShape{16, 2}
       : id <= BW_TE1 ? LayerShape{32, 2}
       : id == BW_FINCAT ? LayerShape{64, 5}
       : id == BW_FIN ? LayerShape{80, 4}
       : id == BW_L5 ? LayerShape{64, 6}
       : id == BW_L1 ? LayerShape{64, 2}
       : LayerShape{64, 4};
}
// Lane (half h', register r) of a C fragment <-> row i of its M-block, inverted.
DFN_HD constexpr int mblock_half_of_row(int i) { return (i >> 2) & 1; }
DFN_HD constexpr int mblock_reg_of_row(int i) { return (i & 3) + 4 * (i >> 3); }
// Column of the 27-wide direction encoding whose gradient lands in C register r of half h of BW_FINCAT's 5th
// M-block: half h owns frequencies 2h, 2h+1 (r = 6k'+c: sin of coord c, 6k'+3+c: cos), half 0 also the raw
// direction in r = 12..14.  -1 = unused.
DFN_HD constexpr int pe_dir_feature(int h, int r) {
  return r < 12 ? 3 + 6 * (2 * h + r / 6) + (r % 6) : (h == 0 && r < 15 ? r - 12 : -1);
}
// Staging buffer of the backward kernel: its largest unit is BW_L5 (6 M-blocks x 64 slots) / L5 forward.
// M-blocks per staging unit of the gradient kernel: a whole layer (f16); exact fp32: one; split-f16: TWO — the per-M-block
// barrier was 15-21 % of these kernels (ablation without barriers); three 49 KB buffers still fit the 160 KB of LDS.  The
// backward units only pay off together with the bias-free backward layers (layer<..., NOBIAS>: 32 registers fewer).
template <class P> DFN_HD constexpr int bwd_fwd_unit_mb() { return (P::kSlotsPerChunk == 8 && !P::kSplit) ? 8 : (P::kSplit ? 2 : 1); }
template <class P> DFN_HD constexpr int bwd_unit_mb() { return (P::kSlotsPerChunk == 8 && !P::kSplit) ? 8 : (P::kSplit ? 2 : 1); }
template <class P>
DFN_HD constexpr uint32_t bwd_max_unit_bytes() {
  constexpr bool whole = P::kSlotsPerChunk == 8 && !P::kSplit;
  constexpr int uf = bwd_fwd_unit_mb<P>(), ub = bwd_unit_mb<P>();
  const uint32_t a = unit_bytes<P>(96, whole ? 4 : uf);  // forward layer 5 (no merged layout here)
  const uint32_t f5 = unit_bytes<P>(64, whole ? 5 : uf); // forward 128-wide layers / xyz_encoding_final
  const uint32_t b = unit_bytes<P>(64, whole ? 6 : ub), c = unit_bytes<P>(80, whole ? 4 : ub);
  const uint32_t m1 = a > f5 ? a : f5, m2 = b > c ? b : c;
  return m1 > m2 ? m1 : m2;
}

// Per-ray bias table written by the ray-bias kernel and read by the fine kernel:
// [ray][table(0 = dir_encoding, 1 = transient_encoding.0)][mb(2)][h(2)][r(16)] fp32.
constexpr int kRayBiasFloats = 2 * 2 * 2 * 16;   // netwidth 128
DFN_HD constexpr int ray_bias_floats(int W) { return W; }   // 2 tables x (W / 64) M-blocks x 2 halves x 16
constexpr int kMaxWidth = 256;   // widest netwidth of the register-resident kernels

}  // namespace dfn
