// dfnet_kernels.h — internal launch interface of the DFNet feature-extractor kernels.
//
// Activation layout in HBM ("blocked-permuted NHWC"): [B][H][W][C/32][2][16] elements of T (f16 or
// f32).  Within a 32-channel block, element (h, s) holds true channel 4h + (s&3) + 8(s>>2): exactly
// the order in which a 32x32 MFMA C fragment leaves lane half h, so a lane stores its 16 results
// as ONE contiguous 32/64-byte run and the next layer reads its B operand as contiguous 16-byte
// (f16: 8 slots) or 4-byte (f32: 1 slot) pieces.  The permutation is absorbed into the packed
// weights on the host (same trick as the NeRF-H MLP, nerfh_layout.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dfn {

// Split-f16 activation scale of the DFNet convolutions: x4 represents |x| up to 32 500 (hi saturates at 65 000 and
// the lo half carries the rest; VGG16 activations of trained checkpoints reach the thousands) while the lo halves of
// O(0.03+) activations stay normal f16.
constexpr float kConvActScale = 4.f;
constexpr int kConvTileH = 8, kConvTileW = 32;  // output pixels per workgroup: 4 waves x 2 rows x 32 columns

struct ConvArgs {
  const void* in;      // [B,H,W,nblk_in,2,SB]
  const char* w;       // packed A fragments: [cout_group][blk][ky][mb][kx][kc][64 lanes][slots/chunk]
  const float* bias;   // [Cout/32][2][16] in C-fragment order
  void* out_act;       // [B,H,W,Cout/32,2,16] after the optional ReLU (may be null)
  void* out_pre;       // same shape, BEFORE the ReLU (hypercolumn tap; may be null)
  int B, H, W;
  int nblk_in;         // input channel blocks
  int cout_blocks;     // Cout/32
  int relu;
  float out_scale;     // prec 2 (split-f16): accumulators are multiplied by this before the epilogue (2^-(s+4))
  int xcd_groups;      // split-f16 kernel: > 0 = 1-D XCD-aware grid (see conv_x3_kernel); = number of logical workgroups
  // Optional fused identity "upsample" (the adaptation conv of a pyramid level whose size already is the requested feature size):
  // the epilogue also/instead writes fp32 NCHW planes, image b at out_nchw + (b / nchw_split) * nchw_group_stride +
  // (b % nchw_split) * Cout * H * W (nchw_split = B, or B/2 for the siamese two-stack output).  Values are what
  // launch_upsample would have produced from out_act.
  // Optional fused 2x2/2 max pool (split-f16 kernel): [B,H/2,W/2,Cout/32,2,16] = maxpool(relu?(out)); the tile origin is even, so
  // every window lies inside one wave's two rows.
  void* out_pool;
  // Split-f16 storage (dfnet_conv.hip, split_piece): in_split != 0: `in` holds hi | lo f16 blocks (conv_x3s_kernel, LDS-DMA staged);
  // out_split bit 0 / 1 / 2: out_act / out_pre / out_pool are written in that storage.  zeros: >= 16 device bytes of zero (the padding
  // pixels' DMA source), required with in_split.
  int in_split, out_split;
  const void* zeros;
  // Optional fused 1x1 convolution + ReLU on this conv's PRE-ReLU output (the level-0 adaptation layer's first conv on the conv1_2
  // tap; needs Cout = 64 = one workgroup's channels, split-f16): fuse_w / fuse_bias / fuse_scale = the 1x1's packed fragments,
  // pre-scaled bias and out_scale; fuse_out [B,H,W,64] in the split storage.  Same operands, same order, same bits as the two-kernel form.
  const char* fuse_w;
  const float* fuse_bias;
  float fuse_scale;
  void* fuse_out;
  float* out_nchw;
  int nchw_split;
  size_t nchw_group_stride;
  const float* dyn_scale;  // prec 2, optional: device [scale, 1/scale] of the INPUT tensor (launch_absmax_scale) replacing
                           // the fixed x16 activation scale — gradient tensors have no a-priori magnitude
  // prec 2, optional: device word that receives max |output| over the image's pixels (unsigned atomicMax on the bit pattern of
  // the non-negative float; the caller zeroes it).  A data-gradient conv leaves the bound its consumer's split needs (gate_split).
  unsigned* absmax_out;
  // conv_x3s_kernel, 3x3: split of the contraction over input half-blocks into `ksplit` slices, one workgroup each, when the grid
  // would leave most of the chip idle (the 30 x 40 and 15 x 20 maps of a 240 x 320 batch: 64-192 workgroups for 512 slots).  Every
  // slice leaves its accumulators as an fp32 slab in ks_slab; the slice that arrives last (ks_count, one word per tile group) sums
  // the slabs in slice order — bit-identical reruns — and runs the epilogue.  Filled in by launch_conv (per-stream workspace).
  int ksplit;
  float* ks_slab;
  unsigned* ks_count;
};

// prec: 0 = f16 MFMA inputs / fp32 accumulate, 1 = exact fp32 MFMA, 2 = split-f16 (hi/lo operands, three f16 MFMAs
// per product, fp32 activations in HBM: fp32-grade results at f16 MFMA rate).  ks in {1,3,5}; sb = slots per
// half per input block (16, or 8/4 for the zero-padded RGB input of conv1_1).
hipError_t launch_conv(int prec, int ks, int sb, const ConvArgs& a, hipStream_t stream);
int conv_mb(int prec, int cout_blocks);  // M-blocks (of 32 output channels) one workgroup computes; fixes the packed layout

// x [B,3,H,W] fp32 in [0,1] -> (x-mean)/std in the conv1_1 input layout [B,H,W,1,2,SB0].
hipError_t launch_dfnet_prep(int prec, const float* x, int B, int H, int W, void* out, hipStream_t stream);
int prep_sb(int prec);
// 2x2/2 max pooling on the blocked layout: [B,H,W,nblk,32] -> [B,H/2,W/2,nblk,32].
hipError_t launch_maxpool(int prec, const void* in, int B, int H, int W, int nblk, void* out, hipStream_t stream);
// bilinear, align_corners=True, blocked [B,h,w,4,32] T -> fp32 NCHW planes out[b*out_bstride + c*UH*UW + Y*UW + X].
// affine (optional, fp32 paths): device sc[128] | sh[128] in stored-position order, out = resize(in) * sc + sh — the
// BatchNorm of a training-mode adaptation layer (dfnet_bn.hip); bilinear weights sum to one, so the order is free.
hipError_t launch_upsample(int prec, const void* in, int B, int h, int w, int UH, int UW, float* out,
                           size_t out_bstride, hipStream_t stream, const float* affine = nullptr);

// Triplet loss of DFNet's training on two feature stacks [L,B,rows,W] (dfnet_loss.hip); ls = level stride in floats.
constexpr size_t kTripletPartDoubles = 2048 * 8;   // [blocks of triplet_rows_kernel][4 mining sums + 4 hinge sums]
hipError_t launch_triplet_forward(const float* f1, size_t ls1, const float* f2, size_t ls2, int L, int B, int rows, int W, float margin,
                                  float eps, int mode, double* part, int* case_out, float* mse_out, float* margin_out, float* row_stat,
                                  float* loss, hipStream_t s);
hipError_t launch_triplet_backward(const float* f1, size_t ls1, const float* f2, size_t ls2, int L, int B, int rows, int W, float eps,
                                   const int* case_in, const float* row_stat, const float* margin_in, const float* grad_loss, float* g1,
                                   size_t gs1, float* g2, size_t gs2, hipStream_t s);

// The caller-owned state buffer of a triplet loss (dfn_triplet_loss_state_bytes): [case:int][4 x mse][margin][pad] | row statistics
// (6 floats per row) | fp64 partials.
struct TripletState { int* case_dev; float* mse; float* margin; float* row_stat; double* part; };
inline size_t triplet_state_bytes(int L, int B, int rows) {
  return 256 + (((size_t)L * B * rows * 6 * sizeof(float) + 255) & ~size_t(255)) + kTripletPartDoubles * sizeof(double);
}
inline TripletState carve_triplet(void* state, int L, int B, int rows) {
  char* base = static_cast<char*>(state);
  TripletState t;
  t.case_dev = reinterpret_cast<int*>(base);
  t.mse = reinterpret_cast<float*>(base + 16);
  t.margin = reinterpret_cast<float*>(base + 32);
  t.row_stat = reinterpret_cast<float*>(base + 256);
  t.part = reinterpret_cast<double*>(base + 256 + (((size_t)L * B * rows * 6 * sizeof(float) + 255) & ~size_t(255)));
  return t;
}
// mode 0 naive / 1 two-case / 2 four-case mining: picks the case from the partial sums, writes case, mse, margin and the loss.
hipError_t launch_triplet_finalize(const double* part, int n_blocks, int mode, double count, double n_rows, float margin, int* case_out,
                                   float* mse_out, float* margin_out, float* loss, hipStream_t s);
// The same loss from the LOW-RESOLUTION pyramid (dfnet_triplet_pyr.hip): z [2 hb][h][w][128] blocked fp32 (the plain 5x5 output of one
// level), bn = the level's BatchNorm work block; f1_half = which half of the batch is the anchor stack.  Forward: row statistics of
// the level (rows ((level * hb + b) * UH + Y) * 128 + channel) and its fp64 partials at part[(part0 + block) * 8 ...]; backward:
// gout [2 hb][h][w][128] = d L / d (BatchNorm output) at low resolution.
int triplet_pyr_blocks(int hb, int UH);
hipError_t launch_triplet_pyr_forward(const float* z, const float* bn, int h, int w, int UH, int UW, int hb, int f1_half, int level, int L,
                                      float margin, float eps, float* row_stat, double* part, int part0, hipStream_t s);
hipError_t launch_triplet_pyr_backward(const float* z, const float* bn, int h, int w, int UH, int UW, int hb, int f1_half, int level, int L,
                                       float eps, const int* case_in, const float* row_stat, const float* margin_in, const float* grad_loss,
                                       float* gout, hipStream_t s);

// BatchNorm work block of one pyramid level: device floats in stored-position order.
constexpr int kBnSc = 0, kBnSh = 128, kBnMean = 256, kBnRstd = 384, kBnMg = 512, kBnMgx = 640, kBnWorkFloats = 768;
constexpr int kBnMaxChunks = 1024;   // fp64 partials: [chunk][2][128]
constexpr size_t kBnPartBytes = size_t(kBnMaxChunks) * 2 * 128 * 8;
// bn: device [4][128] gamma, beta, running_mean, running_var (channel order).  z: blocked [Q,4,32] fp32.
hipError_t launch_bn_batch_stats(const float* z, long long Q, const float* bn, float eps, double* part, float* bw, float* mean_out,
                                 float* var_out, hipStream_t s);
hipError_t launch_bn_running_stats(const float* bn, float eps, float* bw, hipStream_t s);
// g (d L/d y, blocked) -> d L/d z in place; batch != 0 also writes d gamma, d beta [128] (channel order).
// absmax_out (optional): device word receiving max |d L / d z| (atomicMax on the bit pattern; the caller zeroes it)
hipError_t launch_bn_backward(int batch, float* g, const float* z, long long Q, double* part, float* bw, float* dgamma, float* dbeta,
                              hipStream_t s, unsigned* absmax_out = nullptr);
// pose head: relu'd conv5_3 activations [B,h,w,16,32] -> maxpool2 -> global mean -> fc [feat_dim,512].
hipError_t launch_pose_head(int prec, const void* act, int B, int h, int w, const float* fc_w, const float* fc_b,
                            int feat_dim, float* part /* scratch: B * (h / 2) * 512 floats */, float* pose, hipStream_t stream);


// --- input-gradient path (dfnet_grad.hip); all tensors in the blocked layout, element type by `prec`
// out = (act > 0 ? g : 0) + add; g / add may be null.
hipError_t launch_relu_gate(int prec, const void* g, const void* act, const void* add, size_t n, void* out, hipStream_t s);
// act [B,H,W,nblk*32] pre-pool, g [B,H/2,W/2,...] -> out [B,H,W,...]: gradient to the first maximum of each window.
hipError_t launch_maxpool_backward(int prec, const void* act, const void* g, int B, int H, int W, int nblk, void* out,
                                   hipStream_t s);
// ReLU gate (+ max-pool routing, + tap gradient) of the split-f16 training chain, written straight into the split storage at a
// power-of-two scale derived from the producers' |max| bounds (dfnet_grad.hip: gate_split_kernel); scale_out = [2^k, 2^-k].
hipError_t launch_gate_split(const float* g, int pooled_g, const void* act, int act_split, const float* add, int B, int H, int W, int nblk,
                             const unsigned* absmax_g, const unsigned* absmax_add, void* out, int out_split, float* scale_out,
                             hipStream_t s, int add_images = -1);   // add_images >= 0: `add` covers only the leading frames
// adjoint of launch_upsample: fp32 NCHW planes gup[b*bstride + c*UH*UW + ...] -> blocked [B,h,w,4,32].
hipError_t launch_upsample_backward(int prec, const float* gup, size_t bstride, int B, int h, int w, int UH, int UW, void* out,
                                    hipStream_t s, unsigned* absmax_out = nullptr);
// adjoint of launch_dfnet_prep: blocked gradient (RGB = first three elements of a pixel) -> d L/d x [B,3,H,W] fp32.
hipError_t launch_unprep(int prec, const void* g, int B, int H, int W, int nblk, float* gx, hipStream_t s);


// --- parameter gradients of the pose-regression path (dfnet_wgrad.hip); fp32 blocked activations
// dW[cout][cin][ks][ks] of a stride-1 "same" conv from g (gradient w.r.t. its pre-activation) and its input;
// `part` is scratch of part_floats floats.  cin, cout multiples of 32; ks 1 or 3.
// gscale != nullptr: split-f16 product (fp32-grade, f16 MFMA rate); device [scale, 1/scale] of g (launch_absmax_scale).
hipError_t launch_conv_wgrad(int ks, const float* g, const float* in, int B, int H, int W, int cout, int cin, float* part,
                             size_t part_floats, float* dW, hipStream_t s, const float* gscale = nullptr);
// conv1_1: input = the prep output (pix_stride floats per pixel, RGB first), g has 64 channels; dW [64][3][3][3].
hipError_t launch_conv0_wgrad(const float* g, const float* xn, int B, int H, int W, int pix_stride, float* part, size_t part_floats,
                              float* dW, hipStream_t s, const float* gscale = nullptr, float* db = nullptr);   // db (split-f16 only): + the bias gradient
hipError_t launch_bias_grad(const float* g, int B, int H, int W, int cout, float* part, size_t part_floats, float* db, hipStream_t s);
// pose head: relu5_3 activations + d pose -> gradient w.r.t. the activations, fc gradients; pooled = scratch of B * ((h + 1) / 2) * 512 floats.
// absmax_out (optional): device word that receives a bound of max |gact| (atomicMax on the float's bit pattern; caller zeroes it)
hipError_t launch_pose_head_backward(const float* act, int B, int h, int w, const float* fc_w, const float* gpose, int feat_dim,
                                     float* pooled, float* gact, float* dW_fc, float* db_fc, hipStream_t s,
                                     unsigned* absmax_out = nullptr);

// --- weight gradients over PRE-SPLIT operands (dfnet_wgrad_s.hip): g / in in the split row-planar storage of dfnet_conv.hip
// (hi | lo f16 planes; g scaled by gscale[0], in by kConvActScale), cout and cin multiples of 64, ks in {1, 3, 5}.  db may be null.
void conv_wgrad_split_scratch(int ks, int B, int H, int W, int cout, int cin, size_t* part_floats, size_t* part_b_floats);
hipError_t launch_conv_wgrad_split(int ks, const void* g, const void* in, const void* zeros, int B, int H, int W, int cout, int cin,
                                   float* part, size_t part_floats, float* part_b, size_t part_b_floats, float* dW, float* db,
                                   const float* gscale, hipStream_t s);
// fp32 blocked [B,H,W,nblk,32] / fp32 NCHW [B,C,H,W] -> split row-planar storage, scale = scale_dev[0] (device) or `scale`
hipError_t launch_split_rows(const float* x, int B, int H, int W, int nblk, float scale, const float* scale_dev, void* out, hipStream_t s);
hipError_t launch_nchw_to_split(const float* x, int B, int C, int H, int W, float scale, const float* scale_dev, void* out, hipStream_t s);

// out[0] = 2^k with max|x| * 2^k in [2^10, 2^11) (1 if the tensor is all zero), out[1] = 2^-k.  fp32 tensor of n elements;
// `part` is scratch of >= 1024 floats.
hipError_t launch_absmax_scale(const float* x, size_t n, float* part, float* out, hipStream_t s);
hipError_t launch_absmax_finalize(const float* part, int n, float* out, hipStream_t s);   // the second half alone: part[n] block maxima
// launch_relu_gate (fp32) + launch_absmax_scale of its output in one pass: part = 1024 floats of scratch, scale = [scale, 1/scale]
hipError_t launch_relu_gate_scale(const void* g, const void* act, const void* add, size_t n, void* out, float* part, float* scale,
                                  hipStream_t s);

// device-side re-packing of a conv's fp32 master weights (after an optimizer step): same fragment layouts as the host
// packers of dfnet_api.hip.  mode 0 = forward conv, 1 = its data-gradient conv (w_cout / w_cin = the forward shape).
hipError_t launch_pack_conv(int prec, const float* w, int cout, int cin, int ks, int first, int sb, int mb, int mode, int w_cout,
                            int w_cin, float wscale, void* out, hipStream_t s);
hipError_t launch_pack_bias(const float* b, int cout, float scale, float* out, hipStream_t s);
// The same, many tensors per launch.  kind 0: conv fragments of precision `prec` (g as for launch_pack_conv); kind 1: bias
// (g.cout = channels, g.wscale = scale).
struct PackGeom { int cout, cin, ks, first, sb, mb, mode, w_cout, w_cin; float wscale; };
struct PackJob { const float* w; void* out; PackGeom g; int prec, kind; };   // kind 0: conv fragments, 1: bias in C-fragment order, 2: plain copy of g.cout floats
hipError_t launch_pack_multi(const PackJob* jobs, int n_jobs, hipStream_t s);

}  // namespace dfn
