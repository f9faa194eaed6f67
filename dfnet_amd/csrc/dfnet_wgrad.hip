// dfnet_wgrad.hip — parameter gradients of DFNet's pose-regression path (gfx950): what loss.backward() leaves in
// model.parameters() of the pose regressor in the DFNet_dm step (/root/reference/script/feature/
// direct_feature_matching.py:372-374, optimizer over `model`; feature/dfnet.py:121-170 is the forward being
// differentiated: VGG16 convs + ReLU + max-pools, pool5 -> AdaptiveAvgPool2d(1) -> fc_pose).
//
// The data gradients reuse the forward MFMA convolution on flipped / transposed weights (dfnet_api.hip:
// pack_dgrad); here are the pieces that produce PARAMETER gradients:
//   * conv weight gradient   dW[co][ci][ky][kx] = sum_pixels g[p][co] * in[p + (ky-1, kx-1)][ci]
//     as an fp32 MFMA product (v_mfma_f32_32x32x2_f32): M = 32 output channels, N = 32 input channels, the
//     contraction runs over PIXELS, two per MFMA.  In the blocked-permuted NHWC layout the 32 channels of a pixel
//     are contiguous, so both operands are plain coalesced 128-byte loads — no transpose, no LDS staging: lane
//     (c, k) of the A operand reads channel position c of pixel 2t + k of g, lane (c, k) of B the same of the
//     shifted input pixel.  One wave keeps the nine taps' accumulators (144 VGPRs) and re-uses the g fragment for
//     all of them.  Pixel chunks are summed in a fixed order by a second kernel (deterministic), which also
//     un-permutes into the state_dict layout.
//   * conv1_1 (3 input channels) and the bias gradients: small reduction kernels.
//   * the pose head: fc_pose gradients, GAP + pool5 routing back to relu5_3.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dfnet_kernels.h"
#include "mfma_frag.h"

namespace dfn {

// true channel (within a 32-block) held at stored position e
__device__ __forceinline__ int chan_of_pos(int e) { return 4 * (e >> 4) + (e & 3) + 8 * ((e & 15) >> 2); }

// ------------------------------------------------------------------------------------------ conv weight gradient
// g   [B,H,W,MBLK,32]  gradient w.r.t. the conv's pre-activation (fp32, blocked)
// in  [B,H,W,NBLK,32]  the conv's input (fp32, blocked)
// part[chunk][mblk][nblk][KS*KS][32 (co position)][32 (ci position)]  partial sums
// TY kernel rows [ky0, ky0 + TY) per launch: 3x3 keeps all nine taps (144 accumulator registers), 5x5 runs row by row.
template <int KS, int TY>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ in, int B, int H,
                                                            int W, int mblks, int nblks, int n_chunks, int ky0, float* __restrict__ part) {
  constexpr int T = TY * KS, TT = KS * KS, R = KS / 2;
  __shared__ float red[3][16][64];      // waves 1..3 hand their accumulators to wave 0, one tap at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, k = lane >> 5;
  const int pair = blockIdx.x, chunk = blockIdx.y;
  const int mblk = pair / nblks, nblk = pair - mblk * nblks;
  const long long Q = (long long)B * H * W;
  // this workgroup's pixels: [q0, q1), split evenly over the four waves, each wave walks pixel pairs
  const long long per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = chunk * per, q1 = q0 + per < Q ? q0 + per : Q;
  const long long wper = ((q1 - q0 + 3) / 4 + 1) & ~1LL;   // even, so pairs never straddle two waves' ranges
  const long long w0 = q0 + wave * wper, w1 = w0 + wper < q1 ? w0 + wper : q1;
  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // Addressing: the input pixel of tap (ky, kx) for pixel q is q + (ky-R) W + (kx-R) in the flat [B*H*W] order, so
  // every tap gets its own wave-uniform buffer descriptor (base = the workgroup's first pixel shifted by the tap) and
  // all taps share ONE 32-bit per-lane byte offset that advances by a constant.  Border handling costs one select per
  // tap: a masked lane's offset is set past num_records and the hardware bounds check returns 0 — no address
  // arithmetic, no value select, no division in the loop; (x, y) are tracked incrementally for the masks only.
  constexpr uint32_t kOob = 0xFFFFFFF0u, kRecords = 0x80000000u;
  const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + ((size_t)q0 * mblks + mblk) * 32), 0, kRecords, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_in[T];
#pragma unroll
  for (int ty = 0; ty < TY; ++ty)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const long long off = (long long)(ky0 + ty - R) * W + (kx - R);   // uniform; may point before the tensor: masked lanes only
      rs_in[ty * KS + kx] =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + ((q0 + off) * nblks + nblk) * 32), 0, kRecords, 0x00020000);
    }
  const uint32_t gstep = (uint32_t)mblks * 128, istep = (uint32_t)nblks * 128;   // bytes per pixel
  struct Px { long long q; int x, y; };
  auto locate = [&](long long q) { Px p; p.q = q; p.x = int(q % W); p.y = int((q / W) % H); return p; };
  auto advance = [&](Px& p) {   // by four pixels
    p.q += 4;
    p.x += 4;
    while (p.x >= W) { p.x -= W; if (++p.y == H) p.y = 0; }
  };
  // Two pixel pairs per iteration: all 2 x (1 + T) loads are issued before the first MFMA, so the second pair's
  // (and, through the other wave of the SIMD, the next iteration's) latency hides under the first pair's MFMAs.
  auto fetch = [&](const Px& p, float& a, float (&bv)[T]) {
    const bool live = p.q < w1;
    const uint32_t d = (uint32_t)(p.q - q0);
    a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, live ? d * gstep + 4u * c : kOob, 0, 0));
    const uint32_t vo = d * istep + 4u * c;
    bool okx[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) okx[kx] = (unsigned)(p.x + kx - R) < (unsigned)W;
#pragma unroll
    for (int ty = 0; ty < TY; ++ty) {
      const bool rowok = live && (unsigned)(p.y + ky0 + ty - R) < (unsigned)H;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx)
        bv[ty * KS + kx] =
            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in[ty * KS + kx], (rowok && okx[kx]) ? vo : kOob, 0, 0));
    }
  };
  Px pa = locate(w0 + k < Q ? w0 + k : 0), pb = locate(w0 + k + 2 < Q ? w0 + k + 2 : 0);
  pa.q = w0 + k; pb.q = w0 + k + 2;
  for (; pa.q < w1 + k; advance(pa), advance(pb)) {   // every lane of the wave runs the same number of iterations
    float a0, a1, b0[T], b1[T];
    fetch(pa, a0, b0);
    fetch(pb, a1, b1);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
  }
  // reduce the four waves (fixed order), tap by tap, then one plain store per partial
  float* dst = part + ((((size_t)chunk * mblks + mblk) * nblks + nblk) * TT + ky0 * KS) * 1024;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ((acc[t][r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * k;   // row of the C fragment = co position
        dst[(t * 32 + i) * 32 + c] = v;                   // column = lane & 31 = ci position
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ split-f16 weight gradient
// The same contraction on the f16 matrix cores (v_mfma_f32_32x32x16_f16: 16 pixels per instruction, 16x the rate of
// the fp32 MFMA) at fp32-grade accuracy: both operands are split x = hi + lo (f16 each) in registers and a product is
// hi*hi + hi*lo + lo*hi accumulated in fp32, exactly as the split-f16 convolutions do (dfnet_conv.hip).  Operand
// layout of the 32x32x16 MFMA: lane (c, kg) holds channel position c of pixels 8 kg .. 8 kg + 7 of the group — eight
// dword loads per lane (each still a coalesced 128-byte row per pixel), converted with packed instructions.
// Scales: g * gscale[0] (the tensor's measured power of two, launch_absmax_scale — the same one its data-gradient conv
// uses), activations * kConvActScale with the hi part saturated; the finalize kernel multiplies the product of the
// inverse scales back.  Borders as in the fp32 kernel (per-tap descriptors, a masked element's offset is pushed out of
// range), with a wave-uniform fast path for groups that touch no image border.
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <bool SAT>
__device__ __forceinline__ void split8(const f32x8& v, float scale, half8& hi, half8& lo) {
  const f32x8 xs = v * scale;
  f32x8 cl = xs;
  if (SAT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) cl[k] = __builtin_amdgcn_fmed3f(xs[k], -65000.f, 65000.f);
  }
  hi = __builtin_convertvector(cl, half8);
  f32x8 r = xs - __builtin_convertvector(hi, f32x8);
  if (SAT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = __builtin_amdgcn_fmed3f(r[k], -65000.f, 65000.f);
  }
  lo = __builtin_convertvector(r, half8);
}

// MW = 32-channel output blocks per wave (1 or 2): with two, the input window of a pixel group is loaded and split once for 64
// output channels — the kernel is bound by the bytes it pulls through L2 (every (output block, input block) pair re-reads both
// tensors), not by the matrix pipe.
template <int KS, int TY, int MW = 1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const float* __restrict__ g, const float* __restrict__ in, int B, int H,
                                                               int W, int mblks, int nblks, int n_chunks, int ky0,
                                                               const float* __restrict__ gscale, float* __restrict__ part) {
  constexpr int T = TY * KS, TT = KS * KS, R = KS / 2;
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, kg = lane >> 5;
  const int pair = blockIdx.x, chunk = blockIdx.y;
  const int mblk = (pair / nblks) * MW, nblk = pair % nblks;   // this wave's first output block
  const long long Q = (long long)B * H * W;
  const long long per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = chunk * per, q1 = q0 + per < Q ? q0 + per : Q;
  const long long wper = ((q1 - q0 + 3) / 4 + 15) & ~15LL;   // whole 16-pixel groups per wave
  const long long w0 = q0 + wave * wper, w1 = w0 + wper < q1 ? w0 + wper : q1;
  f32x16 acc[MW][T];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  constexpr uint32_t kRecords = 0x80000000u;
  const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + ((size_t)q0 * mblks + mblk) * 32), 0, kRecords, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_in[T];
#pragma unroll
  for (int ty = 0; ty < TY; ++ty)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const long long off = (long long)(ky0 + ty - R) * W + (kx - R);
      rs_in[ty * KS + kx] =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + ((q0 + off) * nblks + nblk) * 32), 0, kRecords, 0x00020000);
    }
  const uint32_t gstep = (uint32_t)mblks * 128, istep = (uint32_t)nblks * 128;   // bytes per pixel
  const float sg = gscale[0];
  long long ql = w0 + 8 * kg;                                 // this lane's first pixel of the current group
  int x = 0, y = 0;
  if (ql < Q) { x = int(ql % W); y = int((ql / W) % H); }
  auto fetch8_plain = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t base, uint32_t step) {
    f32x8 v;
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, base + (uint32_t)t * step, 0, 0));
    return v;
  };
  auto group = [&](auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    uint32_t live = 0xFFu, rowm[TY], colm[KS];
    if (MASKED) {
      live = 0;
#pragma unroll
      for (int ty = 0; ty < TY; ++ty) rowm[ty] = 0;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) colm[kx] = 0;
      int xt = x, yt = y;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        live |= (uint32_t)(ql + t < w1) << t;
#pragma unroll
        for (int ty = 0; ty < TY; ++ty) rowm[ty] |= (uint32_t)((unsigned)(yt + ky0 + ty - R) < (unsigned)H) << t;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) colm[kx] |= (uint32_t)((unsigned)(xt + kx - R) < (unsigned)W) << t;
        if (++xt == W) { xt = 0; if (++yt == H) yt = 0; }
      }
    }
    const uint32_t d = (uint32_t)(ql - q0);
    auto fetch8 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t base, uint32_t step, uint32_t m) {
      f32x8 v;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        uint32_t off = base + (uint32_t)t * step;
        if (MASKED) off |= (uint32_t)(-(int)((~m >> t) & 1u));   // invalid element: offset out of range -> the load returns 0
        v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
      }
      return v;
    };
    // the taps are software-pipelined: tap t+1's eight loads are in flight while tap t is split and multiplied
    f32x8 araw[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) araw[m] = fetch8(rs_g, d * gstep + 4u * c + 128u * m, gstep, live);
    const uint32_t vo = d * istep + 4u * c;
    auto tap_mask = [&](int t) -> uint32_t { return MASKED ? (live & rowm[t / KS] & colm[t % KS]) : 0xFFu; };
    f32x8 nxt = fetch8(rs_in[0], vo, istep, tap_mask(0));
    half8 ah[MW], al[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) split8<false>(araw[m], sg, ah[m], al[m]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const f32x8 cur = nxt;
      if (t + 1 < T) nxt = fetch8(rs_in[t + 1], vo, istep, tap_mask(t + 1));
      half8 bh, bl;
      split8<true>(cur, kConvActScale, bh, bl);
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m][t], 0, 0, 0);
      }
    }
  };
  // Interior groups of a one-row kernel (TY == 1): the KS taps of the row read the SAME pixels shifted by one, so the
  // lane loads its window once — 8 + KS - 1 pixels — splits each element once, and every tap's fragment is a register
  // selection: even shifts are whole registers of the packed (hi | lo) rows, odd shifts one v_alignbit per register.
  constexpr int NROW = 8 + KS - 1 + ((8 + KS - 1) & 1);   // pixels a lane loads for one kernel row (even count)
  struct RowData { f32x8 a[MW]; float raw[NROW]; };
  auto load_row = [&](long long qlane) {
    RowData r;
    const uint32_t d = (uint32_t)(qlane - q0);
#pragma unroll
    for (int m = 0; m < MW; ++m) r.a[m] = fetch8_plain(rs_g, d * gstep + 4u * c + 128u * m, gstep);
    const uint32_t vo = d * istep + 4u * c;
#pragma unroll
    for (int j = 0; j < NROW; ++j)   // tap 0's descriptor starts R pixels to the left of the lane's first pixel
      r.raw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in[0], vo + (uint32_t)(j < 8 + KS - 1 ? j : j - 1) * istep, 0, 0));
    return r;
  };
  auto mac_row = [&](const RowData& rd) {
    half8 ah[MW], al[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) split8<false>(rd.a[m], sg, ah[m], al[m]);
    typedef _Float16 half2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint32_t hi[NROW / 2], lo[NROW / 2];
#pragma unroll
    for (int j = 0; j < NROW / 2; ++j) {
      const f32x2 xs = f32x2{rd.raw[2 * j], rd.raw[2 * j + 1]} * kConvActScale;
      const f32x2 cl = {__builtin_amdgcn_fmed3f(xs[0], -65000.f, 65000.f), __builtin_amdgcn_fmed3f(xs[1], -65000.f, 65000.f)};
      const half2 h = __builtin_convertvector(cl, half2);
      f32x2 r = xs - __builtin_convertvector(h, f32x2);
      r = f32x2{__builtin_amdgcn_fmed3f(r[0], -65000.f, 65000.f), __builtin_amdgcn_fmed3f(r[1], -65000.f, 65000.f)};
      hi[j] = __builtin_bit_cast(uint32_t, h);
      lo[j] = __builtin_bit_cast(uint32_t, (half2)__builtin_convertvector(r, half2));
    }
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 uh, ul;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((kx & 1) == 0) { uh[i] = hi[kx / 2 + i]; ul[i] = lo[kx / 2 + i]; }
        else {
          uh[i] = __builtin_amdgcn_alignbit(hi[(kx + 1) / 2 + i], hi[(kx - 1) / 2 + i], 16);
          ul[i] = __builtin_amdgcn_alignbit(lo[(kx + 1) / 2 + i], lo[(kx - 1) / 2 + i], 16);
        }
      }
      const half8 bh = __builtin_bit_cast(half8, uh), bl = __builtin_bit_cast(half8, ul);
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        acc[m][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m][kx], 0, 0, 0);
        acc[m][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m][kx], 0, 0, 0);
        acc[m][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m][kx], 0, 0, 0);
      }
    }
  };
  // interior (wave-uniform): every pixel of every lane's window is live, no tap leaves the image, no window wraps a row
  auto interior_at = [&](long long qlane, int xx, int yy) {
    const bool in = qlane + 7 < w1 && xx >= R && xx + 7 + R < W && yy + ky0 - R >= 0 && yy + ky0 + TY - 1 - R < H;
    return __builtin_amdgcn_ballot_w64(!in) == 0;
  };
  constexpr bool kRowPath = TY == 1 && KS > 1;
  RowData pre;             // the NEXT group's loads, in flight while the current group is split and multiplied
  bool have_pre = false;   // wave-uniform
  for (long long qg = w0; qg < w1; qg += 16) {
    int xn = x + 16, yn = y;
    while (xn >= W) { xn -= W; if (++yn == H) yn = 0; }
    if (interior_at(ql, x, y)) {
      if constexpr (kRowPath) {
        RowData cur;
        if (have_pre) cur = pre; else cur = load_row(ql);
        have_pre = qg + 16 < w1 && interior_at(ql + 16, xn, yn);
        if (have_pre) pre = load_row(ql + 16);
        mac_row(cur);
      } else {
        group(std::integral_constant<bool, false>{});
      }
    } else {
      have_pre = false;
      group(std::integral_constant<bool, true>{});
    }
    ql += 16;
    x = xn; y = yn;
  }
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    float* dst = part + ((((size_t)chunk * mblks + mblk + m) * nblks + nblk) * TT + ky0 * KS) * 1024;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[m][t][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = ((acc[m][t][r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
          const int i = (r & 3) + 8 * (r >> 2) + 4 * kg;
          dst[(t * 32 + i) * 32 + c] = v;
        }
      }
      __syncthreads();
    }
  }
}

// Sum the chunks in order and write dW[co][ci][ky][kx] (state_dict layout), un-permuting the channel positions.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ part, int mblks, int nblks, int T, int n_chunks,
                                                             int cout, int cin, float* __restrict__ dW, const float* __restrict__ gscale) {
  const size_t n = (size_t)mblks * nblks * T * 1024;
  const float unscale = gscale ? gscale[1] * (1.f / kConvActScale) : 1.f;   // split-f16 operands were pre-scaled by powers of two
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int ch = 0; ch < n_chunks; ++ch) s += part[(size_t)ch * n + i];
    const int cj = int(i & 31), ci_ = int((i >> 5) & 31);
    size_t r = i >> 10;
    const int t = int(r % T); r /= T;
    const int nblk = int(r % nblks), mblk = int(r / nblks);
    const int co = 32 * mblk + chan_of_pos(ci_), ci = 32 * nblk + chan_of_pos(cj);
    if (co < cout && ci < cin) dW[((size_t)co * cin + ci) * T + t] = s * unscale;
  }
}

hipError_t launch_conv_wgrad(int ks, const float* g, const float* in, int B, int H, int W, int cout, int cin, float* part,
                             size_t part_floats, float* dW, hipStream_t s, const float* gscale) {
  const int mblks = cout / 32, nblks = cin / 32, T = ks * ks;
  if (cout % 32 || cin % 32 || (ks != 1 && ks != 3 && ks != 5)) return hipErrorInvalidValue;
  const long long Q = (long long)B * H * W;
  const int pairs = mblks * nblks;
  constexpr int target_wgs = 1536;   // two rounds of the 768 resident workgroups (384 ... 3072 measure the same within noise)
  const int mw = (gscale && ks == 3 && mblks % 2 == 0) ? 2 : 1;   // output blocks per wave (split-f16 3x3: two)
  long long n_chunks = target_wgs / (pairs / mw);         // ~1536 workgroups
  const long long max_by_work = (Q + 1023) / 1024;        // at least ~1024 pixels per workgroup
  if (n_chunks > max_by_work) n_chunks = max_by_work;
  if (n_chunks < 1) n_chunks = 1;
  while (n_chunks > 1 && (size_t)n_chunks * pairs * T * 1024 > part_floats) --n_chunks;
  if ((size_t)n_chunks * pairs * T * 1024 > part_floats) return hipErrorInvalidValue;
  // 32-bit per-lane byte offsets inside a workgroup's pixel range, below the descriptors' num_records (conv_wgrad_kernel)
  if ((unsigned long long)((Q + n_chunks - 1) / n_chunks + 32) * (unsigned)(mblks > nblks ? mblks : nblks) * 128 >= (1ull << 31))
    return hipErrorInvalidValue;
  const dim3 grid(pairs, int(n_chunks));
  if (gscale) {   // split-f16 product (fp32-grade): gscale = device [scale, 1 / scale] of g
    if (mw == 2) {
      const dim3 grid2(pairs / 2, int(n_chunks));
      for (int ky = 0; ky < 3; ++ky)   // one kernel row per launch, two output blocks per wave: 96 accumulator registers
        hipLaunchKernelGGL((conv_wgrad_x3_kernel<3, 1, 2>), grid2, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), ky, gscale, part);
    } else if (ks == 3)
      for (int ky = 0; ky < 3; ++ky)   // one kernel row per launch: 48 accumulator registers, no spills
        hipLaunchKernelGGL((conv_wgrad_x3_kernel<3, 1>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), ky, gscale, part);
    else if (ks == 1) hipLaunchKernelGGL((conv_wgrad_x3_kernel<1, 1>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), 0, gscale, part);
    else
      for (int ky = 0; ky < 5; ++ky)
        hipLaunchKernelGGL((conv_wgrad_x3_kernel<5, 1>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), ky, gscale, part);
  } else
  if (ks == 3) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), 0, part);
  else if (ks == 1) hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), 0, part);
  else
    for (int ky = 0; ky < 5; ++ky)
      hipLaunchKernelGGL((conv_wgrad_kernel<5, 1>), grid, dim3(256), 0, s, g, in, B, H, W, mblks, nblks, int(n_chunks), ky, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const size_t n = (size_t)pairs * T * 1024;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(int((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, part, mblks,
                     nblks, T, int(n_chunks), cout, cin, dW, gscale);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ conv1_1 (3 input channels)
// x_norm: the prep output [B,H,W,2*SB] (RGB = elements 0..2).  dW[co][c][ky][kx], co < 64.  One workgroup per pixel
// chunk, thread = (co, 4 pixel lanes); partials [chunk][64][27] then a fixed-order sum.
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ xn, int B, int H, int W,
                                                          int pix_stride, int n_chunks, float* __restrict__ part) {
  __shared__ float red[4][64][27];
  const int pos = threadIdx.x & 63, pl = threadIdx.x >> 6;   // stored channel position of g (64 = 2 blocks), pixel lane
  const long long Q = (long long)B * H * W, per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = blockIdx.x * per, q1 = q0 + per < Q ? q0 + per : Q;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  for (long long q = q0 + pl; q < q1; q += 4) {
    const int x = int(q % W);
    const long long rr = q / W;
    const int y = int(rr % H);
    const long long b = rr / H;
    const float gv = g[q * 64 + pos];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const float* px = xn + ((b * H + yy) * (long long)W + xx) * pix_stride;
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c * 9 + ky * 3 + kx] += gv * px[c];
        }
      }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) red[pl][pos][t] = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 27; i += 256) {
    const int p = i / 27, t = i - p * 27;
    part[((size_t)blockIdx.x * 64 + p) * 27 + t] = ((red[0][p][t] + red[1][p][t]) + red[2][p][t]) + red[3][p][t];
  }
}
__global__ __launch_bounds__(256) void conv0_finalize_kernel(const float* __restrict__ part, int n_chunks, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 27) return;
  float s = 0.f;
  for (int ch = 0; ch < n_chunks; ++ch) s += part[(size_t)ch * 64 * 27 + i];
  const int p = i / 27, t = i - p * 27;
  const int co = 32 * (p >> 5) + chan_of_pos(p & 31);
  dW[co * 27 + t] = s;   // [co][c][ky][kx]
}
// The same gradient on the matrix cores (split-f16, as conv_wgrad_x3_kernel): M = the 64 output channels (two blocks),
// N = the 27 (c, ky, kx) combinations — lane (j, kg) gathers input channel c_j of the pixel shifted by (ky_j, kx_j) for
// its eight pixels — K = 16 pixels per MFMA.  Six MFMAs per 16 pixels cover the whole layer; the scalar kernel above
// spends 27 loads and FMAs per pixel and thread.  part[chunk][64 (co position)][32 (combination: 27 taps, then a column of ones
// = the bias gradient)].
__global__ __launch_bounds__(256) void conv0_wgrad_x3_kernel(const float* __restrict__ g, const float* __restrict__ xn, int B, int H, int W,
                                                             int pix_stride, int n_chunks, const float* __restrict__ gscale,
                                                             float* __restrict__ part) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kg = lane >> 5;
  const int cj = j / 9, kyj = (j % 9) / 3, kxj = j % 3;
  const bool combo = j < 27;
  const long long Q = (long long)B * H * W;
  const long long per = (Q + n_chunks - 1) / n_chunks;
  const long long q0 = blockIdx.x * per, q1 = q0 + per < Q ? q0 + per : Q;
  const long long wper = ((q1 - q0 + 3) / 4 + 15) & ~15LL;
  const long long w0 = q0 + wave * wper, w1 = w0 + wper < q1 ? w0 + wper : q1;
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  constexpr uint32_t kRecords = 0x80000000u;
  const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + (size_t)q0 * 64), 0, kRecords, 0x00020000);
  // the input descriptor starts one row and one pixel before the chunk: every tap offset is then non-negative
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn + (q0 - W - 1) * (long long)pix_stride), 0, kRecords, 0x00020000);
  const uint32_t xstep = (uint32_t)pix_stride * 4u;
  const uint32_t lane_off = (uint32_t)((kyj * W + kxj) * pix_stride + cj) * 4u;   // tap shift + channel, from the descriptor base
  const float sg = gscale[0];
  long long ql = w0 + 8 * kg;
  int x = 0, y = 0;
  if (ql < Q) { x = int(ql % W); y = int((ql / W) % H); }
  for (long long qg = w0; qg < w1; qg += 16) {
    const uint32_t d = (uint32_t)(ql - q0);
    f32x8 xv, g0, g1;
    int xt = x, yt = y;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool live = ql + t < w1;
      const bool ok = live && combo && (unsigned)(yt + kyj - 1) < (unsigned)H && (unsigned)(xt + kxj - 1) < (unsigned)W;
      xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, ok ? (d + t) * xstep + lane_off : 0xFFFFFFF0u, 0, 0));
      if (j == 27) xv[t] = live ? 1.f : 0.f;   // column 27 multiplies the gradient by one: the bias gradient, from the same MFMAs
      const uint32_t go = live ? (d + t) * 256u + 4u * j : 0xFFFFFFF0u;
      g0[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, go, 0, 0));
      g1[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, live ? go + 128u : go, 0, 0));
      if (++xt == W) { xt = 0; if (++yt == H) yt = 0; }
    }
    half8 bh, bl, ah, al;
    split8<true>(xv, kConvActScale, bh, bl);
    split8<false>(g0, sg, ah, al);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
    split8<false>(g1, sg, ah, al);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[1], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[1], 0, 0, 0);
    ql += 16;
    x += 16;
    while (x >= W) { x -= W; if (++y == H) y = 0; }
  }
  float* dst = part + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[m][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ((acc[m][r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * kg;   // co position within block m
        dst[(m * 32 + i) * 32 + j] = v;
      }
    }
    __syncthreads();
  }
}
// A workgroup owns 16 consecutive elements: 16 chunk segments per element (a 64-byte run per segment and chunk, the partials are L2
// resident), eight loads in flight per thread, segments combined through LDS in a fixed order (deterministic).  (Eight workgroups walking
// all chunks per element, as this was until round 6, took 49 us at the tail of the training steps' backward chain.)
__global__ __launch_bounds__(256) void conv0_x3_finalize_kernel(const float* __restrict__ part, int n_chunks, const float* __restrict__ gscale,
                                                                float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[16][16];
  const int e = threadIdx.x & 15, seg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;          // < 64 * 32 (grid = 128)
  const int per = (n_chunks + 15) / 16, c0 = seg * per, c1 = min(n_chunks, c0 + per);
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int ch = c0;
  for (; ch + 8 <= c1; ch += 8)
#pragma unroll
    for (int k = 0; k < 8; ++k) s8[k] += part[(size_t)(ch + k) * 64 * 32 + i];
  for (int k = 0; ch < c1; ++ch, ++k) s8[k] += part[(size_t)ch * 64 * 32 + i];
  red[seg][e] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __syncthreads();
  if (seg) return;
  float s = 0.f;
  for (int k = 0; k < 16; ++k) s += red[k][e];
  const int p = i >> 5, t = i & 31;
  const int co = 32 * (p >> 5) + chan_of_pos(p & 31);
  const float v = s * gscale[1] * (1.f / kConvActScale);
  if (t < 27) dW[co * 27 + t] = v;   // [co][c][ky][kx]
  else if (t == 27 && db) db[co] = v;
}

hipError_t launch_conv0_wgrad(const float* g, const float* xn, int B, int H, int W, int pix_stride, float* part, size_t part_floats,
                              float* dW, hipStream_t s, const float* gscale, float* db) {
  if (db && !gscale) return hipErrorInvalidValue;   // (the bias gradient rides the split-f16 kernel only)
  if (gscale) {
    const long long Q = (long long)B * H * W;
    long long n_chunks = (Q + 1023) / 1024;
    if (n_chunks > 2048) n_chunks = 2048;
    if ((size_t)n_chunks * 64 * 32 > part_floats) return hipErrorInvalidValue;
    if ((unsigned long long)((Q + n_chunks - 1) / n_chunks + W + 64) * 256ull >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv0_wgrad_x3_kernel, dim3(int(n_chunks)), dim3(256), 0, s, g, xn, B, H, W, pix_stride, int(n_chunks), gscale, part);
    hipLaunchKernelGGL(conv0_x3_finalize_kernel, dim3(64 * 32 / 16), dim3(256), 0, s, part, int(n_chunks), gscale, dW, db);
    return hipGetLastError();
  }

  const long long Q = (long long)B * H * W;
  long long n_chunks = (Q + 511) / 512;
  if (n_chunks > 1024) n_chunks = 1024;
  if ((size_t)n_chunks * 64 * 27 > part_floats) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(int(n_chunks)), dim3(256), 0, s, g, xn, B, H, W, pix_stride, int(n_chunks), part);
  hipLaunchKernelGGL(conv0_finalize_kernel, dim3((64 * 27 + 255) / 256), dim3(256), 0, s, part, int(n_chunks), dW);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ bias gradient
// db[co] = sum over pixels of g[p][co].  Grid = (32-channel block, pixel chunk): 8 pixel lanes x 32 positions per
// workgroup write a partial; a second kernel adds the chunks in order (deterministic).
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ g, long long Q, int blks, int n_chunks,
                                                        float* __restrict__ part) {
  __shared__ float red[8][32];
  const int pos = threadIdx.x & 31, pl = threadIdx.x >> 5, blk = blockIdx.x, chunk = blockIdx.y;
  const long long per = (Q + n_chunks - 1) / n_chunks, q0 = chunk * per, q1 = q0 + per < Q ? q0 + per : Q;
  const size_t step = (size_t)blks * 32;
  const float* p = g + (size_t)blk * 32 + pos;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains: four loads in flight per thread
  long long q = q0 + pl;
  for (; q + 24 < q1; q += 32) {
    s0 += p[(size_t)q * step];
    s1 += p[(size_t)(q + 8) * step];
    s2 += p[(size_t)(q + 16) * step];
    s3 += p[(size_t)(q + 24) * step];
  }
  for (; q < q1; q += 8) s0 += p[(size_t)q * step];
  red[pl][pos] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0) {
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += red[i][pos];
    part[((size_t)chunk * blks + blk) * 32 + pos] = t;
  }
}
// One workgroup per 32-channel block: 32 chunk lanes per position with four loads in flight each, combined in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void bias_grad_finalize_kernel(const float* __restrict__ part, int blks, int n_chunks, int cout,
                                                                  float* __restrict__ db) {
  __shared__ float red[32][32];
  const int pos = threadIdx.x & 31, cl = threadIdx.x >> 5, blk = blockIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int ch = cl;
  for (; ch + 96 < n_chunks; ch += 128) {
    s0 += part[((size_t)ch * blks + blk) * 32 + pos];
    s1 += part[((size_t)(ch + 32) * blks + blk) * 32 + pos];
    s2 += part[((size_t)(ch + 64) * blks + blk) * 32 + pos];
    s3 += part[((size_t)(ch + 96) * blks + blk) * 32 + pos];
  }
  for (; ch < n_chunks; ch += 32) s0 += part[((size_t)ch * blks + blk) * 32 + pos];
  red[cl][pos] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0) {
    float t = 0.f;
    for (int i = 0; i < 32; ++i) t += red[i][pos];
    const int co = 32 * blk + chan_of_pos(pos);
    if (co < cout) db[co] = t;
  }
}
hipError_t launch_bias_grad(const float* g, int B, int H, int W, int cout, float* part, size_t part_floats, float* db, hipStream_t s) {
  const long long Q = (long long)B * H * W;
  const int blks = cout / 32;
  long long n_chunks = (Q + 255) / 256;                       // >= 256 pixels per workgroup,
  const long long cap = 4096 / blks > 8 ? 4096 / blks : 8;    // ~4096 workgroups in all
  if (n_chunks > cap) n_chunks = cap;
  if ((size_t)n_chunks * blks * 32 > part_floats) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(blks, int(n_chunks)), dim3(256), 0, s, g, Q, blks, int(n_chunks), part);
  hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3(blks), dim3(1024), 0, s, part, blks, int(n_chunks), cout, db);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ pose head backward
// forward: relu5_3 act [B,h,w,512] -> pool5 (2x2 max) -> mean over (h/2, w/2) -> fc (feature/dfnet.py:168-170).
// Given d pose [B,F]: the gradient w.r.t. act — each pooled pixel's share d_pooled[ch] / (ho wo) goes to the FIRST maximum of its window
// (torch's max-pool routing) — and the pooled activation for the fc weight gradient.  One workgroup per (image, pair of rows): the rows'
// windows and nothing else; the pooled mean is left as per-row-pair partial sums pooled_part[B][(h+1)/2][512] (channel order) that
// fc_grad_kernel adds in a fixed order.  (One workgroup per image, as this was until round 6, walked the h*w pixels serially: 123 us
// at the head of the backward chain of both training steps.)
__global__ __launch_bounds__(512) void pose_head_backward_kernel(const float* __restrict__ act, int h, int w, const float* __restrict__ fc_w,
                                                                 const float* __restrict__ gpose, int feat_dim,
                                                                 float* __restrict__ pooled_part, float* __restrict__ gact,
                                                                 unsigned* __restrict__ absmax_out) {
  const int cpos = threadIdx.x;   // stored position 0..511
  const size_t b = blockIdx.x;
  const int yo = blockIdx.y, np = gridDim.y;
  const int ho = h / 2, wo = w / 2;
  const int blk = cpos >> 5, e = cpos & 31;
  const int ch = blk * 32 + chan_of_pos(e);
  float dp = 0.f;
  for (int o = 0; o < feat_dim; ++o) dp += gpose[b * feat_dim + o] * fc_w[o * 512 + ch];
  dp /= float(ho * wo);
  if (absmax_out && yo == 0) {   // bound of |gact| for the split of the gated gradient: one atomicMax per wave of the image's first workgroup
    float m = fabsf(dp);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(absmax_out, __float_as_uint(m));
  }
  float sum = 0.f;
  const float* s0 = act + ((b * h + 2 * yo) * (size_t)w) * 512 + cpos;
  float* g0 = gact + ((b * h + 2 * yo) * (size_t)w) * 512 + cpos;
  if (yo < ho) {
    float* g1 = g0 + (size_t)w * 512;
    for (int xo = 0; xo < wo; ++xo) {
      const float* s = s0 + (size_t)xo * 1024;
      const float a0 = s[0], a1 = s[512], a2 = s[(size_t)w * 512], a3 = s[(size_t)w * 512 + 512];
      const float m = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
      const int first = a0 == m ? 0 : (a1 == m ? 1 : (a2 == m ? 2 : 3));
      sum += m;
      g0[(size_t)xo * 1024] = first == 0 ? dp : 0.f;
      g0[(size_t)xo * 1024 + 512] = first == 1 ? dp : 0.f;
      g1[(size_t)xo * 1024] = first == 2 ? dp : 0.f;
      g1[(size_t)xo * 1024 + 512] = first == 3 ? dp : 0.f;
    }
    if (w & 1) { g0[(size_t)(w - 1) * 512] = 0.f; g1[(size_t)(w - 1) * 512] = 0.f; }   // a trailing column belongs to no window
  } else {
    for (int x = 0; x < w; ++x) g0[(size_t)x * 512] = 0.f;                              // and so does a trailing row
  }
  pooled_part[(b * np + yo) * 512 + ch] = sum;
}
// dW_fc[o][ch] = sum_b gpose[b][o] pooled[b][ch];  db[o] = sum_b gpose[b][o];  pooled[b][ch] = sum of the row-pair partials / (ho wo)
__global__ __launch_bounds__(512) void fc_grad_kernel(const float* __restrict__ gpose, const float* __restrict__ pooled_part, int B, int np,
                                                      float inv_windows, int feat_dim, float* __restrict__ dW, float* __restrict__ db) {
  const int ch = threadIdx.x, o = blockIdx.x;
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    float pooled = 0.f;
    for (int r = 0; r < np; ++r) pooled += pooled_part[((size_t)b * np + r) * 512 + ch];
    s += gpose[b * feat_dim + o] * (pooled * inv_windows);
    sb += gpose[b * feat_dim + o];
  }
  dW[o * 512 + ch] = s;
  if (ch == 0) db[o] = sb;
}
// pooled_part: B * ((h + 1) / 2) * 512 floats (pose_head_part_floats)
hipError_t launch_pose_head_backward(const float* act, int B, int h, int w, const float* fc_w, const float* gpose, int feat_dim,
                                     float* pooled_part, float* gact, float* dW_fc, float* db_fc, hipStream_t s, unsigned* absmax_out) {
  const int np = (h + 1) / 2;
  hipLaunchKernelGGL(pose_head_backward_kernel, dim3(B, np), dim3(512), 0, s, act, h, w, fc_w, gpose, feat_dim, pooled_part, gact, absmax_out);
  hipLaunchKernelGGL(fc_grad_kernel, dim3(feat_dim), dim3(512), 0, s, gpose, pooled_part, B, np, 1.f / float((h / 2) * (w / 2)), feat_dim, dW_fc,
                     db_fc);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ device-side weight packer
// After an optimizer step the fp32 master weights live in device memory; these kernels re-create the packed MFMA
// fragments (dfnet_api.hip: pack_conv / pack_conv_x3 / pack_dgrad, same index maps) without a host round trip.
// mode 0: forward conv  w[co][ci][ky][kx];  mode 1: data-gradient conv  w'[co'][ci'][ky][kx] = w[ci'][co'][K-1-ky][K-1-kx]
// prec 0: f16 fragments, 1: fp32 fragments, 2: split-f16 slices [hi|lo] of w * wscale.
__device__ __forceinline__ float pack_source(const float* w, const PackGeom& g, int co, int ci, int ky, int kx) {
  if (co >= g.cout || ci < 0 || ci >= g.cin) return 0.f;
  if (g.mode == 0) return w[(((size_t)co * g.w_cin + ci) * g.ks + ky) * g.ks + kx];
  if (co >= g.w_cin || ci >= g.w_cout) return 0.f;                         // padded output channels of the dgrad conv
  return w[(((size_t)ci * g.w_cin + co) * g.ks + (g.ks - 1 - ky)) * g.ks + (g.ks - 1 - kx)];
}
// A thread packs the SPC (8, or 1 for fp32 fragments) consecutive slots of one fragment lane: one index decomposition and one 16-byte
// store per plane instead of eight of each (the per-element form spent the re-pack of a training step, ~30 M elements, on 64-bit
// divisions and 2-byte stores: 0.17-0.21 ms per step).
template <int PREC>
__device__ __forceinline__ void pack_conv_range(const float* __restrict__ w, const PackGeom& g, void* __restrict__ out, size_t first_i, size_t stride) {
  constexpr int SPC = PREC == 1 ? 1 : 8;
  const unsigned nblk = g.first ? 1 : g.cin / 32, kcb = g.sb / SPC, groups = g.cout / 32 / g.mb, ks = g.ks, mb = g.mb;
  const size_t n = (size_t)groups * nblk * ks * mb * ks * kcb * 64;   // fragment lanes (< 2^32: checked by the launcher)
  for (size_t i = first_i; i < n; i += stride) {
    unsigned r = (unsigned)i;
    const int lane = int(r & 63); r >>= 6;
    const int kc = int(r % kcb); r /= kcb;
    const int kx = int(r % ks); r /= ks;
    const int m = int(r % mb); r /= mb;
    const int ky = int(r % ks); r /= ks;
    const int blk = int(r % nblk);
    const int cg = int(r / nblk);
    const int co = 32 * (cg * g.mb + m) + (lane & 31);
    const int hh = lane >> 5;
    float v[SPC];
#pragma unroll
    for (int j = 0; j < SPC; ++j) {
      const int s = kc * SPC + j;
      const int ci = g.first ? ((hh == 0 && s < 3) ? s : -1) : 32 * blk + 4 * hh + (s & 3) + 8 * (s >> 2);
      v[j] = pack_source(w, g, co, ci, ky, kx);
    }
    if (PREC == 1) static_cast<float*>(out)[i] = v[0];
    else if (PREC == 0) {
      half8 h8;
#pragma unroll
      for (int j = 0; j < 8; ++j) h8[j] = (_Float16)v[j % SPC];
      *reinterpret_cast<half8*>(static_cast<_Float16*>(out) + i * 8) = h8;
    } else {
      // layout [cg][blk][ky][kc][hi|lo][mb][kx][lane][8] (dfnet_api.hip: pack_conv_x3)
      const size_t half_slice = (size_t)g.mb * g.ks * 64 * 8;
      const size_t slice = (((size_t)cg * nblk + blk) * g.ks + ky) * kcb + kc;
      const size_t o = (((size_t)m * g.ks + kx) * 64 + lane) * 8;
      _Float16* sl = static_cast<_Float16*>(out) + slice * 2 * half_slice;
      half8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float vs = v[j % SPC] * g.wscale;
        hi[j] = (_Float16)vs;
        lo[j] = (_Float16)(vs - (float)hi[j]);
      }
      *reinterpret_cast<half8*>(sl + o) = hi;
      *reinterpret_cast<half8*>(sl + half_slice + o) = lo;
    }
  }
}
template <int PREC>
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, PackGeom g, void* __restrict__ out) {
  pack_conv_range<PREC>(w, g, out, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
// bias [cout] -> C-fragment order [cout/32][2][16], times `scale`
__device__ __forceinline__ void pack_bias_range(const float* __restrict__ b, int cout, float scale, float* __restrict__ out, size_t first_i,
                                                size_t stride) {
  for (size_t i = first_i; i < (size_t)cout; i += stride) {
    const int m = int(i >> 5), e = int(i & 31), hh = e >> 4, r = e & 15;
    out[i] = b[32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh] * scale;
  }
}
// Every tensor of a device re-pack in ONE launch (the optimizer step of a training loop re-packs 13-19 convolutions, forward and
// data-gradient fragments, plus their biases: 50-72 small launches per step otherwise).  blockIdx.y = job.
constexpr int kPackJobsPerLaunch = 24;
struct PackJobs { PackJob job[kPackJobsPerLaunch]; int n; };
__global__ __launch_bounds__(256) void pack_multi_kernel(PackJobs js) {
  const PackJob& j = js.job[blockIdx.y];
  const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (j.kind == 1) { pack_bias_range(j.w, j.g.cout, j.g.wscale, static_cast<float*>(j.out), i0, stride); return; }
  if (j.kind == 2) {   // plain copy of g.cout floats (fc weights, BatchNorm vectors: a dozen 5 us device-to-device copies per step otherwise)
    for (size_t i = i0; i < (size_t)j.g.cout; i += stride) static_cast<float*>(j.out)[i] = j.w[i];
    return;
  }
  if (j.prec == 0) pack_conv_range<0>(j.w, j.g, j.out, i0, stride);
  else if (j.prec == 1) pack_conv_range<1>(j.w, j.g, j.out, i0, stride);
  else pack_conv_range<2>(j.w, j.g, j.out, i0, stride);
}
static bool pack_lanes_fit(const PackGeom& g, int prec) {   // pack_conv_range decomposes the fragment-lane index in 32 bits
  const int spc = prec == 1 ? 1 : 8;
  const size_t nblk = g.first ? 1 : g.cin / 32;
  return (size_t)(g.cout / 32 / g.mb) * nblk * g.ks * g.mb * g.ks * (g.sb / spc) * 64 < (1ull << 32);
}
hipError_t launch_pack_multi(const PackJob* jobs, int n_jobs, hipStream_t s) {
  for (int i = 0; i < n_jobs; ++i)
    if (jobs[i].kind == 0 && !pack_lanes_fit(jobs[i].g, jobs[i].prec)) return hipErrorInvalidValue;
  for (int at = 0; at < n_jobs; at += kPackJobsPerLaunch) {
    PackJobs js{};
    js.n = n_jobs - at < kPackJobsPerLaunch ? n_jobs - at : kPackJobsPerLaunch;
    for (int i = 0; i < js.n; ++i) js.job[i] = jobs[at + i];
    hipLaunchKernelGGL(pack_multi_kernel, dim3(512, js.n), dim3(256), 0, s, js);
  }
  return hipGetLastError();
}
hipError_t launch_pack_conv(int prec, const float* w, int cout, int cin, int ks, int first, int sb, int mb, int mode, int w_cout,
                            int w_cin, float wscale, void* out, hipStream_t s) {
  const PackGeom g{cout, cin, ks, first, sb, mb, mode, w_cout, w_cin, wscale};
  if (!pack_lanes_fit(g, prec)) return hipErrorInvalidValue;
  const dim3 grid(2048), block(256);
  if (prec == 0) hipLaunchKernelGGL(pack_conv_kernel<0>, grid, block, 0, s, w, g, out);
  else if (prec == 1) hipLaunchKernelGGL(pack_conv_kernel<1>, grid, block, 0, s, w, g, out);
  else hipLaunchKernelGGL(pack_conv_kernel<2>, grid, block, 0, s, w, g, out);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void pack_bias_kernel(const float* __restrict__ b, int cout, float scale, float* __restrict__ out) {
  pack_bias_range(b, cout, scale, out, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
hipError_t launch_pack_bias(const float* b, int cout, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pack_bias_kernel, dim3((cout + 255) / 256), dim3(256), 0, s, b, cout, scale, out);
  return hipGetLastError();
}

}  // namespace dfn
