// dfnet_triplet_pyr.hip — DFNet's training triplet loss computed FROM THE LOW-RESOLUTION PYRAMID, forward and backward (gfx950).
//
// /root/reference/script/feature/dfnet.py:142-160 enlarges every adapted level [2B,128,h,w] to [.,.,H,W] (UpsamplingBilinear2d,
// align_corners=True) and stacks the levels; /root/reference/script/feature/misc.py:355-435 then takes, per feature ROW (level, image,
// channel, output row Y), the pairwise L2 distances  ||x - y + eps||  and  ||x - z + eps||  over the W output columns, the hinge of
// their difference, and — for the in-triplet hard-negative mining — four full-tensor mean squared differences.  SURVEY 8(f) N2 asks
// for exactly these reductions "fused into the upsample".  The bilinear map is linear and separable, so none of the enlarged tensors
// has to exist:
//   an enlarged row of a DIFFERENCE of two images is  u(X) = sum_j wx_j(X) dv_j,   dv = (1 - ly) D[i0] + ly D[i1]
//   (D = the low-resolution difference, i0 / i1 / ly = the two source rows of output row Y and their blend), hence
//        sum_X u(X)^2            =  dv' G dv                                   G_jj' = sum_X wx_j(X) wx_j'(X)   (tridiagonal)
//        sum_X (u(X) + eps)^2    =  dv' G dv + 2 eps s'dv + W eps^2            s_j   = sum_X wx_j(X)
//   and the adjoint of a row's distance w.r.t. dv is  (G dv + eps s) / distance, carried to the two source rows by (1 - ly, ly).
// The BatchNorm of the adaptation layer is an affine map per channel (y = sc z + sh, dfnet_bn.hip): the shift cancels in every
// difference, the scale multiplies it — the kernels read the plain 5x5 output z the training forward keeps (blocked [image][h][w][128],
// channel = stored position: the loss is a mean over rows, so the channel permutation is immaterial) and produce d L / d y at LOW
// resolution, which is what launch_bn_backward takes.  Per training step of 8 + 4 frames at 240x320 this removes the two
// [3,B,128,240,320] stacks and their gradients (1.9 GB written and read back), the upsample and its adjoint.
//   forward   one workgroup per (level, image pair b, output row Y): four waves = four column segments, a lane = two channels;
//             row statistics (six distances per row, as dfnet_loss.hip) + fp64 partials of the four mining and four hinge sums
//   backward  one workgroup per (level, image pair b, SOURCE row i), gather form (deterministic): every output row that blends
//             source row i contributes through its per-row coefficients (LDS); only the three source rows i-1, i, i+1 are read
// The level-0 maps have the output's size (identity resize): G = I, one source row, the plain element-wise form.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfn_common.h"
#include "dfnet_kernels.h"

namespace dfn {
namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool px_is_f1(int c) { return c == 0 || c == 2; }   // roles per mining case, as dfnet_loss.hip
__device__ __forceinline__ bool pz_is_f1(int c) { return c == 1 || c == 2; }

// Horizontal Gram tables of the align_corners resize w -> UW, by gather (deterministic): Gd[j] = G_jj, Gu[j] = G_j,j+1, sv[j] = s_j.
// The coordinate arithmetic is upsample_kernel's (dfnet_conv.hip), so the weights are the forward's to the bit.
__device__ void pyr_tables(int w, int UW, float* Gd, float* Gu, float* sv, int tid, int nthreads) {
  const float sx = UW > 1 ? float(w - 1) / float(UW - 1) : 0.f;
  for (int j = tid; j < w; j += nthreads) {
    int X0 = 0, X1 = UW - 1;
    if (sx > 0.f) { X0 = max(0, int(floorf(float(j - 1) / sx)) - 1); X1 = min(UW - 1, int(ceilf(float(j + 2) / sx)) + 1); }
    double gd = 0.0, gu = 0.0, s = 0.0;
    for (int X = X0; X <= X1; ++X) {
      const float fx = sx * float(X);
      const int xA = int(fx), xB = xA + (xA < w - 1 ? 1 : 0);
      const float lx = fx - float(xA), w0 = 1.f - lx;
      const float wj = (xA == j ? w0 : 0.f) + (xB == j ? lx : 0.f);
      const float wj1 = (xA == j + 1 ? w0 : 0.f) + (xB == j + 1 ? lx : 0.f);
      gd += (double)wj * wj; gu += (double)wj * wj1; s += (double)wj;
    }
    Gd[j] = (float)gd; Gu[j] = (float)gu; sv[j] = (float)s;
  }
}

struct PyrArgs {
  const float* z;       // [2 hb][h][w][128] fp32, blocked: the plain 5x5 output of the level
  const float* bn;      // BatchNorm work block: sc at kBnSc (stored-position order)
  int h, w, UH, UW;
  int hb;               // images per stream
  int f1_half;          // which half of the batch is f1 (the anchor stack of misc.py): 0 = images [0, hb), 1 = [hb, 2 hb)
  int level, L;
};

__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
}  // namespace

// ------------------------------------------------------------------------------------------ forward: row statistics + partial sums
// row_stat[row][6] = d(f1,f2), d(f2,f1), d(f1, roll f2), d(f2, roll f1), d(f1, roll f1), d(f2, roll f2)   (dfnet_loss.hip's columns),
// row = ((level * hb + b) * UH + Y) * 128 + stored channel.  part[part0 + block][8] (fp64) = four mining sums, four hinge sums.
__global__ __launch_bounds__(256) void triplet_pyr_rows_kernel(PyrArgs a, float margin, float eps, float* __restrict__ row_stat,
                                                               double* __restrict__ part, int part0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gd = sm;                       // [w]
  float* Gu = Gd + a.w;                 // [w]
  float* sv = Gu + a.w;                 // [w]
  float* red = sv + a.w;                // [4 segments][10][128]
  __shared__ double dred[2][8];
  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  pyr_tables(a.w, a.UW, Gd, Gu, sv, tid, 256);
  __syncthreads();
  const float sy = a.UH > 1 ? float(a.h - 1) / float(a.UH - 1) : 0.f;
  const size_t img = (size_t)a.h * a.w * 128, rowf = (size_t)a.w * 128;
  const int j_lo = seg * a.w / 4, j_hi = (seg + 1) * a.w / 4;
  const int j_end = j_hi < a.w - 1 ? j_hi : a.w - 1;       // the cross term (j_hi - 1, j_hi) belongs to this segment
  const float ceps = float(a.UW) * eps * eps;
  double mse[4] = {0.0, 0.0, 0.0, 0.0}, hin[4] = {0.0, 0.0, 0.0, 0.0};
  const int units = a.hb * a.UH;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / a.UH, Y = u - b * a.UH, bm = b == 0 ? a.hb - 1 : b - 1;
    const float fy = sy * float(Y);
    const int yA = int(fy), yB = yA + (yA < a.h - 1 ? 1 : 0);
    const float ly = fy - float(yA), wy0 = 1.f - ly;
    const bool two = ly != 0.f;
    const int o1 = a.f1_half * a.hb, o2 = (1 - a.f1_half) * a.hb;
    const float* pA = a.z + (size_t)(o1 + b) * img + 2 * lane;        // f1[b]
    const float* pP = a.z + (size_t)(o2 + b) * img + 2 * lane;        // f2[b]
    const float* pAn = a.z + (size_t)(o1 + bm) * img + 2 * lane;      // roll f1
    const float* pPn = a.z + (size_t)(o2 + bm) * img + 2 * lane;      // roll f2
    const size_t rA = (size_t)yA * rowf, rB = (size_t)yB * rowf;
    f32x2 Q[5], Lin[5], prev[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { Q[k] = f32x2{0.f, 0.f}; Lin[k] = f32x2{0.f, 0.f}; prev[k] = f32x2{0.f, 0.f}; }
#pragma unroll 2
    for (int j = j_lo; j <= j_end; ++j) {
      const size_t c = (size_t)j * 128;
      f32x2 vA = ld2(pA + rA + c), vP = ld2(pP + rA + c), vAn = ld2(pAn + rA + c), vPn = ld2(pPn + rA + c);
      if (two) {
        const f32x2 wA = ld2(pA + rB + c), wP = ld2(pP + rB + c), wAn = ld2(pAn + rB + c), wPn = ld2(pPn + rB + c);
        vA = wy0 * vA + ly * wA; vP = wy0 * vP + ly * wP; vAn = wy0 * vAn + ly * wAn; vPn = wy0 * vPn + ly * wPn;
      }
      const f32x2 d[5] = {vA - vP, vA - vPn, vP - vAn, vA - vAn, vP - vPn};
      const float gd = j < j_hi ? Gd[j] : 0.f, sj = j < j_hi ? sv[j] : 0.f;
      const float gu2 = j > j_lo ? 2.f * Gu[j - 1] : 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        Q[k] += gd * d[k] * d[k] + gu2 * prev[k] * d[k];
        Lin[k] += sj * d[k];
        prev[k] = d[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      *reinterpret_cast<f32x2*>(red + ((seg * 10 + k) * 128 + 2 * lane)) = Q[k];
      *reinterpret_cast<f32x2*>(red + ((seg * 10 + 5 + k) * 128 + 2 * lane)) = Lin[k];
    }
    __syncthreads();
    if (tid < 128) {
      const float sc = a.bn[kBnSc + tid], sc2 = sc * sc;
      float q[5], ln[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        q[k] = (((red[(0 * 10 + k) * 128 + tid] + red[(1 * 10 + k) * 128 + tid]) + red[(2 * 10 + k) * 128 + tid]) + red[(3 * 10 + k) * 128 + tid]) * sc2;
        ln[k] = (((red[(0 * 10 + 5 + k) * 128 + tid] + red[(1 * 10 + 5 + k) * 128 + tid]) + red[(2 * 10 + 5 + k) * 128 + tid]) +
                 red[(3 * 10 + 5 + k) * 128 + tid]) * sc;
        q[k] = fmaxf(q[k], 0.f);
      }
      const float e2 = 2.f * eps;
      const float dxy[2] = {sqrtf(fmaxf(q[0] + e2 * ln[0] + ceps, 0.f)), sqrtf(fmaxf(q[0] - e2 * ln[0] + ceps, 0.f))};
      float dxz[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) dxz[k] = sqrtf(fmaxf(q[1 + k] + e2 * ln[1 + k] + ceps, 0.f));
      float* st = row_stat + 6 * ((((size_t)a.level * a.hb + b) * a.UH + Y) * 128 + tid);
      st[0] = dxy[0]; st[1] = dxy[1]; st[2] = dxz[0]; st[3] = dxz[1]; st[4] = dxz[2]; st[5] = dxz[3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mse[k] += (double)q[1 + k];
        const float hinge = dxy[px_is_f1(k) ? 0 : 1] - dxz[k] + margin;
        if (hinge > 0.f) hin[k] += (double)hinge;
      }
    }
    __syncthreads();
  }
  if (tid < 128) {   // fixed-order block sums of the eight fp64 accumulators (waves 0 and 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = mse[k], g = hin[k];
      for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); g += __shfl_down(g, o, 64); }
      if (lane == 0) { dred[seg][k] = v; dred[seg][4 + k] = g; }
    }
  }
  __syncthreads();
  if (tid < 8) part[(size_t)(part0 + blockIdx.x) * 8 + tid] = dred[0][tid] + dred[1][tid];
}

// ------------------------------------------------------------------------------------------ backward: d L / d y at low resolution
// gout [2 hb][h][w][128] (blocked, as z): rows of the two images of pair b (its anchor-stack image and its other-stack image) at
// source row i.  coefficient of output row Y and channel p (LDS):  a_xy = scale / d_xy, a_xz = scale / d_xz (0 when the row's hinge
// is inactive or the distance is 0), a_zp = scale / d_xz of image b+1's row (this image as its negative).
template <bool IDENT>
__global__ __launch_bounds__(256) void triplet_pyr_backward_kernel(PyrArgs a, float eps, const int* __restrict__ case_in,
                                                                   const float* __restrict__ row_stat, const float* __restrict__ margin_in,
                                                                   const float* __restrict__ grad_loss, double n_rows, int y_cap,
                                                                   float* __restrict__ gout) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gd = sm;                       // [w]
  float* Gu = Gd + a.w;                 // [w]
  float* sv = Gu + a.w;                 // [w]
  float* yw = sv + a.w;                 // [y_cap][4]: wy0, ly, weight of source row i, (ra | rb << 2) as a float-encoded int
  float* coef = yw + 4 * y_cap;         // [y_cap][3][128]
  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int cse = *case_in;
  const float margin = *margin_in;
  const float scale = (float)((double)*grad_loss / n_rows);
  const bool xf1 = px_is_f1(cse), zf1 = pz_is_f1(cse), z_is_x = xf1 == zf1;
  const int ixy = xf1 ? 0 : 1, ixz = 2 + cse;
  const int b = blockIdx.x / a.h, i = blockIdx.x - b * a.h;
  const int bm = b == 0 ? a.hb - 1 : b - 1, bp = b == a.hb - 1 ? 0 : b + 1;
  const int oX = (xf1 ? a.f1_half : 1 - a.f1_half) * a.hb, oY = (xf1 ? 1 - a.f1_half : a.f1_half) * a.hb, oZ = zf1 == xf1 ? oX : oY;
  pyr_tables(a.w, a.UW, Gd, Gu, sv, tid, 256);
  // output rows that blend source row i
  const float sy = a.UH > 1 ? float(a.h - 1) / float(a.UH - 1) : 0.f;
  int Y0 = 0, Y1 = a.UH - 1;
  if (IDENT) { Y0 = Y1 = i; }
  else if (sy > 0.f) { Y0 = max(0, int(floorf(float(i - 1) / sy)) - 1); Y1 = min(a.UH - 1, int(ceilf(float(i + 1) / sy)) + 1); }
  auto wy_of = [&](int Y, float& wy0, float& ly, int& yA, int& yB) {
    const float fy = sy * float(Y);
    yA = int(fy); yB = yA + (yA < a.h - 1 ? 1 : 0);
    ly = fy - float(yA); wy0 = 1.f - ly;
    return (yA == i ? wy0 : 0.f) + (yB == i ? ly : 0.f);
  };
  {   // trim rows of zero weight (uniform)
    float w0, l0; int ya, yb;
    while (Y0 < Y1 && wy_of(Y0, w0, l0, ya, yb) == 0.f) ++Y0;
    while (Y1 > Y0 && wy_of(Y1, w0, l0, ya, yb) == 0.f) --Y1;
  }
  const int ny_all = Y1 - Y0 + 1;
  const size_t img = (size_t)a.h * a.w * 128, rowf = (size_t)a.w * 128;
  const f32x2 sc = ld2(a.bn + kBnSc + 2 * lane);
  const float* pX = a.z + (size_t)(oX + b) * img + 2 * lane;
  const float* pY = a.z + (size_t)(oY + b) * img + 2 * lane;
  const float* pZm = a.z + (size_t)(oZ + bm) * img + 2 * lane;
  const float* pXp = a.z + (size_t)(oX + bp) * img + 2 * lane;
  float* gX = gout + ((size_t)(oX + b) * a.h + i) * rowf + 2 * lane;
  float* gY = gout + ((size_t)(oY + b) * a.h + i) * rowf + 2 * lane;
  // the output rows in chunks of at most y_cap (the LDS coefficient table): the first chunk writes the gradient rows, later ones add
  // to what the same thread wrote (steep enlargements only: 240 x 320 from 15 x 20 is one chunk of 33 rows)
  for (int yc = 0; yc < ny_all; yc += y_cap) {
  const int ny = min(ny_all - yc, y_cap);
  const int Yb = Y0 + yc;
  __syncthreads();
  for (int t = tid; t < ny; t += 256) {
    float w0, l0; int ya, yb;
    const float wi = wy_of(Yb + t, w0, l0, ya, yb);
    yw[4 * t] = w0; yw[4 * t + 1] = l0; yw[4 * t + 2] = wi;
    yw[4 * t + 3] = __int_as_float((ya - (i - 1)) | ((yb - (i - 1)) << 2));
  }
  for (int t = tid; t < ny * 128; t += 256) {
    const int yy = t >> 7, p = t & 127, Y = Yb + yy;
    const float* st = row_stat + 6 * ((((size_t)a.level * a.hb + b) * a.UH + Y) * 128 + p);
    const float* sp = row_stat + 6 * ((((size_t)a.level * a.hb + bp) * a.UH + Y) * 128 + p);
    const float dxy = st[ixy], dxz = st[ixz];
    const bool act = dxy - dxz + margin >= 0.f;            // clamp_min passes the gradient at hinge == 0, as torch
    const float dxyp = sp[ixy], dxzp = sp[ixz];
    const bool actp = dxyp - dxzp + margin >= 0.f;
    coef[(yy * 3 + 0) * 128 + p] = act && dxy > 0.f ? scale / dxy : 0.f;
    coef[(yy * 3 + 1) * 128 + p] = act && dxz > 0.f ? scale / dxz : 0.f;
    coef[(yy * 3 + 2) * 128 + p] = actp && dxzp > 0.f ? scale / dxzp : 0.f;
  }
  __syncthreads();
  constexpr int NR = IDENT ? 1 : 3;
  size_t roff[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rr = IDENT ? i : min(max(i - 1 + r, 0), a.h - 1);
    roff[r] = (size_t)rr * rowf;
  }
  const int j_lo = seg * a.w / 4, j_hi = (seg + 1) * a.w / 4;
  // low-resolution differences of the three source rows at columns j - 1, j, j + 1 (sliding window)
  f32x2 Dxy[NR][3], Dxz[NR][3], Dzp[NR][3];
  auto load_col = [&](int j, int slot) {
    const bool ok = j >= 0 && j < a.w;
    const size_t c = (size_t)(ok ? j : 0) * 128;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const f32x2 x = ld2(pX + roff[r] + c), y = ld2(pY + roff[r] + c), zm = ld2(pZm + roff[r] + c), xp = ld2(pXp + roff[r] + c);
      const f32x2 zs = z_is_x ? x : y;
      const float m = ok ? 1.f : 0.f;
      Dxy[r][slot] = m * (x - y); Dxz[r][slot] = m * (x - zm); Dzp[r][slot] = m * (xp - zs);
    }
  };
  load_col(j_lo - 1, 0);
  load_col(j_lo, 1);
  for (int j = j_lo; j < j_hi; ++j) {
    load_col(j + 1, 2);
    const float gl = j > 0 ? Gu[j - 1] : 0.f, gd = Gd[j], gr = j + 1 < a.w ? Gu[j] : 0.f, es = eps * sv[j];
    f32x2 Txy[NR], Txz[NR], Tzp[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      Txy[r] = sc * (gl * Dxy[r][0] + gd * Dxy[r][1] + gr * Dxy[r][2]);
      Txz[r] = sc * (gl * Dxz[r][0] + gd * Dxz[r][1] + gr * Dxz[r][2]);
      Tzp[r] = sc * (gl * Dzp[r][0] + gd * Dzp[r][1] + gr * Dzp[r][2]);
    }
    f32x2 ax{0.f, 0.f}, ay{0.f, 0.f};
    for (int yy = 0; yy < ny; ++yy) {
      const float wy0 = yw[4 * yy], ly = yw[4 * yy + 1], wi = yw[4 * yy + 2];
      const int rab = __float_as_int(yw[4 * yy + 3]);
      f32x2 txy, txz, tzp;
      if (IDENT) { txy = Txy[0]; txz = Txz[0]; tzp = Tzp[0]; }
      else {
        const int ra = rab & 3, rb = rab >> 2;
        // (three source rows: select without dynamic register indexing)
        const f32x2 axy = ra == 0 ? Txy[0] : (ra == 1 ? Txy[1 % NR] : Txy[2 % NR]), bxy = rb == 0 ? Txy[0] : (rb == 1 ? Txy[1 % NR] : Txy[2 % NR]);
        const f32x2 axz = ra == 0 ? Txz[0] : (ra == 1 ? Txz[1 % NR] : Txz[2 % NR]), bxz = rb == 0 ? Txz[0] : (rb == 1 ? Txz[1 % NR] : Txz[2 % NR]);
        const f32x2 azp = ra == 0 ? Tzp[0] : (ra == 1 ? Tzp[1 % NR] : Tzp[2 % NR]), bzp = rb == 0 ? Tzp[0] : (rb == 1 ? Tzp[1 % NR] : Tzp[2 % NR]);
        txy = wy0 * axy + ly * bxy; txz = wy0 * axz + ly * bxz; tzp = wy0 * azp + ly * bzp;
      }
      txy += es; txz += es; tzp += es;
      const f32x2 cxy = ld2(coef + (yy * 3 + 0) * 128 + 2 * lane), cxz = ld2(coef + (yy * 3 + 1) * 128 + 2 * lane),
                  czp = ld2(coef + (yy * 3 + 2) * 128 + 2 * lane);
      const f32x2 gxy = cxy * txy, gz = czp * tzp;
      f32x2 gx = gxy - cxz * txz, gy = -gxy;
      if (z_is_x) gx += gz; else gy += gz;
      ax += wi * gx; ay += wi * gy;
    }
    if (yc > 0) { ax += ld2(gX + (size_t)j * 128); ay += ld2(gY + (size_t)j * 128); }
    *reinterpret_cast<f32x2*>(gX + (size_t)j * 128) = ax;
    *reinterpret_cast<f32x2*>(gY + (size_t)j * 128) = ay;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      Dxy[r][0] = Dxy[r][1]; Dxy[r][1] = Dxy[r][2];
      Dxz[r][0] = Dxz[r][1]; Dxz[r][1] = Dxz[r][2];
      Dzp[r][0] = Dzp[r][1]; Dzp[r][1] = Dzp[r][2];
    }
  }
  }   // chunks of output rows
}

// rows of an upsample h -> UH that can blend one source row (bound used for the LDS coefficient table)
static int pyr_y_cap(int h, int UH) {
  if (h == UH) return 1;
  if (h <= 1) return UH;
  const int per = (UH - 1 + h - 2) / (h - 1);   // ceil((UH - 1) / (h - 1)): output rows per source interval
  int cap = 2 * per + 3;
  if (cap > UH) cap = UH;
  return cap < 36 ? cap : 36;      // 36 rows x (3 x 128 coefficients + 4) floats = 56 KB of LDS; more rows run in chunks
}

int triplet_pyr_blocks(int hb, int UH) {   // workgroups of one level's forward launch (<= a third of the fp64 partial slots)
  const int units = hb * UH, cap = int(kTripletPartDoubles / 8 / 3);
  return units < cap ? units : cap;
}

hipError_t launch_triplet_pyr_forward(const float* z, const float* bn, int h, int w, int UH, int UW, int hb, int f1_half, int level, int L,
                                      float margin, float eps, float* row_stat, double* part, int part0, hipStream_t s) {
  PyrArgs a{z, bn, h, w, UH, UW, hb, f1_half, level, L};
  const size_t lds = (size_t)(3 * w + 4 * 10 * 128) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(triplet_pyr_rows_kernel, dim3(triplet_pyr_blocks(hb, UH)), dim3(256), lds, s, a, margin, eps, row_stat, part, part0);
  return hipGetLastError();
}

hipError_t launch_triplet_pyr_backward(const float* z, const float* bn, int h, int w, int UH, int UW, int hb, int f1_half, int level, int L,
                                       float eps, const int* case_in, const float* row_stat, const float* margin_in, const float* grad_loss,
                                       float* gout, hipStream_t s) {
  PyrArgs a{z, bn, h, w, UH, UW, hb, f1_half, level, L};
  const int y_cap = pyr_y_cap(h, UH);
  const size_t lds = (size_t)(3 * w + 4 * y_cap + 3 * 128 * y_cap) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const double n_rows = (double)L * hb * 128 * UH;
  if (h == UH && w == UW)
    hipLaunchKernelGGL(triplet_pyr_backward_kernel<true>, dim3(hb * h), dim3(256), lds, s, a, eps, case_in, row_stat, margin_in, grad_loss, n_rows,
                       y_cap, gout);
  else
    hipLaunchKernelGGL(triplet_pyr_backward_kernel<false>, dim3(hb * h), dim3(256), lds, s, a, eps, case_in, row_stat, margin_in, grad_loss,
                       n_rows, y_cap, gout);
  return hipGetLastError();
}

}  // namespace dfn
