// pose_polar.hip — the pose regressor's rotation re-orthogonalisation and its adjoint, closed form on the device (gfx950).
//
// /root/reference/script/feature/direct_feature_matching.py:85-92 (and feature/misc.py:68-72 in the evaluation loop):
//     R = pose[:, :3, :3];  u, s, v = torch.svd(R);  pose[:, :3, :3] = u @ v^T
// U V^T of the SVD M = U S V^T is the ORTHOGONAL POLAR FACTOR Q of M (M = Q P, P = (M^T M)^(1/2) symmetric positive definite;
// det Q = sign det M as for U V^T), so no SVD is needed: scaled Newton iteration Q <- (c Q + Q^-T / c) / 2 with Higham's
// c = sqrt(|Q^-T|_F / |Q|_F) converges quadratically from Q0 = M (a 3x3 adjugate per iteration), in fp64 per matrix.
// Adjoint (what autograd of torch.svd + matmul computes): with A = Q^T dM the first-order change is dQ = Q [w]x where
// (tr(P) I - P) w = axial(A - A^T); hence for G = d L / d Q
//     d L / d M = Q [u]x,   (tr(P) I - P) u = a(Q^T G),   a(B) = (B32 - B23, B13 - B31, B21 - B12),   P = Q^T M.
// One thread per pose: the DFNet_dm step has 4-8 of them; this replaces rocSOLVER's batched Jacobi SVD, two rocBLAS / hipBLASLt
// GEMMs and the slice / cat copies around them (~20 launches per step) by one launch per direction.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"

namespace dfn {
namespace {
struct M3 { double m[3][3]; };

__device__ inline M3 cofactor_T_over_det(const M3& a, double* det_out) {   // a^-T = cofactor(a) / det(a)
  M3 c;
  c.m[0][0] = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  c.m[0][1] = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  c.m[0][2] = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  c.m[1][0] = a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2];
  c.m[1][1] = a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0];
  c.m[1][2] = a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1];
  c.m[2][0] = a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1];
  c.m[2][1] = a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2];
  c.m[2][2] = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
  const double det = a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[0][1] + a.m[0][2] * c.m[0][2];
  *det_out = det;
  const double inv = 1.0 / det;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.m[i][j] *= inv;
  return c;
}
__device__ inline double fro(const M3& a) {
  double s = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) s += a.m[i][j] * a.m[i][j];
  return sqrt(s);
}
// orthogonal polar factor of a (returns false for a singular matrix: the factor is not unique there, as U V^T is not)
__device__ inline bool polar(const M3& a, M3* q_out) {
  M3 q = a;
  for (int it = 0; it < 40; ++it) {
    double det;
    const M3 y = cofactor_T_over_det(q, &det);
    if (!(fabs(det) > 1e-300) || !isfinite(det)) return false;
    const double c = sqrt(fro(y) / fro(q));
    M3 n;
    double diff = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        n.m[i][j] = 0.5 * (c * q.m[i][j] + y.m[i][j] / c);
        const double d = n.m[i][j] - q.m[i][j];
        diff += d * d;
      }
    q = n;
    if (diff < 1e-30) break;
  }
  *q_out = q;
  return true;
}

// A rank-deficient (or numerically singular) block: the Newton iteration has no inverse to take, while torch.svd — what the reference
// calls — still returns a finite U V^T (an orthogonal factor with M = Q P, P symmetric positive SEMI-definite; unique only up to the
// orientation of the null directions).  The same here, by the route the SVD takes: V and the singular values from a cyclic Jacobi
// eigen-decomposition of M^T M, u_i = M v_i / s_i for the non-zero s_i, the remaining u_i an orthonormal completion (rank 2: the
// cross product; rank 1: any frame around u_1; rank 0: U = V).  A zero-initialised or collapsed fc_pose output therefore gives a
// finite pose, as in the reference, instead of poisoning the DFNet_dm loss with NaN.  Non-finite input stays non-finite.
__device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline bool polar_rank_deficient(const M3& a, M3* q_out) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) if (!isfinite(a.m[i][j])) return false;
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a.m[k][i] * a.m[k][j]; A[i][j] = s; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; ++k) { const double x = A[k][p], y = A[k][q]; A[k][p] = c * x - sn * y; A[k][q] = sn * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = A[p][k], y = A[q][k]; A[p][k] = c * x - sn * y; A[q][k] = sn * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = V[k][p], y = V[k][q]; V[k][p] = c * x - sn * y; V[k][q] = sn * x + c * y; }
      }
  }
  int ord[3] = {0, 1, 2};   // eigenvalues in descending order
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j) if (A[ord[j]][ord[j]] > A[ord[i]][ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double v[3][3], u[3][3], sv[3];
  for (int i = 0; i < 3; ++i) { sv[i] = sqrt(fmax(A[ord[i]][ord[i]], 0.0)); for (int k = 0; k < 3; ++k) v[i][k] = V[k][ord[i]]; }
  const double tol = 1e-12 * sv[0];
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (sv[i] > tol && sv[0] > 0) {
      for (int k = 0; k < 3; ++k) u[i][k] = (a.m[k][0] * v[i][0] + a.m[k][1] * v[i][1] + a.m[k][2] * v[i][2]) / sv[i];
      double n = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
      for (int k = 0; k < 3; ++k) u[i][k] /= n;
      rank = i + 1;
    } else break;
  if (rank == 0) { for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) u[i][k] = v[i][k]; }
  else if (rank == 1) {
    double e[3] = {0, 0, 0};
    int small = fabs(u[0][0]) <= fabs(u[0][1]) ? (fabs(u[0][0]) <= fabs(u[0][2]) ? 0 : 2) : (fabs(u[0][1]) <= fabs(u[0][2]) ? 1 : 2);
    e[small] = 1.0;
    cross3(u[0], e, u[1]);
    const double n = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    for (int k = 0; k < 3; ++k) u[1][k] /= n;
    cross3(u[0], u[1], u[2]);
  } else if (rank == 2) cross3(u[0], u[1], u[2]);
  if (rank < 3) {   // keep the completed frame's handedness that of V's, so that det Q = +1 on the completed directions
    double vx[3];
    cross3(v[0], v[1], vx);
    if (vx[0] * v[2][0] + vx[1] * v[2][1] + vx[2] * v[2][2] < 0) for (int k = 0; k < 3; ++k) u[2][k] = -u[2][k];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) q_out->m[i][j] = u[0][i] * v[0][j] + u[1][i] * v[1][j] + u[2][i] * v[2][j];
  return true;
}
__device__ inline bool polar_any(const M3& a, M3* q_out) { return polar(a, q_out) || polar_rank_deficient(a, q_out); }

// pose_in / pose_out [B][3][4]: rotation block <- its orthogonal polar factor, translation column copied
__global__ void pose_polar_forward_kernel(const float* __restrict__ pin, int B, float* __restrict__ pout, int* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  M3 a, q;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.m[i][j] = pin[b * 12 + i * 4 + j];
  if (!polar_any(a, &q)) {      // only non-finite input is left here
    if (status) atomicOr(status, 1);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) q.m[i][j] = nan("");
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pout[b * 12 + i * 4 + j] = (float)q.m[i][j];
    pout[b * 12 + i * 4 + 3] = pin[b * 12 + i * 4 + 3];
  }
}
// grad_in [B][3][4] = adjoint of the forward at pose_in, given grad_out [B][3][4] (the translation column passes through)
__global__ void pose_polar_backward_kernel(const float* __restrict__ pin, const float* __restrict__ gout, int B, float* __restrict__ gin) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  M3 a, q, g;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { a.m[i][j] = pin[b * 12 + i * 4 + j]; g.m[i][j] = gout[b * 12 + i * 4 + j]; }
  bool ok = polar_any(a, &q);   // recomputed in fp64 (cheaper than carrying a double-precision tape for 9 numbers)
  double P[3][3], QtG[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0, t = 0;
      for (int k = 0; k < 3; ++k) { s += q.m[k][i] * a.m[k][j]; t += q.m[k][i] * g.m[k][j]; }
      P[i][j] = s; QtG[i][j] = t;
    }
  const double tr = P[0][0] + P[1][1] + P[2][2];
  M3 K;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K.m[i][j] = (i == j ? tr : 0.0) - 0.5 * (P[i][j] + P[j][i]);
  const double rhs[3] = {QtG[2][1] - QtG[1][2], QtG[0][2] - QtG[2][0], QtG[1][0] - QtG[0][1]};
  double detK;
  const M3 KiT = cofactor_T_over_det(K, &detK);   // K symmetric: K^-1 = K^-T
  // K's eigenvalues are the pairwise sums s_i + s_j of M's singular values: singular only at rank <= 1, where the factor is not
  // differentiable and torch.svd's own backward divides by zero as well
  ok = ok && fabs(detK) > 1e-300 && isfinite(detK);
  double u[3];
  for (int i = 0; i < 3; ++i) u[i] = KiT.m[i][0] * rhs[0] + KiT.m[i][1] * rhs[1] + KiT.m[i][2] * rhs[2];
  const double ux[3][3] = {{0, -u[2], u[1]}, {u[2], 0, -u[0]}, {-u[1], u[0], 0}};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += q.m[i][k] * ux[k][j];
      gin[b * 12 + i * 4 + j] = ok ? (float)s : nanf("");
    }
    gin[b * 12 + i * 4 + 3] = gout[b * 12 + i * 4 + 3];
  }
}
}  // namespace
}  // namespace dfn

extern "C" int dfn_pose_orthogonalize(const float* pose_in, int B, float* pose_out, void* stream) {
  if (!pose_in || !pose_out || B < 0) return dfn::set_error(DFN_ERR_ARG, "dfn_pose_orthogonalize: bad argument");
  if (!B) return DFN_OK;
  hipLaunchKernelGGL(dfn::pose_polar_forward_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pose_in, B,
                     pose_out, static_cast<int*>(nullptr));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? DFN_OK : dfn::set_error(DFN_ERR_HIP, "dfn_pose_orthogonalize: %s", hipGetErrorString(e));
}

extern "C" int dfn_pose_orthogonalize_backward(const float* pose_in, const float* grad_out, int B, float* grad_in, void* stream) {
  if (!pose_in || !grad_out || !grad_in || B < 0) return dfn::set_error(DFN_ERR_ARG, "dfn_pose_orthogonalize_backward: bad argument");
  if (!B) return DFN_OK;
  hipLaunchKernelGGL(dfn::pose_polar_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pose_in,
                     grad_out, B, grad_in);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? DFN_OK : dfn::set_error(DFN_ERR_HIP, "dfn_pose_orthogonalize_backward: %s", hipGetErrorString(e));
}
