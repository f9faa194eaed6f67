// nerfh_train_api.hip — C ABI of the NeRF-H TRAINING path and of the generic-width render path (include/dfnet_hip.h):
// the layer-by-layer orchestration of nerfh_train.hip's fp32-MFMA products over both networks.
//
// Replaces (reference, /root/reference/script/): run_nerf.py:50-66 (render(**render_kwargs_train), loss.backward()),
// models/rendering.py:245-337 (render_rays, test_time=False), models/nerfw.py:47-95,297-354, models/losses.py:19-57.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"
#include "nerfh_fused_train.h"
#include "nerfh_handle.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"
#include "nerfh_train.h"

using namespace dfn;
using namespace dfn::train;

#define HS(s) reinterpret_cast<hipStream_t>(s)
#define CHECK_HIP(expr, what)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return set_error(DFN_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------------------------------------ canonical parameter order
namespace {
enum LayerIx { L1 = 0, L8 = 7, FIN = 8, DIR = 9, SIG = 10, RGB = 11, TE0 = 12, TE3 = 15, TSIG = 16, TRGB = 17, TBETA = 18 };
constexpr int kCoarseLayers_ = 12, kFineLayers_ = 19;
constexpr int kCoarseParams = 2 * kCoarseLayers_, kFineParams = 2 * kFineLayers_;
constexpr int kParamCount = kCoarseParams + kFineParams + 2;

const char* layer_key(int l) {
  static const char* k[19] = {"xyz_encoding_1.0", "xyz_encoding_2.0", "xyz_encoding_3.0", "xyz_encoding_4.0", "xyz_encoding_5.0",
                              "xyz_encoding_6.0", "xyz_encoding_7.0", "xyz_encoding_8.0", "xyz_encoding_final", "dir_encoding.0",
                              "static_sigma.0", "static_rgb.0", "transient_encoding.0", "transient_encoding.2", "transient_encoding.4",
                              "transient_encoding.6", "transient_sigma.0", "transient_rgb.0", "transient_beta.0"};
  return k[l];
}
std::vector<std::string> make_names() {
  std::vector<std::string> v;
  for (int f = 0; f < 2; ++f)
    for (int l = 0; l < (f ? kFineLayers_ : kCoarseLayers_); ++l) {
      v.push_back(std::string(f ? "fine." : "coarse.") + layer_key(l) + ".weight");
      v.push_back(std::string(f ? "fine." : "coarse.") + layer_key(l) + ".bias");
    }
  v.push_back("embedding_a.weight");
  v.push_back("embedding_t.weight");
  return v;
}
const std::vector<std::string>& names() {
  static const std::vector<std::string> v = make_names();
  return v;
}

struct Net {          // device pointers of one network's layers
  const float* w[19];
  const float* b[19];
  float* gw[19];      // gradients (backward only)
  float* gb[19];
};
Net net_of(const float* const* params, float* const* grads, bool fine) {
  Net n{};
  const int base = fine ? kCoarseParams : 0;
  for (int l = 0; l < (fine ? kFineLayers_ : kCoarseLayers_); ++l) {
    n.w[l] = params[base + 2 * l];
    n.b[l] = params[base + 2 * l + 1];
    if (grads) { n.gw[l] = grads[base + 2 * l]; n.gb[l] = grads[base + 2 * l + 1]; }
  }
  return n;
}

inline size_t al64(size_t floats) { return (floats + 63) & ~size_t(63); }
inline int r4(int v) { return (v + 3) & ~3; }

struct Dims {
  int W, W2, na, nt, kd_c, kd_f, ld_dc, ld_df, ld_t, hist_bin, dim_a, dim_t, n_vocab;
};
Dims dims_of(const dfn_nerfh_desc& d) {
  Dims m{};
  m.W = d.width; m.W2 = d.width / 2;
  m.na = d.hist_bin * d.dim_a; m.nt = d.hist_bin * d.dim_t;
  m.kd_c = kChDir; m.kd_f = kChDir + m.na;
  m.ld_dc = r4(m.kd_c); m.ld_df = r4(m.kd_f); m.ld_t = r4(m.nt);
  m.hist_bin = d.hist_bin; m.dim_a = d.dim_a; m.dim_t = d.dim_t; m.n_vocab = d.n_vocab;
  return m;
}

// Workspace of one training step (floats).  Everything the backward needs stays where the forward left it.
struct TrainWs {
  float *view, *dir_c, *dir_f, *t_in;                  // per ray
  float *z_c, *pe_c, *h_c[8], *fin_c, *dirh_c, *raw_c;  // coarse, per point
  float *z_f, *pe_f, *h_f[8], *fin_f, *dirh_f, *te[4];  // fine, per point
  float *gpre_f, *gpre_c, *gA, *gB, *gfin, *gt0, *gt1, *gsum, *gray, *wscratch;  // backward
  float *gA_c, *gB_c, *gfin_c, *gt0_c, *gsum_c, *gray_c, *wscratch_c;             // the coarse net's own scratch (it runs beside the fine one)
  size_t total;
};
TrainWs carve_train(float* base, const Dims& m, size_t R, int Nc, int Ni, bool with_backward) {
  TrainWs w{};
  const size_t Nf = size_t(Nc) + Ni, Pc = R * Nc, Pf = R * Nf;
  size_t off = 0;
  auto take = [&](size_t floats) { float* p = base ? base + off : nullptr; off += al64(floats); return p; };
  w.view = take(R * 3);
  w.dir_c = take(R * m.ld_dc);
  w.dir_f = take(R * m.ld_df);
  w.t_in = take(R * m.ld_t);
  w.z_c = take(Pc);
  w.pe_c = take(Pc * 64);
  for (auto& p : w.h_c) p = take(Pc * m.W);
  w.fin_c = take(Pc * m.W);
  w.dirh_c = take(Pc * m.W2);
  w.raw_c = take(Pc * 4);
  w.z_f = take(Pf);
  w.pe_f = take(Pf * 64);
  for (auto& p : w.h_f) p = take(Pf * m.W);
  w.fin_f = take(Pf * m.W);
  w.dirh_f = take(Pf * m.W2);
  for (auto& p : w.te) p = take(Pf * m.W2);
  if (with_backward) {
    w.gpre_f = take(Pf * 9);
    w.gpre_c = take(Pc * 4);
    w.gA = take(Pf * m.W);
    w.gB = take(Pf * m.W);
    w.gfin = take(Pf * m.W);
    w.gt0 = take(Pf * m.W2);
    w.gt1 = take(Pf * m.W2);
    w.gsum = take(R * m.W2);
    w.gray = take(R * size_t(m.ld_df > m.ld_t ? m.ld_df : m.ld_t));
    // weight-gradient partials: the largest over the (outputs N, segment width K) pairs the backward really issues — for
    // netwidth < 64 the 63-wide encoding and the 27 + hist_bin * dim_a wide direction tail exceed W x W
    auto wscratch_floats = [&](long long P) {
      const int pairs[][2] = {{m.W, m.W}, {m.W, kChXyz + 1}, {m.W2, m.W}, {m.W2, m.ld_df}, {m.W2, m.ld_dc}, {m.W2, m.ld_t}, {m.W2, m.W2},
                              {4, m.W}, {4, m.W2}};
      size_t best = 0;
      for (const auto& nk : pairs) {
        const size_t f = gemm_wgrad_scratch_floats(nk[0], nk[1], P);
        best = f > best ? f : best;
      }
      return best + 1024;
    };
    w.wscratch = take(wscratch_floats((long long)Pf));
    w.gA_c = take(Pc * m.W);
    w.gB_c = take(Pc * m.W);
    w.gfin_c = take(Pc * m.W);
    w.gt0_c = take(Pc * m.W2);
    w.gsum_c = take(R * m.W2);
    w.gray_c = take(R * size_t(m.ld_dc));
    w.wscratch_c = take(wscratch_floats((long long)Pc));
  }
  w.total = off * sizeof(float);
  return w;
}

struct NetBufs {   // one network's activation buffers inside the workspace
  float *pe, *h[8], *fin, *dirh, *te[4], *raw, *dir_in, *t_in;
  int ld_dir, kd, raw_ld;
  long long P;
  int Ns;
};

// The network forward, layer by layer (models/nerfw.py:297-354).  sigma_only: trunk + static_sigma (coarse test-time query).
int net_forward(const Net& n, const Dims& m, const NetBufs& b, bool fine, bool sigma_only, hipStream_t s) {
  const int W = m.W, W2 = m.W2;
  Seg pe{b.pe, 64, kChXyz, 1, 0};
  for (int l = 0; l < 8; ++l) {
    if (l == 0) {
      CHECK_HIP(gemm_fwd(&pe, 1, n.w[l], kChXyz, n.b[l], W, ACT_RELU, b.h[l], W, b.P, s), "train forward: xyz_encoding_1");
    } else if (l == 4) {
      Seg sg[2] = {pe, {b.h[l - 1], W, W, 1, kChXyz}};   // torch.cat([input_xyz, xyz_], 1)
      CHECK_HIP(gemm_fwd(sg, 2, n.w[l], W + kChXyz, n.b[l], W, ACT_RELU, b.h[l], W, b.P, s), "train forward: xyz_encoding_5");
    } else {
      Seg sg{b.h[l - 1], W, W, 1, 0};
      CHECK_HIP(gemm_fwd(&sg, 1, n.w[l], W, n.b[l], W, ACT_RELU, b.h[l], W, b.P, s), "train forward: xyz_encoding");
    }
  }
  Seg h8{b.h[7], W, W, 1, 0};
  CHECK_HIP(gemm_fwd(&h8, 1, n.w[SIG], W, n.b[SIG], 1, ACT_SOFTPLUS, b.raw + (sigma_only ? 0 : 3), b.raw_ld, b.P, s),
            "train forward: static_sigma");
  if (sigma_only) return DFN_OK;
  CHECK_HIP(gemm_fwd(&h8, 1, n.w[FIN], W, n.b[FIN], W, ACT_NONE, b.fin, W, b.P, s), "train forward: xyz_encoding_final");
  {
    Seg sg[2] = {{b.fin, W, W, 1, 0}, {b.dir_in, b.ld_dir, b.kd, b.Ns, W}};
    CHECK_HIP(gemm_fwd(sg, 2, n.w[DIR], W + b.kd, n.b[DIR], W2, ACT_RELU, b.dirh, W2, b.P, s), "train forward: dir_encoding");
    Seg dh{b.dirh, W2, W2, 1, 0};
    CHECK_HIP(gemm_fwd(&dh, 1, n.w[RGB], W2, n.b[RGB], 3, ACT_SIGMOID, b.raw, b.raw_ld, b.P, s), "train forward: static_rgb");
  }
  if (!fine) return DFN_OK;
  {
    Seg sg[2] = {{b.fin, W, W, 1, 0}, {b.t_in, m.ld_t, m.nt, b.Ns, W}};
    CHECK_HIP(gemm_fwd(sg, 2, n.w[TE0], W + m.nt, n.b[TE0], W2, ACT_RELU, b.te[0], W2, b.P, s), "train forward: transient_encoding.0");
    for (int j = 1; j < 4; ++j) {
      Seg t{b.te[j - 1], W2, W2, 1, 0};
      CHECK_HIP(gemm_fwd(&t, 1, n.w[TE0 + j], W2, n.b[TE0 + j], W2, ACT_RELU, b.te[j], W2, b.P, s), "train forward: transient_encoding");
    }
    Seg t3{b.te[3], W2, W2, 1, 0};
    CHECK_HIP(gemm_fwd(&t3, 1, n.w[TRGB], W2, n.b[TRGB], 3, ACT_SIGMOID, b.raw + 4, b.raw_ld, b.P, s), "train forward: transient_rgb");
    CHECK_HIP(gemm_fwd(&t3, 1, n.w[TSIG], W2, n.b[TSIG], 1, ACT_SOFTPLUS, b.raw + 7, b.raw_ld, b.P, s), "train forward: transient_sigma");
    CHECK_HIP(gemm_fwd(&t3, 1, n.w[TBETA], W2, n.b[TBETA], 1, ACT_SOFTPLUS, b.raw + 8, b.raw_ld, b.P, s), "train forward: transient_beta");
  }
  return DFN_OK;
}

struct BwdBufs { float *gpre, *gA, *gB, *gfin, *gt0, *gt1, *gsum, *gray, *wscratch; };

// Gradients of every parameter of one network from the pre-activation gradients of its outputs (gpre, raw_ld wide).
// weights == false (dfn_nerfh_train_backward_rays): the data-gradient chain alone, continued into the encodings — g_pe [P,64]
// (d L / d pe_xyz: layer 1 + the skip columns of layer 5) and g_dpe [P,28] (d L / d pe_dir: dir_encoding.0's direction columns).
int net_backward(const Net& n, const Dims& m, const NetBufs& b, const BwdBufs& g, bool fine, const float* hist, size_t hist_rows,
                 float* g_emb_a, float* g_emb_t, size_t R, hipStream_t s, bool weights = true, float* g_pe = nullptr, float* g_dpe = nullptr) {
  const int W = m.W, W2 = m.W2, C = b.raw_ld;
  const long long P = b.P;
  auto wg = [&](const float* G, int ldg, int N, Seg x, float* dW, int ldw, float* db, long long PP) {
    return weights ? gemm_wgrad(G, ldg, N, x, dW, ldw, db, g.wscratch, PP, s) : hipSuccess;
  };
  const int ldw_dir = W + b.kd, ldw_te0 = W + m.nt;
  if (fine) {
    // transient heads -> transient_encoding.6 output
    CHECK_HIP(gemm_bwd(g.gpre + 4, C, 3, n.w[TRGB], W2, 0, W2, g.gt0, W2, 0, nullptr, 0, P, s), "train backward: transient_rgb");
    CHECK_HIP(gemm_bwd(g.gpre + 7, C, 1, n.w[TSIG], W2, 0, W2, g.gt0, W2, 1, nullptr, 0, P, s), "train backward: transient_sigma");
    CHECK_HIP(gemm_bwd(g.gpre + 8, C, 1, n.w[TBETA], W2, 0, W2, g.gt0, W2, 1, b.te[3], W2, P, s), "train backward: transient_beta");
    Seg t3{b.te[3], W2, W2, 1, 0};
    CHECK_HIP(wg(g.gpre + 4, C, 3, t3, n.gw[TRGB], W2, n.gb[TRGB], P), "train wgrad: transient_rgb");
    CHECK_HIP(wg(g.gpre + 7, C, 1, t3, n.gw[TSIG], W2, n.gb[TSIG], P), "train wgrad: transient_sigma");
    CHECK_HIP(wg(g.gpre + 8, C, 1, t3, n.gw[TBETA], W2, n.gb[TBETA], P), "train wgrad: transient_beta");
    float* cur = g.gt0;
    float* nxt = g.gt1;
    for (int j = 3; j >= 1; --j) {   // transient_encoding.{6,4,2}
      Seg x{b.te[j - 1], W2, W2, 1, 0};
      CHECK_HIP(wg(cur, W2, W2, x, n.gw[TE0 + j], W2, n.gb[TE0 + j], P), "train wgrad: transient_encoding");
      CHECK_HIP(gemm_bwd(cur, W2, W2, n.w[TE0 + j], W2, 0, W2, nxt, W2, 0, b.te[j - 1], W2, P, s), "train backward: transient_encoding");
      float* t = cur; cur = nxt; nxt = t;
    }
    // transient_encoding.0 on cat([final, t])
    CHECK_HIP(wg(cur, W2, W2, Seg{b.fin, W, W, 1, 0}, n.gw[TE0], ldw_te0, n.gb[TE0], P), "train wgrad: transient_encoding.0");
    if (weights) {
      CHECK_HIP(sum_over_samples(cur, W2, W2, R, b.Ns, g.gsum, W2, s), "train backward: per-ray sum");
      CHECK_HIP(wg(g.gsum, W2, W2, Seg{b.t_in, m.ld_t, m.nt, 1, W}, n.gw[TE0], ldw_te0, nullptr, (long long)R), "train wgrad: transient tail");
      CHECK_HIP(gemm_bwd(g.gsum, W2, W2, n.w[TE0], ldw_te0, W, m.nt, g.gray, m.ld_t, 0, nullptr, 0, (long long)R, s), "train backward: d t");
      CHECK_HIP(embedding_scatter(g.gray, m.ld_t, 0, hist, hist_rows, m.hist_bin, m.dim_t, m.n_vocab, R, g_emb_t, s), "train: embedding_t grad");
    }
    CHECK_HIP(gemm_bwd(cur, W2, W2, n.w[TE0], ldw_te0, 0, W, g.gfin, W, 0, nullptr, 0, P, s), "train backward: d final (transient)");
  }
  // static_rgb -> dir_encoding output
  CHECK_HIP(gemm_bwd(g.gpre, C, 3, n.w[RGB], W2, 0, W2, g.gt0, W2, 0, b.dirh, W2, P, s), "train backward: static_rgb");
  CHECK_HIP(wg(g.gpre, C, 3, Seg{b.dirh, W2, W2, 1, 0}, n.gw[RGB], W2, n.gb[RGB], P), "train wgrad: static_rgb");
  CHECK_HIP(wg(g.gt0, W2, W2, Seg{b.fin, W, W, 1, 0}, n.gw[DIR], ldw_dir, n.gb[DIR], P), "train wgrad: dir_encoding");
  if (g_dpe) CHECK_HIP(gemm_bwd(g.gt0, W2, W2, n.w[DIR], ldw_dir, W, kChDir, g_dpe, 28, 0, nullptr, 0, P, s), "train backward: d pe_dir");
  if (weights) {
    CHECK_HIP(sum_over_samples(g.gt0, W2, W2, R, b.Ns, g.gsum, W2, s), "train backward: per-ray sum");
    CHECK_HIP(wg(g.gsum, W2, W2, Seg{b.dir_in, b.ld_dir, b.kd, 1, W}, n.gw[DIR], ldw_dir, nullptr, (long long)R), "train wgrad: dir tail");
  }
  if (fine && weights) {
    CHECK_HIP(gemm_bwd(g.gsum, W2, W2, n.w[DIR], ldw_dir, W + kChDir, m.na, g.gray, m.ld_df, 0, nullptr, 0, (long long)R, s), "train backward: d a");
    CHECK_HIP(embedding_scatter(g.gray, m.ld_df, 0, hist, hist_rows, m.hist_bin, m.dim_a, m.n_vocab, R, g_emb_a, s), "train: embedding_a grad");
  }
  CHECK_HIP(gemm_bwd(g.gt0, W2, W2, n.w[DIR], ldw_dir, 0, W, g.gfin, W, fine ? 1 : 0, nullptr, 0, P, s), "train backward: d final");
  // xyz_encoding_final + static_sigma on h8
  Seg h8{b.h[7], W, W, 1, 0};
  CHECK_HIP(wg(g.gfin, W, W, h8, n.gw[FIN], W, n.gb[FIN], P), "train wgrad: xyz_encoding_final");
  CHECK_HIP(wg(g.gpre + 3, C, 1, h8, n.gw[SIG], W, n.gb[SIG], P), "train wgrad: static_sigma");
  CHECK_HIP(gemm_bwd(g.gfin, W, W, n.w[FIN], W, 0, W, g.gA, W, 0, nullptr, 0, P, s), "train backward: xyz_encoding_final");
  CHECK_HIP(gemm_bwd(g.gpre + 3, C, 1, n.w[SIG], W, 0, W, g.gA, W, 1, b.h[7], W, P, s), "train backward: static_sigma");
  float* cur = g.gA;
  float* nxt = g.gB;
  Seg pe{b.pe, 64, kChXyz, 1, 0};
  for (int l = 7; l >= 1; --l) {
    if (l == 4) {
      CHECK_HIP(wg(cur, W, W, pe, n.gw[l], W + kChXyz, n.gb[l], P), "train wgrad: xyz_encoding_5 (xyz)");
      CHECK_HIP(wg(cur, W, W, Seg{b.h[3], W, W, 1, kChXyz}, n.gw[l], W + kChXyz, nullptr, P), "train wgrad: xyz_encoding_5 (h)");
      if (g_pe) CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W + kChXyz, 0, kChXyz, g_pe, 64, 0, nullptr, 0, P, s), "train backward: d pe (skip)");
      CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W + kChXyz, kChXyz, W, nxt, W, 0, b.h[3], W, P, s), "train backward: xyz_encoding_5");
    } else {
      CHECK_HIP(wg(cur, W, W, Seg{b.h[l - 1], W, W, 1, 0}, n.gw[l], W, n.gb[l], P), "train wgrad: xyz_encoding");
      CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W, 0, W, nxt, W, 0, b.h[l - 1], W, P, s), "train backward: xyz_encoding");
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  CHECK_HIP(wg(cur, W, W, pe, n.gw[0], kChXyz, n.gb[0], P), "train wgrad: xyz_encoding_1");
  if (g_pe) CHECK_HIP(gemm_bwd(cur, W, W, n.w[0], kChXyz, 0, kChXyz, g_pe, 64, 1, nullptr, 0, P, s), "train backward: d pe");
  return DFN_OK;
}

NetBufs bufs_of(const TrainWs& w, const Dims& m, bool fine, float* raw_f, size_t R, int Nc, int Ni) {
  NetBufs b{};
  if (fine) {
    b.pe = w.pe_f; for (int i = 0; i < 8; ++i) b.h[i] = w.h_f[i];
    b.fin = w.fin_f; b.dirh = w.dirh_f; for (int i = 0; i < 4; ++i) b.te[i] = w.te[i];
    b.raw = raw_f; b.raw_ld = 9; b.dir_in = w.dir_f; b.ld_dir = m.ld_df; b.kd = m.kd_f; b.t_in = w.t_in;
    b.Ns = Nc + Ni; b.P = (long long)R * b.Ns;
  } else {
    b.pe = w.pe_c; for (int i = 0; i < 8; ++i) b.h[i] = w.h_c[i];
    b.fin = w.fin_c; b.dirh = w.dirh_c; b.raw = w.raw_c; b.raw_ld = 4; b.dir_in = w.dir_c; b.ld_dir = m.ld_dc; b.kd = m.kd_c;
    b.Ns = Nc; b.P = (long long)R * Nc;
  }
  return b;
}

int check_train_args(dfn_nerfh_t h, int Nc, int Ni, const char* fn) {
  if (!h) return set_error(DFN_ERR_ARG, "%s: null handle", fn);
  if (Nc < 3 || Ni < 1 || Nc + Ni > 512) return set_error(DFN_ERR_UNSUPPORTED, "%s: need 3 <= N_samples, 1 <= N_importance, sum <= 512", fn);
  return DFN_OK;
}
}  // namespace

extern "C" int dfn_nerfh_train_param_count(void) { return kParamCount; }
extern "C" const char* dfn_nerfh_train_param_name(int i) { return (i >= 0 && i < kParamCount) ? names()[i].c_str() : nullptr; }

// netwidth 128 on the register-resident kernels: the fused chain of nerfh_fused_*.hip; otherwise (any other netwidth, or
// dfn_nerfh_set_train_mode(h, DFN_TRAIN_EXACT)) the layer-by-layer exact-fp32 products below.
static bool use_fused(dfn_nerfh_t h) { return fused::available(h) && !h->train_exact; }

extern "C" int dfn_nerfh_set_train_mode(dfn_nerfh_t h, int mode) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_nerfh_set_train_mode: null handle");
  if (mode != DFN_TRAIN_FUSED && mode != DFN_TRAIN_EXACT && mode != DFN_TRAIN_FUSED_SPLIT)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_set_train_mode: unknown mode %d", mode);
  h->train_exact = mode == DFN_TRAIN_EXACT;
  if (mode != DFN_TRAIN_EXACT) h->train_split_fine = mode == DFN_TRAIN_FUSED_SPLIT;
  return DFN_OK;
}

extern "C" size_t dfn_nerfh_train_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni) {
  if (!h) return 0;
  const size_t exact = carve_train(nullptr, dims_of(h->desc), n_rays ? n_rays : 1, Nc, Ni, true).total;
  if (!fused::available(h)) return exact;
  const size_t fz = fused::workspace_bytes(h, n_rays ? n_rays : 1, Nc, Ni);
  return fz > exact ? fz : exact;   // either mode fits
}

extern "C" int dfn_nerfh_train_forward(dfn_nerfh_t h, const float* const* params, const float* rays_o, const float* rays_d,
                                       const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near, float far,
                                       const float* t_rand, const float* noise, float raw_noise_std, const float* u, float* rgb,
                                       float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                                       float* beta, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_train_args(h, Nc, Ni, "dfn_nerfh_train_forward")) return rc;
  if (!n_rays) return DFN_OK;
  if (!params || !rays_o || !rays_d || !hist || !rgb || !disp || !acc || !raw || !rgb0 || !disp0 || !acc0 || !z_std || !beta ||
      !workspace || (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_forward: bad argument (hist_rows must be 1 or n_rays)");
  h->train_forward_exact = !use_fused(h);
  h->train_forward_split = h->train_split_fine;
  if (use_fused(h))
    return fused::train_forward(h, params, rays_o, rays_d, hist, hist_rows, n_rays, Nc, Ni, near, far, t_rand, noise, raw_noise_std, u, rgb,
                                disp, acc, raw, rgb0, disp0, acc0, z_std, beta, workspace, workspace_bytes, HS(stream));
  const Dims m = dims_of(h->desc);
  const TrainWs w = carve_train(static_cast<float*>(workspace), m, n_rays, Nc, Ni, true);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = HS(stream);
  const size_t R = n_rays;
  const int Nf = Nc + Ni;
  const Net nc = net_of(params, nullptr, false), nf = net_of(params, nullptr, true);
  const float* emb_a = params[kCoarseParams + kFineParams];
  const float* emb_t = params[kCoarseParams + kFineParams + 1];
  CHECK_HIP(launch_viewdirs(rays_d, R, w.view, s), "train forward: viewdirs");
  CHECK_HIP(ray_inputs(w.view, nullptr, 1, nullptr, nullptr, m.hist_bin, m.dim_a, m.dim_t, m.n_vocab, R, w.dir_c, m.ld_dc, nullptr, 0, s),
            "train forward: coarse ray inputs");
  CHECK_HIP(ray_inputs(w.view, hist, hist_rows, emb_a, emb_t, m.hist_bin, m.dim_a, m.dim_t, m.n_vocab, R, w.dir_f, m.ld_df, w.t_in,
                       m.ld_t, s),
            "train forward: fine ray inputs");
  CHECK_HIP(stratified_z(t_rand, R, Nc, near, far, w.z_c, s, h->render_flags & DFN_RENDER_LINDISP), "train forward: stratified z");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_c, R, Nc, w.pe_c, s), "train forward: coarse encoding");
  if (int rc = net_forward(nc, m, bufs_of(w, m, false, nullptr, R, Nc, Ni), false, false, s)) return rc;
  CHECK_HIP(sample_fine_train(w.raw_c, w.z_c, noise, raw_noise_std, u, R, Nc, Ni, w.z_f, rgb0, disp0, acc0, z_std, s),
            "train forward: coarse composite + sampling");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_f, R, Nf, w.pe_f, s), "train forward: fine encoding");
  if (int rc = net_forward(nf, m, bufs_of(w, m, true, raw, R, Nc, Ni), true, false, s)) return rc;
  // training compositing: joint rgb, depth = sum w z, beta = sum w_t beta_t + beta_min (flags = 0)
  CHECK_HIP(launch_composite_fine(raw, w.z_f, R, Nf, 0.1f, 0, rgb, disp, acc, nullptr, nullptr, beta, s), "train forward: fine composite");
  return DFN_OK;
}

extern "C" int dfn_nerfw_loss(const float* rgb, const float* rgb0, const float* beta, const float* raw, const float* target,
                              size_t n_rays, int Nf, float coef, float lambda_u, float* loss5, float* g_rgb, float* g_rgb0,
                              float* g_beta, void* stream) {
  if (!rgb || !rgb0 || !beta || !raw || !target || !loss5 || !g_rgb || !g_rgb0 || !g_beta || !n_rays || Nf < 1)
    return set_error(DFN_ERR_ARG, "dfn_nerfw_loss: bad argument");
  CHECK_HIP(nerfw_loss(rgb, rgb0, beta, raw, target, n_rays, Nf, coef, lambda_u, loss5, g_rgb, g_rgb0, g_beta, HS(stream)), "dfn_nerfw_loss");
  return DFN_OK;
}

extern "C" int dfn_nerfh_train_backward(dfn_nerfh_t h, const float* const* params, const float* hist, size_t hist_rows, size_t n_rays,
                                        int Nc, int Ni, const float* noise, float raw_noise_std, const float* raw, const float* g_rgb,
                                        const float* g_rgb0, const float* g_beta, float g_tsigma, const float* g_tsigma_dense, float* const* grads, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (int rc = check_train_args(h, Nc, Ni, "dfn_nerfh_train_backward")) return rc;
  if (!n_rays) return DFN_OK;
  if (!params || !grads || !hist || !raw || !g_rgb || !g_rgb0 || !g_beta || !workspace || (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward: bad argument");
  for (int i = 0; i < kParamCount; ++i)
    if (!params[i] || !grads[i]) return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward: params[%d] / grads[%d] is null", i, i);
  // the workspace was laid out by the implementation that ran the forward: the other one's carve() would read foreign bytes
  if (h->train_forward_exact != !use_fused(h))
    return set_error(DFN_ERR_STATE, "dfn_nerfh_train_backward: the last dfn_nerfh_train_forward ran the %s step, the handle is now in the %s "
                     "mode (dfn_nerfh_set_train_mode between forward and backward)", h->train_forward_exact ? "exact" : "fused",
                     use_fused(h) ? "fused" : "exact");
  if (use_fused(h) && h->train_forward_split != h->train_split_fine)
    return set_error(DFN_ERR_STATE, "dfn_nerfh_train_backward: DFN_TRAIN_FUSED / DFN_TRAIN_FUSED_SPLIT changed between forward and backward "
                     "(dfn_nerfh_set_train_mode between forward and backward): the stored operands have the other layout");
  if (use_fused(h))
    return fused::train_backward(h, params, hist, hist_rows, n_rays, Nc, Ni, noise, raw_noise_std, raw, g_rgb, g_rgb0, g_beta, g_tsigma,
                                 g_tsigma_dense, grads, workspace, workspace_bytes, HS(stream));
  const Dims m = dims_of(h->desc);
  const TrainWs w = carve_train(static_cast<float*>(workspace), m, n_rays, Nc, Ni, true);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = HS(stream);
  const size_t R = n_rays;
  const int Nf = Nc + Ni;
  float* g_emb_a = grads[kCoarseParams + kFineParams];
  float* g_emb_t = grads[kCoarseParams + kFineParams + 1];
  CHECK_HIP(hipMemsetAsync(g_emb_a, 0, size_t(m.n_vocab) * m.dim_a * 4, s), "train backward: zero embedding_a grad");
  CHECK_HIP(hipMemsetAsync(g_emb_t, 0, size_t(m.n_vocab) * m.dim_t * 4, s), "train backward: zero embedding_t grad");
  CHECK_HIP(composite_fine_backward_train(raw, w.z_f, g_rgb, g_beta, g_tsigma, g_tsigma_dense, R, Nf, w.gpre_f, s), "train backward: fine composite");
  CHECK_HIP(composite_coarse_backward(w.raw_c, w.z_c, noise, raw_noise_std, g_rgb0, R, Nc, w.gpre_c, s), "train backward: coarse composite");
  BwdBufs gf{w.gpre_f, w.gA, w.gB, w.gfin, w.gt0, w.gt1, w.gsum, w.gray, w.wscratch};
  BwdBufs gc{w.gpre_c, w.gA_c, w.gB_c, w.gfin_c, w.gt0_c, nullptr, w.gsum_c, w.gray_c, w.wscratch_c};
  // The two networks' backward passes share nothing (z_samples.detach(), rendering.py:302): the coarse one — a third of the points,
  // 1.5 tiles per persistent workgroup, its launches half-empty in their second round — runs on a side stream beside the fine one.
  if (!h->side_stream) {
    CHECK_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking), "train backward: side stream");
    for (hipEvent_t& e : h->side_ev) CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "train backward: event");
  }
  CHECK_HIP(hipEventRecord(h->side_ev[0], s), "train backward: fork");
  CHECK_HIP(hipStreamWaitEvent(h->side_stream, h->side_ev[0], 0), "train backward: fork");
  if (int rc = net_backward(net_of(params, grads, false), m, bufs_of(w, m, false, nullptr, R, Nc, Ni), gc, false, hist, hist_rows, nullptr,
                            nullptr, R, h->side_stream))
    return rc;
  CHECK_HIP(hipEventRecord(h->side_ev[1], h->side_stream), "train backward: join");
  if (int rc = net_backward(net_of(params, grads, true), m, bufs_of(w, m, true, const_cast<float*>(raw), R, Nc, Ni), gf, true, hist,
                            hist_rows, g_emb_a, g_emb_t, R, s))
    return rc;
  CHECK_HIP(hipStreamWaitEvent(s, h->side_ev[1], 0), "train backward: join");
  return DFN_OK;
}

// The reference's training render is differentiable with respect to its rays as well (models/rendering.py:245-337 under autograd:
// pts = o + d z enter both networks; z itself carries no gradient — near / far bounds, z_samples.detach()).  d L / d rays from the
// same output gradients as dfn_nerfh_train_backward, on the layer-by-layer exact-fp32 products: the data-gradient chains of BOTH
// networks continued into the positional encodings, their Jacobians, the per-ray reduction.  Needs the activations of an
// exact-mode forward (dfn_nerfh_set_train_mode(h, DFN_TRAIN_EXACT)) in `workspace`.
extern "C" size_t dfn_nerfh_train_backward_rays_scratch_bytes(size_t n_rays, int Nc, int Ni) {
  const size_t Pf = (n_rays ? n_rays : 1) * (size_t(Nc) + Ni);
  return (al64(Pf * 64) + al64(Pf * 28) + al64(Pf * 6)) * sizeof(float);
}
extern "C" int dfn_nerfh_train_backward_rays(dfn_nerfh_t h, const float* const* params, const float* rays_o, const float* rays_d,
                                             const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, const float* noise,
                                             float raw_noise_std, const float* raw, const float* g_rgb, const float* g_rgb0,
                                             const float* g_beta, float g_tsigma, const float* g_tsigma_dense, float* grad_rays_o,
                                             float* grad_rays_d, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                             void* stream) {
  if (int rc = check_train_args(h, Nc, Ni, "dfn_nerfh_train_backward_rays")) return rc;
  if (!n_rays) return DFN_OK;
  if (!params || !rays_o || !rays_d || !hist || !raw || !g_rgb || !g_rgb0 || !g_beta || !grad_rays_o || !grad_rays_d || !workspace || !scratch ||
      (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward_rays: bad argument");
  if (!h->train_forward_exact)
    return set_error(DFN_ERR_STATE, "dfn_nerfh_train_backward_rays: the last dfn_nerfh_train_forward of this handle ran the fused chain, which keeps no "
                                    "activations: call dfn_nerfh_set_train_mode(h, DFN_TRAIN_EXACT) before the forward");
  const Dims m = dims_of(h->desc);
  const TrainWs w = carve_train(static_cast<float*>(workspace), m, n_rays, Nc, Ni, true);
  if (w.total > workspace_bytes || dfn_nerfh_train_backward_rays_scratch_bytes(n_rays, Nc, Ni) > scratch_bytes)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward_rays: workspace / scratch too small");
  hipStream_t s = HS(stream);
  const size_t R = n_rays;
  const int Nf = Nc + Ni;
  const size_t Pf = R * size_t(Nf);
  float* g_pe = static_cast<float*>(scratch);
  float* g_dpe = g_pe + al64(Pf * 64);
  float* gpts = g_dpe + al64(Pf * 28);
  CHECK_HIP(composite_fine_backward_train(raw, w.z_f, g_rgb, g_beta, g_tsigma, g_tsigma_dense, R, Nf, w.gpre_f, s), "train backward (rays): fine composite");
  CHECK_HIP(composite_coarse_backward(w.raw_c, w.z_c, noise, raw_noise_std, g_rgb0, R, Nc, w.gpre_c, s), "train backward (rays): coarse composite");
  BwdBufs gf{w.gpre_f, w.gA, w.gB, w.gfin, w.gt0, w.gt1, w.gsum, w.gray, w.wscratch};
  BwdBufs gc{w.gpre_c, w.gA_c, w.gB_c, w.gfin_c, w.gt0_c, nullptr, w.gsum_c, w.gray_c, w.wscratch_c};
  if (int rc = net_backward(net_of(params, nullptr, true), m, bufs_of(w, m, true, const_cast<float*>(raw), R, Nc, Ni), gf, true, hist, hist_rows,
                            nullptr, nullptr, R, s, false, g_pe, g_dpe))
    return rc;
  CHECK_HIP(posenc_backward(rays_o, rays_d, w.view, w.z_f, g_pe, g_dpe, 28, R, Nf, gpts, s), "train backward (rays): fine encodings");
  CHECK_HIP(launch_ray_grad_reduce(gpts, w.z_f, rays_d, R, Nf, 1, grad_rays_o, grad_rays_d, nullptr, s), "train backward (rays): fine reduction");
  if (int rc = net_backward(net_of(params, nullptr, false), m, bufs_of(w, m, false, nullptr, R, Nc, Ni), gc, false, hist, hist_rows, nullptr, nullptr,
                            R, s, false, g_pe, g_dpe))
    return rc;
  CHECK_HIP(posenc_backward(rays_o, rays_d, w.view, w.z_c, g_pe, g_dpe, 28, R, Nc, gpts, s), "train backward (rays): coarse encodings");
  CHECK_HIP(launch_ray_grad_reduce(gpts, w.z_c, rays_d, R, Nc, 1, grad_rays_o, grad_rays_d, nullptr, s, 1), "train backward (rays): coarse reduction");
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ generic-width test-time render
extern "C" size_t dfn_nerfh_generic_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni) {
  if (!h) return 0;
  return carve_train(nullptr, dims_of(h->desc), n_rays ? n_rays : 1, Nc, Ni, false).total + al64((n_rays ? n_rays : 1) * size_t(Nc)) * 4;
}

extern "C" int dfn_nerfh_generic_render_rays(dfn_nerfh_t h, const float* rays_o, const float* rays_d, const float* hist, size_t hist_rows,
                                             size_t n_rays, int Nc, int Ni, float near, float far, float* rgb, float* disp, float* acc,
                                             float* raw, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_train_args(h, Nc, Ni, "dfn_nerfh_generic_render_rays")) return rc;
  if (!h->committed) return set_error(DFN_ERR_STATE, "dfn_nerfh_generic_render_rays: dfn_nerfh_commit() has not been called");
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !hist || !rgb || !disp || !acc || !raw || !workspace || (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_nerfh_generic_render_rays: bad argument (raw [n_rays, Nc+Ni, 9] is required scratch/output)");
  const Dims m = dims_of(h->desc);
  const TrainWs w = carve_train(static_cast<float*>(workspace), m, n_rays, Nc, Ni, false);
  float* sigma = static_cast<float*>(workspace) + w.total / 4;
  if (w.total + al64(n_rays * size_t(Nc)) * 4 > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_generic_render_rays: workspace too small (%zu)", workspace_bytes);
  hipStream_t s = HS(stream);
  const size_t R = n_rays;
  const int Nf = Nc + Ni;
  const float* const* params = h->gen_params.data();
  const Net nc = net_of(params, nullptr, false), nf = net_of(params, nullptr, true);
  CHECK_HIP(launch_viewdirs(rays_d, R, w.view, s), "generic render: viewdirs");
  CHECK_HIP(ray_inputs(w.view, hist, hist_rows, params[kCoarseParams + kFineParams], params[kCoarseParams + kFineParams + 1], m.hist_bin,
                       m.dim_a, m.dim_t, m.n_vocab, R, w.dir_f, m.ld_df, w.t_in, m.ld_t, s),
            "generic render: ray inputs");
  CHECK_HIP(stratified_z(nullptr, R, Nc, near, far, w.z_c, s, h->render_flags & DFN_RENDER_LINDISP), "generic render: z");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_c, R, Nc, w.pe_c, s), "generic render: coarse encoding");
  NetBufs bc = bufs_of(w, m, false, nullptr, R, Nc, Ni);
  bc.raw = sigma;   // sigma only: [R, Nc]
  bc.raw_ld = 1;
  if (int rc = net_forward(nc, m, bc, false, true, s)) return rc;
  CHECK_HIP(launch_sample_fine(sigma, R, Nc, Ni, near, far, w.z_f, nullptr, nullptr, s, h->render_flags & DFN_RENDER_LINDISP), "generic render: sample_fine");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_f, R, Nf, w.pe_f, s), "generic render: fine encoding");
  if (int rc = net_forward(nf, m, bufs_of(w, m, true, raw, R, Nc, Ni), true, false, s)) return rc;
  CHECK_HIP(launch_composite_fine(raw, w.z_f, R, Nf, 0.1f, DFN_COMP_TEST_TIME | DFN_COMP_STATIC_ONLY, rgb, disp, acc, nullptr, nullptr,
                                  nullptr, s),
            "generic render: composite");
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ generic-width render gradient
// dfn_render_rays_backward for ANY even netwidth, exact fp32: the test-time forward of dfn_nerfh_generic_render_rays with every
// fine activation kept, raw2outputs_NeRFW backward (rgb only), head derivatives, then the fine network's data-gradient chain
// (net_backward without the weight gradients, plus the two products that reach the encodings) and the encodings' Jacobians.
namespace {
struct GenBwdWs { TrainWs t; float *sigma, *raw, *g_pe, *g_dpe, *gpts, *view; size_t total; };
GenBwdWs carve_gen_bwd(float* base, const Dims& m, size_t R, int Nc, int Ni) {
  GenBwdWs w{};
  w.t = carve_train(base, m, R, Nc, Ni, true);
  size_t off = w.t.total / sizeof(float);
  const size_t Pf = R * (size_t(Nc) + Ni);
  auto take = [&](size_t floats) { float* p = base ? base + off : nullptr; off += al64(floats); return p; };
  w.sigma = take(R * Nc);
  w.raw = take(Pf * 9);
  w.g_pe = take(Pf * 64);
  w.g_dpe = take(Pf * 28);
  w.gpts = take(Pf * 6);
  w.view = take(R * 3);
  w.total = off * sizeof(float);
  return w;
}
}  // namespace

extern "C" size_t dfn_nerfh_generic_backward_workspace_bytes(dfn_nerfh_t h, size_t n_rays, int Nc, int Ni) {
  if (!h) return 0;
  return carve_gen_bwd(nullptr, dims_of(h->desc), n_rays ? n_rays : 1, Nc, Ni).total;
}

extern "C" int dfn_nerfh_generic_render_rays_backward(dfn_nerfh_t h, const float* rays_o, const float* rays_d, const float* viewdirs,
                                                      const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near,
                                                      float far, const float* grad_rgb, float* grad_rays_o, float* grad_rays_d,
                                                      float* grad_viewdirs, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_train_args(h, Nc, Ni, "dfn_nerfh_generic_render_rays_backward")) return rc;
  if (!h->committed) return set_error(DFN_ERR_STATE, "dfn_nerfh_generic_render_rays_backward: dfn_nerfh_commit() has not been called");
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !hist || !grad_rgb || !grad_rays_o || !grad_rays_d || !workspace || (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_nerfh_generic_render_rays_backward: bad argument (hist_rows must be 1 or n_rays)");
  const Dims m = dims_of(h->desc);
  const GenBwdWs g = carve_gen_bwd(static_cast<float*>(workspace), m, n_rays, Nc, Ni);
  if (g.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_generic_render_rays_backward: workspace too small (%zu < %zu)", workspace_bytes, g.total);
  hipStream_t s = HS(stream);
  const TrainWs& w = g.t;
  const size_t R = n_rays;
  const int Nf = Nc + Ni, W = m.W, W2 = m.W2;
  const long long P = (long long)R * Nf;
  const float* const* params = h->gen_params.data();
  const Net nc = net_of(params, nullptr, false), nf = net_of(params, nullptr, true);
  const float* v = viewdirs;
  if (!v) {
    CHECK_HIP(launch_viewdirs(rays_d, R, g.view, s), "generic render gradient: viewdirs");
    v = g.view;
  }
  // ---- forward (dfn_nerfh_generic_render_rays), every fine activation kept in the workspace
  CHECK_HIP(ray_inputs(v, hist, hist_rows, params[kCoarseParams + kFineParams], params[kCoarseParams + kFineParams + 1], m.hist_bin,
                       m.dim_a, m.dim_t, m.n_vocab, R, w.dir_f, m.ld_df, w.t_in, m.ld_t, s),
            "generic render gradient: ray inputs");
  CHECK_HIP(stratified_z(nullptr, R, Nc, near, far, w.z_c, s, h->render_flags & DFN_RENDER_LINDISP), "generic render gradient: z");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_c, R, Nc, w.pe_c, s), "generic render gradient: coarse encoding");
  NetBufs bc = bufs_of(w, m, false, nullptr, R, Nc, Ni);
  bc.raw = g.sigma;
  bc.raw_ld = 1;
  if (int rc = net_forward(nc, m, bc, false, true, s)) return rc;
  CHECK_HIP(launch_sample_fine(g.sigma, R, Nc, Ni, near, far, w.z_f, nullptr, nullptr, s, h->render_flags & DFN_RENDER_LINDISP), "generic render gradient: sample_fine");
  CHECK_HIP(posenc_points(rays_o, rays_d, w.z_f, R, Nf, w.pe_f, s), "generic render gradient: fine encoding");
  const NetBufs b = bufs_of(w, m, true, g.raw, R, Nc, Ni);
  if (int rc = net_forward(nf, m, b, true, false, s)) return rc;
  // ---- d rgb -> d raw -> d pre-activation (in place)
  float* gpre = w.gpre_f;
  CHECK_HIP(launch_composite_fine_backward(g.raw, w.z_f, grad_rgb, R, Nf, gpre, s), "generic render gradient: composite");
  CHECK_HIP(head_prime(g.raw, gpre, size_t(P), s), "generic render gradient: head derivatives");
  // ---- data gradients (the calls of net_backward, without the weight gradients)
  const int C = 9, ldw_dir = W + b.kd, ldw_te0 = W + m.nt;
  const Net& n = nf;
  CHECK_HIP(gemm_bwd(gpre + 4, C, 3, n.w[TRGB], W2, 0, W2, w.gt0, W2, 0, nullptr, 0, P, s), "generic render gradient: transient_rgb");
  CHECK_HIP(gemm_bwd(gpre + 7, C, 1, n.w[TSIG], W2, 0, W2, w.gt0, W2, 1, nullptr, 0, P, s), "generic render gradient: transient_sigma");
  CHECK_HIP(gemm_bwd(gpre + 8, C, 1, n.w[TBETA], W2, 0, W2, w.gt0, W2, 1, b.te[3], W2, P, s), "generic render gradient: transient_beta");
  float* cur = w.gt0;
  float* nxt = w.gt1;
  for (int j = 3; j >= 1; --j) {
    CHECK_HIP(gemm_bwd(cur, W2, W2, n.w[TE0 + j], W2, 0, W2, nxt, W2, 0, b.te[j - 1], W2, P, s), "generic render gradient: transient_encoding");
    float* t = cur; cur = nxt; nxt = t;
  }
  CHECK_HIP(gemm_bwd(cur, W2, W2, n.w[TE0], ldw_te0, 0, W, w.gfin, W, 0, nullptr, 0, P, s), "generic render gradient: d final (transient)");
  CHECK_HIP(gemm_bwd(gpre, C, 3, n.w[RGB], W2, 0, W2, w.gt0, W2, 0, b.dirh, W2, P, s), "generic render gradient: static_rgb");
  CHECK_HIP(gemm_bwd(w.gt0, W2, W2, n.w[DIR], ldw_dir, W, kChDir, g.g_dpe, 28, 0, nullptr, 0, P, s), "generic render gradient: d pe_dir");
  CHECK_HIP(gemm_bwd(w.gt0, W2, W2, n.w[DIR], ldw_dir, 0, W, w.gfin, W, 1, nullptr, 0, P, s), "generic render gradient: d final");
  CHECK_HIP(gemm_bwd(w.gfin, W, W, n.w[FIN], W, 0, W, w.gA, W, 0, nullptr, 0, P, s), "generic render gradient: xyz_encoding_final");
  CHECK_HIP(gemm_bwd(gpre + 3, C, 1, n.w[SIG], W, 0, W, w.gA, W, 1, b.h[7], W, P, s), "generic render gradient: static_sigma");
  cur = w.gA;
  nxt = w.gB;
  for (int l = 7; l >= 1; --l) {
    if (l == 4) {
      CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W + kChXyz, 0, kChXyz, g.g_pe, 64, 0, nullptr, 0, P, s), "generic render gradient: d pe (skip)");
      CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W + kChXyz, kChXyz, W, nxt, W, 0, b.h[3], W, P, s), "generic render gradient: xyz_encoding_5");
    } else {
      CHECK_HIP(gemm_bwd(cur, W, W, n.w[l], W, 0, W, nxt, W, 0, b.h[l - 1], W, P, s), "generic render gradient: xyz_encoding");
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  CHECK_HIP(gemm_bwd(cur, W, W, n.w[0], kChXyz, 0, kChXyz, g.g_pe, 64, 1, nullptr, 0, P, s), "generic render gradient: d pe");
  // ---- encodings, then the per-ray reduction of dfn_render_rays_backward
  CHECK_HIP(posenc_backward(rays_o, rays_d, v, w.z_f, g.g_pe, g.g_dpe, 28, R, Nf, g.gpts, s), "generic render gradient: encodings");
  CHECK_HIP(launch_ray_grad_reduce(g.gpts, w.z_f, rays_d, R, Nf, viewdirs == nullptr, grad_rays_o, grad_rays_d,
                                   viewdirs ? grad_viewdirs : nullptr, s),
            "generic render gradient: per-ray reduction");
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ the three products, for parity tests
extern "C" int dfn_linear_forward(const float* x, int ldx, int K, const float* w, int ldw, int wcol, const float* b, int N, int act,
                                  float* y, int ldy, size_t n_points, int x_row_div, void* stream) {
  if (!x || !w || !y || K < 1 || N < 1 || act < 0 || act > 3 || x_row_div < 1) return set_error(DFN_ERR_ARG, "dfn_linear_forward: bad argument");
  Seg sg{x, ldx, K, x_row_div, wcol};
  CHECK_HIP(gemm_fwd(&sg, 1, w, ldw, b, N, act, y, ldy, (long long)n_points, HS(stream)), "dfn_linear_forward");
  return DFN_OK;
}
extern "C" int dfn_linear_backward_input(const float* g, int ldg, int N, const float* w, int ldw, int wcol, int K, float* dx, int lddx,
                                         int accumulate, const float* relu_src, int ld_relu, size_t n_points, void* stream) {
  if (!g || !w || !dx || K < 1 || N < 1) return set_error(DFN_ERR_ARG, "dfn_linear_backward_input: bad argument");
  CHECK_HIP(gemm_bwd(g, ldg, N, w, ldw, wcol, K, dx, lddx, accumulate, relu_src, ld_relu, (long long)n_points, HS(stream)),
            "dfn_linear_backward_input");
  return DFN_OK;
}
extern "C" size_t dfn_linear_backward_weight_scratch_bytes(int N, int K, size_t n_points) {
  return (gemm_wgrad_scratch_floats(N, K, (long long)n_points) + 64) * sizeof(float);
}
extern "C" int dfn_linear_backward_weight(const float* g, int ldg, int N, const float* x, int ldx, int K, int x_row_div, float* dw, int ldw,
                                          int wcol, float* db, void* scratch, size_t n_points, void* stream) {
  if (!g || !x || !dw || !scratch || K < 1 || N < 1 || x_row_div < 1 || !n_points)
    return set_error(DFN_ERR_ARG, "dfn_linear_backward_weight: bad argument");
  CHECK_HIP(gemm_wgrad(g, ldg, N, Seg{x, ldx, K, x_row_div, wcol}, dw, ldw, db, static_cast<float*>(scratch), (long long)n_points, HS(stream)),
            "dfn_linear_backward_weight");
  return DFN_OK;
}
