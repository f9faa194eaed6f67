// nerfh_fused_api.hip — host side of the fused NeRF-H training step (netwidth 128): the unit / destination tables that tie the
// master parameter tensors to the chain kernels' staging units and to the weight-gradient blocks, the workspace layout, and the
// two passes dfn_nerfh_train_forward / dfn_nerfh_train_backward run when the handle is on the register-resident kernels
// (nerfh_train_api.hip routes here; every other netwidth keeps the layer-by-layer path of nerfh_train.hip).
//
// Replaces (reference, /root/reference/script/): run_nerf.py:50-66, models/rendering.py:245-337 (render_rays, test_time=False),
// models/nerfw.py:47-95,297-354.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"
#include "nerfh_fused_train.h"
#include "nerfh_handle.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"
#include "nerfh_train.h"

using namespace dfn;
using namespace dfn::fused;
using namespace dfn::train;

#define CHECK_HIP(expr, what)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return set_error(DFN_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

namespace {
// canonical parameter order (dfn_nerfh_train_param_name): coarse layers 0..11, fine layers 0..18, embeddings
enum LayerIx { L1 = 0, L8 = 7, FIN = 8, DIR = 9, SIG = 10, RGB = 11, TE0 = 12, TE3 = 15, TSIG = 16, TRGB = 17, TBETA = 18 };
constexpr int kCoarseLayers_ = 12, kFineLayers_ = 19;
constexpr int kCoarseParams = 2 * kCoarseLayers_, kFineParams = 2 * kFineLayers_;
using P = PrecX3;
constexpr int UMB = 2;

struct Geo { int na, nt, kd_c, kd_f, ld_dc, ld_df, ld_t; };
inline int r4(int v) { return (v + 3) & ~3; }
Geo geo_of(const dfn_nerfh_desc& d) {
  Geo g{};
  g.na = d.hist_bin * d.dim_a; g.nt = d.hist_bin * d.dim_t;
  g.kd_c = kChDir; g.kd_f = kChDir + g.na;
  g.ld_dc = r4(g.kd_c); g.ld_df = r4(g.kd_f); g.ld_t = r4(g.nt);
  return g;
}

// ---- where an element of a staging unit / a weight-gradient block lives in the parameter tensors
struct Row { int layer; int row; };   // LayerIx of the Linear, row of its weight (row < 0: none)
int cols_of(int layer, bool fine, const Geo& g) {
  switch (layer) {
    case 0: return kChXyz;
    case 4: return kWidth + kChXyz;
    case DIR: return kWidth + (fine ? g.kd_f : g.kd_c);
    case RGB: case TSIG: case TRGB: case TBETA: return kWidth / 2;
    case TE0: return kWidth + g.nt;
    case 13: case 14: case 15: return kWidth / 2;
    default: return kWidth;   // xyz_encoding_2..4, 6..8, final, static_sigma
  }
}
int rows_of(int layer) {
  switch (layer) {
    case DIR: case TE0: case 13: case 14: case 15: return kWidth / 2;
    case SIG: case TSIG: case TBETA: return 1;
    case RGB: case TRGB: return 3;
    default: return kWidth;
  }
}
int weight_param(bool fine, int layer) { return (fine ? kCoarseParams : 0) + 2 * layer; }
int32_t enc(int param, int index) { return int32_t((param << 20) | index); }

// forward layer `ly` (nerfh_layout.h: LayerId), M-block mb, row i of the block -> Linear + row (Packer::row_source)
Row fwd_row(int ly, int mb, int i) {
  const int row = 32 * mb + i;
  if (ly >= LY_L1 && ly <= LY_L8) return {ly - LY_L1, row};
  switch (ly) {
    case LY_FIN: return mb < kWidth / 32 ? Row{FIN, row} : Row{SIG, i == 0 ? 0 : -1};
    case LY_DIR: return {DIR, row};
    case LY_RGB: return {RGB, i < 3 ? i : -1};
    case LY_TE0: return {TE0, row};
    case LY_TE1: return {13, row};
    case LY_TE2: return {14, row};
    case LY_TE3: return {15, row};
    case LY_THEAD: return i < 3 ? Row{TRGB, i} : (i == 3 ? Row{TSIG, 0} : (i == 8 ? Row{TBETA, 0} : Row{TBETA, -1}));
  }
  return {0, -1};
}
int fwd_col(int ly, int hh, int s) {   // Packer::col_source
  if (ly == LY_L1) return pe_xyz_feature(hh, s);
  if (ly == LY_L5) return s < 32 ? pe_xyz_feature(hh, s) : kChXyz + hidden_feature(hh, s - 32);
  return hidden_feature(hh, s);
}

// backward layers of the training chain (the transposed Linears; rows = INPUT features of the forward layer, contraction slots =
// its OUTPUT features in C-fragment order).  TBW_CAT: fine [transient_encoding.0 ; dir_encoding.0]^T on the `final` columns, coarse
// dir_encoding.0^T.  No layer-1 / encoding-column products: the rays carry no gradient in the optimisation step.
enum TBwd { TBW_THEAD = 0, TBW_TE3, TBW_TE2, TBW_TE1, TBW_RGB, TBW_CAT, TBW_FIN, TBW_L8, TBW_L7, TBW_L6, TBW_L5, TBW_L4, TBW_L3, TBW_L2, TBW_COUNT };
LayerShape tbwd_shape(int id, bool fine) {
  if (id == TBW_THEAD || id == TBW_RGB) return {16, 2};
  if (id <= TBW_TE1) return {32, 2};
  if (id == TBW_CAT) return {fine ? 64 : 32, 4};
  if (id == TBW_FIN) return {80, 4};
  return {64, 4};
}
// (Linear, row, column) of the backward element at input feature k, slot (hh, s); row < 0: zero
struct Elem3 { int layer, row, col; };
Elem3 tbwd_elem(int id, bool fine, int k, int hh, int s) {
  const int j = hidden_feature(hh, s);
  switch (id) {
    case TBW_THEAD: return j < 3 ? Elem3{TRGB, j, k} : (j == 3 ? Elem3{TSIG, 0, k} : (j == 8 ? Elem3{TBETA, 0, k} : Elem3{0, -1, 0}));
    case TBW_TE3: return {15, j, k};
    case TBW_TE2: return {14, j, k};
    case TBW_TE1: return {13, j, k};
    case TBW_RGB: return j < 3 ? Elem3{RGB, j, k} : Elem3{0, -1, 0};
    case TBW_CAT:
      if (!fine) return {DIR, j, k};
      return s < 32 ? Elem3{TE0, hidden_feature(hh, s), k} : Elem3{DIR, hidden_feature(hh, s - 32), k};
    case TBW_FIN:
      if (s < 64) return {FIN, j, k};
      return (s == 64 && hh == 0) ? Elem3{SIG, 0, k} : Elem3{0, -1, 0};
    case TBW_L5: return {4, j, kChXyz + k};
    default: return {8 - (id - TBW_L8) - 1, j, k};   // TBW_L8 -> xyz_encoding_8 (LayerIx 7) ... TBW_L2 -> LayerIx 1
  }
}

struct BlobPlan {
  std::vector<uint32_t> tab;        // (offset, bytes) per unit
  std::vector<PackElem> welem, belem;
  uint32_t bytes = 0;
};
BlobPlan plan_forward(bool fine, const Geo& g) {
  BlobPlan bp;
  const int seq_f[] = {LY_L1, LY_L2, LY_L3, LY_L4, LY_L5, LY_L6, LY_L7, LY_L8, LY_FIN, LY_DIR, LY_RGB, LY_TE0, LY_TE1, LY_TE2, LY_TE3, LY_THEAD};
  const int nl = fine ? 16 : 11;
  for (int li = 0; li < nl; ++li) {
    const int ly = seq_f[li];
    const LayerShape sh = layer_shape(ly);
    const int KC = sh.slots / 8;
    for (int u0 = 0; u0 < sh.mb; u0 += UMB) {
      const int group = sh.mb - u0 < UMB ? sh.mb - u0 : UMB;
      const uint32_t ub = unit_bytes<P>(sh.slots, group), off = bp.bytes;
      bp.tab.push_back(off);
      bp.tab.push_back(ub);
      bp.bytes += ub;
      for (int gi = 0; gi < group; ++gi) {
        const int mb = u0 + gi;
        for (int lane = 0; lane < 64; ++lane) {
          const Row rw = fwd_row(ly, mb, lane & 31);
          if (rw.row < 0 || rw.row >= rows_of(rw.layer)) continue;
          const int cols = cols_of(rw.layer, fine, g);
          for (int kc = 0; kc < KC; ++kc)
            for (int jj = 0; jj < 8; ++jj) {
              const int col = fwd_col(ly, lane >> 5, kc * 8 + jj);
              if (col < 0 || col >= cols) continue;
              bp.welem.push_back({off + uint32_t((gi * KC + kc) * 2048 + lane * 16 + jj * 2), enc(weight_param(fine, rw.layer), rw.row * cols + col)});
            }
        }
        if (ly == LY_DIR || ly == LY_TE0) continue;   // these biases are folded into the per-ray table
        for (int hh = 0; hh < 2; ++hh)
          for (int r = 0; r < 16; ++r) {
            const Row rw = fwd_row(ly, mb, mblock_row(hh, r));
            if (rw.row < 0 || rw.row >= rows_of(rw.layer)) continue;
            bp.belem.push_back({off + uint32_t(group * KC * 2048 + ((gi * 2 + hh) * 16 + r) * 4), enc(weight_param(fine, rw.layer) + 1, rw.row)});
          }
      }
    }
  }
  return bp;
}
BlobPlan plan_backward(bool fine, const Geo& g) {
  BlobPlan bp;
  for (int id = fine ? TBW_THEAD : TBW_RGB; id < TBW_COUNT; ++id) {
    const LayerShape sh = tbwd_shape(id, fine);
    const int KC = sh.slots / 8;
    for (int u0 = 0; u0 < sh.mb; u0 += UMB) {
      const int group = sh.mb - u0 < UMB ? sh.mb - u0 : UMB;
      const uint32_t ub = unit_bytes<P>(sh.slots, group), off = bp.bytes;
      bp.tab.push_back(off);
      bp.tab.push_back(ub);
      bp.bytes += ub;
      for (int gi = 0; gi < group; ++gi)
        for (int lane = 0; lane < 64; ++lane)
          for (int kc = 0; kc < KC; ++kc)
            for (int jj = 0; jj < 8; ++jj) {
              const Elem3 e = tbwd_elem(id, fine, 32 * (u0 + gi) + (lane & 31), lane >> 5, kc * 8 + jj);
              if (e.row < 0 || e.row >= rows_of(e.layer)) continue;
              const int cols = cols_of(e.layer, fine, g);
              if (e.col >= cols) continue;
              bp.welem.push_back({off + uint32_t((gi * KC + kc) * 2048 + lane * 16 + jj * 2), enc(weight_param(fine, e.layer), e.row * cols + e.col)});
            }
    }
  }
  return bp;
}

// ---- weight-gradient jobs
// One job = one stored gradient array against one stored input array.  Layer 5 reads cat([pe, h4]): two jobs on the same gradients
// (the encoding columns, then the h4 columns at weight column 63; the second without the bias sums).
// (GA_CAT2 = [d final (chunks 0..7) ; d sigma_s (chunks 8, 9)] is read by two jobs as well, so that no job has more than 20 blocks:
// five per wave of the stream kernel.)  gc0 / gkc: the chunks of the gradient array the job reads.
struct JobPlan { int garr, gc0, gkc, x0; int col0; bool bias; };
const JobPlan kJobsFine[] = {{GA_L1, 0, 8, XA_PE, 0, true}, {GA_L2, 0, 8, XA_H1, 0, true}, {GA_L3, 0, 8, XA_H2, 0, true}, {GA_L4, 0, 8, XA_H3, 0, true},
                             {GA_L5, 0, 8, XA_PE, 0, true}, {GA_L5, 0, 8, XA_H4, kChXyz, false}, {GA_L6, 0, 8, XA_H5, 0, true}, {GA_L7, 0, 8, XA_H6, 0, true},
                             {GA_L8, 0, 8, XA_H7, 0, true}, {GA_CAT2, 0, 8, XA_H8, 0, true}, {GA_CAT2, 8, 2, XA_H8, 0, true}, {GA_CAT, 0, 0, XA_FIN, 0, true},
                             {GA_DRGB, 0, 2, XA_DE, 0, true},
                             {GA_T1, 0, 4, XA_T0, 0, true}, {GA_T2, 0, 4, XA_T1, 0, true}, {GA_T3, 0, 4, XA_T2, 0, true}, {GA_DTH, 0, 2, XA_T3, 0, true}};
constexpr int kNumJobsFine = 17, kNumJobsCoarse = 13;
int job_kcg(const JobPlan& jp, bool fine) { return jp.gkc ? jp.gkc : (fine ? kGChunksFine : kGChunksCoarse)[jp.garr]; }   // GA_CAT: 8 fine / 4 coarse
// G-side slot (hh, s) of array `garr` -> Linear + row
Row g_row(int garr, bool fine, int hh, int s) {
  const int j = hidden_feature(hh, s & 31);   // feature inside a 64-wide (two M-block) vector
  if (garr <= GA_L8) return {garr - GA_L1, hidden_feature(hh, s)};
  switch (garr) {
    case GA_CAT2: return s < 64 ? Row{FIN, hidden_feature(hh, s)} : ((s == 64 && hh == 0) ? Row{SIG, 0} : Row{0, -1});
    case GA_CAT:
      if (!fine) return {DIR, hidden_feature(hh, s)};
      return s < 32 ? Row{TE0, j} : Row{DIR, j};
    case GA_DRGB: return j < 3 && s < 16 ? Row{RGB, j} : Row{0, -1};
    case GA_T1: return {13, hidden_feature(hh, s)};
    case GA_T2: return {14, hidden_feature(hh, s)};
    case GA_T3: return {15, hidden_feature(hh, s)};
    case GA_DTH: return s >= 16 ? Row{0, -1} : (j < 3 ? Row{TRGB, j} : (j == 3 ? Row{TSIG, 0} : (j == 8 ? Row{TBETA, 0} : Row{0, -1})));
  }
  return {0, -1};
}
// X-side slot (hh, s) of the job's concatenated X chunks -> weight column
int x_col(const JobPlan& jp, int hh, int s) {
  if (jp.x0 == XA_PE) return pe_xyz_feature(hh, s);
  return jp.col0 + hidden_feature(hh, s);
}
int job_kcx(const JobPlan& jp) { return kXChunks[jp.x0]; }
// destination map of one job: [nb_g * (nb_x + 1)][1024], index r * 64 + lane
void job_map(const JobPlan& jp, bool fine, const Geo& g, std::vector<int32_t>& out) {
  const int kcg = job_kcg(jp, fine), kcx = job_kcx(jp);
  const int nb_g = kcg / 2, nb_x = kcx / 2;
  for (int gb = 0; gb < nb_g; ++gb)
    for (int xb = 0; xb < nb_x + (jp.bias ? 1 : 0); ++xb)
      for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = mblock_row(lane >> 5, r), n = lane & 31;
          const int sg = 8 * (jp.gc0 + 2 * gb + (m >> 4)) + (m & 7), hg = (m >> 3) & 1;
          const Row rw = g_row(jp.garr, fine, hg, sg);
          int32_t dst = -1;
          if (rw.row >= 0 && rw.row < rows_of(rw.layer)) {
            if (xb == nb_x) {
              if (n == 0) dst = enc(weight_param(fine, rw.layer) + 1, rw.row);
            } else {
              const int sx = 8 * (2 * xb + (n >> 4)) + (n & 7), hx = (n >> 3) & 1;
              const int col = x_col(jp, hx, sx), cols = cols_of(rw.layer, fine, g);
              if (col >= 0 && col < cols) dst = enc(weight_param(fine, rw.layer), rw.row * cols + col);
            }
          }
          out.push_back(dst);
        }
}

struct DevBlob { char* blob = nullptr; uint32_t* tab = nullptr; PackElem* welem = nullptr; PackElem* belem = nullptr; int n_units = 0, n_w = 0, n_b = 0; uint32_t bytes = 0; };
}  // namespace

namespace dfn {
namespace fused {
struct State {
  DevBlob blob[2][2];      // [coarse / fine][forward / backward]
  int32_t* map = nullptr;  // destination maps: coarse jobs then fine jobs
  int map_off[2][kMaxJobs] = {};
};
}  // namespace fused
}  // namespace dfn

namespace {
template <class T>
int upload_vec(const std::vector<T>& v, T** dst) {
  const size_t bytes = v.size() * sizeof(T);
  if (hipMalloc(reinterpret_cast<void**>(dst), bytes ? bytes : 16) != hipSuccess) return set_error(DFN_ERR_HIP, "fused training: hipMalloc(%zu) failed", bytes);
  if (bytes && hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return set_error(DFN_ERR_HIP, "fused training: upload failed");
  return DFN_OK;
}
// h->fused is published only when every allocation and upload has succeeded: a failure frees what was built and leaves the handle
// without a state, so the next dfn_nerfh_train_forward builds again instead of launching on null blobs.
int build_state_into(dfn_nerfh_s* h, State* st) {
  const Geo g = geo_of(h->desc);
  for (int f = 0; f < 2; ++f)
    for (int pass = 0; pass < 2; ++pass) {
      const BlobPlan bp = pass ? plan_backward(f, g) : plan_forward(f, g);
      DevBlob& d = st->blob[f][pass];
      d.bytes = bp.bytes;
      d.n_units = int(bp.tab.size() / 2);
      d.n_w = int(bp.welem.size());
      d.n_b = int(bp.belem.size());
      if (hipMalloc(reinterpret_cast<void**>(&d.blob), bp.bytes) != hipSuccess || hipMemset(d.blob, 0, bp.bytes) != hipSuccess)
        return set_error(DFN_ERR_HIP, "fused training: blob allocation failed");
      if (int rc = upload_vec(bp.tab, &d.tab)) return rc;
      if (int rc = upload_vec(bp.welem, &d.welem)) return rc;
      if (int rc = upload_vec(bp.belem, &d.belem)) return rc;
    }
  std::vector<int32_t> maps;
  for (int f = 0; f < 2; ++f)
    for (int j = 0; j < (f ? kNumJobsFine : kNumJobsCoarse); ++j) {
      st->map_off[f][j] = int(maps.size());
      job_map(kJobsFine[j], f, g, maps);
    }
  return upload_vec(maps, &st->map);
}
int build_state(dfn_nerfh_s* h) {
  auto* st = new State();
  const int rc = build_state_into(h, st);
  if (rc != DFN_OK) {
    h->fused = st;                       // destroy_state frees the partial state through the handle
    dfn::fused::destroy_state(h);        // ... and leaves h->fused null
    return rc;
  }
  h->fused = st;
  return DFN_OK;
}
}  // namespace

// Host-only consistency check of the tables above (no GPU work): every weight / bias element the chains read is packed exactly
// once per pass, and every gradient element the stream owns is written by exactly one (job, block, lane, register).
extern "C" int dfn_nerfh_train_tables_selfcheck(const dfn_nerfh_desc* desc) {
  if (!desc || desc->width != kWidth) return set_error(DFN_ERR_ARG, "dfn_nerfh_train_tables_selfcheck: netwidth 128 only");
  const Geo g = geo_of(*desc);
  for (int f = 0; f < 2; ++f) {
    const int nl = f ? kFineLayers_ : kCoarseLayers_;
    // expected coverage per Linear: columns [c0, c1) of every row
    auto expect = [&](int pass, int layer, int& c0, int& c1) {   // pass 0 forward blob, 1 backward blob, 2 gradient maps
      const int cols = cols_of(layer, f, g);
      c0 = 0; c1 = cols;
      if (layer == DIR || layer == TE0) c1 = kWidth;                 // the tail columns are per-ray work
      if (pass == 1 && layer == 0) c1 = 0;                           // no data gradient through layer 1
      if (pass == 1 && layer == 4) c0 = kChXyz;                      // ... nor through the encoding columns of layer 5
    };
    for (int pass = 0; pass < 3; ++pass) {
      std::vector<std::vector<int>> cnt(2 * nl);
      for (int l = 0; l < nl; ++l) {
        cnt[2 * l].assign(size_t(rows_of(l)) * cols_of(l, f, g), 0);
        cnt[2 * l + 1].assign(rows_of(l), 0);
      }
      auto hit = [&](int32_t src) {
        const int pid = (src >> 20) - (f ? kCoarseParams : 0), idx = src & 0xfffff;
        if (pid < 0 || pid >= 2 * nl || idx >= int(cnt[pid].size())) return false;
        ++cnt[pid][idx];
        return true;
      };
      if (pass < 2) {
        const BlobPlan bp = pass ? plan_backward(f, g) : plan_forward(f, g);
        if (bp.tab.empty()) return set_error(DFN_ERR_STATE, "selfcheck: empty plan");
        for (const auto& e : bp.welem) if (e.off + 1024 + 2 > bp.bytes || !hit(e.src)) return set_error(DFN_ERR_STATE, "selfcheck: bad weight element (net %d pass %d)", f, pass);
        for (const auto& e : bp.belem) if (e.off + 4 > bp.bytes || !hit(e.src)) return set_error(DFN_ERR_STATE, "selfcheck: bad bias element (net %d pass %d)", f, pass);
      } else {
        std::vector<int32_t> maps;
        for (int j = 0; j < (f ? kNumJobsFine : kNumJobsCoarse); ++j) job_map(kJobsFine[j], f, g, maps);
        for (int32_t m : maps) if (m >= 0 && !hit(m)) return set_error(DFN_ERR_STATE, "selfcheck: bad gradient destination (net %d)", f);
      }
      for (int l = 0; l < nl; ++l) {
        int c0, c1;
        expect(pass, l, c0, c1);
        const int cols = cols_of(l, f, g);
        for (int r = 0; r < rows_of(l); ++r)
          for (int c = 0; c < cols; ++c) {
            const int want = (c >= c0 && c < c1) ? 1 : 0;
            if (cnt[2 * l][size_t(r) * cols + c] != want)
              return set_error(DFN_ERR_STATE, "selfcheck: net %d pass %d layer %d weight[%d][%d] covered %d times, expected %d", f, pass, l, r, c,
                               cnt[2 * l][size_t(r) * cols + c], want);
          }
        const int want_b = pass == 0 ? ((l == DIR || l == TE0) ? 0 : 1) : (pass == 1 ? 0 : 1);
        for (int r = 0; r < rows_of(l); ++r)
          if (cnt[2 * l + 1][r] != want_b)
            return set_error(DFN_ERR_STATE, "selfcheck: net %d pass %d layer %d bias[%d] covered %d times, expected %d", f, pass, l, r, cnt[2 * l + 1][r], want_b);
      }
    }
  }
  return DFN_OK;
}

namespace dfn {
namespace fused {

void destroy_state(dfn_nerfh_s* h) {
  auto* st = static_cast<State*>(h->fused);
  if (!st) return;
  for (auto& a : st->blob)
    for (auto& d : a) {
      if (d.blob) (void)hipFree(d.blob);
      if (d.tab) (void)hipFree(d.tab);
      if (d.welem) (void)hipFree(d.welem);
      if (d.belem) (void)hipFree(d.belem);
    }
  if (st->map) (void)hipFree(st->map);
  delete st;
  h->fused = nullptr;
}

bool available(const dfn_nerfh_s* h) {
  if (!h || !h->fast || h->desc.width != kWidth) return false;
  const Geo g = geo_of(h->desc);
  return size_t(g.kd_f + g.nt) * 64 * sizeof(float) <= 64 * 1024;   // the per-ray bias kernel stages both weight tails in LDS
}

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
struct NetWs {
  char* x;  size_t x_off[XA_COUNT];
  char* g;  size_t g_off[GA_COUNT];
  float* gscale; uint32_t* masks; float* ray_bias; float* gpre; float* z;
  size_t n_wt; long long P;
};
struct Ws {
  float *view, *dir_c, *dir_f, *t_in, *raw_c, *gsum_f, *gsum_c, *gray, *wscratch, *partial;
  NetWs net[2];
  size_t partial_floats, total;
};
size_t tail_scratch_floats(const Geo& g, size_t R) {
  const int pairs[][2] = {{kWidth / 2, g.ld_df}, {kWidth / 2, g.ld_dc}, {kWidth / 2, g.ld_t}};
  size_t sum = 0;     // the three products' partials side by side (gemm_wgrad_multi: one launch for all three)
  for (const auto& nk : pairs) sum += gemm_wgrad_scratch_floats(nk[0], nk[1], (long long)R);
  return sum + 1024;
}
// Point chunking of one network's jobs: `target` workgroups per launch (a whole number of rounds of the 2 x CU resident
// workgroups), shared among the jobs in proportion to the bytes they stream.  Depends on the batch shape only (deterministic).
void plan_chunks(bool fine, size_t n_wt, int target, int* wpc, int* n_chunks) {
  const int* kcgs = fine ? kGChunksFine : kGChunksCoarse;
  const int nj = fine ? kNumJobsFine : kNumJobsCoarse;
  double total = 0;
  for (int j = 0; j < nj; ++j) total += job_kcg(kJobsFine[j], fine) + job_kcx(kJobsFine[j]);
  for (int j = 0; j < nj; ++j) {
    const double share = (job_kcg(kJobsFine[j], fine) + job_kcx(kJobsFine[j])) / total;
    long long c = (long long)(share * target);
    if (c < 1) c = 1;
    if (c > (long long)n_wt) c = (long long)n_wt;
    wpc[j] = int((n_wt + c - 1) / c);
    n_chunks[j] = int((n_wt + wpc[j] - 1) / wpc[j]);
  }
}
constexpr int kStreamCus = 256;   // the chunking is sized for MI355X's 256 CUs (a shape constant: the sums must not depend on the device)
int launch_target(bool fine) { return (fine ? 4 : 2) * kStreamCus; }   // fine: two rounds of 2 workgroups per CU, coarse: one
// job list of one network against a carved workspace; returns the number of workgroups of the launch
int make_jobs(bool fine, const NetWs& n, const State& st, float*& partial, WJob* jobs, int& n_jobs) {
  const int* kcgs = fine ? kGChunksFine : kGChunksCoarse;
  int wpc[kMaxJobs], nch[kMaxJobs];
  plan_chunks(fine, n.n_wt, launch_target(fine), wpc, nch);
  int wgs = 0;
  for (int j = 0; j < (fine ? kNumJobsFine : kNumJobsCoarse); ++j) {
    const JobPlan& jp = kJobsFine[j];
    WJob& w = jobs[n_jobs++];
    w.kcg = job_kcg(jp, fine);
    w.g_stride = kcgs[jp.garr];
    w.g_chunk0 = jp.gc0;
    w.kcx0 = kXChunks[jp.x0];
    w.kcx1 = 0;
    w.g = n.g + n.g_off[jp.garr];
    w.x0 = n.x + n.x_off[jp.x0];
    w.x1 = nullptr;
    w.gscale = n.gscale + size_t(jp.garr) * n.n_wt;
    w.nb_g = w.kcg / 2;
    w.nb_x = w.kcx0 / 2;
    w.has_bias = jp.bias ? 1 : 0;
    w.wt_per_chunk = wpc[j];
    w.n_chunks = nch[j];
    w.first_wg = wgs;
    w.map_off = st.map_off[fine][j];
    w.partial = partial;
    if (partial) partial += size_t(w.n_chunks) * w.nb_g * (w.nb_x + w.has_bias) * 1024;
    wgs += w.n_chunks;
  }
  return wgs;
}
Ws carve(char* base, const dfn_nerfh_desc& d, size_t R, int Nc, int Ni, bool split_fine) {
  Ws w{};
  const Geo g = geo_of(d);
  const size_t Nf = size_t(Nc) + Ni;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  auto takef = [&](size_t floats) { return reinterpret_cast<float*>(take(floats * 4)); };
  w.view = takef(R * 3);
  w.dir_c = takef(R * g.ld_dc);
  w.dir_f = takef(R * g.ld_df);
  w.t_in = takef(R * g.ld_t);
  w.raw_c = takef(R * Nc * 4);
  w.gsum_f = takef(R * kWidth);
  w.gsum_c = takef(R * (kWidth / 2));
  w.gray = takef(R * size_t(g.ld_df > g.ld_t ? g.ld_df : g.ld_t));
  w.wscratch = takef(tail_scratch_floats(g, R));
  for (int f = 0; f < 2; ++f) {
    NetWs& n = w.net[f];
    n.P = (long long)R * (f ? Nf : Nc);
    n.n_wt = chain_wave_tiles(n.P);
    n.z = takef(size_t(n.P));
    n.gpre = takef(size_t(n.P) * (f ? 9 : 4));
    n.ray_bias = takef(R * kRayBiasFloats);
    n.masks = reinterpret_cast<uint32_t*>(take(n.n_wt * kMaskWords * 64 * 4));
    n.gscale = takef(size_t(GA_COUNT) * n.n_wt);
    size_t xb = 0, gb = 0;
    for (int a = 0; a < (f ? int(XA_COUNT) : kXCountCoarse); ++a) { n.x_off[a] = xb; xb += n.n_wt * kXChunks[a] * size_t(1024 * planes_of(f, split_fine)); }
    for (int a = 0; a < (f ? int(GA_COUNT) : kGCountCoarse); ++a) { n.g_off[a] = gb; gb += n.n_wt * (f ? kGChunksFine : kGChunksCoarse)[a] * size_t(1024 * planes_of(f, split_fine)); }
    n.x = take(xb);
    n.g = take(gb);
  }
  // weight-gradient partials
  {
    State dummy;
    WJob jobs[kMaxJobs];
    int nj = 0;
    float* part = nullptr;
    size_t floats = 0;
    for (int f = 0; f < 2; ++f) {
      nj = 0;
      make_jobs(f, w.net[f], dummy, part, jobs, nj);
      for (int j = 0; j < nj; ++j) floats += size_t(jobs[j].n_chunks) * jobs[j].nb_g * (jobs[j].nb_x + jobs[j].has_bias) * 1024;
    }
    w.partial_floats = floats;
    w.partial = takef(floats);
  }
  w.total = off;
  return w;
}

// Operand scale of the training chains: 2^10 above the render kernels' (whose pipelined conversion narrows before it scales and
// therefore keeps the largest |w| at ~1): the chains scale in fp32 first, so their weights sit where every lo half is a normal f16.
float train_scale(const dfn_nerfh_s* h, bool fine) { return h->net[fine][2][0].in_scale * 1024.f; }

ChainArgs chain_args(const dfn_nerfh_s* h, const State& st, bool fine, int pass, const NetWs& n, const float* o, const float* d, size_t R,
                     int Ns) {
  ChainArgs a{};
  const DevBlob& b = st.blob[fine][pass];
  a.blob = b.blob; a.tab = b.tab; a.n_units = b.n_units;
  a.rays_o = o; a.rays_d = d; a.z = n.z; a.ray_bias = n.ray_bias;
  a.masks = n.masks;
  a.arrays = pass ? n.g : n.x;
  for (int i = 0; i < 16; ++i) a.arr_off[i] = pass ? (i < GA_COUNT ? n.g_off[i] : 0) : (i < XA_COUNT ? n.x_off[i] : 0);
  a.gscale = n.gscale;
  a.gpre = n.gpre;
  a.n_rays = (long long)R;
  a.n_samples = Ns;
  a.in_scale = train_scale(h, fine);
  a.status = h->range_flag + 2;   // the step's own word (GuardArgs)
  return a;
}
}  // namespace

// (sized for the larger layout, so that a workspace stays valid across dfn_nerfh_set_train_mode)
size_t workspace_bytes(const dfn_nerfh_s* h, size_t R, int Nc, int Ni) { return carve(nullptr, h->desc, R, Nc, Ni, true).total; }

int train_forward(dfn_nerfh_s* h, const float* const* params, const float* rays_o, const float* rays_d, const float* hist, size_t hist_rows,
                  size_t R, int Nc, int Ni, float near, float far, const float* t_rand, const float* noise, float raw_noise_std,
                  const float* u, float* rgb, float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                  float* beta, void* workspace, size_t workspace_bytes_, hipStream_t s) {
  if (!h->fused)
    if (int rc = build_state(h)) return rc;
  const State& st = *static_cast<State*>(h->fused);
  const Ws w = carve(static_cast<char*>(workspace), h->desc, R, Nc, Ni, h->train_split_fine);
  if (w.total > workspace_bytes_) return set_error(DFN_ERR_ARG, "dfn_nerfh_train_forward: workspace too small (%zu < %zu)", workspace_bytes_, w.total);
  const Geo g = geo_of(h->desc);
  const dfn_nerfh_desc& d = h->desc;
  const int Nf = Nc + Ni, W = kWidth;
  const float* const* pc = params;
  const float* const* pf = params + kCoarseParams;
  const float* emb_a = params[kCoarseParams + kFineParams];
  const float* emb_t = params[kCoarseParams + kFineParams + 1];
  const int n_cu = device_cu_count();
  {   // view directions, both networks' per-ray input rows, the stratified coarse depths, the step's range word cleared: one launch
    TrainRayPrepArgs pa{};
    pa.rays_d = rays_d; pa.R = R;
    pa.hist = hist; pa.hist_rows = hist_rows; pa.emb_a = emb_a; pa.emb_t = emb_t;
    pa.hist_bin = d.hist_bin; pa.dim_a = d.dim_a; pa.dim_t = d.dim_t; pa.n_vocab = d.n_vocab;
    pa.view = w.view; pa.dir_c = w.dir_c; pa.ld_dc = g.ld_dc; pa.dir_f = w.dir_f; pa.ld_df = g.ld_df; pa.t_in = w.t_in; pa.ld_t = g.ld_t;
    pa.t_rand = t_rand; pa.Nc = Nc; pa.near = near; pa.far = far; pa.lindisp = (h->render_flags & DFN_RENDER_LINDISP) ? 1 : 0;
    pa.z = w.net[0].z;
    pa.range_word = h->range_flag + 2;
    CHECK_HIP(train_ray_prep(pa, s), "train forward: per-ray inputs");
  }
  // the step's weights -> staging units of the four chain passes (hi | lo split at the handle's operand scale)
  {
    PackArgs4 pa{};
    for (int f = 0; f < 2; ++f)
      for (int pass = 0; pass < 2; ++pass) {
        const DevBlob& d = st.blob[f][pass];
        const float in_scale = train_scale(h, f);
        pa.b[2 * f + pass] = PackBlob{d.welem, d.n_w, d.belem, d.n_b, d.blob, in_scale / kX3ActScale, in_scale};
      }
    for (int i = 0; i < 64; ++i) pa.params[i] = params[i];
    pa.status = h->range_flag + 2;
    CHECK_HIP(launch_pack4(pa, s), "fused training: weight packing");
  }
  CHECK_HIP(launch_ray_bias_train_pair(pc[2 * DIR], pc[2 * DIR + 1], W + g.kd_c, g.kd_c, w.dir_c, g.ld_dc, w.net[0].ray_bias,
                                       pf[2 * DIR], pf[2 * DIR + 1], W + g.kd_f, g.kd_f, w.dir_f, g.ld_df, pf[2 * TE0], pf[2 * TE0 + 1], W + g.nt,
                                       g.nt, w.t_in, g.ld_t, w.net[1].ray_bias, R, s),
            "train forward: per-ray bias tables");
  {
    ChainArgs a = chain_args(h, st, false, 0, w.net[0], rays_o, rays_d, R, Nc);
    a.raw_out = w.raw_c;
    CHECK_HIP(launch_train_forward_chain(false, planes_of(false, h->train_split_fine), a, n_cu, s), "train forward: coarse chain");
  }
  CHECK_HIP(sample_fine_train(w.raw_c, w.net[0].z, noise, raw_noise_std, u, R, Nc, Ni, w.net[1].z, rgb0, disp0, acc0, z_std, s),
            "train forward: coarse composite + sampling");
  {
    ChainArgs a = chain_args(h, st, true, 0, w.net[1], rays_o, rays_d, R, Nf);
    a.raw_out = raw;
    CHECK_HIP(launch_train_forward_chain(true, planes_of(true, h->train_split_fine), a, n_cu, s), "train forward: fine chain");
  }
  CHECK_HIP(launch_composite_fine(raw, w.net[1].z, R, Nf, 0.1f, 0, rgb, disp, acc, nullptr, nullptr, beta, s), "train forward: fine composite");
  return DFN_OK;
}

int train_backward(dfn_nerfh_s* h, const float* const* params, const float* hist, size_t hist_rows, size_t R, int Nc, int Ni,
                   const float* noise, float raw_noise_std, const float* raw, const float* g_rgb, const float* g_rgb0, const float* g_beta,
                   float g_tsigma, const float* g_tsigma_dense, float* const* grads, void* workspace, size_t workspace_bytes_, hipStream_t s) {
  if (!h->fused) return set_error(DFN_ERR_STATE, "dfn_nerfh_train_backward: no forward pass on this handle");
  const State& st = *static_cast<State*>(h->fused);
  const Ws w = carve(static_cast<char*>(workspace), h->desc, R, Nc, Ni, h->train_split_fine);
  if (w.total > workspace_bytes_) return set_error(DFN_ERR_ARG, "dfn_nerfh_train_backward: workspace too small (%zu < %zu)", workspace_bytes_, w.total);
  const Geo g = geo_of(h->desc);
  const dfn_nerfh_desc& d = h->desc;
  const int Nf = Nc + Ni, W = kWidth, W2 = kWidth / 2;
  const float* const* pc = params;
  const float* const* pf = params + kCoarseParams;
  float* const* gc = grads;
  float* const* gf = grads + kCoarseParams;
  float* g_emb_a = grads[kCoarseParams + kFineParams];
  float* g_emb_t = grads[kCoarseParams + kFineParams + 1];
  const int n_cu = device_cu_count();
  CHECK_HIP(composite_fine_backward_train(raw, w.net[1].z, g_rgb, g_beta, g_tsigma, g_tsigma_dense, R, Nf, w.net[1].gpre, s), "train backward: fine composite");
  // (the coarse compositor also zeroes the two embedding gradients that the scatter kernels at the end of the pass accumulate into)
  CHECK_HIP(composite_coarse_backward(w.raw_c, w.net[0].z, noise, raw_noise_std, g_rgb0, R, Nc, w.net[0].gpre, s, g_emb_a,
                                      size_t(d.n_vocab) * d.dim_a, g_emb_t, size_t(d.n_vocab) * d.dim_t),
            "train backward: coarse composite");
  // data-gradient chains: every pre-activation gradient stored once, in the operand layout the weight-gradient stream reads
  // (the two networks' chains are independent — the coarse loss alone reaches the coarse network, rendering.py:302 detaches the
  // samples — and run as the two halves of ONE grid, the coarse chain's workgroups starting on the CUs the fine chain leaves first:
  // train_bwd_chain_pair_kernel.  DFN_TRAIN_BWD_PAIR=0: one launch each, the A/B switch; same kernels' bodies, bit-identical.)
  static const bool pair = [] { const char* e = getenv("DFN_TRAIN_BWD_PAIR"); return !e || atoi(e) != 0; }();
  if (pair) {
    ChainArgs af = chain_args(h, st, true, 1, w.net[1], nullptr, nullptr, R, Nf);
    ChainArgs ac = chain_args(h, st, false, 1, w.net[0], nullptr, nullptr, R, Nc);
    CHECK_HIP(launch_train_backward_chain_pair(planes_of(true, h->train_split_fine), af, ac, n_cu, s), "train backward: chains");
  } else {
    {
      ChainArgs a = chain_args(h, st, true, 1, w.net[1], nullptr, nullptr, R, Nf);
      CHECK_HIP(launch_train_backward_chain(true, planes_of(true, h->train_split_fine), a, n_cu, s), "train backward: fine chain");
    }
    {
      ChainArgs a = chain_args(h, st, false, 1, w.net[0], nullptr, nullptr, R, Nc);
      CHECK_HIP(launch_train_backward_chain(false, planes_of(false, h->train_split_fine), a, n_cu, s), "train backward: coarse chain");
    }
  }
  // weight gradients: one stream launch per network (n_wt is a launch constant; disjoint partial buffers, the fine one first), then ONE
  // fixed-order reduction of both networks' partials into the .grad tensors (it was one per network, the fine one between the two
  // streams: a tiny launch less on the one-stream step, and the two reductions side by side)
  {
    WgradArgs wa{};
    ReduceArgs ra{};
    ra.n_jobs = 0;
    float* part = w.partial;
    for (int f = 1; f >= 0; --f) {
      wa.n_jobs = 0;
      const int wgs = make_jobs(f, w.net[f], st, part, wa.job, wa.n_jobs);
      wa.n_wt = int(w.net[f].n_wt);
      CHECK_HIP(launch_wgrad_stream(wa, wgs, planes_of(f, h->train_split_fine), s), "train backward: weight-gradient stream");
      for (int j = 0; j < wa.n_jobs; ++j) {
        const WJob& q = wa.job[j];
        ra.job[ra.n_jobs++] = RJob{q.partial, q.nb_g, q.nb_x, q.has_bias, q.n_chunks, q.map_off};
      }
    }
    ra.map = st.map;
    for (int i = 0; i < 64; ++i) ra.grads[i] = grads[i];
    CHECK_HIP(launch_wgrad_reduce(ra, s), "train backward: weight-gradient reduction");
  }
  // columns beyond `final` of dir_encoding.0 / transient_encoding.0 multiply per-ray inputs: per-ray sums of the stored gradients,
  // then the small products of the layer-by-layer path (nerfh_train.hip) over rays
  CHECK_HIP(launch_frag_ray_sum_pair(w.net[1].g + w.net[1].g_off[GA_CAT], 8, planes_of(true, h->train_split_fine),
                                     w.net[1].gscale + size_t(GA_CAT) * w.net[1].n_wt, Nf, w.gsum_f, W,
                                     w.net[0].g + w.net[0].g_off[GA_CAT], 4, w.net[0].gscale + size_t(GA_CAT) * w.net[0].n_wt, Nc, w.gsum_c, W2, R, s),
            "train backward: per-ray sums");
  const int ldw_dir_f = W + g.kd_f, ldw_dir_c = W + g.kd_c, ldw_te0 = W + g.nt;
  // the three small weight products over rays — transient_encoding.0 tail (gsum_f[:, 0:64]), dir_encoding.0 tail of the fine network
  // (gsum_f[:, 64:128]) and of the coarse one — in one launch + one reduction launch (six tiny launches in a row otherwise)
  const WgradJob tails[3] = {{w.gsum_f, W, W2, Seg{w.t_in, g.ld_t, g.nt, 1, W}, gf[2 * TE0], ldw_te0, nullptr},
                             {w.gsum_f + W2, W, W2, Seg{w.dir_f, g.ld_df, g.kd_f, 1, W}, gf[2 * DIR], ldw_dir_f, nullptr},
                             {w.gsum_c, W2, W2, Seg{w.dir_c, g.ld_dc, g.kd_c, 1, W}, gc[2 * DIR], ldw_dir_c, nullptr}};
  const bool multi = gemm_wgrad_multi_ok(tails, 3, (long long)R);
  if (multi) CHECK_HIP(gemm_wgrad_multi(tails, 3, w.wscratch, (long long)R, s), "train wgrad: per-ray tails");
  else CHECK_HIP(gemm_wgrad(w.gsum_f, W, W2, Seg{w.t_in, g.ld_t, g.nt, 1, W}, gf[2 * TE0], ldw_te0, nullptr, w.wscratch, (long long)R, s), "train wgrad: transient tail");
  CHECK_HIP(gemm_bwd(w.gsum_f, W, W2, pf[2 * TE0], ldw_te0, W, g.nt, w.gray, g.ld_t, 0, nullptr, 0, (long long)R, s), "train backward: d t");
  CHECK_HIP(embedding_scatter(w.gray, g.ld_t, 0, hist, hist_rows, d.hist_bin, d.dim_t, d.n_vocab, R, g_emb_t, s), "train: embedding_t grad");
  if (!multi) CHECK_HIP(gemm_wgrad(w.gsum_f + W2, W, W2, Seg{w.dir_f, g.ld_df, g.kd_f, 1, W}, gf[2 * DIR], ldw_dir_f, nullptr, w.wscratch, (long long)R, s), "train wgrad: dir tail");
  CHECK_HIP(gemm_bwd(w.gsum_f + W2, W, W2, pf[2 * DIR], ldw_dir_f, W + kChDir, g.na, w.gray, g.ld_df, 0, nullptr, 0, (long long)R, s), "train backward: d a");
  CHECK_HIP(embedding_scatter(w.gray, g.ld_df, 0, hist, hist_rows, d.hist_bin, d.dim_a, d.n_vocab, R, g_emb_a, s), "train: embedding_a grad");
  if (!multi) CHECK_HIP(gemm_wgrad(w.gsum_c, W2, W2, Seg{w.dir_c, g.ld_dc, g.kd_c, 1, W}, gc[2 * DIR], ldw_dir_c, nullptr, w.wscratch, (long long)R, s), "train wgrad: coarse dir tail");
  {   // a step whose operands left the split-f16 range leaves zeros, not clamped gradients (GuardArgs)
    GuardArgs ga{};
    for (int f = 0; f < 2; ++f)
      for (int l = 0; l < (f ? kFineLayers_ : kCoarseLayers_); ++l) {
        const int p = weight_param(f, l);
        ga.grads[p] = grads[p]; ga.numel[p] = uint32_t(rows_of(l)) * uint32_t(cols_of(l, f, g));
        ga.grads[p + 1] = grads[p + 1]; ga.numel[p + 1] = uint32_t(rows_of(l));
      }
    const int pe = kCoarseParams + kFineParams;
    ga.grads[pe] = g_emb_a; ga.numel[pe] = uint32_t(d.n_vocab) * uint32_t(d.dim_a);
    ga.grads[pe + 1] = g_emb_t; ga.numel[pe + 1] = uint32_t(d.n_vocab) * uint32_t(d.dim_t);
    ga.n = pe + 2;
    ga.step_flag = h->range_flag + 2;
    ga.range_flag = h->range_flag;
    CHECK_HIP(launch_grads_guard(ga, s), "train backward: range guard");
  }
  return DFN_OK;
}

}  // namespace fused
}  // namespace dfn
