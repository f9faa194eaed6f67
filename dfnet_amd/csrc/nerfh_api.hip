// nerfh_api.hip — the C ABI (include/dfnet_hip.h) of the NeRF-H render path: handle, host-side
// packing of the reference's state_dict tensors into MFMA A-fragments, stage entry points and
// the whole-path dfn_render_rays / dfn_render_image drivers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"
#include "nerfh_fused_train.h"
#include "nerfh_handle.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"

using namespace dfn;

// ------------------------------------------------------------------------------------------ errors
namespace dfn {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
int device_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
}  // namespace dfn

extern "C" const char* dfn_last_error(void) { return g_err; }
extern "C" int dfn_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------ handle (nerfh_handle.h)
static std::map<std::string, std::vector<size_t>> expected_shapes(const dfn_nerfh_desc& d) {
  std::map<std::string, std::vector<size_t>> m;
  const size_t W = d.width, na = size_t(d.hist_bin) * d.dim_a, nt = size_t(d.hist_bin) * d.dim_t;
  for (int f = 0; f < 2; ++f) {
    const std::string pre = f ? "fine." : "coarse.";
    for (int i = 0; i < d.depth; ++i) {
      const size_t k = i == 0 ? kChXyz : (i == 4 ? W + kChXyz : W);
      m[pre + "xyz_encoding_" + std::to_string(i + 1) + ".0.weight"] = {W, k};
      m[pre + "xyz_encoding_" + std::to_string(i + 1) + ".0.bias"] = {W};
    }
    m[pre + "xyz_encoding_final.weight"] = {W, W};
    m[pre + "xyz_encoding_final.bias"] = {W};
    m[pre + "dir_encoding.0.weight"] = {W / 2, W + kChDir + (f ? na : 0)};
    m[pre + "dir_encoding.0.bias"] = {W / 2};
    m[pre + "static_sigma.0.weight"] = {1, W};
    m[pre + "static_sigma.0.bias"] = {1};
    m[pre + "static_rgb.0.weight"] = {3, W / 2};
    m[pre + "static_rgb.0.bias"] = {3};
    if (f) {
      const size_t ks[4] = {W + nt, W / 2, W / 2, W / 2};
      for (int j = 0; j < 4; ++j) {
        m[pre + "transient_encoding." + std::to_string(2 * j) + ".weight"] = {W / 2, ks[j]};
        m[pre + "transient_encoding." + std::to_string(2 * j) + ".bias"] = {W / 2};
      }
      m[pre + "transient_sigma.0.weight"] = {1, W / 2};
      m[pre + "transient_sigma.0.bias"] = {1};
      m[pre + "transient_rgb.0.weight"] = {3, W / 2};
      m[pre + "transient_rgb.0.bias"] = {3};
      m[pre + "transient_beta.0.weight"] = {1, W / 2};
      m[pre + "transient_beta.0.bias"] = {1};
    }
  }
  m["embedding_a.weight"] = {size_t(d.n_vocab), size_t(d.dim_a)};
  m["embedding_t.weight"] = {size_t(d.n_vocab), size_t(d.dim_t)};
  return m;
}

extern "C" int dfn_nerfh_create(const dfn_nerfh_desc* desc, dfn_nerfh_t* out) {
  if (!desc || !out) return set_error(DFN_ERR_ARG, "dfn_nerfh_create: null argument");
  if (desc->depth != 8 || desc->multires != kLxyz || desc->multires_views != kLdir)
    return set_error(DFN_ERR_UNSUPPORTED,
                     "dfn_nerfh_create: kernels are specialised for netdepth=8 multires=%d multires_views=%d (got %d/%d/%d)",
                     kLxyz, kLdir, desc->depth, desc->multires, desc->multires_views);
  // netwidth: %d runs on the register-resident MFMA kernels; any other even width runs on the generic layer-by-layer
  // fp32-MFMA path (nerfh_train_api.hip), which is also the training path.
  if (desc->width < 2 || desc->width > 1024 || (desc->width & 1))
    return set_error(DFN_ERR_UNSUPPORTED, "dfn_nerfh_create: netwidth must be even and in [2, 1024] (got %d)", desc->width);
  if (desc->hist_bin <= 0 || desc->dim_a <= 0 || desc->dim_t <= 0 || desc->n_vocab <= 0 ||
      desc->hist_bin * (desc->dim_a + desc->dim_t) > 1024)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_create: bad embedding geometry");
  auto* h = new dfn_nerfh_s();
  h->desc = *desc;
  *out = h;
  return DFN_OK;
}

static void free_packed(dfn_nerfh_s* h) {
  for (auto& a : h->net)
    for (auto& b : a)
      for (auto& n : b) {
        if (n.blob) (void)hipFree(n.blob);
        if (n.tab) (void)hipFree(n.tab);
        n = PackedNet();
      }
  for (auto& n : h->bwd) {
    if (n.blob) (void)hipFree(n.blob);
    if (n.tab) (void)hipFree(n.tab);
    n = PackedNet();
  }
  if (h->extra) (void)hipFree(h->extra);
  h->extra = nullptr;
  if (h->gen_blob) (void)hipFree(h->gen_blob);
  h->gen_blob = nullptr;
  h->gen_params.clear();
  h->fast = false;
}

extern "C" int dfn_nerfh_destroy(dfn_nerfh_t h) {
  if (!h) return DFN_OK;
  free_packed(h);
  dfn::fused::destroy_state(h);
  for (hipEvent_t e : h->side_ev) if (e) (void)hipEventDestroy(e);
  if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
  if (h->range_flag) (void)hipFree(h->range_flag);
  delete h;
  return DFN_OK;
}

extern "C" int dfn_nerfh_set_param(dfn_nerfh_t h, const char* name, const float* host, size_t numel) {
  if (!h || !name || !host) return set_error(DFN_ERR_ARG, "dfn_nerfh_set_param: null argument");
  const auto shapes = expected_shapes(h->desc);
  const auto it = shapes.find(name);
  if (it == shapes.end()) return set_error(DFN_ERR_ARG, "dfn_nerfh_set_param: unknown parameter '%s'", name);
  size_t want = 1;
  for (size_t s : it->second) want *= s;
  if (want != numel)
    return set_error(DFN_ERR_ARG, "dfn_nerfh_set_param: '%s' has %zu elements, expected %zu", name, numel, want);
  h->params[name].assign(host, host + numel);
  h->committed = false;
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ packing
namespace {

struct Mat {  // a state_dict Linear
  const float* w = nullptr;
  const float* b = nullptr;
  int rows = 0, cols = 0;
};

struct Packer {
  const dfn_nerfh_s* h;
  std::string pre;
  int width() const { return h->desc.width; }
  mutable int fwd_units = 0;   // pack_bwd: number of forward units at the head of the table
  Mat mat(const std::string& key) const {
    Mat m;
    const auto& w = h->params.at(pre + key + ".weight");
    const auto& b = h->params.at(pre + key + ".bias");
    m.w = w.data();
    m.b = b.data();
    m.rows = int(b.size());
    m.cols = int(w.size() / b.size());
    return m;
  }
  // Source (matrix, row) of row i of M-block mb of a layer; row < 0 = zero row.
  void row_source(int layer, int mb, int i, Mat& m, int& row) const {
    row = 32 * mb + i;
    switch (layer) {
      case LY_L1: case LY_L2: case LY_L3: case LY_L4: case LY_L5: case LY_L6: case LY_L7: case LY_L8:
        m = mat("xyz_encoding_" + std::to_string(layer - LY_L1 + 1) + ".0");
        break;
      case LY_FIN:
        if (mb < width() / 32) m = mat("xyz_encoding_final");
        else { m = mat("static_sigma.0"); row = i == 0 ? 0 : -1; }
        break;
      case LY_DIR: m = mat("dir_encoding.0"); break;
      case LY_RGB: m = mat("static_rgb.0"); row = i < 3 ? i : -1; break;
      case LY_TE0: m = mat("transient_encoding.0"); break;
      case LY_TE1: m = mat("transient_encoding.2"); break;
      case LY_TE2: m = mat("transient_encoding.4"); break;
      case LY_TE3: m = mat("transient_encoding.6"); break;
      case LY_THEAD:
        if (i < 3) { m = mat("transient_rgb.0"); row = i; }
        else if (i == 3) { m = mat("transient_sigma.0"); row = 0; }
        else if (i == 8) { m = mat("transient_beta.0"); row = 0; }
        else { m = mat("transient_beta.0"); row = -1; }
        break;
      case LY_SIG: m = mat("static_sigma.0"); row = i == 0 ? 0 : -1; break;
    }
  }
  // Source column of slot s of half h; < 0 = zero.
  static int col_source(int layer, int hh, int s) {
    if (layer == LY_L1) return pe_xyz_feature(hh, s);
    if (layer == LY_L5) return s < 32 ? pe_xyz_feature(hh, s) : kChXyz + hidden_feature(hh, s - 32);
    return hidden_feature(hh, s);
  }
  // The bias folded per ray (DIR, TE0) is NOT packed into the unit.
  static bool unit_has_bias(int layer) { return layer != LY_DIR && layer != LY_TE0; }

  // One layer's M-blocks [mb0, mb0+group) as [A fragments][bias fragments], appended at `base`.
  // split-f16 (P::kSplit): a fragment is a hi plane then a lo plane of w * wscale; the bias is pre-multiplied by
  // wscale * kX3ActScale (what the accumulators carry).
  float wscale = 1.f;
  template <class P>
  void pack_blocks(int layer, int mb0, int group, uint8_t* base) const {
    using Elem = typename std::conditional<P::kSlotsPerChunk == 8, _Float16, float>::type;
    const LayerShape sh = layer_shape(layer, width());
    const int KC = sh.slots / P::kSlotsPerChunk;
    Elem* frag = reinterpret_cast<Elem*>(base);
    float* bias = reinterpret_cast<float*>(base + size_t(group) * KC * 64 * P::kLaneBytes);
    for (int g = 0; g < group; ++g) {
      const int mb = mb0 + g;
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, hh = lane >> 5;
        Mat m;
        int row;
        row_source(layer, mb, i, m, row);
        if (row >= m.rows) row = -1;
        for (int kc = 0; kc < KC; ++kc)
          for (int j = 0; j < P::kSlotsPerChunk; ++j) {
            const int col = col_source(layer, hh, kc * P::kSlotsPerChunk + j);
            const float v = (row >= 0 && col >= 0 && col < m.cols) ? m.w[size_t(row) * m.cols + col] : 0.f;
            if constexpr (P::kSplit) {
              Elem* fr = frag + (size_t(g) * KC + kc) * 64 * 16;   // 1024 halves per fragment: [hi plane][lo plane]
              const Elem hi = Elem(v * wscale);
              fr[lane * 8 + j] = hi;
              fr[512 + lane * 8 + j] = Elem(v * wscale - float(hi));
            } else {
              frag[((size_t(g) * KC + kc) * 64 + lane) * P::kSlotsPerChunk + j] = Elem(v);
            }
          }
      }
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) {
          Mat m;
          int row;
          row_source(layer, mb, mblock_row(hh, r), m, row);
          if (row >= m.rows) row = -1;
          const float b = (row >= 0 && unit_has_bias(layer)) ? m.b[row] : 0.f;
          bias[(g * 2 + hh) * 16 + r] = P::kSplit ? b * wscale * kX3ActScale : b;
        }
    }
  }
  // Element of the backward (W^T) layer `layer` at row i of M-block mb, contraction slot s of half hh
  // (nerfh_layout.h: BwdLayerId).  Rows are INPUT features of the forward Linear, slots its OUTPUT features.
  float bwd_elem(int layer, int mb, int i, int hh, int s) const {
    const int k = 32 * mb + i;
    auto W = [](const Mat& m, int row, int col) {
      return (row >= 0 && row < m.rows && col >= 0 && col < m.cols) ? m.w[size_t(row) * m.cols + col] : 0.f;
    };
    auto pe_col = [&](int blk) { return pe_xyz_feature(mblock_half_of_row(i), 16 * blk + mblock_reg_of_row(i)); };
    const int j = hidden_feature(hh, s);
    switch (layer) {
      case BW_THEAD:
        if (j < 3) return W(mat("transient_rgb.0"), j, k);
        if (j == 3) return W(mat("transient_sigma.0"), 0, k);
        if (j == 8) return W(mat("transient_beta.0"), 0, k);
        return 0.f;
      case BW_TE3: return W(mat("transient_encoding.6"), j, k);
      case BW_TE2: return W(mat("transient_encoding.4"), j, k);
      case BW_TE1: return W(mat("transient_encoding.2"), j, k);
      case BW_RGB: return j < 3 ? W(mat("static_rgb.0"), j, k) : 0.f;
      case BW_FINCAT: {
        const bool dirpart = s >= 32;
        const int jj = hidden_feature(hh, s & 31);
        if (mb < 4) return W(mat(dirpart ? "dir_encoding.0" : "transient_encoding.0"), jj, k);
        if (!dirpart) return 0.f;
        const int c = pe_dir_feature(mblock_half_of_row(i), mblock_reg_of_row(i));
        return c >= 0 ? W(mat("dir_encoding.0"), jj, kWidth + c) : 0.f;
      }
      case BW_FIN:
        if (s < 64) return W(mat("xyz_encoding_final"), j, k);
        return (s == 64 && hh == 0) ? W(mat("static_sigma.0"), 0, k) : 0.f;
      case BW_L5:
        if (mb < 4) return W(mat("xyz_encoding_5.0"), j, kChXyz + k);
        return W(mat("xyz_encoding_5.0"), j, pe_col(mb - 4));
      case BW_L1: return W(mat("xyz_encoding_1.0"), j, pe_col(mb));
      default: {  // BW_L8..BW_L6, BW_L4..BW_L2: 128 -> 128
        const int n = 8 - (layer - BW_L8);
        return W(mat("xyz_encoding_" + std::to_string(n) + ".0"), j, k);
      }
    }
  }
  template <class P>
  void pack_bwd_blocks(int layer, int mb0, int group, uint8_t* base) const {
    using Elem = typename std::conditional<P::kSlotsPerChunk == 8, _Float16, float>::type;
    const LayerShape sh = bwd_layer_shape(layer);
    const int KC = sh.slots / P::kSlotsPerChunk;
    Elem* frag = reinterpret_cast<Elem*>(base);  // bias fragments after the A fragments stay zero
    for (int g = 0; g < group; ++g)
      for (int lane = 0; lane < 64; ++lane)
        for (int kc = 0; kc < KC; ++kc)
          for (int j = 0; j < P::kSlotsPerChunk; ++j) {
            const float v = bwd_elem(layer, mb0 + g, lane & 31, lane >> 5, kc * P::kSlotsPerChunk + j);
            if constexpr (P::kSplit) {
              Elem* fr = frag + (size_t(g) * KC + kc) * 64 * 16;   // [hi plane][lo plane], as pack_blocks
              const Elem hi = Elem(v * wscale);
              fr[lane * 8 + j] = hi;
              fr[512 + lane * 8 + j] = Elem(v * wscale - float(hi));
            } else {
              frag[((size_t(g) * KC + kc) * 64 + lane) * P::kSlotsPerChunk + j] = Elem(v);
            }
          }
  }
  // Blob of the gradient kernel: the fine net's forward layers, one unit per layer (f16) / per M-block (f32),
  // then the backward layers the same way.
  template <class PF, class P>
  void pack_bwd(std::vector<uint8_t>& blob, std::vector<uint32_t>& tab) const {
    pack<PF>(true, bwd_fwd_unit_mb<PF>(), false, blob, tab);
    fwd_units = int(tab.size() / 2);
    const int umb = bwd_unit_mb<P>();
    for (int layer = 0; layer < BW_COUNT; ++layer) {
      const LayerShape sh = bwd_layer_shape(layer);
      for (int u0 = 0; u0 < sh.mb; u0 += umb) {
        // units of one or two M-blocks: BW_L5's two d pe M-blocks (4, 5) are staged BEFORE its four d h4 M-blocks — the kernel
        // folds d pe to three floats while only the layer's input is live (nerfh_bwd.hip)
        const int mb0 = (umb <= 2 && layer == BW_L5) ? (u0 < 2 ? 4 + u0 : u0 - 2) : u0;
        const int group = sh.mb - mb0 < umb ? sh.mb - mb0 : umb;
        const uint32_t bytes = unit_bytes<P>(sh.slots, group);
        const uint32_t off = uint32_t(blob.size());
        blob.resize(off + bytes, 0);
        tab.push_back(off);
        tab.push_back(bytes);
        pack_bwd_blocks<P>(layer, mb0, group, blob.data() + off);
      }
    }
  }

  template <class P>
  uint32_t blocks_bytes(int layer, int group) const {
    const LayerShape sh = layer_shape(layer, width());
    return uint32_t(group) * (sh.slots / P::kSlotsPerChunk) * 64 * P::kLaneBytes + uint32_t(group) * 128;
  }

  // The packed blob: staging units in execution order.  umb >= 8: a unit holds whole layers and the small
  // layers of one group (kFineGroup / kCoarseGroup) share a unit; otherwise a unit is <= umb M-blocks of a layer.
  // l5_umb: M-blocks per unit of layer 5 (< 0: the same as the other layers)
  template <class P>
  void pack(bool fine, int umb, bool merge, std::vector<uint8_t>& blob, std::vector<uint32_t>& tab, int l5_umb = -1) const {
    const int* seq = fine ? kFineSeq : kCoarseSeq;
    const int* grp = fine ? kFineGroup : kCoarseGroup;
    const int nl = fine ? kFineLayers : kCoarseLayers;
    if (merge) {
      for (int li = 0; li < nl;) {
        int lj = li;
        uint32_t bytes = 0;
        while (lj < nl && grp[lj] == grp[li]) bytes += blocks_bytes<P>(seq[lj], layer_shape(seq[lj], width()).mb), ++lj;
        if (seq[li] == LY_L5) {  // split so that three staging buffers fit in LDS (nerfh_layout.h: l5_unit_mb)
          const LayerShape sh = layer_shape(LY_L5, width());
          const int g = l5_unit_mb(umb);
          for (int mb0 = 0; mb0 < sh.mb; mb0 += g) {
            const uint32_t ub = unit_bytes<P>(sh.slots, g), uo = uint32_t(blob.size());
            blob.resize(uo + ub, 0);
            tab.push_back(uo);
            tab.push_back(ub);
            pack_blocks<P>(LY_L5, mb0, g, blob.data() + uo);
          }
          li = lj;
          continue;
        }
        const uint32_t off = uint32_t(blob.size());
        blob.resize(off + align_piece(bytes), 0);
        tab.push_back(off);
        tab.push_back(align_piece(bytes));
        uint32_t at = off;
        for (int l = li; l < lj; ++l) {
          pack_blocks<P>(seq[l], 0, layer_shape(seq[l], width()).mb, blob.data() + at);
          at += blocks_bytes<P>(seq[l], layer_shape(seq[l], width()).mb);
        }
        li = lj;
      }
      return;
    }
    for (int li = 0; li < nl; ++li) {
      const LayerShape sh = layer_shape(seq[li], width());
      const int lu = (seq[li] == LY_L5 && l5_umb > 0) ? l5_umb : umb;
      for (int mb0 = 0; mb0 < sh.mb; mb0 += lu) {
        const int group = sh.mb - mb0 < lu ? sh.mb - mb0 : lu;
        const uint32_t bytes = unit_bytes<P>(sh.slots, group);
        const uint32_t off = uint32_t(blob.size());
        blob.resize(off + bytes, 0);
        tab.push_back(off);
        tab.push_back(bytes);
        pack_blocks<P>(seq[li], mb0, group, blob.data() + off);
      }
    }
  }
};

int upload(const void* src, size_t bytes, void** dst) {
  if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return set_error(DFN_ERR_HIP, "hipMalloc(%zu) failed", bytes);
  if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
    return set_error(DFN_ERR_HIP, "hipMemcpy H2D (%zu bytes) failed", bytes);
  return DFN_OK;
}

}  // namespace

__global__ void range_fetch_kernel(int* flag) { flag[1] = atomicExch(flag, 0); }

extern "C" int dfn_nerfh_set_render_options(dfn_nerfh_t h, int flags) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_nerfh_set_render_options: null handle");
  if (flags & ~(DFN_RENDER_LINDISP | DFN_RENDER_COARSE_F16)) return set_error(DFN_ERR_UNSUPPORTED, "dfn_nerfh_set_render_options: unknown option bits 0x%x", flags);
  h->render_flags = flags;
  return DFN_OK;
}

extern "C" int dfn_nerfh_range_status(dfn_nerfh_t h, int* flags, void* stream) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_nerfh_range_status: null handle");
  int v = 0;
  if (h->range_flag) {
    // read-and-clear in ONE atomic exchange on the device: bits raised by kernels of another stream between a copy and a
    // separate clear would be lost
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(range_fetch_kernel, dim3(1), dim3(1), 0, s, h->range_flag);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&v, h->range_flag + 1, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return set_error(DFN_ERR_HIP, "dfn_nerfh_range_status: reading the flag failed");
  }
  if (flags) { *flags = v; return DFN_OK; }
  if (v)
    return set_error(DFN_ERR_RANGE, "NeRF-H render: activations left the range of the %s arithmetic since the last check (%s); the frames rendered "
                     "since then are not the network's output: render them with DFN_PREC_F32",
                     (v & DFN_RANGE_F16_OVERFLOW) ? "f16" : "split-f16",
                     (v & DFN_RANGE_F16_OVERFLOW) ? "an f16 layer output overflowed to inf: |activation| > 65504"
                                                  : "a split-f16 hi half saturated: |activation| >= 4094 (hidden layers of the render kernels: "
                                                    ">= 4094 x the network's largest |weight|, the accumulators are narrowed before they are scaled)");
  return DFN_OK;
}

extern "C" int dfn_nerfh_range_status_async(dfn_nerfh_t h, int* host_flags, void* stream) {
  if (!h || !host_flags) return set_error(DFN_ERR_ARG, "dfn_nerfh_range_status_async: null handle / destination");
  if (!h->range_flag) { *host_flags = 0; return DFN_OK; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(range_fetch_kernel, dim3(1), dim3(1), 0, s, h->range_flag);   // read-and-clear, as dfn_nerfh_range_status
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(host_flags, h->range_flag + 1, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess)
    return set_error(DFN_ERR_HIP, "dfn_nerfh_range_status_async: enqueueing the read failed");
  return DFN_OK;
}

extern "C" int dfn_nerfh_commit(dfn_nerfh_t h) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_nerfh_commit: null handle");
  for (const auto& kv : expected_shapes(h->desc))
    if (!h->params.count(kv.first)) return set_error(DFN_ERR_STATE, "dfn_nerfh_commit: parameter '%s' not set", kv.first.c_str());
  free_packed(h);
  dfn::fused::destroy_state(h);   // its tables follow the handle's geometry and scales: rebuilt on the next training step
  if (!h->range_flag) {   // range guard of the narrow arithmetic modes: three device ints (flag, fetch slot, the training step's own word), cleared here
    if (hipMalloc(reinterpret_cast<void**>(&h->range_flag), 3 * sizeof(int)) != hipSuccess) {
      h->range_flag = nullptr;
      return set_error(DFN_ERR_HIP, "dfn_nerfh_commit: allocating the range-guard flag failed");
    }
    if (hipMemset(h->range_flag, 0, 3 * sizeof(int)) != hipSuccess) return set_error(DFN_ERR_HIP, "dfn_nerfh_commit: clearing the range-guard flag failed");
  }
  {  // generic-width path: plain device copies in canonical order
    std::vector<float> all;
    std::vector<size_t> offs;
    const int np = dfn_nerfh_train_param_count();
    for (int i = 0; i < np; ++i) {
      const auto& v = h->params.at(dfn_nerfh_train_param_name(i));
      offs.push_back(all.size());
      all.insert(all.end(), v.begin(), v.end());
      all.resize((all.size() + 3) & ~size_t(3));
    }
    int rc = upload(all.data(), all.size() * 4, reinterpret_cast<void**>(&h->gen_blob));
    if (rc) return rc;
    for (size_t o : offs) h->gen_params.push_back(h->gen_blob + o);
  }
  const int Wd = h->desc.width;
  if (Wd != kWidth && Wd != kMaxWidth) {   // generic-width path only
    h->committed = true;
    return DFN_OK;
  }
  if (Wd == kMaxWidth) {   // netwidth 256: one kernel variant per precision, plain staging units (nerfh_mlp.hip: launch_mlp)
    for (int f = 0; f < 2; ++f)
      for (int prec = 0; prec < 3; ++prec) {
        Packer pk{h, f ? "fine." : "coarse."};
        std::vector<uint8_t> blob;
        std::vector<uint32_t> tab;
        PackedNet& n = h->net[f][prec][0];
        if (prec == DFN_PREC_F16) pk.pack<PrecF16>(f, unit_mb_w256<PrecF16>(), false, blob, tab);
        else if (prec == DFN_PREC_F32) pk.pack<PrecF32>(f, unit_mb_w256<PrecF32>(), false, blob, tab);
        else {
          float wmax = 0.f;
          for (const auto& kv : h->params)
            if (kv.first.compare(0, pk.pre.size(), pk.pre) == 0 && kv.first.find(".weight") != std::string::npos)
              for (float v : kv.second) wmax = std::fmax(wmax, std::fabs(v));
          int sexp = wmax > 0.f ? -int(std::ceil(std::log2(wmax))) : 0;
          sexp = sexp < -8 ? -8 : (sexp > 24 ? 24 : sexp);
          pk.wscale = std::ldexp(1.f, sexp);
          n.in_scale = pk.wscale * kX3ActScale;
          pk.pack<PrecX3>(f, unit_mb_w256<PrecX3>(), false, blob, tab);
        }
        int rc = upload(blob.data(), blob.size(), reinterpret_cast<void**>(&n.blob));
        if (rc) return rc;
        rc = upload(tab.data(), tab.size() * 4, reinterpret_cast<void**>(&n.tab));
        if (rc) return rc;
        n.n_units = int(tab.size() / 2);
      }
  } else {
  for (int f = 0; f < 2; ++f)
    for (int prec = 0; prec < 3; ++prec)
      for (int var = 0; var < kVariants; ++var) {
        Packer pk{h, f ? "fine." : "coarse."};
        std::vector<uint8_t> blob;
        std::vector<uint32_t> tab;
        PackedNet& n = h->net[f][prec][var];
        if (prec == DFN_PREC_F16) pk.pack<PrecF16>(f, unit_mb<PrecF16>(var), unit_mb<PrecF16>(var) >= 8, blob, tab);
        else if (prec == DFN_PREC_F32) pk.pack<PrecF32>(f, unit_mb<PrecF32>(var), false, blob, tab);
        else {
          // one power-of-two weight scale per network: the largest |w| lands in (0.5, 1].  The lo halves of most weights are then f16
          // subnormals — still exact to 2^-24 of the largest weight (the matrix cores do not flush f16 denormals), which is fp32's
          // resolution — and the accumulators (in_scale x the value = 16 x value / max|w|) fit f16 up to |value| = 4094 max|w|: that is
          // what lets the render kernels narrow BEFORE they scale (X3Piece: one packed multiply per result pair instead of two)
          float wmax = 0.f;
          for (const auto& kv : h->params)
            if (kv.first.compare(0, pk.pre.size(), pk.pre) == 0 && kv.first.find(".weight") != std::string::npos)
              for (float v : kv.second) wmax = std::fmax(wmax, std::fabs(v));
          int sexp = wmax > 0.f ? -int(std::ceil(std::log2(wmax))) : 0;
          sexp = sexp < -8 ? -8 : (sexp > 24 ? 24 : sexp);
          pk.wscale = std::ldexp(1.f, sexp);
          n.in_scale = pk.wscale * kX3ActScale;
          pk.pack<PrecX3>(f, unit_mb<PrecX3>(var), false, blob, tab, l5_unit_mb_p<PrecX3>(unit_mb<PrecX3>(var)));
        }
        int rc = upload(blob.data(), blob.size(), reinterpret_cast<void**>(&n.blob));
        if (rc) return rc;
        rc = upload(tab.data(), tab.size() * 4, reinterpret_cast<void**>(&n.tab));
        if (rc) return rc;
        n.n_units = int(tab.size() / 2);
      }
  for (int prec = 0; prec < 3; ++prec) {
    Packer pk{h, "fine."};
    std::vector<uint8_t> blob;
    std::vector<uint32_t> tab;
    PackedNet& n = h->bwd[prec];
    if (prec == DFN_PREC_F16) pk.pack_bwd<PrecF16, PrecF16>(blob, tab);
    else if (prec == DFN_PREC_F32) pk.pack_bwd<PrecF32, PrecF32>(blob, tab);
    else {
      // the gradient kernels scale their accumulators in fp32 BEFORE narrowing them (store_hidden), so their weights can sit where
      // every lo half is a normal f16 (largest |w| near 2^10: 22 bits per weight) instead of at the render kernels' unit scale
      n.in_scale = h->net[1][2][0].in_scale * 1024.f;
      pk.wscale = n.in_scale / kX3ActScale;
      pk.pack_bwd<PrecX3, PrecX3>(blob, tab);
    }
    int rc = upload(blob.data(), blob.size(), reinterpret_cast<void**>(&n.blob));
    if (rc) return rc;
    rc = upload(tab.data(), tab.size() * 4, reinterpret_cast<void**>(&n.tab));
    if (rc) return rc;
    n.n_units = int(tab.size() / 2);
    n.n_fwd_units = pk.fwd_units;
  }
  }   // netwidth 128
  // per-ray-bias weights: transposed tails of dir_encoding.0 / transient_encoding.0 + embeddings
  const dfn_nerfh_desc& d = h->desc;
  const int na = d.hist_bin * d.dim_a, nt = d.hist_bin * d.dim_t, kd = kChDir + na, NO = Wd / 2;
  const auto& wd = h->params.at("fine.dir_encoding.0.weight");
  const auto& bd = h->params.at("fine.dir_encoding.0.bias");
  const auto& wt = h->params.at("fine.transient_encoding.0.weight");
  const auto& bt = h->params.at("fine.transient_encoding.0.bias");
  const auto& ea = h->params.at("embedding_a.weight");
  const auto& et = h->params.at("embedding_t.weight");
  std::vector<float> ex;
  const size_t o_wd = 0, o_bd = o_wd + size_t(kd) * NO, o_wt = o_bd + NO, o_bt = o_wt + size_t(nt) * NO,
               o_ea = o_bt + NO, o_et = o_ea + ea.size();
  ex.resize(o_et + et.size());
  const int ld_d = Wd + kd, ld_t = Wd + nt;
  for (int j = 0; j < kd; ++j)
    for (int f = 0; f < NO; ++f) ex[o_wd + size_t(j) * NO + f] = wd[size_t(f) * ld_d + Wd + j];
  for (int j = 0; j < nt; ++j)
    for (int f = 0; f < NO; ++f) ex[o_wt + size_t(j) * NO + f] = wt[size_t(f) * ld_t + Wd + j];
  std::memcpy(&ex[o_bd], bd.data(), NO * 4);
  std::memcpy(&ex[o_bt], bt.data(), NO * 4);
  std::memcpy(&ex[o_ea], ea.data(), ea.size() * 4);
  std::memcpy(&ex[o_et], et.data(), et.size() * 4);
  int rc = upload(ex.data(), ex.size() * 4, reinterpret_cast<void**>(&h->extra));
  if (rc) return rc;
  h->rb.w_dir = h->extra + o_wd;
  h->rb.b_dir = h->extra + o_bd;
  h->rb.w_tr = h->extra + o_wt;
  h->rb.b_tr = h->extra + o_bt;
  h->rb.emb_a = h->extra + o_ea;
  h->rb.emb_t = h->extra + o_et;
  h->rb.hist_bin = d.hist_bin;
  h->rb.dim_a = d.dim_a;
  h->rb.dim_t = d.dim_t;
  h->rb.n_vocab = d.n_vocab;
  h->rb.nout = NO;
  h->committed = true;
  h->fast = true;
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ profiling aid
namespace {
struct KernelTimer {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[DFN_PROF_SLOTS];
} g_prof;
struct ScopedTimer {
  int which;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(int w, hipStream_t st) : which(w), s(st) {
    if (!g_prof.on) return;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
  }
  ~ScopedTimer() {
    if (!a) return;
    (void)hipEventRecord(b, s);
    g_prof.ev[which].emplace_back(a, b);
  }
};
}  // namespace

extern "C" int dfn_profile_enable(int on) {
  for (auto& v : g_prof.ev) {
    for (auto& p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    v.clear();
  }
  g_prof.on = on != 0;
  return DFN_OK;
}
extern "C" int dfn_profile_read(int which, double* avg_ms, int* launches) {
  if (which < 0 || which >= DFN_PROF_SLOTS || !avg_ms || !launches) return set_error(DFN_ERR_ARG, "dfn_profile_read: bad argument");
  double tot = 0;
  for (auto& p : g_prof.ev[which]) {
    float ms = 0;
    if (hipEventSynchronize(p.second) != hipSuccess || hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess)
      return set_error(DFN_ERR_HIP, "dfn_profile_read: event query failed");
    tot += ms;
  }
  *launches = int(g_prof.ev[which].size());
  *avg_ms = *launches ? tot / *launches : 0.0;
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ stage entry points
#define HS(s) reinterpret_cast<hipStream_t>(s)
#define CHECK_HIP(expr, what)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return set_error(DFN_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

static int check_net(dfn_nerfh_t h, int prec, const char* fn, bool allow_x3 = false) {
  if (!h) return set_error(DFN_ERR_ARG, "%s: null handle", fn);
  if (!h->committed) return set_error(DFN_ERR_STATE, "%s: dfn_nerfh_commit() has not been called", fn);
  if (!h->fast)
    return set_error(DFN_ERR_UNSUPPORTED, "%s: the register-resident kernels exist for netwidth %d and %d (this handle: %d); "
                     "use the generic-width entry points (dfn_nerfh_generic_*)", fn, kWidth, kMaxWidth, h->desc.width);
  if (prec != DFN_PREC_F16 && prec != DFN_PREC_F32 && !(allow_x3 && prec == DFN_PREC_F16X3))
    return set_error(DFN_ERR_ARG, "%s: unknown / unsupported precision %d", fn, prec);
  return DFN_OK;
}

// Gradient entry points: fp32-grade arithmetic only (DFN_PREC_F32, DFN_PREC_F16X3).
static int check_grad_width(dfn_nerfh_t h, const char* fn) {
  if (h->desc.width != kWidth)
    return set_error(DFN_ERR_UNSUPPORTED, "%s: the input-gradient kernels are built for netwidth %d only (this handle: %d)", fn, kWidth,
                     h->desc.width);
  return DFN_OK;
}
static int check_grad_prec(int prec, const char* fn) {
  if (prec == DFN_PREC_F16)
    return set_error(DFN_ERR_UNSUPPORTED, "%s: gradients run in DFN_PREC_F16X3 or DFN_PREC_F32 only (plain-f16 ReLU gates are 3e-2 off "
                     "autograd)", fn);
  return DFN_OK;
}

// Kernel variant: DFN_MLP_VARIANT=0|1|2 (A/B aid, see nerfh_layout.h).
static int mlp_variant_128();
static int mlp_variant_of(dfn_nerfh_t h) { return h->desc.width == kWidth ? mlp_variant_128() : 0; }
static int mlp_variant_128() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DFN_MLP_VARIANT");
    v = (e && e[0] >= '0' && e[0] < '0' + kVariants) ? e[0] - '0' : 0;
  }
  return v;
}

// The handle's range-guard flag (device int): allocated and cleared by dfn_nerfh_commit, which fails loudly when it cannot be —
// every kernel launch that carries the guard runs on a committed handle.
static int* range_flag_of(dfn_nerfh_t h) { return h->range_flag; }

static unsigned long long* g_timing_buf = nullptr;  // DFN_TIMING builds only (tools/gpu_timing.py)
extern "C" void dfn_debug_set_timing_buffer(void* p) { g_timing_buf = static_cast<unsigned long long*>(p); }


extern "C" int dfn_raygen(int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d,
                          float* viewdirs, void* stream) {
  if (H < 0 || W < 0 || !c2w || !rays_o || !rays_d || !(focal > 0)) return set_error(DFN_ERR_ARG, "dfn_raygen: bad argument");
  CHECK_HIP(launch_raygen(H, W, focal, c2w, rays_o, rays_d, viewdirs, HS(stream)), "dfn_raygen");
  return DFN_OK;
}

extern "C" int dfn_posenc(const float* x, size_t n, int L, int mode, float* out, void* stream) {
  if (!x || !out || L < 0 || L > 16 || (mode != 0 && mode != 1)) return set_error(DFN_ERR_ARG, "dfn_posenc: bad argument");
  CHECK_HIP(launch_posenc(x, n, L, mode, out, HS(stream)), "dfn_posenc");
  return DFN_OK;
}

extern "C" int dfn_mlp_coarse(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, size_t n_rays,
                              int Nc, float near, float far, float* sigma, void* stream) {
  if (int rc = check_net(h, prec, "dfn_mlp_coarse", true)) return rc;
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !sigma || Nc < 1) return set_error(DFN_ERR_ARG, "dfn_mlp_coarse: bad argument");
  const PackedNet& n = h->net[0][prec][mlp_variant_of(h)];
  MlpArgs a{n.blob, n.tab, n.n_units, rays_o, rays_d, nullptr, nullptr, sigma, nullptr, (long long)n_rays, Nc, near, far, nullptr, n.in_scale,
            h->render_flags & DFN_RENDER_LINDISP};
  a.status = range_flag_of(h);
  ScopedTimer t(0, HS(stream));
  CHECK_HIP(launch_mlp(false, prec, mlp_variant_of(h), a, device_cu_count(), HS(stream), h->desc.width), "dfn_mlp_coarse");
  return DFN_OK;
}

extern "C" int dfn_coarse_weights(const float* sigma, const float* z, size_t n, int N, float* weights, void* stream) {
  if (!sigma || !z || !weights || N < 1 || N > 2048) return set_error(DFN_ERR_ARG, "dfn_coarse_weights: bad argument");
  CHECK_HIP(launch_coarse_weights(sigma, z, n, N, weights, HS(stream)), "dfn_coarse_weights");
  return DFN_OK;
}

extern "C" int dfn_sample_pdf(const float* bins, const float* weights, size_t n, int nb, int Ni, const float* u,
                              float* out, void* stream) {
  if (!bins || !weights || !out || nb < 2 || Ni < 1 || 3 * nb + 2 * Ni > 8192)
    return set_error(DFN_ERR_ARG, "dfn_sample_pdf: bad argument");
  CHECK_HIP(launch_sample_pdf(bins, weights, n, nb, Ni, u, out, HS(stream)), "dfn_sample_pdf");
  return DFN_OK;
}

extern "C" int dfn_sample_fine_opt(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far, int render_flags,
                                   float* z_fine, float* weights_coarse, float* z_samples, void* stream) {
  if (!sigma || !z_fine || Nc < 3 || Ni < 1 || 6 * Nc + 2 * Ni > 8192)
    return set_error(DFN_ERR_ARG, "dfn_sample_fine: bad argument (need N_samples >= 3, N_importance >= 1)");
  if ((render_flags & DFN_RENDER_LINDISP) && !(near > 0.f))
    return set_error(DFN_ERR_ARG, "dfn_sample_fine: lindisp needs near > 0");
  CHECK_HIP(launch_sample_fine(sigma, n_rays, Nc, Ni, near, far, z_fine, weights_coarse, z_samples, HS(stream),
                               render_flags & DFN_RENDER_LINDISP),
            "dfn_sample_fine");
  return DFN_OK;
}
extern "C" int dfn_sample_fine(const float* sigma, size_t n_rays, int Nc, int Ni, float near, float far,
                               float* z_fine, float* weights_coarse, float* z_samples, void* stream) {
  return dfn_sample_fine_opt(sigma, n_rays, Nc, Ni, near, far, 0, z_fine, weights_coarse, z_samples, stream);
}

extern "C" int dfn_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, size_t n, float* out_o,
                            float* out_d, void* stream) {
  if (!n) return DFN_OK;
  if (!rays_o || !rays_d || !out_o || !out_d || H < 1 || W < 1 || !(focal > 0.f)) return set_error(DFN_ERR_ARG, "dfn_ndc_rays: bad argument");
  CHECK_HIP(launch_ndc_rays(H, W, focal, near, rays_o, rays_d, n, out_o, out_d, HS(stream)), "dfn_ndc_rays");
  return DFN_OK;
}

extern "C" int dfn_upsample_bicubic(const float* in, int H, int W, int C, int outH, int outW, float* out, void* stream) {
  if (!in || !out || H < 1 || W < 1 || C < 1 || outH < 1 || outW < 1) return set_error(DFN_ERR_ARG, "dfn_upsample_bicubic: bad argument");
  CHECK_HIP(launch_bicubic(in, H, W, C, outH, outW, out, HS(stream)), "dfn_upsample_bicubic");
  return DFN_OK;
}

extern "C" int dfn_upsample_bicubic_frames(const float* in, int B, int H, int W, int C, int outH, int outW, int out_nchw, float* out,
                                           void* stream) {
  if (!in || !out || B < 1 || H < 1 || W < 1 || C < 1 || outH < 1 || outW < 1) return set_error(DFN_ERR_ARG, "dfn_upsample_bicubic_frames: bad argument");
  CHECK_HIP(launch_bicubic(in, H, W, C, outH, outW, out, HS(stream), B, out_nchw != 0), "dfn_upsample_bicubic_frames");
  return DFN_OK;
}
extern "C" int dfn_upsample_bicubic_frames_backward(const float* grad_out, int B, int H, int W, int C, int outH, int outW, int out_nchw,
                                                    float* grad_in, void* stream) {
  if (!grad_out || !grad_in || B < 1 || H < 1 || W < 1 || C < 1 || outH < 1 || outW < 1)
    return set_error(DFN_ERR_ARG, "dfn_upsample_bicubic_frames_backward: bad argument");
  CHECK_HIP(launch_bicubic_backward(grad_out, H, W, C, outH, outW, grad_in, HS(stream), B, out_nchw != 0), "dfn_upsample_bicubic_frames_backward");
  return DFN_OK;
}

extern "C" int dfn_upsample_bicubic_backward(const float* grad_out, int H, int W, int C, int outH, int outW, float* grad_in,
                                             void* stream) {
  if (!grad_out || !grad_in || H < 1 || W < 1 || C < 1 || outH < 1 || outW < 1)
    return set_error(DFN_ERR_ARG, "dfn_upsample_bicubic_backward: bad argument");
  CHECK_HIP(launch_bicubic_backward(grad_out, H, W, C, outH, outW, grad_in, HS(stream)), "dfn_upsample_bicubic_backward");
  return DFN_OK;
}

extern "C" size_t dfn_fine_bias_bytes(size_t n_rays) { return (n_rays ? n_rays : 1) * ray_bias_floats(kMaxWidth) * sizeof(float); }

extern "C" int dfn_mlp_fine(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                            const float* hist, size_t hist_rows, size_t n_rays, const float* z_fine, int Nf,
                            float* raw, void* bias_ws, void* stream) {
  if (int rc = check_net(h, prec, "dfn_mlp_fine", true)) return rc;
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !viewdirs || !hist || !z_fine || !raw || !bias_ws || Nf < 1 ||
      (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_mlp_fine: bad argument (hist_rows must be 1 or n_rays)");
  float* table = static_cast<float*>(bias_ws);
  CHECK_HIP(launch_ray_bias(h->rb, viewdirs, hist, hist_rows, n_rays, table, HS(stream)), "dfn_mlp_fine(ray_bias)");
  const PackedNet& n = h->net[1][prec][mlp_variant_of(h)];
  MlpArgs a{n.blob, n.tab, n.n_units, rays_o, rays_d, z_fine, table, raw, nullptr, (long long)n_rays, Nf, 0.f, 0.f, g_timing_buf, n.in_scale};
  a.status = range_flag_of(h);
  ScopedTimer t(1, HS(stream));
  CHECK_HIP(launch_mlp(true, prec, mlp_variant_of(h), a, device_cu_count(), HS(stream), h->desc.width), "dfn_mlp_fine");
  return DFN_OK;
}

extern "C" int dfn_composite_fine(const float* raw, const float* z, size_t n_rays, int Nf, float beta_min, int flags,
                                  float* rgb, float* disp, float* acc, float* depth, float* weights, float* beta,
                                  void* stream) {
  if (!raw || !z || !rgb || !disp || !acc || Nf < 1 || Nf > 512)
    return set_error(DFN_ERR_ARG, "dfn_composite_fine: bad argument (1 <= Nf <= 512)");
  CHECK_HIP(launch_composite_fine(raw, z, n_rays, Nf, beta_min, flags, rgb, disp, acc, depth, weights, beta, HS(stream)),
            "dfn_composite_fine");
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ whole path
namespace {
constexpr size_t kMaxChunkRays = 65536;  // rays per internal pass (bounds the raw buffer)
inline size_t al(size_t b) { return (b + 255) & ~size_t(255); }
// Equal passes of at most kMaxChunkRays rays (multiple of 64 rays so MLP tiles stay aligned).
inline size_t chunk_rays(size_t n_rays) {
  const size_t passes = (n_rays + kMaxChunkRays - 1) / kMaxChunkRays;
  const size_t c = passes ? (n_rays + passes - 1) / passes : 1;
  return (c + 63) & ~size_t(63);
}
struct Workspace {
  float *o, *d, *v, *sigma, *z, *raw, *bias, *partial;
  size_t total;
};
Workspace carve(char* base, size_t n_rays, int Nc, int Ni, bool own_rays) {
  const size_t chunk = chunk_rays(n_rays);
  const size_t Nf = size_t(Nc) + Ni;
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return reinterpret_cast<float*>(p); };
  w.o = take(own_rays ? n_rays * 12 : 0);
  w.d = take(own_rays ? n_rays * 12 : 0);
  w.v = take(n_rays * 12);
  w.sigma = take(chunk * Nc * 4);
  w.z = take(chunk * Nf * 4);
  w.raw = take(chunk * Nf * 9 * 4);
  w.bias = take(chunk * ray_bias_floats(kMaxWidth) * 4);
  w.partial = take(chunk * ((Nf + 31) / 32) * 12 * 4);
  w.total = off;
  return w;
}

int render_core(dfn_nerfh_t h, int prec, const float* o, const float* d, const float* v, const float* hist,
                size_t hist_rows, size_t n_rays, int Nc, int Ni, float near, float far, float* rgb, float* disp,
                float* acc, float* raw_out, const Workspace& w, hipStream_t s) {
  const int Nf = Nc + Ni;
  // Compositing is fused into the fine kernel when a wave's points are one ray segment and raw is not wanted: 64 samples per wave in
  // the f16 variants 0/1/3, 32 in the split-f16 and fp32 kernels (one point block per wave); the 3-block f16 variant and netwidth
  // 256 keep the separate compositor.
  const int var = mlp_variant_of(h);
  const int seg = prec == DFN_PREC_F16 ? 64 : 32;
  const bool fused = !raw_out && Nf % seg == 0 && !(prec == DFN_PREC_F16 && var == 2) && h->desc.width == kWidth;
  const int cprec = (h->render_flags & DFN_RENDER_COARSE_F16) ? DFN_PREC_F16 : prec;   // the coarse pass only places the fine samples
  const PackedNet& nc = h->net[0][cprec][var];
  const PackedNet& nf = h->net[1][prec][var];
  const int cus = device_cu_count();
  const size_t chunk = chunk_rays(n_rays);
  for (size_t r0 = 0; r0 < n_rays; r0 += chunk) {
    const size_t n = n_rays - r0 < chunk ? n_rays - r0 : chunk;
    const float* co = o + r0 * 3;
    const float* cd = d + r0 * 3;
    const float* cv = v + r0 * 3;
    const float* ch = hist_rows == 1 ? hist : hist + r0 * h->desc.hist_bin;
    float* raw = raw_out ? raw_out + r0 * size_t(Nf) * 9 : w.raw;
    {
      MlpArgs a{nc.blob, nc.tab, nc.n_units, co, cd, nullptr, nullptr, w.sigma, nullptr, (long long)n, Nc, near, far, nullptr, nc.in_scale,
                 h->render_flags & DFN_RENDER_LINDISP};
      a.status = range_flag_of(h);
      ScopedTimer t(0, s);
      CHECK_HIP(launch_mlp(false, cprec, var, a, cus, s, h->desc.width), "render: coarse MLP");
    }
    {
      ScopedTimer t(DFN_PROF_SAMPLE_FINE, s);
      CHECK_HIP(launch_sample_fine(w.sigma, n, Nc, Ni, near, far, w.z, nullptr, nullptr, s, h->render_flags & DFN_RENDER_LINDISP),
                "render: sample_fine");
    }
    {
      ScopedTimer t(DFN_PROF_RAY_BIAS, s);
      CHECK_HIP(launch_ray_bias(h->rb, cv, ch, hist_rows, n, w.bias, s), "render: ray_bias");
    }
    {
      MlpArgs a{nf.blob, nf.tab, nf.n_units, co, cd, w.z, w.bias, raw, fused ? w.partial : nullptr, (long long)n, Nf, 0.f, 0.f, g_timing_buf, nf.in_scale};
      a.status = range_flag_of(h);
      ScopedTimer t(1, s);
      CHECK_HIP(launch_mlp(true, prec, var, a, cus, s, h->desc.width), "render: fine MLP");
    }
    if (fused) {
      ScopedTimer t(DFN_PROF_COMBINE, s);
      CHECK_HIP(launch_composite_combine(w.partial, n, Nf / seg, 0.1f, DFN_COMP_TEST_TIME | DFN_COMP_STATIC_ONLY, rgb + r0 * 3,
                                         disp + r0, acc + r0, s),
                "render: composite combine");
    } else {
      ScopedTimer t(DFN_PROF_COMPOSITE, s);
      CHECK_HIP(launch_composite_fine(raw, w.z, n, Nf, 0.1f, DFN_COMP_TEST_TIME | DFN_COMP_STATIC_ONLY, rgb + r0 * 3,
                                      disp + r0, acc + r0, nullptr, nullptr, nullptr, s),
                "render: composite");
    }
  }
  return DFN_OK;
}

int check_render_args(int Nc, int Ni, const char* fn) {
  if (Nc < 3 || Ni < 1 || Nc + Ni > 512 || 6 * Nc + 2 * Ni > 8192)
    return set_error(DFN_ERR_UNSUPPORTED, "%s: need 3 <= N_samples, 1 <= N_importance, N_samples+N_importance <= 512", fn);
  return DFN_OK;
}
}  // namespace

extern "C" size_t dfn_render_workspace_bytes(size_t n_rays, int Nc, int Ni) {
  return carve(nullptr, n_rays ? n_rays : 1, Nc, Ni, true).total;
}

extern "C" int dfn_render_rays(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                               const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near, float far,
                               float* rgb, float* disp, float* acc, float* raw, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (int rc = check_net(h, prec, "dfn_render_rays", true)) return rc;
  if (int rc = check_render_args(Nc, Ni, "dfn_render_rays")) return rc;
  if (!n_rays) return DFN_OK;  // an empty batch is valid (its buffers may be null)
  if (!rays_o || !rays_d || !hist || !rgb || !disp || !acc || !workspace || (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_render_rays: bad argument (hist_rows must be 1 or n_rays)");
  if (!n_rays) return DFN_OK;
  const Workspace w = carve(static_cast<char*>(workspace), n_rays, Nc, Ni, true);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_render_rays: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  const float* v = viewdirs;
  if (!v) {
    CHECK_HIP(launch_viewdirs(rays_d, n_rays, w.v, HS(stream)), "dfn_render_rays: viewdirs");
    v = w.v;
  }
  return render_core(h, prec, rays_o, rays_d, v, hist, hist_rows, n_rays, Nc, Ni, near, far, rgb, disp, acc, raw, w,
                     HS(stream));
}

extern "C" int dfn_render_image(dfn_nerfh_t h, int prec, const float* c2w, int H, int W, float focal, float near,
                                float far, int Nc, int Ni, const float* hist, float* rgb, float* disp, float* acc,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_net(h, prec, "dfn_render_image", true)) return rc;
  if (int rc = check_render_args(Nc, Ni, "dfn_render_image")) return rc;
  if (!c2w || !hist || !rgb || !disp || !acc || !workspace || H < 1 || W < 1 || !(focal > 0))
    return set_error(DFN_ERR_ARG, "dfn_render_image: bad argument");
  const size_t n_rays = size_t(H) * W;
  const Workspace w = carve(static_cast<char*>(workspace), n_rays, Nc, Ni, true);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_render_image: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  CHECK_HIP(launch_raygen(H, W, focal, c2w, w.o, w.d, w.v, HS(stream)), "dfn_render_image: raygen");
  return render_core(h, prec, w.o, w.d, w.v, hist, 1, n_rays, Nc, Ni, near, far, rgb, disp, acc, nullptr, w, HS(stream));
}

// ------------------------------------------------------------------------------------------ gradient path
extern "C" int dfn_composite_fine_backward(const float* raw, const float* z, const float* grad_rgb, size_t n_rays, int Nf,
                                           float* grad_raw, void* stream) {
  if (!raw || !z || !grad_rgb || !grad_raw || Nf < 1 || Nf > 512)
    return set_error(DFN_ERR_ARG, "dfn_composite_fine_backward: bad argument (1 <= Nf <= 512)");
  CHECK_HIP(launch_composite_fine_backward(raw, z, grad_rgb, n_rays, Nf, grad_raw, HS(stream)), "dfn_composite_fine_backward");
  return DFN_OK;
}

extern "C" int dfn_mlp_fine_backward(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                                     const float* hist, size_t hist_rows, size_t n_rays, const float* z_fine, int Nf,
                                     const float* grad_raw, float* grad_pts, void* bias_ws, void* stream) {
  if (int rc = check_net(h, prec, "dfn_mlp_fine_backward", true)) return rc;
  if (int rc = check_grad_prec(prec, "dfn_mlp_fine_backward")) return rc;
  if (int rc = check_grad_width(h, "dfn_mlp_fine_backward")) return rc;
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !viewdirs || !hist || !z_fine || !grad_raw || !grad_pts || !bias_ws || Nf < 1 ||
      (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_mlp_fine_backward: bad argument (hist_rows must be 1 or n_rays)");
  float* table = static_cast<float*>(bias_ws);
  CHECK_HIP(launch_ray_bias(h->rb, viewdirs, hist, hist_rows, n_rays, table, HS(stream)), "dfn_mlp_fine_backward(ray_bias)");
  const PackedNet& n = h->bwd[prec];
  BwdArgs a{n.blob, n.tab, n.n_units, rays_o, rays_d, viewdirs, z_fine, table, grad_raw, grad_pts, (long long)n_rays, Nf, n.in_scale};
  CHECK_HIP(launch_mlp_fine_backward(prec, a, device_cu_count(), HS(stream)), "dfn_mlp_fine_backward");
  return DFN_OK;
}

extern "C" size_t dfn_mlp_fine_mask_bytes(size_t n_points) {
  return (n_points + kBwdTilePoints - 1) / kBwdTilePoints * (kBwdTilePoints / 32) * kBwdMaskWords * 64 * sizeof(uint32_t);
}

extern "C" int dfn_mlp_fine_saving(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                                   const float* hist, size_t hist_rows, size_t n_rays, const float* z_fine, int Nf, float* raw,
                                   void* masks, void* bias_ws, void* stream) {
  if (int rc = check_net(h, prec, "dfn_mlp_fine_saving", true)) return rc;
  if (int rc = check_grad_width(h, "dfn_mlp_fine_saving")) return rc;
  if (prec != DFN_PREC_F16X3) return set_error(DFN_ERR_UNSUPPORTED, "dfn_mlp_fine_saving: split-f16 (DFN_PREC_F16X3) only");
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !viewdirs || !hist || !z_fine || !raw || !masks || !bias_ws || Nf < 1 ||
      (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_mlp_fine_saving: bad argument (hist_rows must be 1 or n_rays)");
  float* table = static_cast<float*>(bias_ws);
  CHECK_HIP(launch_ray_bias(h->rb, viewdirs, hist, hist_rows, n_rays, table, HS(stream)), "dfn_mlp_fine_saving(ray_bias)");
  // the test-time fine kernel itself (pipelined epilogue, unit-scale weights), recording the ReLU signs as it goes
  const PackedNet& n = h->net[1][prec][0];
  MlpArgs a{n.blob, n.tab, n.n_units, rays_o, rays_d, z_fine, table, raw, nullptr, (long long)n_rays, Nf, 0.f, 0.f, nullptr, n.in_scale};
  a.status = range_flag_of(h);
  a.masks = static_cast<uint32_t*>(masks);
  CHECK_HIP(launch_mlp(true, prec, 0, a, device_cu_count(), HS(stream), h->desc.width), "dfn_mlp_fine_saving");
  return DFN_OK;
}

extern "C" int dfn_mlp_fine_backward_saved(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                                           size_t n_rays, const float* z_fine, int Nf, const float* raw, const void* masks,
                                           const float* grad_raw, float* grad_pts, void* stream) {
  if (int rc = check_net(h, prec, "dfn_mlp_fine_backward_saved", true)) return rc;
  if (int rc = check_grad_width(h, "dfn_mlp_fine_backward_saved")) return rc;
  if (prec != DFN_PREC_F16X3) return set_error(DFN_ERR_UNSUPPORTED, "dfn_mlp_fine_backward_saved: split-f16 (DFN_PREC_F16X3) only");
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !viewdirs || !z_fine || !raw || !masks || !grad_raw || !grad_pts || Nf < 1)
    return set_error(DFN_ERR_ARG, "dfn_mlp_fine_backward_saved: bad argument");
  const PackedNet& n = h->bwd[prec];
  BwdArgs a{n.blob, n.tab + 2 * n.n_fwd_units, n.n_units - n.n_fwd_units, rays_o, rays_d, viewdirs, z_fine, nullptr, grad_raw, grad_pts,
            (long long)n_rays, Nf, n.in_scale, nullptr, raw, static_cast<uint32_t*>(const_cast<void*>(masks))};
  CHECK_HIP(launch_mlp_fine_backward(prec, a, device_cu_count(), HS(stream), 2), "dfn_mlp_fine_backward_saved");
  return DFN_OK;
}

extern "C" int dfn_ray_grad_reduce(const float* grad_pts, const float* z_fine, const float* rays_d, size_t n_rays, int Nf,
                                   int derive_viewdirs, float* grad_rays_o, float* grad_rays_d, float* grad_viewdirs,
                                   void* stream) {
  if (!grad_pts || !z_fine || !rays_d || !grad_rays_o || !grad_rays_d || Nf < 1)
    return set_error(DFN_ERR_ARG, "dfn_ray_grad_reduce: bad argument");
  CHECK_HIP(launch_ray_grad_reduce(grad_pts, z_fine, rays_d, n_rays, Nf, derive_viewdirs, grad_rays_o, grad_rays_d, grad_viewdirs,
                                   HS(stream)),
            "dfn_ray_grad_reduce");
  return DFN_OK;
}

extern "C" int dfn_raygen_frames(int B, int H, int W, float focal, const float* c2w, float* rays_o, float* rays_d, float* viewdirs,
                                 void* stream) {
  if (B < 1 || H < 0 || W < 0 || !c2w || !rays_o || !rays_d || !(focal > 0)) return set_error(DFN_ERR_ARG, "dfn_raygen_frames: bad argument");
  CHECK_HIP(launch_raygen(H, W, focal, c2w, rays_o, rays_d, viewdirs, HS(stream), B), "dfn_raygen_frames");
  return DFN_OK;
}
extern "C" int dfn_raygen_frames_backward(int B, int H, int W, float focal, const float* grad_rays_o, const float* grad_rays_d,
                                          float* grad_c2w, void* stream) {
  if (B < 1 || H < 1 || W < 1 || !grad_rays_o || !grad_rays_d || !grad_c2w || !(focal > 0))
    return set_error(DFN_ERR_ARG, "dfn_raygen_frames_backward: bad argument");
  CHECK_HIP(launch_raygen_backward(H, W, focal, grad_rays_o, grad_rays_d, grad_c2w, HS(stream), B), "dfn_raygen_frames_backward");
  return DFN_OK;
}

extern "C" int dfn_raygen_backward(int H, int W, float focal, const float* grad_rays_o, const float* grad_rays_d, float* grad_c2w,
                                   void* stream) {
  if (H < 1 || W < 1 || !(focal > 0) || !grad_rays_o || !grad_rays_d || !grad_c2w)
    return set_error(DFN_ERR_ARG, "dfn_raygen_backward: bad argument");
  CHECK_HIP(launch_raygen_backward(H, W, focal, grad_rays_o, grad_rays_d, grad_c2w, HS(stream)), "dfn_raygen_backward");
  return DFN_OK;
}

namespace {
struct BwdWorkspace {
  Workspace f;
  float *graw, *gpts, *go, *gd;
  size_t total;
};
BwdWorkspace carve_bwd(char* base, size_t n_rays, int Nc, int Ni) {
  BwdWorkspace b{};
  b.f = carve(base, n_rays, Nc, Ni, true);
  const size_t chunk = chunk_rays(n_rays), Nf = size_t(Nc) + Ni;
  size_t off = b.f.total;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return reinterpret_cast<float*>(p); };
  b.graw = take(chunk * Nf * 9 * 4);
  b.gpts = take(chunk * Nf * 6 * 4);
  b.go = take(n_rays * 12);
  b.gd = take(n_rays * 12);
  b.total = off;
  return b;
}

// Forward recompute (coarse -> sampler -> fine raw) then the gradient kernels, chunk by chunk.
int render_backward_core(dfn_nerfh_t h, int prec, const float* o, const float* d, const float* v, bool derive_v,
                         const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near, float far,
                         const float* grad_rgb, float* go, float* gd, float* gv, const BwdWorkspace& w, hipStream_t s) {
  const int Nf = Nc + Ni, var = mlp_variant_of(h), cus = device_cu_count();
  const PackedNet& nc = h->net[0][prec][var];
  const PackedNet& nf = h->net[1][prec][var];
  const PackedNet& nb = h->bwd[prec];
  const size_t chunk = chunk_rays(n_rays);
  for (size_t r0 = 0; r0 < n_rays; r0 += chunk) {
    const size_t n = n_rays - r0 < chunk ? n_rays - r0 : chunk;
    const float* co = o + r0 * 3;
    const float* cd = d + r0 * 3;
    const float* cv = v + r0 * 3;
    const float* ch = hist_rows == 1 ? hist : hist + r0 * h->desc.hist_bin;
    MlpArgs ac{nc.blob, nc.tab, nc.n_units, co, cd, nullptr, nullptr, w.f.sigma, nullptr, (long long)n, Nc, near, far, nullptr, nc.in_scale,
                 h->render_flags & DFN_RENDER_LINDISP};
    ac.status = range_flag_of(h);
    CHECK_HIP(launch_mlp(false, prec, var, ac, cus, s, h->desc.width), "render backward: coarse MLP");
    CHECK_HIP(launch_sample_fine(w.f.sigma, n, Nc, Ni, near, far, w.f.z, nullptr, nullptr, s, h->render_flags & DFN_RENDER_LINDISP),
              "render backward: sample_fine");
    CHECK_HIP(launch_ray_bias(h->rb, cv, ch, hist_rows, n, w.f.bias, s), "render backward: ray_bias");
    MlpArgs af{nf.blob, nf.tab, nf.n_units, co, cd, w.f.z, w.f.bias, w.f.raw, nullptr, (long long)n, Nf, 0.f, 0.f, nullptr, nf.in_scale};
    af.status = range_flag_of(h);
    CHECK_HIP(launch_mlp(true, prec, var, af, cus, s, h->desc.width), "render backward: fine MLP");
    CHECK_HIP(launch_composite_fine_backward(w.f.raw, w.f.z, grad_rgb + r0 * 3, n, Nf, w.graw, s), "render backward: composite");
    BwdArgs ab{nb.blob, nb.tab, nb.n_units, co, cd, cv, w.f.z, w.f.bias, w.graw, w.gpts, (long long)n, Nf, nb.in_scale};
    CHECK_HIP(launch_mlp_fine_backward(prec, ab, cus, s), "render backward: fine MLP gradient");
    CHECK_HIP(launch_ray_grad_reduce(w.gpts, w.f.z, cd, n, Nf, derive_v ? 1 : 0, go + r0 * 3, gd + r0 * 3,
                                     gv ? gv + r0 * 3 : nullptr, s),
              "render backward: ray reduction");
  }
  return DFN_OK;
}
}  // namespace

extern "C" size_t dfn_render_backward_workspace_bytes(size_t n_rays, int Nc, int Ni) {
  return carve_bwd(nullptr, n_rays ? n_rays : 1, Nc, Ni).total;
}

extern "C" int dfn_render_rays_backward(dfn_nerfh_t h, int prec, const float* rays_o, const float* rays_d, const float* viewdirs,
                                        const float* hist, size_t hist_rows, size_t n_rays, int Nc, int Ni, float near,
                                        float far, const float* grad_rgb, float* grad_rays_o, float* grad_rays_d,
                                        float* grad_viewdirs, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_net(h, prec, "dfn_render_rays_backward", true)) return rc;
  if (int rc = check_grad_prec(prec, "dfn_render_rays_backward")) return rc;
  if (int rc = check_grad_width(h, "dfn_render_rays_backward")) return rc;
  if (int rc = check_render_args(Nc, Ni, "dfn_render_rays_backward")) return rc;
  if (!n_rays) return DFN_OK;
  if (!rays_o || !rays_d || !hist || !grad_rgb || !grad_rays_o || !grad_rays_d || !workspace ||
      (hist_rows != 1 && hist_rows != n_rays))
    return set_error(DFN_ERR_ARG, "dfn_render_rays_backward: bad argument (hist_rows must be 1 or n_rays)");
  if (!n_rays) return DFN_OK;
  const BwdWorkspace w = carve_bwd(static_cast<char*>(workspace), n_rays, Nc, Ni);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_render_rays_backward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  const float* v = viewdirs;
  if (!v) {
    CHECK_HIP(launch_viewdirs(rays_d, n_rays, w.f.v, HS(stream)), "dfn_render_rays_backward: viewdirs");
    v = w.f.v;
  }
  return render_backward_core(h, prec, rays_o, rays_d, v, viewdirs == nullptr, hist, hist_rows, n_rays, Nc, Ni, near, far,
                              grad_rgb, grad_rays_o, grad_rays_d, viewdirs ? grad_viewdirs : nullptr, w, HS(stream));
}

extern "C" int dfn_render_image_backward(dfn_nerfh_t h, int prec, const float* c2w, int H, int W, float focal, float near,
                                         float far, int Nc, int Ni, const float* hist, const float* grad_rgb,
                                         float* grad_c2w, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_net(h, prec, "dfn_render_image_backward", true)) return rc;
  if (int rc = check_grad_prec(prec, "dfn_render_image_backward")) return rc;
  if (int rc = check_grad_width(h, "dfn_render_image_backward")) return rc;
  if (int rc = check_render_args(Nc, Ni, "dfn_render_image_backward")) return rc;
  if (!c2w || !hist || !grad_rgb || !grad_c2w || !workspace || H < 1 || W < 1 || !(focal > 0))
    return set_error(DFN_ERR_ARG, "dfn_render_image_backward: bad argument");
  const size_t n_rays = size_t(H) * W;
  const BwdWorkspace w = carve_bwd(static_cast<char*>(workspace), n_rays, Nc, Ni);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_render_image_backward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  CHECK_HIP(launch_raygen(H, W, focal, c2w, w.f.o, w.f.d, w.f.v, HS(stream)), "dfn_render_image_backward: raygen");
  if (int rc = render_backward_core(h, prec, w.f.o, w.f.d, w.f.v, true, hist, 1, n_rays, Nc, Ni, near, far, grad_rgb, w.go,
                                    w.gd, nullptr, w, HS(stream)))
    return rc;
  CHECK_HIP(launch_raygen_backward(H, W, focal, w.go, w.gd, grad_c2w, HS(stream)), "dfn_render_image_backward: raygen");
  return DFN_OK;
}
