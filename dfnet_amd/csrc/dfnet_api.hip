// dfnet_api.hip — C ABI of the DFNet feature extractor (include/dfnet_hip.h, dfn_dfnet_*):
// parameter intake with the reference's state_dict names, BatchNorm folding, packing of every
// convolution into MFMA A-fragments (both precisions), and the forward driver.
#include <hip/hip_runtime.h>
#include <functional>

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dfnet_hip.h"
#include "dfn_common.h"
#include "dfnet_kernels.h"
#include "nerfh_layout.h"

using namespace dfn;

namespace {

struct ConvSpec {
  std::string key;  // state_dict prefix, e.g. "encoder.5"
  int cin, cout, ks;
  bool relu;        // activation written to out_act
  bool pool_after;  // VGG max-pool follows
  int tap;          // hypercolumn level tapped BEFORE the ReLU, or -1
  int enc_index;    // index in the 31-module encoder (for error messages)
};

struct PackedConv {
  char* w[3] = {nullptr, nullptr, nullptr};  // per precision (2 = split-f16: [hi fragments][lo fragments] per slice)
  float* bias = nullptr;            // [cout/32][2][16]
  float* bias_x3 = nullptr;         // the same, pre-multiplied by the split-f16 operand scale
  float out_scale = 1.f;            // 1 / (2^s * kConvActScale): undoes the split-f16 operand scaling
};
inline size_t elem_size(int prec) { return prec == 0 ? 2 : 4; }  // activations in HBM: f16 only on the f16 path

const int kVgg[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, -1, 512, 512, 512, -1, 512, 512, 512, -1};

}  // namespace

struct dfn_dfnet_s {
  int n_taps = 3, feat_dim = 12;
  std::vector<ConvSpec> enc;        // 13 encoder convs
  std::vector<int> tap_channels;    // 64, 256, 512
  std::map<std::string, std::vector<float>> params;
  std::map<std::string, std::vector<size_t>> shapes;
  bool committed = false;
  std::vector<PackedConv> enc_packed;
  std::vector<PackedConv> ad1, ad5;  // per tap: 1x1 and BN-folded 5x5
  // input-gradient convolutions: the same layers with in/out channels swapped and taps flipped (dgrad = conv)
  std::vector<PackedConv> enc_dgrad, ad1_dgrad, ad5_dgrad;
  float* fc = nullptr;               // fc_w [feat_dim,512] | fc_b
  // training DFNet itself (BatchNorm not folded): the plain 5x5 convs, their data-gradient convs, and per tap the
  // device block [4][128] = gamma, beta, running_mean, running_var
  std::vector<PackedConv> ad5_raw, ad5_raw_dgrad;
  std::vector<float*> bn_dev;
  // dfn_dfnet_forward_train(keep = 1) left its activations in this workspace (consumed by backward_all_params)
  // (a few: a training step may hold the siamese forward and a pose-only forward before its backward runs)
  // precisions whose packed fragments are current: all three after dfn_dfnet_commit, the requested ones after a device re-pack
  int fresh_mask = 0;
  struct Kept { const void* ws; int prec, B, H, W, bn_batch; };
  std::vector<Kept> kept;
  // split-f16 inference: the adaptation branches of pyramid levels >= 1 (1x1, 5x5, bilinear resize: mostly HBM writes) run on this
  // stream beside the rest of the encoder (forward_core); created on first use
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_buf[3] = {nullptr, nullptr, nullptr};   // backward_params_core: the side stream's last read of a gradient buffer
  hipEvent_t ev_adapt = nullptr;                         // backward_params_split: the side stream's last read of a level's split operands
};

static size_t zeros_offset(int feat_dim) { return (size_t(feat_dim) * 513 + 3) & ~size_t(3); }   // floats into h->fc, 16-byte aligned

static void build_specs(dfn_dfnet_s* h) {
  int cin = 3, idx = 0;
  const int tap_at[3] = {2, 14, 28};
  for (int v : kVgg) {
    if (v < 0) { h->enc.back().pool_after = true; ++idx; continue; }
    ConvSpec s{"encoder." + std::to_string(idx), cin, v, 3, true, false, -1, idx};
    for (int t = 0; t < h->n_taps; ++t) if (tap_at[t] == idx) s.tap = t;
    h->enc.push_back(s);
    h->shapes[s.key + ".weight"] = {size_t(v), size_t(cin), 3, 3};
    h->shapes[s.key + ".bias"] = {size_t(v)};
    cin = v;
    idx += 2;
  }
  const int chans[3] = {64, 256, 512};
  for (int t = 0; t < h->n_taps; ++t) {
    h->tap_channels.push_back(chans[t]);
    const std::string p = "adaptation_layers.adapt_layer_" + std::to_string(t);
    h->shapes[p + ".0.weight"] = {64, size_t(chans[t]), 1, 1};
    h->shapes[p + ".0.bias"] = {64};
    h->shapes[p + ".2.weight"] = {128, 64, 5, 5};
    h->shapes[p + ".2.bias"] = {128};
    for (const char* k : {".3.weight", ".3.bias", ".3.running_mean", ".3.running_var"}) h->shapes[p + k] = {128};
  }
  h->shapes["fc_pose.weight"] = {size_t(h->feat_dim), 512};
  h->shapes["fc_pose.bias"] = {size_t(h->feat_dim)};
}

extern "C" int dfn_dfnet_create(int n_taps, int feat_dim, dfn_dfnet_t* out) {
  if (!out || (n_taps != 1 && n_taps != 3) || feat_dim < 1 || feat_dim > 512)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_create: n_taps must be 3 (DFNet) or 1 (DFNet_s), 1 <= feat_dim <= 512");
  auto* h = new dfn_dfnet_s();
  h->n_taps = n_taps;
  h->feat_dim = feat_dim;
  build_specs(h);
  *out = h;
  return DFN_OK;
}

static void free_dev(dfn_dfnet_s* h) {
  auto drop = [](std::vector<PackedConv>& v) {
    for (auto& p : v) {
      for (auto& w : p.w) if (w) (void)hipFree(w);
      if (p.bias) (void)hipFree(p.bias);
      if (p.bias_x3) (void)hipFree(p.bias_x3);
    }
    v.clear();
  };
  drop(h->enc_packed);
  drop(h->ad1);
  drop(h->ad5);
  drop(h->enc_dgrad);
  drop(h->ad1_dgrad);
  drop(h->ad5_dgrad);
  if (h->fc) (void)hipFree(h->fc);
  h->fc = nullptr;
  drop(h->ad5_raw);
  drop(h->ad5_raw_dgrad);
  for (float* p : h->bn_dev) if (p) (void)hipFree(p);
  h->bn_dev.clear();
}

extern "C" int dfn_dfnet_destroy(dfn_dfnet_t h) {
  if (!h) return DFN_OK;
  free_dev(h);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_adapt) (void)hipEventDestroy(h->ev_adapt);
  for (hipEvent_t e : h->ev_buf) if (e) (void)hipEventDestroy(e);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
  return DFN_OK;
}

extern "C" int dfn_dfnet_set_param(dfn_dfnet_t h, const char* name, const float* host, size_t numel) {
  if (!h || !name || !host) return set_error(DFN_ERR_ARG, "dfn_dfnet_set_param: null argument");
  const auto it = h->shapes.find(name);
  if (it == h->shapes.end()) return set_error(DFN_ERR_ARG, "dfn_dfnet_set_param: unknown parameter '%s'", name);
  size_t want = 1;
  for (size_t s : it->second) want *= s;
  if (want != numel)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_set_param: '%s' has %zu elements, expected %zu", name, numel, want);
  h->params[name].assign(host, host + numel);
  h->committed = false;
  return DFN_OK;
}

namespace {

// Channel held in slot s of half hh of input block blk (first layer: only RGB in half 0, slots 0..2).
inline int in_channel(bool first, int blk, int hh, int s) {
  if (first) return (hh == 0 && s < 3) ? s : -1;
  return 32 * blk + 4 * hh + (s & 3) + 8 * (s >> 2);
}

template <class P>
void pack_conv(const float* w, int cout, int cin, int ks, bool first, int sb, int mb, std::vector<uint8_t>& blob) {
  using Elem = typename std::conditional<P::kSlotsPerChunk == 8, _Float16, float>::type;
  constexpr int SPC = P::kSlotsPerChunk;
  const int nblk = first ? 1 : cin / 32, kcb = sb / SPC, groups = cout / 32 / mb;
  blob.assign(size_t(groups) * nblk * ks * mb * ks * kcb * 64 * P::kLaneBytes, 0);
  Elem* out = reinterpret_cast<Elem*>(blob.data());
  size_t o = 0;
  for (int cg = 0; cg < groups; ++cg)
    for (int blk = 0; blk < nblk; ++blk)
      for (int ky = 0; ky < ks; ++ky)
        for (int m = 0; m < mb; ++m)
          for (int kx = 0; kx < ks; ++kx)
            for (int kc = 0; kc < kcb; ++kc)
              for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < SPC; ++j, ++o) {
                  const int co = 32 * (cg * mb + m) + (lane & 31);
                  const int ci = in_channel(first, blk, lane >> 5, kc * SPC + j);
                  out[o] = Elem(ci >= 0 && ci < cin ? w[((size_t(co) * cin + ci) * ks + ky) * ks + kx] : 0.f);
                }
}

// Split-f16 packing: w * 2^s = hi + lo (both f16); layout [cg][blk][ky][kc][hi|lo][mb][kx][lane][8] — a (blk, ky) slice is
// contiguous and splits into one sub-slice per K-chunk kc, the unit the kernel double-buffers (dfnet_conv.hip).
void pack_conv_x3(const float* w, int cout, int cin, int ks, bool first, int sb, int mb, float wscale, std::vector<uint8_t>& blob) {
  const int nblk = first ? 1 : cin / 32, kcb = sb / 8, groups = cout / 32 / mb;
  const size_t frag = 64 * 8;                                   // elements of one fragment
  const size_t sub = size_t(2) * mb * ks * frag;                // one (blk, ky, kc) sub-slice: hi and lo
  blob.assign(size_t(groups) * nblk * ks * kcb * sub * 2, 0);
  _Float16* out = reinterpret_cast<_Float16*>(blob.data());
  for (int cg = 0; cg < groups; ++cg)
    for (int blk = 0; blk < nblk; ++blk)
      for (int ky = 0; ky < ks; ++ky)
        for (int kc = 0; kc < kcb; ++kc) {
          _Float16* sl = out + (((size_t(cg) * nblk + blk) * ks + ky) * kcb + kc) * sub;
          for (int m = 0; m < mb; ++m)
            for (int kx = 0; kx < ks; ++kx)
              for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                  const int co = 32 * (cg * mb + m) + (lane & 31);
                  const int ci = in_channel(first, blk, lane >> 5, kc * 8 + j);
                  const float v = (ci >= 0 && ci < cin ? w[((size_t(co) * cin + ci) * ks + ky) * ks + kx] : 0.f) * wscale;
                  const _Float16 hi = _Float16(v);
                  const size_t o = ((size_t(m) * ks + kx) * 64 + lane) * 8 + j;
                  sl[o] = hi;
                  sl[size_t(mb) * ks * frag + o] = _Float16(v - float(hi));
                }
        }
}

int upload_bytes(const void* src, size_t bytes, void** dst) {
  if (hipMalloc(dst, bytes ? bytes : 16) != hipSuccess) return set_error(DFN_ERR_HIP, "hipMalloc(%zu) failed", bytes);
  if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
    return set_error(DFN_ERR_HIP, "hipMemcpy H2D (%zu bytes) failed", bytes);
  return DFN_OK;
}

int pack_and_upload(const float* w, const float* b, int cout, int cin, int ks, bool first, PackedConv& pc) {
  for (int prec = 0; prec < 2; ++prec) {
    const int sb = first ? prep_sb(prec) : 16;
    const int mb = conv_mb(prec, cout / 32);
    std::vector<uint8_t> blob;
    if (prec == 0) pack_conv<PrecF16>(w, cout, cin, ks, first, sb, mb, blob);
    else pack_conv<PrecF32>(w, cout, cin, ks, first, sb, mb, blob);
    if (int rc = upload_bytes(blob.data(), blob.size(), reinterpret_cast<void**>(&pc.w[prec]))) return rc;
  }
  // split-f16: scale the weights by a power of two so that the largest is ~2^10 (lo parts stay normal f16)
  float wmax = 0.f;
  for (size_t i = 0; i < size_t(cout) * cin * ks * ks; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
  int sexp = 0;
  if (wmax > 0.f) sexp = 10 - int(std::ceil(std::log2(wmax)));
  sexp = sexp < -8 ? -8 : (sexp > 24 ? 24 : sexp);
  const float wscale = std::ldexp(1.f, sexp);
  {
    std::vector<uint8_t> blob;
    pack_conv_x3(w, cout, cin, ks, first, first ? prep_sb(2) : 16, 2, wscale, blob);
    if (int rc = upload_bytes(blob.data(), blob.size(), reinterpret_cast<void**>(&pc.w[2]))) return rc;
  }
  pc.out_scale = 1.f / (wscale * kConvActScale);
  std::vector<float> bias(size_t(cout / 32) * 32), bias3(size_t(cout / 32) * 32);
  for (int m = 0; m < cout / 32; ++m)
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) {
        bias[(m * 2 + hh) * 16 + r] = b[32 * m + mblock_row(hh, r)];
        bias3[(m * 2 + hh) * 16 + r] = b[32 * m + mblock_row(hh, r)] * wscale * kConvActScale;
      }
  if (int rc = upload_bytes(bias3.data(), bias3.size() * 4, reinterpret_cast<void**>(&pc.bias_x3))) return rc;
  return upload_bytes(bias.data(), bias.size() * 4, reinterpret_cast<void**>(&pc.bias));
}

// The data gradient of a stride-1 "same" convolution is itself one: w'[ci][co][ky][kx] = w[co][ci][K-1-ky][K-1-kx].
// Output channels are padded to a multiple of 64 (two M-blocks, conv_mb()); no bias.
int pack_dgrad(const float* w, int cout, int cin, int ks, PackedConv& pc) {
  const int cop = (cin + 63) / 64 * 64;
  std::vector<float> wt(size_t(cop) * cout * ks * ks, 0.f), zero(cop, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          wt[((size_t(ci) * cout + co) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)] = w[((size_t(co) * cin + ci) * ks + ky) * ks + kx];
  return pack_and_upload(wt.data(), zero.data(), cop, cout, ks, false, pc);
}

}  // namespace

extern "C" int dfn_dfnet_commit(dfn_dfnet_t h) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_dfnet_commit: null handle");
  for (const auto& kv : h->shapes)
    if (!h->params.count(kv.first)) return set_error(DFN_ERR_STATE, "dfn_dfnet_commit: parameter '%s' not set", kv.first.c_str());
  free_dev(h);
  h->enc_packed.resize(h->enc.size());
  for (size_t i = 0; i < h->enc.size(); ++i) {
    const ConvSpec& s = h->enc[i];
    if (int rc = pack_and_upload(h->params[s.key + ".weight"].data(), h->params[s.key + ".bias"].data(), s.cout, s.cin, 3,
                                 i == 0, h->enc_packed[i]))
      return rc;
  }
  h->ad1.resize(h->n_taps);
  h->ad5.resize(h->n_taps);
  for (int t = 0; t < h->n_taps; ++t) {
    const std::string p = "adaptation_layers.adapt_layer_" + std::to_string(t);
    if (int rc = pack_and_upload(h->params[p + ".0.weight"].data(), h->params[p + ".0.bias"].data(), 64, h->tap_channels[t], 1,
                                 false, h->ad1[t]))
      return rc;
    // fold eval-mode BatchNorm2d (eps = 1e-5) into the 5x5 conv: y = (conv + b - mean) * g/sqrt(var+eps) + beta
    std::vector<float> w5 = h->params[p + ".2.weight"], b5 = h->params[p + ".2.bias"];
    const auto& g = h->params[p + ".3.weight"];
    const auto& beta = h->params[p + ".3.bias"];
    const auto& mu = h->params[p + ".3.running_mean"];
    const auto& var = h->params[p + ".3.running_var"];
    for (int co = 0; co < 128; ++co) {
      const float sc = g[co] / std::sqrt(var[co] + 1e-5f);
      for (int k = 0; k < 64 * 25; ++k) w5[size_t(co) * 64 * 25 + k] *= sc;
      b5[co] = (b5[co] - mu[co]) * sc + beta[co];
    }
    if (int rc = pack_and_upload(w5.data(), b5.data(), 128, 64, 5, false, h->ad5[t])) return rc;
    {
      h->ad5_raw.resize(h->n_taps);
      h->ad5_raw_dgrad.resize(h->n_taps);
      h->bn_dev.resize(h->n_taps, nullptr);
      if (int rc = pack_and_upload(h->params[p + ".2.weight"].data(), h->params[p + ".2.bias"].data(), 128, 64, 5, false, h->ad5_raw[t]))
        return rc;
      if (int rc = pack_dgrad(h->params[p + ".2.weight"].data(), 128, 64, 5, h->ad5_raw_dgrad[t])) return rc;
      std::vector<float> bn(g);
      bn.insert(bn.end(), beta.begin(), beta.end());
      bn.insert(bn.end(), mu.begin(), mu.end());
      bn.insert(bn.end(), var.begin(), var.end());
      if (int rc = upload_bytes(bn.data(), bn.size() * 4, reinterpret_cast<void**>(&h->bn_dev[t]))) return rc;
    }
    h->ad1_dgrad.resize(h->n_taps);
    h->ad5_dgrad.resize(h->n_taps);
    if (int rc = pack_dgrad(h->params[p + ".0.weight"].data(), 64, h->tap_channels[t], 1, h->ad1_dgrad[t])) return rc;
    if (int rc = pack_dgrad(w5.data(), 128, 64, 5, h->ad5_dgrad[t])) return rc;  // BN scale is part of the folded weights
  }
  h->enc_dgrad.resize(h->enc.size());
  for (size_t i = 0; i < h->enc.size(); ++i)
    if (int rc = pack_dgrad(h->params[h->enc[i].key + ".weight"].data(), h->enc[i].cout, h->enc[i].cin, 3, h->enc_dgrad[i]))
      return rc;
  std::vector<float> fc = h->params["fc_pose.weight"];
  const auto& fb = h->params["fc_pose.bias"];
  fc.insert(fc.end(), fb.begin(), fb.end());
  fc.resize(zeros_offset(h->feat_dim) + 64, 0.f);   // + 256 bytes of zeros: the DMA source of padding pixels (ConvArgs::zeros)
  if (int rc = upload_bytes(fc.data(), fc.size() * 4, reinterpret_cast<void**>(&h->fc))) return rc;
  h->committed = true;
  h->fresh_mask = 7;
  return DFN_OK;
}

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
struct DfWs {
  char *prep, *actA, *actB, *tap[3], *tmp64, *ad128;
  double* bn_part;   // BatchNorm reductions (dfn_dfnet_forward_train)
  float* bn_work;
  size_t total;
};
DfWs carve_df(const dfn_dfnet_s* h, char* base, int prec, int B, int H, int W) {
  const size_t es = elem_size(prec);
  const size_t px = size_t(B) * H * W;
  DfWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  w.prep = take(px * 2 * prep_sb(prec) * es);
  w.actA = take(px * 64 * es);
  w.actB = take(px * 64 * es);
  const int div[3] = {1, 4, 16};
  for (int t = 0; t < h->n_taps; ++t) w.tap[t] = take(size_t(B) * (H / div[t]) * (W / div[t]) * h->tap_channels[t] * es);
  w.tmp64 = take(px * 64 * es);
  w.ad128 = take(px * 128 * es);
  w.bn_part = reinterpret_cast<double*>(take(kBnPartBytes));
  w.bn_work = reinterpret_cast<float*>(take(kBnWorkFloats * 4));
  w.total = off;
  return w;
}
}  // namespace

// A device re-pack (dfn_dfnet_refresh_*_params_device) renews only the precisions it was asked for.
static int check_fresh(dfn_dfnet_t h, int prec, const char* fn) {
  if (prec >= 0 && prec < 3 && !((h->fresh_mask >> prec) & 1))
    return set_error(DFN_ERR_STATE, "%s: the fragments of precision %d are stale (last device re-pack left them out): re-pack with its bit "
                     "in prec_mask or dfn_dfnet_commit()", fn, prec);
  return DFN_OK;
}

extern "C" size_t dfn_dfnet_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W) {
  if (!h || B < 1 || H < 1 || W < 1) return 0;
  return carve_df(h, nullptr, prec, B, H, W).total;
}

#define HS(s) reinterpret_cast<hipStream_t>(s)
static int ensure_side(dfn_dfnet_s* h);
// A function that has put work on the handle's side stream leaves through this: whatever the exit (an error return included), the
// caller's stream waits for the side stream — nothing the caller may free or reuse afterwards is still being read or written there.
struct SideJoin {
  dfn_dfnet_s* h;
  hipStream_t s;
  bool armed = false;
  ~SideJoin() {
    if (armed && h->side && hipEventRecord(h->ev_join, h->side) == hipSuccess) (void)hipStreamWaitEvent(s, h->ev_join, 0);
  }
};
#define CHECK_HIP(expr, what)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return set_error(DFN_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

// bn_mode 0: inference (BatchNorm folded into the 5x5 convs).  1, 2: training — the plain 5x5 conv, then BatchNorm as
// an affine map applied by the upsample kernel: 1 = running statistics (frozen), 2 = batch statistics over all B
// images, written to bn_stats [n_taps][2][128] (mean, biased variance).
// the handle's side stream and its events, created on first use
static int ensure_side(dfn_dfnet_s* h) {
  if (h->side) return DFN_OK;
  CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking), "dfnet: side stream");
  CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming), "dfnet: side stream");
  CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming), "dfnet: side stream");
  for (hipEvent_t& e : h->ev_buf) CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "dfnet: side stream");
  CHECK_HIP(hipEventCreateWithFlags(&h->ev_adapt, hipEventDisableTiming), "dfnet: side stream");
  return DFN_OK;
}

// level_mask: bit t = pyramid level t is computed (its planes of `features` written).  Levels outside the mask are left untouched, and
// with return_pose == 0 the encoder stops after the deepest level asked for.
static int forward_core(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int return_feature, int siamese,
                        int return_pose, int upH, int upW, float* features, float* pose, int bn_mode, float* bn_stats,
                        void* workspace, size_t workspace_bytes, void* stream, int level_mask = ~0) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: null handle");
  if (!h->committed) return set_error(DFN_ERR_STATE, "dfn_dfnet_forward: dfn_dfnet_commit() has not been called");
  if (prec != DFN_PREC_F16 && prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: unknown precision %d", prec);
  if (int rc = check_fresh(h, prec, "dfn_dfnet_forward")) return rc;
  if (!x || !workspace || B < 1 || H < 32 || W < 32 || (return_feature && (!features || upH < 1 || upW < 1)) ||
      (return_pose && !pose) || (return_feature && siamese && (B & 1)))
    return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: bad argument (need H,W >= 32; even batch for siamese)");
  const DfWs w = carve_df(h, static_cast<char*>(workspace), prec, B, H, W);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = HS(stream);
  // Split-f16 inference keeps every intermediate activation as hi | lo f16 blocks (dfnet_conv.hip, split_piece): same bytes as fp32,
  // split once by the producer, LDS-DMA staged by the consumer.  Only what leaves the convs stays fp32: the 5x5 adaptation output
  // (resized / written as NCHW planes) and relu5_3 when the pose head reads it.
  // The training-mode forward (bn_mode != 0) does the same, conv for conv, as the activation-keeping forward of the training path
  // (encoder_keep / adapt_keep: conv1_1 from the fp32 prepared frame, its weight gradient gathers from it) — bit-identical results.
  const bool split = prec == 2;
  const bool split_in0 = split && bn_mode == 0;          // conv1_1's input in the split storage too (inference)
  const void* zeros = h->fc + zeros_offset(h->feat_dim);
  CHECK_HIP(launch_dfnet_prep(split_in0 ? 3 : prec, x, B, H, W, w.prep, s), "dfnet: prep");
  const void* cur = w.prep;
  char* ping[2] = {w.actA, w.actB};
  int pp = 0, ch = H, cw = W, nblk = 1;
  int tap_h[3] = {0, 0, 0}, tap_w[3] = {0, 0, 0};
  const void* last_act = nullptr;
  int last_h = 0, last_w = 0;
  int fused_1x1 = -1;                 // tap whose 1x1 adaptation conv ran inside its encoder conv (its output sits in w.tmp64)
  // One pyramid level's adaptation branch: [1x1 + ReLU unless it ran inside the tap's conv] -> BN-folded 5x5 -> align_corners resize into
  // the caller's NCHW stack (or written there by the 5x5 itself when the level already has the requested size).
  const size_t es = prec == 0 ? 2 : 4;
  const size_t plane = size_t(128) * upH * upW;
  auto adapt_level = [&](int t, hipStream_t st, char* tmp64, char* ad128) -> int {
    ConvArgs a{};
    a.in = w.tap[t]; a.w = h->ad1[t].w[prec]; a.bias = prec == 2 ? h->ad1[t].bias_x3 : h->ad1[t].bias; a.out_scale = h->ad1[t].out_scale; a.out_act = tmp64; a.out_pre = nullptr;
    a.B = B; a.H = tap_h[t]; a.W = tap_w[t]; a.nblk_in = h->tap_channels[t] / 32; a.cout_blocks = 2; a.relu = 1;
    if (split) { a.in_split = 1; a.out_split = 1; a.zeros = zeros; }
    if (t != fused_1x1) CHECK_HIP(launch_conv(prec, 1, 16, a, st), "dfnet: adapt 1x1");
    ConvArgs c{};
    const PackedConv& p5 = bn_mode ? h->ad5_raw[t] : h->ad5[t];
    c.in = tmp64; c.w = p5.w[prec]; c.bias = prec == 2 ? p5.bias_x3 : p5.bias; c.out_scale = p5.out_scale; c.out_act = ad128; c.out_pre = nullptr;
    c.B = B; c.H = tap_h[t]; c.W = tap_w[t]; c.nblk_in = 2; c.cout_blocks = 4; c.relu = 0;
    if (split) { c.in_split = 1; c.zeros = zeros; }
    // a level that already has the requested size (level 0 when the features are asked for at the frame size, as every caller of
    // the reference does): the align_corners resize is the identity, the 5x5 conv writes the caller's NCHW stack itself
    const bool identity = bn_mode == 0 && tap_h[t] == upH && tap_w[t] == upW;
    if (identity) {
      c.out_act = nullptr;
      c.out_nchw = siamese ? features + size_t(t) * (B / 2) * plane : features + size_t(t) * B * plane;
      c.nchw_split = siamese ? B / 2 : B;
      c.nchw_group_stride = size_t(h->n_taps) * (B / 2) * plane;
    }
    CHECK_HIP(launch_conv(prec, 5, 16, c, st), "dfnet: adapt 5x5");
    if (identity) return DFN_OK;
    const float* affine = nullptr;
    if (bn_mode == 1) {
      CHECK_HIP(launch_bn_running_stats(h->bn_dev[t], 1e-5f, w.bn_work, st), "dfnet: BatchNorm running statistics");
      affine = w.bn_work;
    } else if (bn_mode == 2) {
      CHECK_HIP(launch_bn_batch_stats(reinterpret_cast<const float*>(ad128), (long long)B * tap_h[t] * tap_w[t], h->bn_dev[t], 1e-5f,
                                      w.bn_part, w.bn_work, bn_stats + size_t(t) * 256, bn_stats + size_t(t) * 256 + 128, st),
                "dfnet: BatchNorm batch statistics");
      affine = w.bn_work;
    }
    if (!siamese) {
      CHECK_HIP(launch_upsample(prec, ad128, B, tap_h[t], tap_w[t], upH, upW, features + size_t(t) * B * plane, plane, st, affine),
                "dfnet: upsample");
    } else {
      const int hb = B / 2;
      for (int half = 0; half < 2; ++half) {
        const char* src = ad128 + size_t(half) * hb * tap_h[t] * tap_w[t] * 128 * es;
        float* dst = features + (size_t(half) * h->n_taps + t) * hb * plane;
        CHECK_HIP(launch_upsample(prec, src, hb, tap_h[t], tap_w[t], upH, upW, dst, plane, st, affine), "dfnet: upsample");
      }
    }
    return DFN_OK;
  };
  // Split-f16 inference with level 0 at the requested size (every caller of the reference): level 0's 5x5 writes the caller's stack
  // itself and leaves w.ad128 unused, so the branches of levels >= 1 — two small convs and a resize that is pure HBM writes (629 MB
  // per level for 4 frames of 480x640) — get their buffers out of it and run on the handle's side stream as soon as their tap is
  // written, beside the matrix-bound encoder layers whose grids leave CUs idle (conv4_x: 1.25 rounds, conv5_x: 0.4).
  level_mask &= (1 << h->n_taps) - 1;
  if (return_feature && !level_mask) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: empty level mask");
  int deepest = 0;
  for (int t = 0; t < h->n_taps; ++t) if (level_mask >> t & 1) deepest = t;
  auto wanted = [&](int t) { return return_feature && (level_mask >> t & 1); };
  const bool side_levels = split && bn_mode == 0 && return_feature && h->n_taps > 1 && H == upH && W == upW;
  char* side_tmp64 = w.ad128;
  char* side_ad128 = w.ad128 + al256(size_t(B) * (H / 4) * (W / 4) * 64 * es);
  if (side_levels)
    if (int rc = ensure_side(h)) return rc;
  SideJoin side_join{h, s};
  for (size_t i = 0; i < h->enc.size(); ++i) {
    const ConvSpec& sp = h->enc[i];
    const bool is_last_tap = sp.tap == (return_feature ? deepest : h->n_taps - 1);
    const bool stop_here = is_last_tap && !return_pose;
    ConvArgs a{};
    a.in = cur;
    a.w = h->enc_packed[i].w[prec];
    a.bias = prec == 2 ? h->enc_packed[i].bias_x3 : h->enc_packed[i].bias; a.out_scale = h->enc_packed[i].out_scale;
    a.out_act = stop_here ? nullptr : ping[pp];
    a.out_pre = (sp.tap >= 0 && wanted(sp.tap)) ? w.tap[sp.tap] : nullptr;
    a.B = B; a.H = ch; a.W = cw;
    a.nblk_in = nblk;
    a.cout_blocks = sp.cout / 32;
    a.relu = 1;
    // split-f16: a conv followed by the 2x2 max pool writes the pooled activation itself (nothing else reads the full-size one)
    const bool pool_fused = prec == 2 && i > 0 && !stop_here && sp.pool_after && i + 1 < h->enc.size();
    if (pool_fused) { a.out_pool = ping[pp]; a.out_act = nullptr; }
    if (split) {
      const bool last = i + 1 == h->enc.size();          // relu5_3: read by the pose head as fp32
      a.in_split = (i > 0 || split_in0) ? 1 : 0; a.zeros = zeros;
      a.out_split = (last ? 0 : 1) | 2 | 4;
      // a 64-channel tap is one workgroup's channels: its adaptation layer's 1x1 conv + ReLU runs in this conv's epilogue and the
      // tap itself is never stored (level 0: 315 MB written and read back per 4 frames of 480x640, plus the 1x1 kernel)
      if (a.out_pre && sp.cout == 64 && h->tap_channels[sp.tap] == 64 && bn_mode == 0) {
        const PackedConv& p1 = h->ad1[sp.tap];
        a.fuse_w = p1.w[2]; a.fuse_bias = p1.bias_x3; a.fuse_scale = p1.out_scale; a.fuse_out = w.tmp64;
        a.out_pre = nullptr;
        fused_1x1 = sp.tap;
      }
    }
    if (a.out_act || a.out_pre || a.out_pool || a.fuse_out)
      CHECK_HIP(launch_conv(prec, 3, i == 0 ? prep_sb(prec) : 16, a, s), "dfnet: encoder conv");
    if (sp.tap >= 0) { tap_h[sp.tap] = ch; tap_w[sp.tap] = cw; }
    if (side_levels && sp.tap >= 1 && wanted(sp.tap)) {     // this level's tap is on its way: its branch follows it on the side stream
      side_join.armed = true;
      CHECK_HIP(hipEventRecord(h->ev_fork, s), "dfnet: side stream");
      CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0), "dfnet: side stream");
      if (int rc = adapt_level(sp.tap, h->side, side_tmp64, side_ad128)) return rc;
    }
    if (stop_here) break;
    cur = ping[pp];
    last_act = cur; last_h = ch; last_w = cw;
    pp ^= 1;
    nblk = sp.cout / 32;
    if (pool_fused) {
      ch /= 2; cw /= 2;
    } else if (sp.pool_after && i + 1 < h->enc.size()) {
      CHECK_HIP(launch_maxpool(prec, cur, B, ch, cw, nblk, ping[pp], s), "dfnet: maxpool");
      cur = ping[pp];
      pp ^= 1;
      ch /= 2; cw /= 2;
    }
  }
  if (return_feature) {
    for (int t = 0; t < h->n_taps; ++t)
      if (wanted(t) && !(side_levels && t >= 1)) {
        if (int rc = adapt_level(t, s, w.tmp64, w.ad128)) return rc;
      }
    if (side_join.armed) {              // the branches launched beside the encoder rejoin the caller's stream
      side_join.armed = false;
      CHECK_HIP(hipEventRecord(h->ev_join, h->side), "dfnet: side stream");
      CHECK_HIP(hipStreamWaitEvent(s, h->ev_join, 0), "dfnet: side stream");
    }
  }
  if (return_pose) {
    if (!last_act || last_h < 2 || last_w < 2) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward: image too small for pool5");
    // (scratch: w.prep — conv1_1 was its only reader; B * (H / 32) * 512 floats fit in the B * H * W * 16 bytes it has at least)
    CHECK_HIP(launch_pose_head(prec, last_act, B, last_h, last_w, h->fc, h->fc + size_t(h->feat_dim) * 512, h->feat_dim,
                               reinterpret_cast<float*>(w.prep), pose, s),
              "dfnet: pose head");
  }
  return DFN_OK;
}

extern "C" int dfn_dfnet_forward(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int return_feature,
                                 int siamese, int return_pose, int upH, int upW, float* features, float* pose,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return forward_core(h, prec, x, B, H, W, return_feature, siamese, return_pose, upH, upW, features, pose, 0, nullptr, workspace,
                      workspace_bytes, stream);
}

extern "C" int dfn_dfnet_forward_levels(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese, int level_mask,
                                        int upH, int upW, float* features, void* workspace, size_t workspace_bytes, void* stream) {
  if (h && !(level_mask & ((1 << h->n_taps) - 1))) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward_levels: empty level mask");
  return forward_core(h, prec, x, B, H, W, 1, siamese, 0, upH, upW, features, nullptr, 0, nullptr, workspace, workspace_bytes, stream,
                      level_mask);
}

static int forward_train_keep(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese, int return_pose, int bn_batch,
                              int upH, int upW, float* features, float* pose, float* bn_stats, void* workspace, size_t workspace_bytes,
                              void* stream, int pyramid_only = 0, int feat_images = 0);

extern "C" int dfn_dfnet_forward_train(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese, int return_pose,
                                       int bn_batch, int keep, int upH, int upW, float* features, float* pose, float* bn_stats,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (bn_batch && features && !bn_stats) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward_train: null bn_stats");
  if (!features && !keep) return set_error(DFN_ERR_ARG, "dfn_dfnet_forward_train: null features (pose only) needs keep = 1");
  if (keep) {
    if (prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3)
      return set_error(DFN_ERR_UNSUPPORTED, "dfn_dfnet_forward_train: batch statistics need fp32 activations (precision F32 or F16X3)");
    return forward_train_keep(h, prec, x, B, H, W, siamese, return_pose, bn_batch, upH, upW, features, pose, bn_stats, workspace,
                              workspace_bytes, stream);
  }
  if (prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3)
    return set_error(DFN_ERR_UNSUPPORTED, "dfn_dfnet_forward_train: batch statistics need fp32 activations (precision F32 or F16X3)");
  return forward_core(h, prec, x, B, H, W, 1, siamese, return_pose, upH, upW, features, pose, bn_batch ? 2 : 1, bn_stats, workspace,
                      workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------ input gradient
namespace {
struct DfBwdWs {
  char* prep;
  char* act[13];     // post-ReLU output of every encoder conv (ReLU gates, max-pool routing)
  char* tap[3];      // pre-ReLU taps (inputs of the adaptation layers)
  char *pooled, *tmp64, *g128, *g64, *gtap, *gA, *gB;
  char* z128;        // unfolded 5x5 output of the level being processed (training BatchNorm)
  double* bn_part;   // fp64 chunk partials of the BatchNorm reductions
  float* bn_work;    // kBnWorkFloats
  float* scl;        // [scale, 1/scale] + 1024 partials of launch_absmax_scale (split-f16 gradient convs)
  // split-storage input gradient (backward_input_split): split copy of a level's d z, a third trunk buffer, |max| words, scale slots
  char *g128S, *gC;
  unsigned* amax;    // [64]
  float* slots;      // [32][2]
  size_t total;
};
DfBwdWs carve_df_bwd(const dfn_dfnet_s* h, char* base, int prec, int B, int H, int W, bool input_split = false) {
  const size_t es = prec == 0 ? 2 : 4, px = size_t(B) * H * W;
  DfBwdWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  w.prep = take(px * 2 * prep_sb(prec) * es);
  int ch = H, cw = W;
  for (size_t i = 0; i < h->enc.size(); ++i) {
    w.act[i] = take(size_t(B) * ch * cw * h->enc[i].cout * es);
    if (h->enc[i].tap >= 0) w.tap[h->enc[i].tap] = take(size_t(B) * ch * cw * h->enc[i].cout * es);
    if (h->enc[i].pool_after) { ch /= 2; cw /= 2; }
  }
  w.pooled = take(px * 64 / 4 * es);   // largest pooled tensor: 64 channels at half resolution
  w.tmp64 = take(px * 64 * es);
  w.g128 = take(px * 128 * es);
  w.g64 = take(px * 64 * es);
  w.gtap = take(px * 64 * es);
  w.gA = take(px * 64 * es);
  w.gB = take(px * 64 * es);
  w.scl = reinterpret_cast<float*>(take((1024 + 8) * 4));
  if (input_split) {
    w.g128S = take(px * 128 * 4);
    w.gC = take(px * 64 * 4);
    w.amax = reinterpret_cast<unsigned*>(take(64 * 4));
    w.slots = reinterpret_cast<float*>(take(64 * 4));
  }
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t dfn_dfnet_backward_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W) {
  if (!h || B < 1 || H < 1 || W < 1) return 0;
  return carve_df_bwd(h, nullptr, prec, B, H, W, prec == 2).total;
}

// The input gradient on the SPLIT storage (prec 2): the forward recompute keeps every activation as the hi | lo f16 operand planes
// the next conv stages by LDS-DMA (the 2x2 max pool fused into the producing conv's epilogue), and every gradient w.r.t. a conv's
// pre-activation is written once by gate_split_kernel (ReLU gate + max-pool routing + the tap's gradient + split) at a power-of-two
// scale bounded by the |max| word its producer left behind (a conv epilogue, the upsampling adjoint) — the chain of
// backward_params_split without the weight gradients.  Replaces fp32 tensors re-split inside every conv, three relu_gate_absmax
// passes and the absmax_partial / finalize pairs.
static int backward_input_split(dfn_dfnet_t h, const float* x, int B, int H, int W, int upH, int upW, const float* grad_features,
                                int level_mask, float* grad_x, const DfBwdWs& w, hipStream_t s) {
  constexpr int prec = 2;
  const int n_enc = int(h->enc.size());
  const void* zeros = h->fc + zeros_offset(h->feat_dim);
  int deepest = 0;
  for (int t = 0; t < h->n_taps; ++t) if (level_mask >> t & 1) deepest = t;
  // ---- forward up to the deepest requested tap
  CHECK_HIP(launch_dfnet_prep(prec, x, B, H, W, w.prep, s), "dfnet bwd: prep");
  const void* cur = w.prep;
  int ch = H, cw = W, nblk = 1, last = -1;
  int lay_h[13], lay_w[13];
  for (int i = 0; i < n_enc; ++i) {
    const ConvSpec& sp = h->enc[i];
    lay_h[i] = ch; lay_w[i] = cw;
    const bool is_last = sp.tap == deepest;
    ConvArgs a{};
    a.in = cur; a.w = h->enc_packed[i].w[prec]; a.bias = h->enc_packed[i].bias_x3; a.out_scale = h->enc_packed[i].out_scale;
    a.out_act = w.act[i];
    a.out_pre = (sp.tap >= 0 && (level_mask >> sp.tap & 1)) ? w.tap[sp.tap] : nullptr;
    a.B = B; a.H = ch; a.W = cw; a.nblk_in = nblk; a.cout_blocks = sp.cout / 32; a.relu = 1;
    a.in_split = i > 0; a.zeros = zeros;
    a.out_split = 1 | 2 | 4;
    const bool pool_next = sp.pool_after && !is_last;
    if (pool_next) a.out_pool = w.pooled;   // (no two consecutive convs of the VGG stack are followed by a pool: one transient buffer)
    CHECK_HIP(launch_conv(prec, 3, i == 0 ? prep_sb(prec) : 16, a, s), "dfnet bwd: encoder conv");
    last = i;
    if (is_last) break;
    cur = w.act[i];
    nblk = sp.cout / 32;
    if (pool_next) { cur = w.pooled; ch /= 2; cw /= 2; }
  }
  CHECK_HIP(hipMemsetAsync(w.amax, 0, 64 * sizeof(unsigned), s), "dfnet bwd: clear |max| words");
  const size_t plane = size_t(128) * upH * upW;
  char* gbuf[3] = {w.gA, w.gB, w.gC};
  int act_idx = -1;
  bool g_pooled = false;
  for (int i = last; i >= 0; --i) {
    const ConvSpec& sp = h->enc[i];
    const int hh = lay_h[i], ww = lay_w[i];
    const float* g_tap = nullptr;
    const unsigned* am_tap = nullptr;
    if (sp.tap >= 0 && (level_mask >> sp.tap & 1)) {
      const int t = sp.tap;
      unsigned* am = w.amax + 16 + 4 * t;
      float* sl128 = w.slots + 32 + 4 * t, * sl64 = sl128 + 2;
      // adaptation layer forward (ReLU gate of its 1x1) on the split tap, kept split
      ConvArgs a{};
      a.in = w.tap[t]; a.w = h->ad1[t].w[prec]; a.bias = h->ad1[t].bias_x3; a.out_scale = h->ad1[t].out_scale; a.out_act = w.tmp64;
      a.B = B; a.H = hh; a.W = ww; a.nblk_in = sp.cout / 32; a.cout_blocks = 2; a.relu = 1;
      a.in_split = 1; a.out_split = 1; a.zeros = zeros;
      CHECK_HIP(launch_conv(prec, 1, 16, a, s), "dfnet bwd: adapt 1x1");
      CHECK_HIP(launch_upsample_backward(1, grad_features + size_t(t) * B * plane, plane, B, hh, ww, upH, upW, w.g128, s, am),
                "dfnet bwd: upsample");
      CHECK_HIP(launch_gate_split(reinterpret_cast<const float*>(w.g128), 0, nullptr, 0, nullptr, B, hh, ww, 4, am, nullptr, w.g128S, 1, sl128, s),
                "dfnet bwd: split d features");
      ConvArgs c{};
      c.in = w.g128S; c.w = h->ad5_dgrad[t].w[prec]; c.bias = h->ad5_dgrad[t].bias; c.out_scale = h->ad5_dgrad[t].out_scale; c.out_pre = w.g64;
      c.in_split = 1; c.zeros = zeros; c.dyn_scale = sl128; c.absmax_out = am + 1;
      c.B = B; c.H = hh; c.W = ww; c.nblk_in = 4; c.cout_blocks = 2; c.relu = 0;
      CHECK_HIP(launch_conv(prec, 5, 16, c, s), "dfnet bwd: adapt 5x5 dgrad");
      // ReLU gate of the 1x1's output; the split result goes where d z's split copy was (dead by now: 64 of its 128 channels' worth)
      CHECK_HIP(launch_gate_split(reinterpret_cast<const float*>(w.g64), 0, w.tmp64, 1, nullptr, B, hh, ww, 2, am + 1, nullptr, w.g128S, 1, sl64, s),
                "dfnet bwd: adapt gate");
      ConvArgs d{};
      d.in = w.g128S; d.w = h->ad1_dgrad[t].w[prec]; d.bias = h->ad1_dgrad[t].bias; d.out_scale = h->ad1_dgrad[t].out_scale; d.out_pre = w.gtap;
      d.in_split = 1; d.zeros = zeros; d.dyn_scale = sl64; d.absmax_out = am + 2;
      d.B = B; d.H = hh; d.W = ww; d.nblk_in = 2; d.cout_blocks = sp.cout / 32; d.relu = 0;
      CHECK_HIP(launch_conv(prec, 1, 16, d, s), "dfnet bwd: adapt 1x1 dgrad");
      g_tap = reinterpret_cast<const float*>(w.gtap);
      am_tap = am + 2;
    }
    // gate: gradient w.r.t. conv i's pre-activation (ReLU gate, max-pool routing, + the tap's), split
    const int pre_idx = act_idx == 0 ? 1 : 0;
    float* slot = w.slots + 2 * i;
    CHECK_HIP(launch_gate_split(act_idx < 0 ? nullptr : reinterpret_cast<const float*>(gbuf[act_idx]), g_pooled ? 1 : 0, w.act[i], 1, g_tap, B, hh,
                                ww, sp.cout / 32, act_idx < 0 ? nullptr : w.amax + i, am_tap, gbuf[pre_idx], 1, slot, s),
              "dfnet bwd: gate");
    // data gradient of conv i (w.r.t. conv i-1's output, pooled if a max pool sits between)
    int in_idx = 0;
    while (in_idx == pre_idx || in_idx == act_idx) ++in_idx;
    const int cin_p = (sp.cin + 63) / 64 * 64;
    ConvArgs e{};
    e.in = gbuf[pre_idx]; e.w = h->enc_dgrad[i].w[prec]; e.bias = h->enc_dgrad[i].bias; e.out_scale = h->enc_dgrad[i].out_scale;
    e.out_pre = gbuf[in_idx];
    e.in_split = 1; e.zeros = zeros; e.dyn_scale = slot; if (i > 0) e.absmax_out = w.amax + (i - 1);
    e.B = B; e.H = hh; e.W = ww; e.nblk_in = sp.cout / 32; e.cout_blocks = cin_p / 32; e.relu = 0;
    CHECK_HIP(launch_conv(prec, 3, 16, e, s), "dfnet bwd: encoder conv dgrad");
    if (i == 0) {
      CHECK_HIP(launch_unprep(prec, gbuf[in_idx], B, H, W, cin_p / 32, grad_x, s), "dfnet bwd: unprep");
      break;
    }
    act_idx = in_idx;
    g_pooled = h->enc[i - 1].pool_after;
  }
  return DFN_OK;
}

extern "C" int dfn_dfnet_backward_input(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int upH, int upW,
                                        const float* grad_features, int level_mask, float* grad_x, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (!h) return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_input: null handle");
  if (!h->committed) return set_error(DFN_ERR_STATE, "dfn_dfnet_backward_input: dfn_dfnet_commit() has not been called");
  if (prec == DFN_PREC_F16)
    return set_error(DFN_ERR_UNSUPPORTED, "dfn_dfnet_backward_input: gradients run in DFN_PREC_F16X3 or DFN_PREC_F32 only (f16 rounding of "
                     "every activation flips ReLU / max-pool gates: 0.1 relative L2)");
  if (prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3) return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_input: unknown precision %d", prec);
  if (h && h->committed)
    if (int rc = check_fresh(h, prec, "dfn_dfnet_backward_input")) return rc;
  level_mask &= (1 << h->n_taps) - 1;
  if (!x || !grad_features || !grad_x || !workspace || B < 1 || H < 32 || W < 32 || upH < 1 || upW < 1 || !level_mask)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_input: bad argument (need H,W >= 32 and a non-empty level_mask)");
  const DfBwdWs w = carve_df_bwd(h, static_cast<char*>(workspace), prec, B, H, W, prec == 2);
  if (w.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_input: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = HS(stream);
  if (prec == 2) return backward_input_split(h, x, B, H, W, upH, upW, grad_features, level_mask, grad_x, w, s);
  int deepest = 0;
  for (int t = 0; t < h->n_taps; ++t) if (level_mask >> t & 1) deepest = t;
  // ---- forward up to the deepest requested tap, keeping every activation
  CHECK_HIP(launch_dfnet_prep(prec, x, B, H, W, w.prep, s), "dfnet bwd: prep");
  const void* cur = w.prep;
  int ch = H, cw = W, nblk = 1, last = -1;
  int lay_h[13], lay_w[13];
  for (size_t i = 0; i < h->enc.size(); ++i) {
    const ConvSpec& sp = h->enc[i];
    lay_h[i] = ch; lay_w[i] = cw;
    ConvArgs a{};
    a.in = cur; a.w = h->enc_packed[i].w[prec]; a.bias = prec == 2 ? h->enc_packed[i].bias_x3 : h->enc_packed[i].bias; a.out_scale = h->enc_packed[i].out_scale;
    a.out_act = w.act[i];
    a.out_pre = (sp.tap >= 0 && (level_mask >> sp.tap & 1)) ? w.tap[sp.tap] : nullptr;
    a.B = B; a.H = ch; a.W = cw; a.nblk_in = nblk; a.cout_blocks = sp.cout / 32; a.relu = 1;
    CHECK_HIP(launch_conv(prec, 3, i == 0 ? prep_sb(prec) : 16, a, s), "dfnet bwd: encoder conv");
    last = int(i);
    if (sp.tap == deepest) break;
    cur = w.act[i];
    nblk = sp.cout / 32;
    if (sp.pool_after) {
      CHECK_HIP(launch_maxpool(prec, cur, B, ch, cw, nblk, w.pooled, s), "dfnet bwd: maxpool");
      cur = w.pooled;
      ch /= 2; cw /= 2;
    }
  }
  // ---- backward.  In split-f16 mode the gradient convs are split-f16 too: gradient magnitudes are arbitrary, so each
  // conv's input tensor gets a measured power-of-two operand scale (launch_absmax_scale -> ConvArgs::dyn_scale).
  const int gprec = prec;
  auto dyn = [&](const void* t, size_t n) -> const float* {
    if (gprec != 2) return nullptr;
    (void)launch_absmax_scale(static_cast<const float*>(t), n, w.scl + 8, w.scl, s);
    return w.scl;
  };
  // the ReLU gate in front of a split-f16 gradient conv measures that conv's operand scale (w.scl) in the same pass
  auto gate = [&](const void* g, const void* act, const void* add, size_t n, void* out) -> hipError_t {
    if (gprec == 2) return launch_relu_gate_scale(g, act, add, n, out, w.scl + 8, w.scl, s);
    return launch_relu_gate(prec, g, act, add, n, out, s);
  };
  const size_t plane = size_t(128) * upH * upW;
  // Two gradient buffers: the ReLU gate runs in place on the buffer holding g_act, the conv's data gradient goes to
  // the other one, and a max-pool's routed gradient reuses the (by then dead) gated buffer.
  char* gbuf[2] = {w.gA, w.gB};
  const void* g_act = nullptr;  // gradient w.r.t. the post-ReLU output of conv i (null = none yet)
  int act_idx = -1;
  for (int i = last; i >= 0; --i) {
    const ConvSpec& sp = h->enc[i];
    const int hh = lay_h[i], ww = lay_w[i];
    const size_t n_out = size_t(B) * hh * ww * sp.cout;
    const void* g_tap = nullptr;
    if (sp.tap >= 0 && (level_mask >> sp.tap & 1)) {
      const int t = sp.tap;
      // adaptation layer forward (ReLU gate of its 1x1), then its transposed convolutions
      ConvArgs a{};
      a.in = w.tap[t]; a.w = h->ad1[t].w[prec]; a.bias = prec == 2 ? h->ad1[t].bias_x3 : h->ad1[t].bias; a.out_scale = h->ad1[t].out_scale; a.out_act = w.tmp64;
      a.B = B; a.H = hh; a.W = ww; a.nblk_in = sp.cout / 32; a.cout_blocks = 2; a.relu = 1;
      CHECK_HIP(launch_conv(prec, 1, 16, a, s), "dfnet bwd: adapt 1x1");
      CHECK_HIP(launch_upsample_backward(prec, grad_features + size_t(t) * B * plane, plane, B, hh, ww, upH, upW, w.g128, s),
                "dfnet bwd: upsample");
      ConvArgs c{};
      c.in = w.g128; c.w = h->ad5_dgrad[t].w[gprec]; c.bias = h->ad5_dgrad[t].bias; c.out_scale = h->ad5_dgrad[t].out_scale; c.out_pre = w.g64;
      c.dyn_scale = dyn(w.g128, size_t(B) * hh * ww * 128);
      c.B = B; c.H = hh; c.W = ww; c.nblk_in = 4; c.cout_blocks = 2; c.relu = 0;
      CHECK_HIP(launch_conv(gprec, 5, 16, c, s), "dfnet bwd: adapt 5x5 dgrad");
      CHECK_HIP(gate(w.g64, w.tmp64, nullptr, size_t(B) * hh * ww * 64, w.g64), "dfnet bwd: adapt gate");
      ConvArgs d{};
      d.in = w.g64; d.w = h->ad1_dgrad[t].w[gprec]; d.bias = h->ad1_dgrad[t].bias; d.out_scale = h->ad1_dgrad[t].out_scale; d.out_pre = w.gtap;
      d.dyn_scale = gprec == 2 ? w.scl : nullptr;
      d.B = B; d.H = hh; d.W = ww; d.nblk_in = 2; d.cout_blocks = sp.cout / 32; d.relu = 0;
      CHECK_HIP(launch_conv(gprec, 1, 16, d, s), "dfnet bwd: adapt 1x1 dgrad");
      g_tap = w.gtap;
    }
    // gradient w.r.t. the conv's pre-activation: ReLU-gated trunk gradient + the tap's
    const int pre_idx = act_idx < 0 ? 0 : act_idx, in_idx = pre_idx ^ 1;
    CHECK_HIP(gate(g_act, w.act[i], g_tap, n_out, gbuf[pre_idx]), "dfnet bwd: relu gate");
    // data gradient of the conv
    const int cin_p = (sp.cin + 63) / 64 * 64;
    ConvArgs e{};
    e.in = gbuf[pre_idx]; e.w = h->enc_dgrad[i].w[gprec]; e.bias = h->enc_dgrad[i].bias; e.out_scale = h->enc_dgrad[i].out_scale; e.out_pre = gbuf[in_idx];
    e.dyn_scale = gprec == 2 ? w.scl : nullptr;
    e.B = B; e.H = hh; e.W = ww; e.nblk_in = sp.cout / 32; e.cout_blocks = cin_p / 32; e.relu = 0;
    CHECK_HIP(launch_conv(gprec, 3, 16, e, s), "dfnet bwd: encoder conv dgrad");
    if (i == 0) {
      CHECK_HIP(launch_unprep(prec, gbuf[in_idx], B, H, W, cin_p / 32, grad_x, s), "dfnet bwd: unprep");
      break;
    }
    if (h->enc[i - 1].pool_after) {  // the data gradient is w.r.t. the pooled tensor: route it to conv i-1's resolution
      CHECK_HIP(launch_maxpool_backward(prec, w.act[i - 1], gbuf[in_idx], B, lay_h[i - 1], lay_w[i - 1], h->enc[i - 1].cout / 32,
                                        gbuf[pre_idx], s),
                "dfnet bwd: maxpool");
      g_act = gbuf[pre_idx];
      act_idx = pre_idx;
    } else {
      g_act = gbuf[in_idx];
      act_idx = in_idx;
    }
  }
  return DFN_OK;
}

// ------------------------------------------------------------------------------------------ parameter gradients (pose path)
namespace {
constexpr size_t kWgradPartFloats = size_t(2048) * 9 * 1024;   // partial sums of launch_conv_wgrad (75.5 MB)
struct DfParamWs {
  DfBwdWs b;
  float *part, *pooled;
  char* gC;              // third gradient buffer (the side stream may still read the one a layer's gate wrote)
  float* scl_layer;      // [13][8]: per encoder conv [scale, 1/scale] of its gated gradient (read by the side stream's weight gradient)
  float* lvl_tmp64[3];   // per pyramid level: ReLU'd 1x1 output,
  float* lvl_z[3];       //   plain 5x5 output (BatchNorm input),
  float* lvl_bn[3];      //   BatchNorm work block (kBnWorkFloats)
  double* bn_part;
  // split-f16 training path (prec 2): activations / taps / 1x1 outputs above are kept in the split row-planar storage, and
  char* pooledS[13];     //   the 2x2-pooled activation of every conv a max-pool follows (input of the next conv and of its weight gradient),
  char* g128S;           //   the split copy of a level's BatchNorm-backward output (operand of the 5x5 weight / data gradient),
  float* part_b;         //   bias-gradient partials of the weight-gradient stream,
  unsigned* amax;        //   [64] |max| words the producers of gradient tensors leave behind (zeroed once per backward),
  float* scl_lvl;        //   [3][16]: per level [scale, 1/scale] of the split g128 and g64
  size_t part_floats, part_b_floats;
  size_t total;
};
DfParamWs carve_df_params(const dfn_dfnet_s* h, char* base, int prec, int B, int H, int W) {
  DfParamWs w{};
  w.b = carve_df_bwd(h, base, prec, B, H, W);
  size_t off = w.b.total;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return reinterpret_cast<float*>(p); };
  // weight-gradient partials: the fixed allowance of the fp32-input kernels, or what the split stream's chunk plan asks for
  w.part_floats = kWgradPartFloats;
  w.part_b_floats = 0;
  const int div[3] = {1, 4, 16};
  if (prec == 2) {
    auto need = [&](int ks, int hh, int ww, int cout, int cin, int b_min) {
      if (cout % 64 || cin % 64) return;
      // (the chunk count of a layer's plan is not monotonic in the batch: frames per workgroup vs chunks per frame)
      for (int b = b_min; b <= B; ++b) {
        size_t pf, pb;
        conv_wgrad_split_scratch(ks, b, hh, ww, cout, cin, &pf, &pb);
        if (pf > w.part_floats) w.part_floats = pf;
        if (pb > w.part_b_floats) w.part_b_floats = pb;
      }
    };
    int ch = H, cw = W;
    for (size_t i = 0; i < h->enc.size(); ++i) {
      if (i > 0) need(3, ch, cw, h->enc[i].cout, h->enc[i].cin, B);
      if (h->enc[i].pool_after) { ch /= 2; cw /= 2; }
    }
    for (int t = 0; t < h->n_taps; ++t) {
      // the adaptation layers may run on the leading feature_images <= B frames of the batch (forward_train_keep)
      need(5, H / div[t], W / div[t], 128, 64, 1);
      need(1, H / div[t], W / div[t], 64, h->tap_channels[t], 1);
    }
  }
  w.part = take(w.part_floats * 4);
  w.pooled = take(size_t(B) * ((H / 16 + 1) / 2) * 512 * 4);   // pose head: row-pair partials of pool5's mean (relu5_3 is H/16 rows)
  w.gC = reinterpret_cast<char*>(take(size_t(B) * H * W * 64 * (prec == 0 ? 2 : 4)));
  w.scl_layer = take(13 * 8 * 4);
  for (int t = 0; t < h->n_taps; ++t) {
    const size_t q = size_t(B) * (H / div[t]) * (W / div[t]);
    w.lvl_tmp64[t] = take(q * 64 * 4);
    w.lvl_z[t] = take(q * 128 * 4);
    w.lvl_bn[t] = take(kBnWorkFloats * 4);
  }
  w.bn_part = reinterpret_cast<double*>(take(kBnPartBytes));
  if (prec == 2) {
    int ch = H, cw = W;
    for (size_t i = 0; i < h->enc.size(); ++i) {
      if (h->enc[i].pool_after) {
        w.pooledS[i] = reinterpret_cast<char*>(take(size_t(B) * (ch / 2) * (cw / 2) * h->enc[i].cout * 4));
        ch /= 2; cw /= 2;
      }
    }
    w.g128S = reinterpret_cast<char*>(take(size_t(B) * H * W * 128 * 4));
    w.part_b = take((w.part_b_floats ? w.part_b_floats : 64) * 4);
    w.amax = reinterpret_cast<unsigned*>(take(64 * 4));
    w.scl_lvl = take(3 * 16 * 4);
  }
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t dfn_dfnet_backward_params_workspace_bytes(dfn_dfnet_t h, int prec, int B, int H, int W) {
  if (!h || B < 1 || H < 1 || W < 1) return 0;
  return carve_df_params(h, nullptr, prec, B, H, W).total;
}

namespace {
// Encoder forward keeping every activation (and the pre-ReLU taps of the levels in tap_mask) in the params workspace.
// prec 2 (split-f16): everything between the convs is kept in the split row-planar storage (dfnet_conv.hip: split_piece) — the
// operands the next conv, the data-gradient chain and the weight-gradient stream multiply, split once by the producing epilogue —
// except conv1_1's fp32 input (its weight gradient gathers RGB taps) and relu5_3 (the pose head reads fp32); a conv followed by
// the 2x2 max pool also writes the pooled activation (pw.pooledS).
// on_tap (optional): called right after the conv whose pre-ReLU output is level t's tap has been launched — (layer index) -> rc.
int encoder_keep(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int tap_mask, const DfParamWs& pw, hipStream_t s,
                 int* lay_h, int* lay_w, const std::function<int(int)>* on_tap = nullptr) {
  const DfBwdWs& w = pw.b;
  const int n_enc = int(h->enc.size());
  const bool split = prec == 2;
  const void* zeros = h->fc + zeros_offset(h->feat_dim);
  CHECK_HIP(launch_dfnet_prep(prec, x, B, H, W, w.prep, s), "dfnet train: prep");
  const void* cur = w.prep;
  int ch = H, cw = W, nblk = 1;
  for (int i = 0; i < n_enc; ++i) {
    const ConvSpec& sp = h->enc[i];
    lay_h[i] = ch; lay_w[i] = cw;
    ConvArgs a{};
    a.in = cur; a.w = h->enc_packed[i].w[prec]; a.bias = prec == 2 ? h->enc_packed[i].bias_x3 : h->enc_packed[i].bias;
    a.out_scale = h->enc_packed[i].out_scale;
    a.out_act = w.act[i];
    a.out_pre = (sp.tap >= 0 && (tap_mask >> sp.tap & 1)) ? w.tap[sp.tap] : nullptr;
    a.B = B; a.H = ch; a.W = cw; a.nblk_in = nblk; a.cout_blocks = sp.cout / 32; a.relu = 1;
    const bool pool_next = sp.pool_after && i + 1 < n_enc;
    if (split) {
      a.in_split = i > 0; a.zeros = zeros;
      a.out_split = (i + 1 == n_enc ? 0 : 1) | 2 | 4;
      if (pool_next) a.out_pool = pw.pooledS[i];
    }
    CHECK_HIP(launch_conv(prec, 3, i == 0 ? prep_sb(prec) : 16, a, s), "dfnet train: encoder conv");
    if (on_tap && a.out_pre)
      if (int rc = (*on_tap)(i)) return rc;
    cur = w.act[i];
    nblk = sp.cout / 32;
    if (pool_next) {
      if (split) cur = pw.pooledS[i];
      else {
        CHECK_HIP(launch_maxpool(prec, cur, B, ch, cw, nblk, w.pooled, s), "dfnet train: maxpool");
        cur = w.pooled;
      }
      ch /= 2; cw /= 2;
    }
  }
  return DFN_OK;
}

// Adaptation layer of level t on the kept tap: lvl_tmp64 = ReLU(1x1); need_z: lvl_z = plain 5x5; BatchNorm work block
// from batch (bn_batch, optionally reported) or running statistics.  prec 2: tap and lvl_tmp64 in the split storage, lvl_z fp32.
int adapt_keep(dfn_dfnet_t h, int prec, int t, int B, int hh, int ww, int cin, bool bn_batch, bool need_z, const DfParamWs& pw,
               hipStream_t s, float* mean_out, float* var_out) {
  const bool split = prec == 2;
  const void* zeros = h->fc + zeros_offset(h->feat_dim);
  ConvArgs a{};
  a.in = pw.b.tap[t]; a.w = h->ad1[t].w[prec]; a.bias = prec == 2 ? h->ad1[t].bias_x3 : h->ad1[t].bias; a.out_scale = h->ad1[t].out_scale;
  a.out_act = pw.lvl_tmp64[t];
  a.B = B; a.H = hh; a.W = ww; a.nblk_in = cin / 32; a.cout_blocks = 2; a.relu = 1;
  if (split) { a.in_split = 1; a.out_split = 1; a.zeros = zeros; }
  CHECK_HIP(launch_conv(prec, 1, 16, a, s), "dfnet train: adapt 1x1");
  if (need_z) {
    ConvArgs z{};
    z.in = pw.lvl_tmp64[t]; z.w = h->ad5_raw[t].w[prec]; z.bias = prec == 2 ? h->ad5_raw[t].bias_x3 : h->ad5_raw[t].bias;
    z.out_scale = h->ad5_raw[t].out_scale; z.out_act = pw.lvl_z[t];
    z.B = B; z.H = hh; z.W = ww; z.nblk_in = 2; z.cout_blocks = 4; z.relu = 0;
    if (split) { z.in_split = 1; z.zeros = zeros; }
    CHECK_HIP(launch_conv(prec, 5, 16, z, s), "dfnet train: adapt 5x5");
  }
  if (bn_batch)
    CHECK_HIP(launch_bn_batch_stats(pw.lvl_z[t], (long long)B * hh * ww, h->bn_dev[t], 1e-5f, pw.bn_part, pw.lvl_bn[t], mean_out, var_out, s),
              "dfnet train: BatchNorm batch statistics");
  else
    CHECK_HIP(launch_bn_running_stats(h->bn_dev[t], 1e-5f, pw.lvl_bn[t], s), "dfnet train: BatchNorm running statistics");
  return DFN_OK;
}

// The backward of backward_params_core on the SPLIT storage (prec 2, the default): every gradient w.r.t. a conv's pre-activation is
// written once by gate_split_kernel — ReLU gate, max-pool routing and the tap's gradient in one pass — as the hi | lo f16 operand
// planes that BOTH consumers multiply: the data-gradient conv (conv_x3s_kernel: LDS-DMA staged, no conversion) and the
// weight-gradient stream (dfnet_wgrad_s.hip: no conversion, bias gradient as an extra column).  Its power-of-two operand scale comes
// from the |max| word the PRODUCER of the incoming gradient left behind (a conv epilogue's atomicMax, pw.amax), so no extra pass
// measures anything.  The old chain wrote fp32, measured it, and re-split it per (block pair, kernel row) in every consumer.
// Where the gradient w.r.t. the features comes from when it is NOT a pair of enlarged stacks: the pyramid triplet loss
// (dfnet_triplet_pyr.hip) writes d L / d (BatchNorm output) of every level at low resolution from its own row statistics.
struct TripletSrc { const float* grad_loss; TripletState st; int f1_half; int feat_images; };
static hipError_t level_feature_gradient(const TripletSrc* ts, const float* grad_features, const DfParamWs& pw, int t, int L, int B, int hh,
                                         int ww, int upH, int upW, void* g128, hipStream_t s) {
  if (ts)   // (B here = the siamese images, the leading ts->feat_images of the batch)
    return launch_triplet_pyr_backward(pw.lvl_z[t], pw.lvl_bn[t], hh, ww, upH, upW, B / 2, ts->f1_half, t, L, 1e-6f, ts->st.case_dev,
                                       ts->st.row_stat, ts->st.margin, ts->grad_loss, static_cast<float*>(g128), s);
  return launch_upsample_backward(1, grad_features + size_t(t) * B * (size_t(128) * upH * upW), size_t(128) * upH * upW, B, hh, ww, upH, upW,
                                  g128, s);
}

int backward_params_split(dfn_dfnet_t h, const float* x, int B, int H, int W, const float* grad_pose, const float* grad_features, int upH,
                          int upW, int level_mask, int bn_batch, int have_forward, float* const* grads, const DfParamWs& pw,
                          const int* lay_h, const int* lay_w, hipStream_t s, const char* fn, const TripletSrc* ts) {
  constexpr int prec = 2;
  const DfBwdWs& w = pw.b;
  const int n_enc = int(h->enc.size());
  const int per_tap = bn_batch ? 6 : 4;
  const void* zeros = h->fc + zeros_offset(h->feat_dim);
  // Bf: the leading images that went through the adaptation layers — all of them, or the siamese pair of a batch that also carries
  // extra pose-only frames (run_feature.py's synthesised views in ONE encoder pass): the levels' branches run on Bf images, and their
  // tap gradient joins the trunk's for those images only
  const int Bf = ts && ts->feat_images > 0 ? ts->feat_images : B;
  if (int rc = ensure_side(h)) return rc;
  hipStream_t side = h->side;
  SideJoin side_join{h, s, true};
  char* gbuf[3] = {w.gA, w.gB, pw.gC};
  bool side_reads[3] = {false, false, false};
  int side_seq[3] = {0, 0, 0}, seq = 0;
  static const bool adapt_on_side = [] { const char* e = getenv("DFN_ADAPT_WGRAD_SIDE"); return !(e && e[0] == '0'); }();
  bool side_reads_adapt = false;
  auto fork_side = [&]() -> hipError_t {
    hipError_t e = hipEventRecord(h->ev_fork, s);
    return e != hipSuccess ? e : hipStreamWaitEvent(side, h->ev_fork, 0);
  };
  auto side_done_with = [&](int k) -> hipError_t {
    if (k < 0 || !side_reads[k]) return hipSuccess;
    side_reads[k] = false;
    return hipStreamWaitEvent(s, h->ev_buf[k], 0);
  };
  auto join_side = [&]() -> hipError_t {
    hipError_t e = hipEventRecord(h->ev_join, side);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, h->ev_join, 0);
    side_reads[0] = side_reads[1] = side_reads[2] = false;
    return e;
  };
  auto pick_out = [&](int not_a, int not_b) {
    int k = -1;
    for (int c = 0; c < 3; ++c) {
      if (c == not_a || c == not_b) continue;
      if (k < 0 || (side_reads[k] && (!side_reads[c] || side_seq[c] < side_seq[k]))) k = c;
    }
    return k;
  };
  // |max| words: [i] = bound of the gradient w.r.t. conv i's (pooled) output, [16 + 4 t + {0, 1, 2}] = level t's d z, g64, g tap
  CHECK_HIP(hipMemsetAsync(pw.amax, 0, 64 * sizeof(unsigned), s), "dfnet params: clear |max| words");
  CHECK_HIP(fork_side(), "dfnet params: side stream");
  int act_idx = -1, last = -1;
  bool g_pooled = false;   // the gradient in gbuf[act_idx] is w.r.t. the POOLED output of conv i
  if (grad_pose) {
    if (lay_h[n_enc - 1] < 2 || lay_w[n_enc - 1] < 2) return set_error(DFN_ERR_ARG, "%s: image too small for pool5", fn);
    CHECK_HIP(launch_pose_head_backward(reinterpret_cast<const float*>(w.act[n_enc - 1]), B, lay_h[n_enc - 1], lay_w[n_enc - 1], h->fc,
                                        grad_pose, h->feat_dim, pw.pooled, reinterpret_cast<float*>(gbuf[0]), grads[2 * n_enc],
                                        grads[2 * n_enc + 1], s, pw.amax + (n_enc - 1)),
              "dfnet params: pose head");
    act_idx = 0;
    last = n_enc - 1;
  } else {
    CHECK_HIP(hipMemsetAsync(grads[2 * n_enc], 0, size_t(h->feat_dim) * 512 * 4, s), "dfnet params: zero fc gradient");
    CHECK_HIP(hipMemsetAsync(grads[2 * n_enc + 1], 0, size_t(h->feat_dim) * 4, s), "dfnet params: zero fc gradient");
    for (int i = 0; i < n_enc; ++i)
      if (h->enc[i].tap >= 0 && (level_mask >> h->enc[i].tap & 1)) last = i;
    for (int i = last + 1; i < n_enc; ++i) {
      CHECK_HIP(hipMemsetAsync(grads[2 * i], 0, size_t(h->enc[i].cout) * h->enc[i].cin * 9 * 4, s), "dfnet params: zero");
      CHECK_HIP(hipMemsetAsync(grads[2 * i + 1], 0, size_t(h->enc[i].cout) * 4, s), "dfnet params: zero");
    }
  }
  const size_t plane = size_t(128) * upH * upW;
  for (int i = last; i >= 0; --i) {
    const ConvSpec& sp = h->enc[i];
    const int hh = lay_h[i], ww = lay_w[i];
    const float* g_tap = nullptr;
    const unsigned* am_tap = nullptr;
    if (sp.tap >= 0 && (level_mask >> sp.tap & 1)) {
      const int t = sp.tap;
      float* const* ag = grads + 2 * n_enc + 2 + per_tap * t;
      const long long Q = (long long)Bf * hh * ww;
      unsigned* am = pw.amax + 16 + 4 * t;
      float* sl128 = pw.scl_lvl + 16 * t, * sl64 = sl128 + 8;
      // The level's two weight gradients run on the side stream like the encoder's (pw.part stays the side stream's alone), each as soon
      // as the chain has written its split operand; the chain itself goes on with the data gradients.  Before it overwrites the split
      // operands of the level before (pw.g128S, w.tmp64) it waits for the side stream's last read of them.
      if (adapt_on_side && side_reads_adapt) { CHECK_HIP(hipStreamWaitEvent(s, h->ev_adapt, 0), "dfnet params: side stream"); side_reads_adapt = false; }
      if (!adapt_on_side) CHECK_HIP(join_side(), "dfnet params: side stream");   // (A/B: the block's weight gradients on the chain's stream)
      hipStream_t ws = adapt_on_side ? side : s;
      if (!have_forward)
        if (int rc = adapt_keep(h, prec, t, Bf, hh, ww, sp.cout, bn_batch != 0, bn_batch != 0, pw, s, nullptr, nullptr)) return rc;
      CHECK_HIP(level_feature_gradient(ts, grad_features, pw, t, h->n_taps, Bf, hh, ww, upH, upW, w.g128, s), "dfnet params: upsample backward");
      // BatchNorm backward: d L/d y -> d L/d z (z = the plain 5x5 output), in place; leaves max |d z| behind
      CHECK_HIP(launch_bn_backward(bn_batch, reinterpret_cast<float*>(w.g128), pw.lvl_z[t], Q, pw.bn_part, pw.lvl_bn[t], bn_batch ? ag[4] : nullptr,
                                   bn_batch ? ag[5] : nullptr, s, am),
                "dfnet params: BatchNorm backward");
      CHECK_HIP(launch_gate_split(reinterpret_cast<const float*>(w.g128), 0, nullptr, 0, nullptr, Bf, hh, ww, 4, am, nullptr, pw.g128S, 1, sl128, s),
                "dfnet params: split d z");
      if (adapt_on_side) CHECK_HIP(fork_side(), "dfnet params: side stream");
      CHECK_HIP(launch_conv_wgrad_split(5, pw.g128S, pw.lvl_tmp64[t], zeros, Bf, hh, ww, 128, 64, pw.part, pw.part_floats, pw.part_b,
                                        pw.part_b_floats, ag[2], ag[3], sl128, ws),
                "dfnet params: adapt 5x5 weight gradient");
      ConvArgs c{};
      const PackedConv& d5 = h->ad5_raw_dgrad[t];
      c.in = pw.g128S; c.w = d5.w[prec]; c.bias = d5.bias; c.out_scale = d5.out_scale; c.out_pre = w.g64;
      c.in_split = 1; c.zeros = zeros; c.dyn_scale = sl128; c.absmax_out = am + 1;
      c.B = Bf; c.H = hh; c.W = ww; c.nblk_in = 4; c.cout_blocks = 2; c.relu = 0;
      CHECK_HIP(launch_conv(prec, 5, 16, c, s), "dfnet params: adapt 5x5 dgrad");
      // ReLU gate of the 1x1's output into the split storage (w.tmp64 is free: the kept 1x1 output lives in pw.lvl_tmp64)
      CHECK_HIP(launch_gate_split(reinterpret_cast<const float*>(w.g64), 0, pw.lvl_tmp64[t], 1, nullptr, Bf, hh, ww, 2, am + 1, nullptr, w.tmp64, 1,
                                  sl64, s),
                "dfnet params: adapt gate");
      if (adapt_on_side) CHECK_HIP(fork_side(), "dfnet params: side stream");
      CHECK_HIP(launch_conv_wgrad_split(1, w.tmp64, w.tap[t], zeros, Bf, hh, ww, 64, sp.cout, pw.part, pw.part_floats, pw.part_b,
                                        pw.part_b_floats, ag[0], ag[1], sl64, ws),
                "dfnet params: adapt 1x1 weight gradient");
      if (adapt_on_side) { CHECK_HIP(hipEventRecord(h->ev_adapt, side), "dfnet params: side stream"); side_reads_adapt = true; }
      ConvArgs d{};
      d.in = w.tmp64; d.w = h->ad1_dgrad[t].w[prec]; d.bias = h->ad1_dgrad[t].bias; d.out_scale = h->ad1_dgrad[t].out_scale; d.out_pre = w.gtap;
      d.in_split = 1; d.zeros = zeros; d.dyn_scale = sl64; d.absmax_out = am + 2;
      d.B = Bf; d.H = hh; d.W = ww; d.nblk_in = 2; d.cout_blocks = sp.cout / 32; d.relu = 0;
      CHECK_HIP(launch_conv(prec, 1, 16, d, s), "dfnet params: adapt 1x1 dgrad");
      g_tap = reinterpret_cast<const float*>(w.gtap);
      am_tap = am + 2;
      CHECK_HIP(fork_side(), "dfnet params: side stream");   // pw.part is the side stream's again, after this block's weight gradients
    }
    // ---- gate: gradient w.r.t. conv i's pre-activation, split (fp32 for conv1_1, whose weight gradient gathers from fp32)
    const int pre_idx = pick_out(act_idx, -1);
    CHECK_HIP(side_done_with(pre_idx), "dfnet params: side stream");
    float* slot = pw.scl_layer + 8 * i;
    const bool act_is_split = i + 1 < n_enc;
    CHECK_HIP(launch_gate_split(act_idx < 0 ? nullptr : reinterpret_cast<const float*>(gbuf[act_idx]), g_pooled ? 1 : 0, w.act[i], act_is_split,
                                g_tap, B, hh, ww, sp.cout / 32, act_idx < 0 ? nullptr : pw.amax + i, am_tap, gbuf[pre_idx], i > 0, slot, s, Bf),
              "dfnet params: gate");
    // ---- side stream: weight + bias gradient of conv i
    CHECK_HIP(fork_side(), "dfnet params: side stream");
    if (i == 0) {
      const float* g_pre = reinterpret_cast<const float*>(gbuf[pre_idx]);
      // (conv1_1's bias gradient is the 28th column of the same matrix product: a column of ones beside the 27 taps)
      CHECK_HIP(launch_conv0_wgrad(g_pre, reinterpret_cast<const float*>(w.prep), B, hh, ww, 2 * prep_sb(prec), pw.part, pw.part_floats,
                                   grads[0], side, slot, grads[1]),
                "dfnet params: conv1_1 weight + bias gradient");
      break;
    }
    const void* input = h->enc[i - 1].pool_after ? pw.pooledS[i - 1] : w.act[i - 1];
    CHECK_HIP(launch_conv_wgrad_split(3, gbuf[pre_idx], input, zeros, B, hh, ww, sp.cout, sp.cin, pw.part, pw.part_floats, pw.part_b,
                                      pw.part_b_floats, grads[2 * i], grads[2 * i + 1], slot, side),
              "dfnet params: conv weight gradient");
    CHECK_HIP(hipEventRecord(h->ev_buf[pre_idx], side), "dfnet params: side stream");
    side_reads[pre_idx] = true;
    side_seq[pre_idx] = ++seq;
    // ---- the chain: data gradient of conv i (w.r.t. its input: conv i-1's output, pooled if a max-pool sits between)
    const int in_idx = pick_out(pre_idx, -1);
    CHECK_HIP(side_done_with(in_idx), "dfnet params: side stream");
    ConvArgs e{};
    e.in = gbuf[pre_idx]; e.w = h->enc_dgrad[i].w[prec]; e.bias = h->enc_dgrad[i].bias; e.out_scale = h->enc_dgrad[i].out_scale;
    e.out_pre = gbuf[in_idx];
    e.in_split = 1; e.zeros = zeros; e.dyn_scale = slot; e.absmax_out = pw.amax + (i - 1);
    e.B = B; e.H = hh; e.W = ww; e.nblk_in = sp.cout / 32; e.cout_blocks = ((sp.cin + 63) / 64 * 64) / 32; e.relu = 0;
    CHECK_HIP(launch_conv(prec, 3, 16, e, s), "dfnet params: encoder conv dgrad");
    act_idx = in_idx;
    g_pooled = h->enc[i - 1].pool_after;
  }
  side_join.armed = false;
  CHECK_HIP(join_side(), "dfnet params: side stream");
  return DFN_OK;
}

// Parameter gradients of DFNet for d L/d pose (optional) and d L/d features (optional, single-stream layout, levels by
// level_mask).  grads: [0, 2 n_enc) encoder conv weight, bias; then fc_pose weight, bias; then (only when n_grads says
// so) per tap: adapt 1x1 weight [64,C,1,1], bias, adapt 5x5 weight [128,64,5,5], bias (the plain conv parameters, not
// the BatchNorm-folded ones) and, with bn_batch (BatchNorm on batch statistics), BatchNorm weight and bias.
int backward_params_core(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, const float* grad_pose,
                         const float* grad_features, int upH, int upW, int level_mask, int bn_batch, int have_forward,
                         float* const* grads, int n_grads, void* workspace, size_t workspace_bytes, hipStream_t s, const char* fn,
                         const TripletSrc* ts = nullptr) {
  if (!h) return set_error(DFN_ERR_ARG, "%s: null handle", fn);
  if (!h->committed) return set_error(DFN_ERR_STATE, "%s: dfn_dfnet_commit() has not been called", fn);
  if (prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3)
    return set_error(DFN_ERR_UNSUPPORTED, "%s: parameter gradients need fp32 activations (precision F32 or F16X3)", fn);
  if (int rc = check_fresh(h, prec, fn)) return rc;
  const int n_enc = int(h->enc.size());
  const bool feature_grads = grad_features || ts;
  level_mask = feature_grads ? (level_mask & ((1 << h->n_taps) - 1)) : 0;
  const int per_tap = bn_batch ? 6 : 4;
  const int want = 2 * n_enc + 2 + (feature_grads ? per_tap * h->n_taps : 0);
  if (ts && (!have_forward || ts->feat_images < 2 || (ts->feat_images & 1) || ts->feat_images > B || (ts->feat_images != B && prec != 2)))
    return set_error(DFN_ERR_ARG, "%s: the pyramid triplet gradient needs the kept siamese forward (feature_images even, <= B; a batch with extra "
                     "pose-only frames on the split-f16 path only)", fn);
  if (!x || (!grad_pose && !level_mask) || !grads || !workspace || B < 1 || H < 32 || W < 32 || n_grads != want ||
      (level_mask && (upH < 1 || upW < 1)))
    return set_error(DFN_ERR_ARG, "%s: bad argument (%d gradient pointers expected)", fn, want);
  for (int i = 0; i < n_grads; ++i)
    if (!grads[i]) return set_error(DFN_ERR_ARG, "%s: null gradient pointer %d", fn, i);
  const DfParamWs pw = carve_df_params(h, static_cast<char*>(workspace), prec, B, H, W);
  if (pw.total > workspace_bytes) return set_error(DFN_ERR_ARG, "%s: workspace too small (%zu < %zu)", fn, workspace_bytes, pw.total);
  const DfBwdWs& w = pw.b;
  int lay_h[13], lay_w[13];
  if (have_forward) {
    // the activations are already in the workspace: dfn_dfnet_forward_train(keep = 1) on this handle, same shape
    const dfn_dfnet_s::Kept* k = nullptr;
    for (const auto& e : h->kept) if (e.ws == workspace) k = &e;
    if (!k || k->prec != prec || k->B != B || k->H != H || k->W != W || (level_mask && k->bn_batch != (bn_batch ? 1 : 0)))
      return set_error(DFN_ERR_STATE, "%s: the workspace does not hold the state of a matching dfn_dfnet_forward_train(keep = 1)", fn);
    int ch = H, cw = W;
    for (int i = 0; i < n_enc; ++i) {
      lay_h[i] = ch; lay_w[i] = cw;
      if (h->enc[i].pool_after) { ch /= 2; cw /= 2; }
    }
  } else {
    // forward recompute, keeping every activation (and the pre-ReLU taps of the levels that carry gradient)
    if (int rc = encoder_keep(h, prec, x, B, H, W, level_mask, pw, s, lay_h, lay_w)) return rc;
  }
  if (prec == 2)
    return backward_params_split(h, x, B, H, W, grad_pose, grad_features, upH, upW, level_mask, bn_batch, have_forward, grads, pw, lay_h, lay_w,
                                 s, fn, ts);
  // (exact fp32) Three gradient buffers and the handle's side stream: a layer's bias gradient, the re-pooled conv input and its weight gradient
  // read the gated gradient g_pre(i) and nothing the data-gradient chain waits for, so they run on the side stream beside the
  // data-gradient conv of the same and of the next layer (small grids at training resolutions: both leave CUs idle).  The chain
  // picks its output buffers among those the side stream is done with (ev_buf[k] = its last read of buffer k); pw.part and w.pooled
  // belong to the side stream, a per-layer [scale, 1/scale] slot replaces the shared one.  Same kernels, same operands, same
  // summation order as the one-stream form: bit-identical gradients.
  if (int rc = ensure_side(h)) return rc;
  hipStream_t side = h->side;
  SideJoin side_join{h, s, true};
  char* gbuf[3] = {w.gA, w.gB, pw.gC};
  bool side_reads[3] = {false, false, false};
  int side_seq[3] = {0, 0, 0}, seq = 0;      // which of the side stream's reads is the oldest
  auto fork_side = [&]() -> hipError_t {       // the side stream continues after what the chain has launched so far
    hipError_t e = hipEventRecord(h->ev_fork, s);
    return e != hipSuccess ? e : hipStreamWaitEvent(side, h->ev_fork, 0);
  };
  auto side_done_with = [&](int k) -> hipError_t {   // the chain is about to overwrite buffer k
    if (!side_reads[k]) return hipSuccess;
    side_reads[k] = false;
    return hipStreamWaitEvent(s, h->ev_buf[k], 0);
  };
  auto join_side = [&]() -> hipError_t {
    hipError_t e = hipEventRecord(h->ev_join, side);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, h->ev_join, 0);
    side_reads[0] = side_reads[1] = side_reads[2] = false;
    return e;
  };
  auto pick_out = [&](int not_a, int not_b) {  // an output buffer: one the side stream is not reading if there is one
    int k = -1;
    for (int c = 0; c < 3; ++c) {
      if (c == not_a || c == not_b) continue;
      if (k < 0 || (side_reads[k] && (!side_reads[c] || side_seq[c] < side_seq[k]))) k = c;
    }
    return k;
  };
  CHECK_HIP(fork_side(), "dfnet params: side stream");   // pw.part / w.pooled: after whatever used them before this call
  int act_idx = -1, last = -1;   // act_idx: buffer holding the gradient w.r.t. conv i's ReLU output (none yet)
  if (grad_pose) {
    if (lay_h[n_enc - 1] < 2 || lay_w[n_enc - 1] < 2) return set_error(DFN_ERR_ARG, "%s: image too small for pool5", fn);
    CHECK_HIP(launch_pose_head_backward(reinterpret_cast<const float*>(w.act[n_enc - 1]), B, lay_h[n_enc - 1], lay_w[n_enc - 1], h->fc,
                                        grad_pose, h->feat_dim, pw.pooled, reinterpret_cast<float*>(gbuf[0]), grads[2 * n_enc],
                                        grads[2 * n_enc + 1], s),
              "dfnet params: pose head");
    act_idx = 0;
    last = n_enc - 1;
  } else {
    CHECK_HIP(hipMemsetAsync(grads[2 * n_enc], 0, size_t(h->feat_dim) * 512 * 4, s), "dfnet params: zero fc gradient");
    CHECK_HIP(hipMemsetAsync(grads[2 * n_enc + 1], 0, size_t(h->feat_dim) * 4, s), "dfnet params: zero fc gradient");
    for (int i = 0; i < n_enc; ++i)
      if (h->enc[i].tap >= 0 && (level_mask >> h->enc[i].tap & 1)) last = i;
    for (int i = last + 1; i < n_enc; ++i) {   // convs above the deepest level that carries gradient: zero
      CHECK_HIP(hipMemsetAsync(grads[2 * i], 0, size_t(h->enc[i].cout) * h->enc[i].cin * 9 * 4, s), "dfnet params: zero");
      CHECK_HIP(hipMemsetAsync(grads[2 * i + 1], 0, size_t(h->enc[i].cout) * 4, s), "dfnet params: zero");
    }
  }
  const size_t plane = size_t(128) * upH * upW;
  auto dyn = [&](const void* t, size_t n) -> const float* {   // measured operand scale of a gradient tensor (split-f16 convs)
    if (prec != 2) return nullptr;
    (void)launch_absmax_scale(static_cast<const float*>(t), n, w.scl + 8, w.scl, s);
    return w.scl;
  };
  // the ReLU gate in front of split-f16 gradient products measures their operand scale (slot) in the same pass
  auto gate = [&](const void* g, const void* act, const void* add, size_t n, void* out, float* slot) -> hipError_t {
    if (prec == 2) return launch_relu_gate_scale(g, act, add, n, out, w.scl + 8, slot, s);
    return launch_relu_gate(1, g, act, add, n, out, s);
  };
  // ---- encoder, last conv first; the adaptation layers of a level join at its tap
  for (int i = last; i >= 0; --i) {
    const ConvSpec& sp = h->enc[i];
    const int hh = lay_h[i], ww = lay_w[i];
    const void* g_tap = nullptr;
    if (sp.tap >= 0 && (level_mask >> sp.tap & 1)) {
      const int t = sp.tap;
      float* const* ag = grads + 2 * n_enc + 2 + per_tap * t;
      const long long Q = (long long)B * hh * ww;
      CHECK_HIP(join_side(), "dfnet params: side stream");   // this block's reductions use pw.part on the chain's stream
      if (!have_forward)
        if (int rc = adapt_keep(h, prec, t, B, hh, ww, sp.cout, bn_batch != 0, bn_batch != 0, pw, s, nullptr, nullptr)) return rc;
      const float* tmp64 = pw.lvl_tmp64[t];
      CHECK_HIP(level_feature_gradient(ts, grad_features, pw, t, h->n_taps, B, hh, ww, upH, upW, w.g128, s), "dfnet params: upsample backward");
      // BatchNorm backward: d L/d y -> d L/d z (z = the plain 5x5 output), in place
      CHECK_HIP(launch_bn_backward(bn_batch, reinterpret_cast<float*>(w.g128), pw.lvl_z[t], Q, pw.bn_part, pw.lvl_bn[t], bn_batch ? ag[4] : nullptr,
                                   bn_batch ? ag[5] : nullptr, s),
                "dfnet params: BatchNorm backward");
      const float* g128 = reinterpret_cast<const float*>(w.g128);
      CHECK_HIP(launch_bias_grad(g128, B, hh, ww, 128, pw.part, kWgradPartFloats, ag[3], s), "dfnet params: adapt 5x5 bias gradient");
      const float* sc128 = dyn(w.g128, size_t(B) * hh * ww * 128);   // one measured scale for the weight AND data gradient products
      CHECK_HIP(launch_conv_wgrad(5, g128, tmp64, B, hh, ww, 128, 64, pw.part, kWgradPartFloats, ag[2], s, sc128),
                "dfnet params: adapt 5x5 weight gradient");
      ConvArgs c{};
      const PackedConv& d5 = h->ad5_raw_dgrad[t];
      c.in = w.g128; c.w = d5.w[prec]; c.bias = d5.bias; c.out_scale = d5.out_scale; c.out_pre = w.g64;
      c.dyn_scale = sc128;
      c.B = B; c.H = hh; c.W = ww; c.nblk_in = 4; c.cout_blocks = 2; c.relu = 0;
      CHECK_HIP(launch_conv(prec, 5, 16, c, s), "dfnet params: adapt 5x5 dgrad");
      CHECK_HIP(gate(w.g64, tmp64, nullptr, size_t(B) * hh * ww * 64, w.g64, w.scl), "dfnet params: adapt gate");
      const float* g64 = reinterpret_cast<const float*>(w.g64);
      CHECK_HIP(launch_bias_grad(g64, B, hh, ww, 64, pw.part, kWgradPartFloats, ag[1], s), "dfnet params: adapt 1x1 bias gradient");
      const float* sc64 = prec == 2 ? w.scl : nullptr;
      CHECK_HIP(launch_conv_wgrad(1, g64, reinterpret_cast<const float*>(w.tap[t]), B, hh, ww, 64, sp.cout, pw.part, kWgradPartFloats, ag[0], s,
                                  sc64),
                "dfnet params: adapt 1x1 weight gradient");
      ConvArgs d{};
      d.in = w.g64; d.w = h->ad1_dgrad[t].w[prec]; d.bias = h->ad1_dgrad[t].bias; d.out_scale = h->ad1_dgrad[t].out_scale; d.out_pre = w.gtap;
      d.dyn_scale = sc64;
      d.B = B; d.H = hh; d.W = ww; d.nblk_in = 2; d.cout_blocks = sp.cout / 32; d.relu = 0;
      CHECK_HIP(launch_conv(prec, 1, 16, d, s), "dfnet params: adapt 1x1 dgrad");
      g_tap = w.gtap;
      CHECK_HIP(fork_side(), "dfnet params: side stream");   // pw.part is the side stream's again, after this block's reductions
    }
    const int pre_idx = act_idx < 0 ? 0 : act_idx;
    float* slot = pw.scl_layer + 8 * i;
    if (act_idx < 0) CHECK_HIP(side_done_with(pre_idx), "dfnet params: side stream");
    CHECK_HIP(gate(act_idx < 0 ? nullptr : gbuf[act_idx], w.act[i], g_tap, size_t(B) * hh * ww * sp.cout, gbuf[pre_idx], slot),
              "dfnet params: relu gate");
    const float* g_pre = reinterpret_cast<const float*>(gbuf[pre_idx]);
    const float* sc_pre = prec == 2 ? slot : nullptr;
    // ---- side stream: bias gradient, re-pooled input, weight gradient of conv i
    CHECK_HIP(fork_side(), "dfnet params: side stream");
    CHECK_HIP(launch_bias_grad(g_pre, B, hh, ww, sp.cout, pw.part, kWgradPartFloats, grads[2 * i + 1], side), "dfnet params: bias gradient");
    if (i == 0) {
      CHECK_HIP(launch_conv0_wgrad(g_pre, reinterpret_cast<const float*>(w.prep), B, hh, ww, 2 * prep_sb(prec), pw.part, kWgradPartFloats,
                                   grads[0], side, sc_pre),
                "dfnet params: conv1_1 weight gradient");
      break;
    }
    const void* input = w.act[i - 1];
    if (h->enc[i - 1].pool_after) {
      CHECK_HIP(launch_maxpool(prec, w.act[i - 1], B, lay_h[i - 1], lay_w[i - 1], h->enc[i - 1].cout / 32, w.pooled, side),
                "dfnet params: maxpool (conv input)");
      input = w.pooled;
    }
    CHECK_HIP(launch_conv_wgrad(3, g_pre, reinterpret_cast<const float*>(input), B, hh, ww, sp.cout, sp.cin, pw.part, kWgradPartFloats,
                                grads[2 * i], side, sc_pre),
              "dfnet params: conv weight gradient");
    CHECK_HIP(hipEventRecord(h->ev_buf[pre_idx], side), "dfnet params: side stream");
    side_reads[pre_idx] = true;
    side_seq[pre_idx] = ++seq;
    // ---- the chain: data gradient of conv i
    const int in_idx = pick_out(pre_idx, -1);
    CHECK_HIP(side_done_with(in_idx), "dfnet params: side stream");
    ConvArgs e{};
    e.in = gbuf[pre_idx]; e.w = h->enc_dgrad[i].w[prec]; e.bias = h->enc_dgrad[i].bias; e.out_scale = h->enc_dgrad[i].out_scale;
    e.out_pre = gbuf[in_idx];
    e.dyn_scale = sc_pre;
    e.B = B; e.H = hh; e.W = ww; e.nblk_in = sp.cout / 32; e.cout_blocks = ((sp.cin + 63) / 64 * 64) / 32; e.relu = 0;
    CHECK_HIP(launch_conv(prec, 3, 16, e, s), "dfnet params: encoder conv dgrad");
    if (h->enc[i - 1].pool_after) {
      const int up_idx = pick_out(in_idx, -1);
      CHECK_HIP(side_done_with(up_idx), "dfnet params: side stream");
      CHECK_HIP(launch_maxpool_backward(1, w.act[i - 1], gbuf[in_idx], B, lay_h[i - 1], lay_w[i - 1], h->enc[i - 1].cout / 32,
                                        gbuf[up_idx], s),
                "dfnet params: maxpool backward");
      act_idx = up_idx;
    } else {
      act_idx = in_idx;
    }
  }
  side_join.armed = false;
  CHECK_HIP(join_side(), "dfnet params: side stream");
  return DFN_OK;
}
}  // namespace

// dfn_dfnet_forward_train(keep = 1): the training forward on the params-workspace layout, leaving every activation, the
// pre-ReLU taps, each level's 1x1 / plain 5x5 outputs and BatchNorm work block in place for the backward.
// pyramid_only: the adaptation layers run and their plain 5x5 outputs + BatchNorm work blocks stay in the workspace (what the
// pyramid triplet loss and the backward read), but no level is enlarged into a feature stack (`features` is not touched).
static int forward_train_keep(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, int siamese, int return_pose, int bn_batch,
                              int upH, int upW, float* features, float* pose, float* bn_stats, void* workspace, size_t workspace_bytes,
                              void* stream, int pyramid_only, int feat_images) {
  const char* fn = pyramid_only ? "dfn_dfnet_forward_train_pyramid" : "dfn_dfnet_forward_train";
  const int Bf = feat_images > 0 ? feat_images : B;    // leading images that go through the adaptation layers (the siamese pair)
  if (Bf > B || (pyramid_only && (Bf & 1)) || (Bf != B && !pyramid_only))
    return set_error(DFN_ERR_ARG, "%s: feature_images must be even and <= B", fn);
  if (!h) return set_error(DFN_ERR_ARG, "%s: null handle", fn);
  if (!h->committed) return set_error(DFN_ERR_STATE, "%s: dfn_dfnet_commit() has not been called", fn);
  const bool levels = features || pyramid_only;     // the adaptation layers run
  if (!x || !workspace || (!levels && !return_pose) || B < 1 || H < 32 || W < 32 || (features && (upH < 1 || upW < 1)) ||
      (return_pose && !pose) || (levels && siamese && ((feat_images > 0 ? feat_images : B) & 1)) || (pyramid_only && bn_batch && !bn_stats))
    return set_error(DFN_ERR_ARG, "%s: bad argument (need H,W >= 32; even batch for siamese)", fn);
  if (int rc = check_fresh(h, prec, fn)) return rc;
  const DfParamWs pw = carve_df_params(h, static_cast<char*>(workspace), prec, B, H, W);
  if (pw.total > workspace_bytes)
    return set_error(DFN_ERR_ARG, "%s: keep = 1 needs dfn_dfnet_backward_params_workspace_bytes (%zu < %zu)", fn, workspace_bytes, pw.total);
  hipStream_t s = HS(stream);
  for (size_t i = 0; i < h->kept.size();)   // this workspace is being overwritten
    if (h->kept[i].ws == workspace) h->kept.erase(h->kept.begin() + i); else ++i;
  int lay_h[13], lay_w[13];
  const int n_enc = int(h->enc.size());
  const size_t plane = size_t(128) * upH * upW;
  // A level's adaptation branch (1x1, 5x5, BatchNorm statistics, enlargement) needs its tap only: it starts on the handle's side stream
  // as soon as the tapped conv is launched and runs beside the rest of the encoder, whose deep layers' small grids leave CUs idle
  // (as the inference forward does for levels 1-2, forward_core); the branches share pw.bn_part in the side stream's order.
  // DFN_ADAPT_FWD_SIDE=0: after the encoder on the caller's stream, for A/B.
  static const bool branches_beside = [] { const char* e = getenv("DFN_ADAPT_FWD_SIDE"); return !(e && e[0] == '0'); }();
  hipStream_t bs = s;
  SideJoin side_join{h, s, false};
  if (levels && branches_beside) {
    if (int rc = ensure_side(h)) return rc;
    bs = h->side;
  }
  auto level_branch = [&](int i) -> int {
    const int t = h->enc[i].tap;
    const int hh = lay_h[i], ww = lay_w[i];
    if (bs != s) {
      CHECK_HIP(hipEventRecord(h->ev_fork, s), "dfnet train: side stream");
      CHECK_HIP(hipStreamWaitEvent(bs, h->ev_fork, 0), "dfnet train: side stream");
      side_join.armed = true;
    }
    if (int rc = adapt_keep(h, prec, t, Bf, hh, ww, h->enc[i].cout, bn_batch != 0, true, pw, bs, bn_batch ? bn_stats + size_t(t) * 256 : nullptr,
                            bn_batch ? bn_stats + size_t(t) * 256 + 128 : nullptr))
      return rc;
    if (pyramid_only) return DFN_OK;
    if (!siamese) {
      CHECK_HIP(launch_upsample(prec, pw.lvl_z[t], B, hh, ww, upH, upW, features + size_t(t) * B * plane, plane, bs, pw.lvl_bn[t]),
                "dfnet train: upsample");
    } else {
      const int hb = B / 2;
      for (int half = 0; half < 2; ++half)
        CHECK_HIP(launch_upsample(prec, pw.lvl_z[t] + size_t(half) * hb * hh * ww * 128, hb, hh, ww, upH, upW,
                                  features + (size_t(half) * h->n_taps + t) * hb * plane, plane, bs, pw.lvl_bn[t]),
                  "dfnet train: upsample");
    }
    return DFN_OK;
  };
  const std::function<int(int)> on_tap = level_branch;
  if (int rc = encoder_keep(h, prec, x, B, H, W, levels ? (1 << h->n_taps) - 1 : 0, pw, s, lay_h, lay_w, levels ? &on_tap : nullptr)) return rc;
  if (return_pose) {
    if (lay_h[n_enc - 1] < 2 || lay_w[n_enc - 1] < 2) return set_error(DFN_ERR_ARG, "%s: image too small for pool5", fn);
    CHECK_HIP(launch_pose_head(prec, pw.b.act[n_enc - 1], B, lay_h[n_enc - 1], lay_w[n_enc - 1], h->fc, h->fc + size_t(h->feat_dim) * 512,
                               h->feat_dim, pw.pooled, pose, s),   // (pw.pooled: the backward's pose-head scratch, free until then)
              "dfnet train: pose head");
  }
  if (side_join.armed) {
    side_join.armed = false;
    CHECK_HIP(hipEventRecord(h->ev_join, h->side), "dfnet train: side stream");
    CHECK_HIP(hipStreamWaitEvent(s, h->ev_join, 0), "dfnet train: side stream");
  }
  if (h->kept.size() >= 8) h->kept.erase(h->kept.begin());
  h->kept.push_back({workspace, prec, B, H, W, levels ? (bn_batch ? 1 : 0) : -1});   // -1: no feature gradients from this state
  return DFN_OK;
}

extern "C" int dfn_dfnet_backward_params(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, const float* grad_pose,
                                         float* const* grads, int n_grads, void* workspace, size_t workspace_bytes, void* stream) {
  if (!grad_pose) return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_params: null grad_pose");
  return backward_params_core(h, prec, x, B, H, W, grad_pose, nullptr, 0, 0, 0, 0, 0, grads, n_grads, workspace, workspace_bytes,
                              HS(stream), "dfn_dfnet_backward_params");
}

extern "C" int dfn_dfnet_backward_all_params(dfn_dfnet_t h, int prec, const float* x, int B, int H, int W, const float* grad_pose,
                                             const float* grad_features, int upH, int upW, int level_mask, int bn_batch,
                                             int have_forward, float* const* grads, int n_grads, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  if (!grad_features && !have_forward)
    return set_error(DFN_ERR_ARG, "dfn_dfnet_backward_all_params: null grad_features (use dfn_dfnet_backward_params)");
  return backward_params_core(h, prec, x, B, H, W, grad_pose, grad_features, upH, upW, level_mask, bn_batch != 0, have_forward != 0,
                              grads, n_grads, workspace, workspace_bytes, HS(stream), "dfn_dfnet_backward_all_params");
}

// ------------------------------------------------------------------------------------------ triplet loss on the kept pyramid
// The siamese training forward without the enlarged stacks, the triplet loss of misc.py:355-435 from the low-resolution levels it
// keeps, and the backward that starts from that loss (dfnet_triplet_pyr.hip).  f1_half: which half of the batch misc.py's f1 (the
// anchor stack) is — run_feature.py:154 passes (features_rgb, features_target) = (second half, first half): f1_half = 1.
extern "C" int dfn_dfnet_forward_train_pyramid(dfn_dfnet_t h, int prec, const float* x, int B, int feature_images, int H, int W,
                                               int return_pose, int bn_batch, float* pose, float* bn_stats, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  if (prec != DFN_PREC_F32 && prec != DFN_PREC_F16X3)
    return set_error(DFN_ERR_UNSUPPORTED, "dfn_dfnet_forward_train_pyramid: batch statistics need fp32 activations (precision F32 or F16X3)");
  if (feature_images < 2 || (feature_images & 1) || feature_images > B || (feature_images != B && prec != DFN_PREC_F16X3))
    return set_error(DFN_ERR_ARG, "dfn_dfnet_forward_train_pyramid: feature_images must be even, >= 2 and <= B (< B: precision F16X3 only)");
  return forward_train_keep(h, prec, x, B, H, W, 1, return_pose, bn_batch, 0, 0, nullptr, pose, bn_stats, workspace, workspace_bytes, stream, 1,
                            feature_images);
}

extern "C" size_t dfn_dfnet_triplet_pyramid_state_bytes(dfn_dfnet_t h, int feature_images, int upH) {
  if (!h || feature_images < 2 || (feature_images & 1) || upH < 1) return 0;
  return triplet_state_bytes(h->n_taps, feature_images / 2, 128 * upH);
}

namespace {
// the kept state of `workspace` and the level geometry; nullptr (with the error set) when the workspace holds no matching forward
const dfn_dfnet_s::Kept* kept_levels(dfn_dfnet_t h, int prec, int B, int H, int W, const void* workspace, int* tap_h, int* tap_w,
                                     const char* fn) {
  const dfn_dfnet_s::Kept* k = nullptr;
  for (const auto& e : h->kept) if (e.ws == workspace) k = &e;
  if (!k || k->prec != prec || k->B != B || k->H != H || k->W != W || k->bn_batch < 0) {
    set_error(DFN_ERR_STATE, "%s: the workspace does not hold the levels of a matching dfn_dfnet_forward_train(_pyramid) (keep = 1)", fn);
    return nullptr;
  }
  int ch = H, cw = W;
  for (size_t i = 0; i < h->enc.size(); ++i) {
    if (h->enc[i].tap >= 0) { tap_h[h->enc[i].tap] = ch; tap_w[h->enc[i].tap] = cw; }
    if (h->enc[i].pool_after) { ch /= 2; cw /= 2; }
  }
  return k;
}
}  // namespace

extern "C" int dfn_dfnet_triplet_pyramid_forward(dfn_dfnet_t h, int prec, int B, int feature_images, int H, int W, int upH, int upW,
                                                 int f1_half, float margin, int mining, float* loss, void* state, size_t state_bytes,
                                                 void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "dfn_dfnet_triplet_pyramid_forward";
  if (!h || !loss || !state || !workspace || feature_images < 2 || (feature_images & 1) || feature_images > B || upH < 1 || upW < 1 ||
      mining < 0 || mining > 2 || (f1_half & ~1))
    return set_error(DFN_ERR_ARG, "%s: bad argument", fn);
  if (state_bytes < dfn_dfnet_triplet_pyramid_state_bytes(h, feature_images, upH)) return set_error(DFN_ERR_ARG, "%s: state too small", fn);
  int th[3], tw[3];
  if (!kept_levels(h, prec, B, H, W, workspace, th, tw, fn)) return DFN_ERR_STATE;
  const DfParamWs pw = carve_df_params(h, static_cast<char*>(workspace), prec, B, H, W);
  if (pw.total > workspace_bytes) return set_error(DFN_ERR_ARG, "%s: workspace too small", fn);
  hipStream_t s = HS(stream);
  const int L = h->n_taps, hb = feature_images / 2, blocks = triplet_pyr_blocks(hb, upH);
  const TripletState t = carve_triplet(state, L, hb, 128 * upH);
  for (int l = 0; l < L; ++l)
    CHECK_HIP(launch_triplet_pyr_forward(pw.lvl_z[l], pw.lvl_bn[l], th[l], tw[l], upH, upW, hb, f1_half, l, L, margin, 1e-6f, t.row_stat,
                                         t.part, l * blocks, s),
              "dfnet triplet: level rows");
  const double n_rows = (double)L * hb * 128 * upH;
  CHECK_HIP(launch_triplet_finalize(t.part, L * blocks, mining, n_rows * upW, n_rows, margin, t.case_dev, mining ? t.mse : nullptr, t.margin, loss, s),
            "dfnet triplet: finalize");
  return DFN_OK;
}

extern "C" int dfn_dfnet_backward_all_params_triplet(dfn_dfnet_t h, int prec, const float* x, int B, int feature_images, int H, int W,
                                                     const float* grad_pose, const float* grad_loss, const void* state, size_t state_bytes,
                                                     int f1_half, int upH, int upW, int bn_batch, float* const* grads, int n_grads,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "dfn_dfnet_backward_all_params_triplet";
  if (!h || !grad_loss || !state || feature_images < 2 || (feature_images & 1) || feature_images > B || upH < 1 || upW < 1 || (f1_half & ~1))
    return set_error(DFN_ERR_ARG, "%s: bad argument", fn);
  if (state_bytes < dfn_dfnet_triplet_pyramid_state_bytes(h, feature_images, upH)) return set_error(DFN_ERR_ARG, "%s: state too small", fn);
  const TripletSrc ts{grad_loss, carve_triplet(const_cast<void*>(state), h->n_taps, feature_images / 2, 128 * upH), f1_half, feature_images};
  return backward_params_core(h, prec, x, B, H, W, grad_pose, nullptr, upH, upW, (1 << h->n_taps) - 1, bn_batch != 0, 1, grads, n_grads,
                              workspace, workspace_bytes, HS(stream), fn, &ts);
}

// ------------------------------------------------------------------------------------------ device-side parameter refresh
// After optimizer.step() the pose path's fp32 master weights are in device memory: re-pack the 13 encoder convs (forward
// and data-gradient fragments, all three arithmetic modes) and fc_pose on the device.  `params`: HOST array of
// 2 * 13 + 2 DEVICE pointers in the order of dfn_dfnet_backward_params.  The split-f16 weight scale of each conv is
// kept from dfn_dfnet_commit (weights move slowly under fine-tuning; re-commit from the host to re-derive it).
namespace {
// Re-pack one convolution (forward fragments and data-gradient fragments, all three arithmetic modes) from device fp32
// master weights; the split-f16 weight scales are kept from dfn_dfnet_commit.
int refresh_conv(PackedConv& f, PackedConv& d, const float* wt, const float* bias, int cout, int cin, int ks, bool first, int prec_mask,
                 std::vector<PackJob>& jobs) {
  const float wscale = 1.f / (f.out_scale * kConvActScale);
  const int cop = (cin + 63) / 64 * 64;
  const float dscale = 1.f / (d.out_scale * kConvActScale);
  for (int prec = 0; prec < 3; ++prec) {
    if (!((prec_mask >> prec) & 1)) continue;
    const int sb = first ? prep_sb(prec) : 16;
    const int mbf = prec == 2 ? 2 : conv_mb(prec, cout / 32);
    jobs.push_back(PackJob{wt, f.w[prec], PackGeom{cout, cin, ks, first ? 1 : 0, sb, mbf, 0, cout, cin, wscale}, prec, 0});
    const int mbd = prec == 2 ? 2 : conv_mb(prec, cop / 32);
    jobs.push_back(PackJob{wt, d.w[prec], PackGeom{cop, cout, ks, 0, 16, mbd, 1, cout, cin, dscale}, prec, 0});
  }
  jobs.push_back(PackJob{bias, f.bias, PackGeom{cout, 0, 0, 0, 0, 0, 0, 0, 0, 1.f}, 0, 1});
  jobs.push_back(PackJob{bias, f.bias_x3, PackGeom{cout, 0, 0, 0, 0, 0, 0, 0, 0, wscale * kConvActScale}, 0, 1});
  return DFN_OK;
}

int refresh_core(dfn_dfnet_t h, const float* const* params, int n_params, bool with_adapt, int prec_mask, hipStream_t s, const char* fn) {
  if (!h) return set_error(DFN_ERR_ARG, "%s: null handle", fn);
  if (!h->committed) return set_error(DFN_ERR_STATE, "%s: dfn_dfnet_commit() has not been called", fn);
  const int n_enc = int(h->enc.size());
  const int want = 2 * n_enc + 2 + (with_adapt ? 8 * h->n_taps : 0);
  if (!params || n_params != want) return set_error(DFN_ERR_ARG, "%s: need %d pointers", fn, want);
  if (prec_mask < 1 || prec_mask > 7) return set_error(DFN_ERR_ARG, "%s: prec_mask must select at least one of the three precisions", fn);
  for (int i = 0; i < n_params; ++i)
    if (!params[i]) return set_error(DFN_ERR_ARG, "%s: null pointer %d", fn, i);
  // every tensor's fragments in a few multi-tensor launches (launch_pack_multi) instead of four launches per convolution
  std::vector<PackJob> jobs;
  for (int i = 0; i < n_enc; ++i) {
    const ConvSpec& sp = h->enc[i];
    if (int rc = refresh_conv(h->enc_packed[i], h->enc_dgrad[i], params[2 * i], params[2 * i + 1], sp.cout, sp.cin, 3, i == 0, prec_mask, jobs)) return rc;
  }
  if (with_adapt)
    for (int t = 0; t < h->n_taps; ++t) {
      const float* const* ap = params + 2 * n_enc + 2 + 8 * t;
      if (int rc = refresh_conv(h->ad1[t], h->ad1_dgrad[t], ap[0], ap[1], 64, h->tap_channels[t], 1, false, prec_mask, jobs)) return rc;
      if (int rc = refresh_conv(h->ad5_raw[t], h->ad5_raw_dgrad[t], ap[2], ap[3], 128, 64, 5, false, prec_mask, jobs)) return rc;
    }
  // the plain fp32 copies the kernels read (fc weight | bias, per level gamma, beta, running_mean, running_var) ride in the same launches
  auto copy_job = [&](const float* src, float* dst, int n) { jobs.push_back(PackJob{src, dst, PackGeom{n, 0, 0, 0, 0, 0, 0, 0, 0, 1.f}, 0, 2}); };
  copy_job(params[2 * n_enc], h->fc, h->feat_dim * 512);
  copy_job(params[2 * n_enc + 1], h->fc + size_t(h->feat_dim) * 512, h->feat_dim);
  if (with_adapt)
    for (int t = 0; t < h->n_taps; ++t) {
      const float* const* ap = params + 2 * n_enc + 2 + 8 * t;
      for (int k = 0; k < 4; ++k) copy_job(ap[4 + k], h->bn_dev[t] + 128 * k, 128);
    }
  CHECK_HIP(launch_pack_multi(jobs.data(), int(jobs.size()), s), "refresh: multi-tensor pack");
  h->fresh_mask = prec_mask;   // the weights moved: fragments of the other precisions are stale from here on
  return DFN_OK;
}
}  // namespace

extern "C" int dfn_dfnet_refresh_pose_params_device(dfn_dfnet_t h, const float* const* params, int n_params, int prec_mask, void* stream) {
  return refresh_core(h, params, n_params, false, prec_mask, HS(stream), "dfn_dfnet_refresh_pose_params_device");
}

extern "C" int dfn_dfnet_refresh_train_params_device(dfn_dfnet_t h, const float* const* params, int n_params, int prec_mask, void* stream) {
  return refresh_core(h, params, n_params, true, prec_mask, HS(stream), "dfn_dfnet_refresh_train_params_device");
}

// ------------------------------------------------------------------------------------------ conv weight gradient, stage level
// The weight / bias gradient of ONE convolution from torch-shaped tensors (parity tests against F.conv2d's autograd): the operands
// are split once (g at its measured power-of-two scale, the input at kConvActScale) into the row-planar storage and streamed by
// conv_wgrad_s_kernel (dfnet_wgrad_s.hip) — the kernel the training steps run on their stored activations and gated gradients.
namespace {
struct WgStageWs { char *g, *in, *zeros; float *scl, *part, *part_b; size_t part_floats, part_b_floats, total; };
WgStageWs carve_wg_stage(char* base, int B, int H, int W, int cout, int cin, int ks) {
  WgStageWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  const size_t q = size_t(B) * H * W;
  w.g = take(q * cout * 4);
  w.in = take(q * cin * 4);
  w.zeros = take(256);
  w.scl = reinterpret_cast<float*>(take((1024 + 8) * 4));
  conv_wgrad_split_scratch(ks, B, H, W, cout, cin, &w.part_floats, &w.part_b_floats);
  w.part = reinterpret_cast<float*>(take(w.part_floats * 4));
  w.part_b = reinterpret_cast<float*>(take(w.part_b_floats * 4));
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t dfn_conv_wgrad_scratch_bytes(int B, int H, int W, int cout, int cin, int ks) {
  if (B < 1 || H < 1 || W < 1 || cout < 64 || cin < 64 || cout % 64 || cin % 64 || (ks != 1 && ks != 3 && ks != 5)) return 0;
  return carve_wg_stage(nullptr, B, H, W, cout, cin, ks).total;
}

extern "C" int dfn_conv_wgrad(const float* grad_out, const float* input, int B, int H, int W, int cout, int cin, int ks, float* dW,
                              float* db, void* scratch, size_t scratch_bytes, void* stream) {
  const char* fn = "dfn_conv_wgrad";
  if (!grad_out || !input || !dW || !scratch || B < 1 || H < 1 || W < 1)
    return set_error(DFN_ERR_ARG, "%s: bad argument", fn);
  if (cout < 64 || cin < 64 || cout % 64 || cin % 64 || (ks != 1 && ks != 3 && ks != 5))
    return set_error(DFN_ERR_UNSUPPORTED, "%s: channel counts must be multiples of 64, kernel size 1, 3 or 5", fn);
  const WgStageWs w = carve_wg_stage(static_cast<char*>(scratch), B, H, W, cout, cin, ks);
  if (w.total > scratch_bytes) return set_error(DFN_ERR_ARG, "%s: scratch too small (%zu < %zu)", fn, scratch_bytes, w.total);
  hipStream_t s = HS(stream);
  const size_t q = size_t(B) * H * W;
  CHECK_HIP(hipMemsetAsync(w.zeros, 0, 256, s), "dfn_conv_wgrad: zeros");
  CHECK_HIP(launch_absmax_scale(grad_out, q * cout, w.scl + 8, w.scl, s), "dfn_conv_wgrad: gradient scale");
  CHECK_HIP(launch_nchw_to_split(grad_out, B, cout, H, W, 1.f, w.scl, w.g, s), "dfn_conv_wgrad: split gradient");
  CHECK_HIP(launch_nchw_to_split(input, B, cin, H, W, kConvActScale, nullptr, w.in, s), "dfn_conv_wgrad: split input");
  CHECK_HIP(launch_conv_wgrad_split(ks, w.g, w.in, w.zeros, B, H, W, cout, cin, w.part, w.part_floats, w.part_b, w.part_b_floats, dW, db,
                                    w.scl, s),
            "dfn_conv_wgrad: weight-gradient stream");
  return DFN_OK;
}
