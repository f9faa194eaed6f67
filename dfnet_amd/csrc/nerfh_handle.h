// nerfh_handle.h — the NeRF-H handle behind dfn_nerfh_t, shared by nerfh_api.hip (test-time render path) and
// nerfh_train_api.hip (training path, generic-width path).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/dfnet_hip.h"
#include "nerfh_kernels.h"
#include "nerfh_layout.h"

struct PackedNet {
  char* blob = nullptr;
  uint32_t* tab = nullptr;
  int n_units = 0;
  int n_fwd_units = 0;   // gradient nets: the forward units come first in the table
  float in_scale = 1.f;  // split-f16: weight scale x activation scale carried by the accumulators
};

struct dfn_nerfh_s {
  dfn_nerfh_desc desc;
  std::map<std::string, std::vector<float>> params;
  bool committed = false;
  PackedNet net[2][3][dfn::kVariants];  // [coarse/fine][prec][kernel variant]
  PackedNet bwd[3];                // [prec] fine forward units + backward (W^T) units of the gradient kernel
                                   // (prec 2: split-f16 forward and backward units)
  float* extra = nullptr;  // w_dir^T | b_dir | w_tr^T | b_tr | emb_a | emb_t
  dfn::RayBiasWeights rb{};
  bool fast = false;       // the register-resident MFMA kernels are packed (netwidth == dfn::kWidth)
  // Generic-width path (nerfh_train_api.hip): every parameter as a plain row-major fp32 device tensor, in the canonical
  // order of dfn_nerfh_train_param_name(); one allocation.
  float* gen_blob = nullptr;
  std::vector<const float*> gen_params;
  // Training step: the coarse network's backward runs beside the fine one's on this stream (created on first use), fenced by the
  // two events (nerfh_train_api.hip: dfn_nerfh_train_backward).
  int render_flags = 0;    // DFN_RENDER_* options of every render entry point (dfn_nerfh_set_render_options)
  int* range_flag = nullptr;   // device int the MLP kernels OR their range-guard bits into (dfn_nerfh_range_status)
  void* fused = nullptr;       // dfn::fused::State: staging-unit / destination tables of the fused training step (nerfh_fused_api.hip)
  bool train_forward_exact = false;   // which implementation the last dfn_nerfh_train_forward ran (dfn_nerfh_train_backward_rays needs the exact one)
  bool train_split_fine = false;   // DFN_TRAIN_FUSED_SPLIT: the fused step stores the fine network's operands as hi | lo planes too (nerfh_fused_train.h)
  bool train_forward_split = false; // ... as the last dfn_nerfh_train_forward laid the workspace out
  bool train_exact = false;    // dfn_nerfh_set_train_mode: run the training step on the layer-by-layer exact-fp32 products even at netwidth 128
  hipStream_t side_stream = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
};

