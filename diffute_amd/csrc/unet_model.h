// Model description shared by the inference graph (unet.hip) and the training graph (unet_train.hip): where every
// parameter of the SD2-inpainting UNet lives in the packed weights arena.
#pragma once
#include <map>
#include <memory>
#include <tuple>
#include <vector>
#include "exec.h"
#include "../../include/diffute_hip.h"

struct XfW {
  int C = 0, heads = 0, ctx_slot = -1;
  size_t ng, nb, wpi, bpi, l1g, l1b, l2g, l2b, l3g, l3b;
  size_t wqkv, wo1, bo1, wq2, wkv2, wo2, bo2, wf1, bf1, wf2, bf2, wpo, bpo;
  // folded LayerNorm: raw (as loaded) copies of the three LN-consuming weights + the derived c1 / c2 vectors
  size_t wqkv_raw, wq2_raw, wf1_raw, c1_qkv, c2_qkv, c1_q2, c2_q2, c1_f1, c2_f1;
};
struct ConvW { size_t w, b; int c; size_t wp = 0; /* upsamplers: derived [4][c][4c] phase weights */ };

struct dmx_unet {
  dmx_unet_config cfg;
  ParamTable pt;
  char* arena = nullptr;
  const void* masters_f32 = nullptr;   // fp32 validation forward only (dmx_unet_forward_f32): the caller's fp32 master arena for the duration of the call
  int temb_dim = 0, tproj_total = 0;
  size_t te_w1, te_b1, te_w2, te_b2, tp_w, tp_b, freq;
  size_t ci_w, ci_b; int ci_kpad = 0;
  size_t co_w, co_b, cno_g, cno_b;
  std::vector<ResW> down_res[4], up_res[4]; std::vector<XfW> down_xf[4], up_xf[4];
  ConvW down_ds[4], up_us[4];
  ResW mid_res[2]; XfW mid_xf;
  std::vector<XfW*> xf_all;          // cross-attention layers in graph order (context cache slots)
  std::vector<TrJob> tr_cache;       // the transpose job table dmx_unet_train_prepare uploaded last (kernels.h TrBatch)
  bool finalized = false;
  std::shared_ptr<void> train_state;   // live training pass (unet_train.hip)
  // hipGraph cache: one captured UNet step per distinct argument tuple (pointers are baked into the nodes)
  typedef std::tuple<const void*, const void*, const void*, const void*, const void*, const void*, const void*,
                     int, int, int, int, int, int, int, int, const void*, const void*, int> GraphKey;       // (last: dmx_plan_epoch() - every dmx_set_* switch changes the plans baked into the graph)
  // optional source of the time-embedding projections (dmx_unet_use_temb_table): row *temb_step of a table computed for all
  // timesteps of a denoise loop in one batched pass, instead of four small launches per step
  const float* temb_table = nullptr; const int* temb_step = nullptr;
  struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; };
  std::map<GraphKey, GraphEntry> graphs;
  void drop_graphs() { for (auto& kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec); graphs.clear(); }
  ~dmx_unet() { drop_graphs(); }

  template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
};


static inline int dmx_ctx_pad(int ctx_len) { return (int)align_up((size_t)ctx_len, 64); }
