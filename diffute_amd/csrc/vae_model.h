// Model description shared by the inference graphs (vae.hip) and the training graph (vae_train.hip) of the AutoencoderKL.
#pragma once
#include <memory>
#include <vector>
#include "exec.h"
#include "../../include/diffute_hip.h"

struct AttnW { int C = 0; size_t gg, gb, wq, bq, wk, bk, wv, bv, wo, bo; };
struct CW { size_t w = 0, b = 0; int cin = 0, cout = 0, kpad = 0; size_t wp = 0; /* decoder upsamplers: derived [4][cout][4*cin] phase weights */ };

struct dmx_vae {
  dmx_vae_config cfg;
  ParamTable pt;
  char* arena = nullptr;
  bool finalized = false;
  // encoder
  CW e_in, e_out, quant; std::vector<ResW> e_res[4]; CW e_ds[4]; ResW e_mid[2]; AttnW e_attn; size_t e_ng, e_nb;
  // decoder
  CW pquant, d_in, d_out; ResW d_mid[2]; AttnW d_attn; std::vector<ResW> d_res[4]; CW d_us[4]; size_t d_ng, d_nb;
  std::shared_ptr<void> train_state;   // live training pass (vae_train.hip)
  template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
  // fp32 VALIDATION mode (dmx_vae_encode_f32 / dmx_vae_decode_f32, tests only): for the duration of the call the parameters
  // come from the caller's fp32 master arena, where weights-arena byte o lives at byte 2*o (as in the UNet's)
  const char* masters_f32 = nullptr;
  template <typename T> const T* W(size_t off) const { return masters_f32 ? (const T*)(masters_f32 + 2 * off) : (const T*)(arena + off); }
  const char* wbase() const { return masters_f32 ? masters_f32 : arena; }
  int wmul() const { return masters_f32 ? 2 : 1; }
};

