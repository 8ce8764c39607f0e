// UNet2DConditionModel executor (SD2-inpainting layout): the graph behind
// `unet(sample, timestep, encoder_hidden_states).sample` (reference call sites
// app.ipynb:814, train_diffute_v1.py:913; module structure SURVEY.md Appendix A.1).
//
// The graph is walked on the host and every op is one of the hand-written gfx950 kernels.
// Fusions relative to the eager diffusers graph:
//   - torch.cat([latents, mask, masked_latents]) + NCHW->NHWC + fp32->bf16 + conv_in im2col: one kernel
//   - skip-connection torch.cat: never materialised (GroupNorm and the shortcut read two sources)
//   - conv1 + bias + time-embedding broadcast add: one launch
//   - conv2 + 1x1 conv_shortcut + residual add: one launch (shortcut appended as extra K)
//   - nearest x2 upsample: folded into the following conv's gather
//   - to_q|to_k|to_v one GEMM (N = 3C); the attention kernel reads V row-major through LDS transpose reads
//   - GEGLU: Linear(C,8C) + a*gelu(b) in one launch; every Linear bias / residual in the GEMM epilogue
//   - all 22 time_emb_proj Linear layers: one GEMV launch
//   - cross-attention K / V^T of the glyph context computed once per image (set_context)
#include <stdlib.h>
#include <math.h>
#include "unet_model.h"

namespace {

void build_resnet(dmx_unet* u, ResW& r, const std::string& p, int cin, int cout, bool temb) {
  resnet_build(u->pt, r, p, cin, cout);
  if (temb) { r.temb_off = u->tproj_total; u->tproj_total += cout; }   // rows of the batched time_emb_proj matrix
}

void build_xf(dmx_unet* u, XfW& x, const std::string& p, int C, int heads) {
  ParamTable& pt = u->pt;
  const int ctx = u->cfg.cross_attention_dim;
  x.C = C; x.heads = heads;
  x.ng = pt.f32(p + "norm.weight", C); x.nb = pt.f32(p + "norm.bias", C);
  x.wpi = pt.linear(p + "proj_in.weight", C, C); x.bpi = pt.f32(p + "proj_in.bias", C);
  const std::string t = p + "transformer_blocks.0.";
  x.l1g = pt.f32(t + "norm1.weight", C); x.l1b = pt.f32(t + "norm1.bias", C);
  x.wqkv_raw = pt.reserve((size_t)3 * C * C * 2);                  // to_q | to_k | to_v stacked: one GEMM, N = 3C
  pt.linear_at(t + "attn1.to_q.weight", C, C, x.wqkv_raw, C);
  pt.linear_at(t + "attn1.to_k.weight", C, C, x.wqkv_raw + (size_t)C * C * 2, C);
  pt.linear_at(t + "attn1.to_v.weight", C, C, x.wqkv_raw + (size_t)2 * C * C * 2, C);
  x.wqkv = pt.reserve((size_t)3 * C * C * 2);                      // norm1's gamma folded in (finalize)
  x.c1_qkv = pt.reserve((size_t)3 * C * 4); x.c2_qkv = pt.reserve((size_t)3 * C * 4);
  x.wo1 = pt.linear(t + "attn1.to_out.0.weight", C, C); x.bo1 = pt.f32(t + "attn1.to_out.0.bias", C);
  x.l2g = pt.f32(t + "norm2.weight", C); x.l2b = pt.f32(t + "norm2.bias", C);
  x.wq2_raw = pt.linear(t + "attn2.to_q.weight", C, C);
  x.wq2 = pt.reserve((size_t)C * C * 2); x.c1_q2 = pt.reserve((size_t)C * 4); x.c2_q2 = pt.reserve((size_t)C * 4);
  x.wkv2 = pt.reserve((size_t)2 * C * ctx * 2);                    // to_k | to_v stacked (context projections)
  pt.linear_at(t + "attn2.to_k.weight", C, ctx, x.wkv2, ctx);
  pt.linear_at(t + "attn2.to_v.weight", C, ctx, x.wkv2 + (size_t)C * ctx * 2, ctx);
  x.wo2 = pt.linear(t + "attn2.to_out.0.weight", C, C); x.bo2 = pt.f32(t + "attn2.to_out.0.bias", C);
  x.l3g = pt.f32(t + "norm3.weight", C); x.l3b = pt.f32(t + "norm3.bias", C);
  { PackRule r; r.kind = PackRule::GEGLU_W; r.dst = pt.reserve((size_t)8 * C * C * 2); r.rows = 8 * C; r.cols = C; r.ld = C;
    pt.add(t + "ff.net.0.proj.weight", {8 * C, C}, r); x.wf1_raw = r.dst; }
  x.wf1 = pt.reserve((size_t)8 * C * C * 2); x.c1_f1 = pt.reserve((size_t)8 * C * 4); x.c2_f1 = pt.reserve((size_t)8 * C * 4);
  { PackRule r; r.kind = PackRule::GEGLU_B; r.dst = pt.reserve((size_t)8 * C * 4); r.rows = 8 * C;
    pt.add(t + "ff.net.0.proj.bias", {8 * C}, r); x.bf1 = r.dst; }
  x.wf2 = pt.linear(t + "ff.net.2.weight", C, 4 * C); x.bf2 = pt.f32(t + "ff.net.2.bias", C);
  x.wpo = pt.linear(p + "proj_out.weight", C, C); x.bpo = pt.f32(p + "proj_out.bias", C);
}

}  // namespace

extern "C" dmx_unet* dmx_unet_create(const dmx_unet_config* cfg) {
  if (!cfg) { dmx_set_error("unet_create: null config"); return nullptr; }
  for (int i = 0; i < 4; ++i) {
    const int c = cfg->block_out_channels[i];
    if (c % 64 != 0 || c % cfg->norm_num_groups != 0 || (cfg->heads[i] > 0 && c / cfg->heads[i] != 64)) {
      dmx_set_error("unet_create: block_out_channels[%d]=%d must be a multiple of 64 with head dim 64", i, c);
      return nullptr;
    }
  }
  if (cfg->cross_attention_dim % 64 != 0) { dmx_set_error("unet_create: cross_attention_dim must be a multiple of 64"); return nullptr; }
  auto u = std::make_unique<dmx_unet>();
  u->cfg = *cfg;
  ParamTable& pt = u->pt;
  const int* boc = cfg->block_out_channels; const int L = cfg->layers_per_block;
  const int temb = boc[0] * 4; u->temb_dim = temb;
  u->te_w1 = pt.linear("time_embedding.linear_1.weight", temb, boc[0]); u->te_b1 = pt.f32("time_embedding.linear_1.bias", temb);
  u->te_w2 = pt.linear("time_embedding.linear_2.weight", temb, temb); u->te_b2 = pt.f32("time_embedding.linear_2.bias", temb);
  u->freq = pt.reserve((size_t)(boc[0] / 2) * 4);
  u->ci_kpad = (int)align_up((size_t)9 * cfg->in_channels, 64);
  u->ci_w = pt.reserve((size_t)boc[0] * u->ci_kpad * 2);
  pt.conv_at("conv_in.weight", boc[0], cfg->in_channels, 3, u->ci_w, u->ci_kpad, 0);
  u->ci_b = pt.f32("conv_in.bias", boc[0]);
  std::vector<int> skips; skips.push_back(boc[0]);
  int cprev = boc[0];
  for (int i = 0; i < 4; ++i) {
    const int c = boc[i];
    u->down_res[i].resize(L); if (cfg->down_has_attn[i]) u->down_xf[i].resize(L);
    for (int j = 0; j < L; ++j) {
      const std::string p = "down_blocks." + std::to_string(i);
      build_resnet(u.get(), u->down_res[i][j], p + ".resnets." + std::to_string(j) + ".", cprev, c, true);
      if (cfg->down_has_attn[i]) build_xf(u.get(), u->down_xf[i][j], p + ".attentions." + std::to_string(j) + ".", c, cfg->heads[i]);
      cprev = c; skips.push_back(c);
    }
    if (i < 3) {
      const std::string p = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.";
      u->down_ds[i].c = c; u->down_ds[i].w = pt.reserve((size_t)c * 9 * c * 2);
      pt.conv_at(p + "weight", c, c, 3, u->down_ds[i].w, 9 * c, 0);
      u->down_ds[i].b = pt.f32(p + "bias", c);
      skips.push_back(c);
    }
  }
  build_resnet(u.get(), u->mid_res[0], "mid_block.resnets.0.", cprev, cprev, true);
  build_xf(u.get(), u->mid_xf, "mid_block.attentions.0.", cprev, cfg->heads[3]);
  build_resnet(u.get(), u->mid_res[1], "mid_block.resnets.1.", cprev, cprev, true);
  for (int i = 0; i < 4; ++i) {
    const int c = boc[3 - i];
    u->up_res[i].resize(L + 1); if (cfg->up_has_attn[i]) u->up_xf[i].resize(L + 1);
    for (int j = 0; j < L + 1; ++j) {
      const int cs = skips.back(); skips.pop_back();
      const std::string p = "up_blocks." + std::to_string(i);
      build_resnet(u.get(), u->up_res[i][j], p + ".resnets." + std::to_string(j) + ".", cprev + cs, c, true);
      if (cfg->up_has_attn[i]) build_xf(u.get(), u->up_xf[i][j], p + ".attentions." + std::to_string(j) + ".", c, cfg->heads[3 - i]);
      cprev = c;
    }
    if (i < 3) {
      const std::string p = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
      u->up_us[i].c = c; u->up_us[i].w = pt.reserve((size_t)c * 9 * c * 2);
      pt.conv_at(p + "weight", c, c, 3, u->up_us[i].w, 9 * c, 0);
      u->up_us[i].b = pt.f32(p + "bias", c);
      u->up_us[i].wp = pt.reserve((size_t)4 * c * 4 * c * 2);      // derived: phase weights of the upsample conv (refresh_derived)
    }
  }
  u->cno_g = pt.f32("conv_norm_out.weight", boc[0]); u->cno_b = pt.f32("conv_norm_out.bias", boc[0]);
  u->co_w = pt.reserve((size_t)cfg->out_channels * 9 * boc[0] * 2);
  pt.conv_at("conv_out.weight", cfg->out_channels, boc[0], 3, u->co_w, 9 * boc[0], 0);
  u->co_b = pt.f32("conv_out.bias", cfg->out_channels);

  // time_emb_proj of every resnet: rows of one [tproj_total][temb] matrix (+ bias vector)
  u->tp_w = pt.reserve((size_t)u->tproj_total * temb * 2);
  u->tp_b = pt.reserve((size_t)u->tproj_total * 4);
  auto reg_tp = [&](const ResW& r, const std::string& p) {
    pt.linear_at(p + "time_emb_proj.weight", r.cout, temb, u->tp_w + (size_t)r.temb_off * temb * 2, temb);
    pt.f32_at(p + "time_emb_proj.bias", r.cout, u->tp_b + (size_t)r.temb_off * 4);
  };
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < L; ++j) reg_tp(u->down_res[i][j], "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".");
  reg_tp(u->mid_res[0], "mid_block.resnets.0."); reg_tp(u->mid_res[1], "mid_block.resnets.1.");
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < L + 1; ++j) reg_tp(u->up_res[i][j], "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".");

  // cross-attention context slots in graph order
  for (int i = 0; i < 4; ++i) for (auto& x : u->down_xf[i]) u->xf_all.push_back(&x);
  u->xf_all.push_back(&u->mid_xf);
  for (int i = 0; i < 4; ++i) for (auto& x : u->up_xf[i]) u->xf_all.push_back(&x);
  for (size_t s = 0; s < u->xf_all.size(); ++s) u->xf_all[s]->ctx_slot = (int)s;
  return u.release();
}

extern "C" void dmx_unet_destroy(dmx_unet* u) { delete u; }
extern "C" int dmx_unet_param_count(const dmx_unet* u) { return u ? (int)u->pt.entries().size() : 0; }
extern "C" int dmx_unet_param_info(const dmx_unet* u, int index, const char** name, int shape[4]) {
  DMX_REQUIRE(u && index >= 0 && index < (int)u->pt.entries().size(), "unet_param_info: bad index %d", index);
  const ParamEntry& e = u->pt.entries()[index];
  if (name) *name = e.name.c_str();
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = e.shape[k];
  return DMX_OK;
}
extern "C" size_t dmx_unet_arena_bytes(const dmx_unet* u) { return u ? u->pt.total() : 0; }
extern "C" int dmx_unet_bind_arena(dmx_unet* u, void* arena, size_t bytes) {
  DMX_REQUIRE(u && arena && bytes >= u->pt.total(), "unet_bind_arena: need %zu bytes", u ? u->pt.total() : (size_t)0);
  u->arena = (char*)arena; u->finalized = false; u->drop_graphs();
  DMX_HIP(hipMemset(arena, 0, u->pt.total()));      // zero the K padding of conv_in
  return DMX_OK;
}
extern "C" int dmx_unet_load_param(dmx_unet* u, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(u != nullptr, "unet_load_param: null handle");
  u->finalized = false;
  return u->pt.load(u->arena, name, src, (hipStream_t)stream);
}

// Recompute everything derived from the raw weights in the arena (folded shortcut biases, LayerNorm-folded GEMM
// weights and their c1 / c2 vectors) after the raw weights changed in place (fused optimizer); asynchronous.
static int refresh_derived(dmx_unet* u, hipStream_t s) {
  auto fuse = [&](const ResW& r) -> int { return resnet_finalize(r, u->arena, s); };
  int rc = 0;
  for (int i = 0; i < 4 && !rc; ++i) { for (auto& r : u->down_res[i]) if (!rc) rc = fuse(r); for (auto& r : u->up_res[i]) if (!rc) rc = fuse(r); }
  if (!rc) rc = fuse(u->mid_res[0]); if (!rc) rc = fuse(u->mid_res[1]);
  for (int i = 0; i < 3 && !rc; ++i)
    rc = dmx_ups_phase_weights_launch(u->at<bf16>(u->up_us[i].w), 9 * u->up_us[i].c, u->at<bf16>(u->up_us[i].wp), u->up_us[i].c, u->up_us[i].c, s);
  // fold norm1/2/3 of every BasicTransformerBlock into the GEMM that consumes it (W' = W*gamma, c1, c2)
  for (const XfW* x : u->xf_all) {
    const int C = x->C;
    if (!rc) rc = dmx_ln_fold_launch(u->at<bf16>(x->wqkv_raw), u->at<bf16>(x->wqkv), u->at<float>(x->l1g), u->at<float>(x->l1b), nullptr,
                                     u->at<float>(x->c1_qkv), u->at<float>(x->c2_qkv), 3 * C, C, s);
    if (!rc) rc = dmx_ln_fold_launch(u->at<bf16>(x->wq2_raw), u->at<bf16>(x->wq2), u->at<float>(x->l2g), u->at<float>(x->l2b), nullptr,
                                     u->at<float>(x->c1_q2), u->at<float>(x->c2_q2), C, C, s);
    if (!rc) rc = dmx_ln_fold_launch(u->at<bf16>(x->wf1_raw), u->at<bf16>(x->wf1), u->at<float>(x->l3g), u->at<float>(x->l3b), u->at<float>(x->bf1),
                                     u->at<float>(x->c1_f1), u->at<float>(x->c2_f1), 8 * C, C, s);
  }
  return rc;
}
extern "C" int dmx_unet_refresh_derived(dmx_unet* u, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->arena && u->finalized, "unet_refresh_derived: weights not finalized");
  u->drop_graphs();
  return refresh_derived(u, (hipStream_t)stream);
}

extern "C" int dmx_unet_finalize(dmx_unet* u, const float* h_freq, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->arena, "unet_finalize: arena not bound");
  DMX_REQUIRE(h_freq != nullptr, "unet_finalize: null frequency table");
  hipStream_t s = (hipStream_t)stream;
  DMX_HIP(hipMemcpyAsync(u->arena + u->freq, h_freq, (size_t)(u->cfg.block_out_channels[0] / 2) * 4, hipMemcpyHostToDevice, s));
  int rc = refresh_derived(u, s);
  DMX_HIP(hipStreamSynchronize(s));
  const bf16* zp = nullptr;
  if (!rc) rc = dmx_zero_page(&zp);                  // allocate the padding page now, never inside a stream capture
  u->finalized = (rc == 0);
  u->drop_graphs();
  return rc;
}

// ----------------------------------------------------------------------------- context
static int ctx_pad(int ctx_len) { return dmx_ctx_pad(ctx_len); }

extern "C" size_t dmx_unet_context_bytes(const dmx_unet* u, int B, int ctx_len) {
  if (!u) return 0;
  size_t tot = 0;
  const int sp = ctx_pad(ctx_len);
  for (const XfW* x : u->xf_all) tot += align_up((size_t)B * sp * 2 * x->C * 2, 256);
  return tot;
}
// context cache slot of one cross-attention layer: [B*sp][2C] bf16, columns [0,C) = K, [C,2C) = V
static const bf16* ctx_slot_ptr(const dmx_unet* u, const void* cache, int B, int ctx_len, int slot) {
  size_t off = 0; const int sp = ctx_pad(ctx_len);
  for (int s = 0; s < slot; ++s) off += align_up((size_t)B * sp * 2 * u->xf_all[s]->C * 2, 256);
  return (const bf16*)((const char*)cache + off);
}

extern "C" int dmx_unet_set_context(dmx_unet* u, const void* ctx, int ctx_is_bf16, int B, int ctx_len,
                                    void* cache, size_t cache_bytes, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_set_context: weights not finalized");
  DMX_REQUIRE(ctx && cache && cache_bytes >= dmx_unet_context_bytes(u, B, ctx_len), "unet_set_context: context cache too small");
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  const int D = u->cfg.cross_attention_dim, sp = ctx_pad(ctx_len);
  bf16* cp = (bf16*)ex.raw((size_t)B * sp * D * 2);
  if (ex.rc) return ex.rc;
  int rc = dmx_cast_pad_rows_launch(ctx, ctx_is_bf16, cp, B, ctx_len, sp, D, ex.stream);
  if (rc) return rc;
  for (const XfW* x : u->xf_all) {
    const bf16* kv = ctx_slot_ptr(u, cache, B, ctx_len, x->ctx_slot);
    // [K | V][b*sp + s][2C] = ctx [W_k ; W_v]^T   (padded context rows are zero -> finite K/V rows)
    ex.gemm_raw(cp, D, B * sp, u->at<bf16>(x->wkv2), D, 2 * x->C, D, nullptr, (void*)kv, 2 * x->C, 0);
    if (ex.rc) return ex.rc;
  }
  return ex.rc;
}

// ----------------------------------------------------------------------------- forward
namespace {

struct Fwd {
  dmx_unet* u; Exec& ex; int B; const float* tproj; int tp_ld; const void* cache; int ctx_len;
  const char* wbase; int wmul;      // weights: the packed bf16 arena (x1) or, in fp32 validation mode, the fp32 master arena (byte offsets x2)
  template <typename T> const T* W(size_t off) const { return (const T*)(wbase + off * (size_t)wmul); }

  Tn resnet(const ResW& r, const Tn& x0, const Tn* x1) {
    return resnet_run(ex, wbase, r, x0, x1, u->cfg.norm_num_groups, 1e-5f, tproj, tp_ld, wmul);
  }

  // fp32 validation mode: the same block with explicit LayerNorms on the RAW weights (the folded copies are derived data of
  // the bf16 path) and the context K / V projected in place (`cache` is the fp32 context [B*ctx_len][cross_attention_dim])
  Tn xformer_f32(const XfW& w, const Tn& x) {
    const int G = u->cfg.norm_num_groups, C = w.C, S = x.H * x.W, D = u->cfg.cross_attention_dim;
    auto F = [&](size_t off) { return W<float>(off); };
    auto H = [&](size_t off) { return W<bf16>(off); };
    Tn t = ex.groupnorm(x, nullptr, F(w.ng), F(w.nb), G, 1e-6f, false);
    Tn h = ex.linear(t, H(w.wpi), C, F(w.bpi), nullptr, false); ex.drop(t);
    Tn n = ex.layernorm(h, F(w.l1g), F(w.l1b), 1e-5f);
    Tn qkv = ex.linear(n, H(w.wqkv_raw), 3 * C, nullptr, nullptr, false); ex.drop(n);
    Tn a = ex.make(x.B, x.H, x.W, C);
    ex.attention(qkv.p, 3 * C, ex.col(qkv, C), 3 * C, ex.col(qkv, 2 * C), 3 * C, S, a.p, C, x.B, w.heads, S, S, 0.125f);
    ex.drop(qkv);
    Tn h2 = ex.linear(a, H(w.wo1), C, F(w.bo1), &h, false); ex.drop(a); ex.drop(h);
    n = ex.layernorm(h2, F(w.l2g), F(w.l2b), 1e-5f);
    Tn q = ex.linear(n, H(w.wq2_raw), C, nullptr, nullptr, false); ex.drop(n);
    Tn cx; cx.p = (bf16*)cache; cx.B = x.B; cx.H = 1; cx.W = ctx_len; cx.C = D; cx.ld = D;
    Tn kv = ex.linear(cx, H(w.wkv2), 2 * C, nullptr, nullptr, false);
    a = ex.make(x.B, x.H, x.W, C);
    ex.attention(q.p, C, kv.p, 2 * C, ex.col(kv, C), 2 * C, ctx_len, a.p, C, x.B, w.heads, S, ctx_len, 0.125f);
    ex.drop(q); ex.drop(kv);
    Tn h3 = ex.linear(a, H(w.wo2), C, F(w.bo2), &h2, false); ex.drop(a); ex.drop(h2);
    n = ex.layernorm(h3, F(w.l3g), F(w.l3b), 1e-5f);
    Tn g = ex.linear(n, H(w.wf1_raw), 8 * C, F(w.bf1), nullptr, true); ex.drop(n);
    Tn h4 = ex.linear(g, H(w.wf2), C, F(w.bf2), &h3, false); ex.drop(g); ex.drop(h3);
    Tn y = ex.linear(h4, H(w.wpo), C, F(w.bpo), &x, false); ex.drop(h4);
    return y;
  }

  Tn xformer(const XfW& w, const Tn& x) {
    if (ex.f32) return xformer_f32(w, x);
    const int G = u->cfg.norm_num_groups, C = w.C, S = x.H * x.W;
    const bool chain = ex.chain_ok(x);     // C = 320 levels: the per-row GEMM chains around the two attention cores are three kernels (xf_chain.hip)
    // the entry GroupNorm rides in the first chain's operand load when x came with its statistics records (xf_chain.hip mode 2)
    const bool gn_fold = chain && ex.chain_gn_fold(x);
    if (gn_fold) ex.flush(x);                // (the chains read x through raw pointers; the GroupNorm below completes a pending x itself)
    Tn t = gn_fold ? x : ex.groupnorm(x, nullptr, u->at<float>(w.ng), u->at<float>(w.nb), G, 1e-6f, false);
    // LayerNorms are folded: each residual-stream producer also emits per-row (sum, sumsq) partials and the
    // consuming GEMM multiplies the raw rows by W*gamma and normalises in its epilogue - no LN kernels, no LN tensors.
    Exec::RowStats st1, st2, st3;
    Tn h, qkv;
    if (chain) {
      // [proj_in -> LN1 -> to_q | to_k | to_v]
      h = ex.make(x.B, x.H, x.W, C); qkv = ex.make(x.B, x.H, x.W, 3 * C);
      XfChainArgs c{};
      c.M = x.rows(); c.C = C; c.eps = 1e-5f;
      c.x = t.p; c.ldx = t.ld; c.w0 = u->at<bf16>(w.wpi); c.b0 = u->at<float>(w.bpi); c.h_out = h.p; c.ldh = h.ld;
      c.w1 = u->at<bf16>(w.wqkv); c.c1 = u->at<float>(w.c1_qkv); c.c2 = u->at<float>(w.c2_qkv); c.y = qkv.p; c.ldy = qkv.ld;
      if (gn_fold) { c.gn_st = x.cst; c.gn_gamma = u->at<float>(w.ng); c.gn_beta = u->at<float>(w.nb); c.gn_groups = G; c.gn_rows = S; c.gn_eps = 1e-6f; }
      ex.xf_chain(2, c);
      if (!gn_fold) ex.drop(t);
    } else {
      h = ex.linear(t, u->at<bf16>(w.wpi), C, u->at<float>(w.bpi), nullptr, false, &st1);
      ex.drop(t);
      // ---- self attention
      Exec::LnIn ln1; ln1.stats = st1.buf; ln1.tiles = st1.tiles; ln1.c1 = u->at<float>(w.c1_qkv); ln1.c2 = u->at<float>(w.c2_qkv);
      qkv = ex.linear(h, u->at<bf16>(w.wqkv), 3 * C, nullptr, nullptr, false, nullptr, &ln1);
      ex.drop(st1.buf);
    }
    Tn a = ex.make(x.B, x.H, x.W, C);
    ex.attention(qkv.p, 3 * C, qkv.p + C, 3 * C, qkv.p + 2 * C, 3 * C, S, a.p, C, x.B, w.heads, S, S, 0.125f);
    ex.drop(qkv);
    const int sp = ctx_pad(ctx_len);
    const bf16* kvc = ctx_slot_ptr(u, cache, x.B, ctx_len, w.ctx_slot);
    if (chain) {
      // [to_out + res -> LN2 -> to_q] and [to_out + res -> LN3 -> FF1 / GEGLU -> FF2 + res -> proj_out + res] around the cross-attention
      XfChainArgs c{};
      c.M = x.rows(); c.C = C; c.eps = 1e-5f;
      Tn h2 = ex.make(x.B, x.H, x.W, C), q = ex.make(x.B, x.H, x.W, C);
      c.x = a.p; c.ldx = a.ld; c.res = h.p; c.ldres = h.ld; c.w0 = u->at<bf16>(w.wo1); c.b0 = u->at<float>(w.bo1);
      c.h_out = h2.p; c.ldh = h2.ld; c.w1 = u->at<bf16>(w.wq2); c.c1 = u->at<float>(w.c1_q2); c.c2 = u->at<float>(w.c2_q2);
      c.y = q.p; c.ldy = q.ld;
      ex.xf_chain(0, c);
      ex.drop(a); ex.drop(h);
      a = ex.make(x.B, x.H, x.W, C);
      ex.attention(q.p, C, kvc, 2 * C, kvc + C, 2 * C, sp, a.p, C, x.B, w.heads, S, ctx_len, 0.125f, true);
      ex.drop(q);
      Tn h3 = ex.make(x.B, x.H, x.W, C), y = ex.make(x.B, x.H, x.W, C);
      XfChainArgs d{};
      d.M = x.rows(); d.C = C; d.eps = 1e-5f;
      d.x = a.p; d.ldx = a.ld; d.res = h2.p; d.ldres = h2.ld; d.w0 = u->at<bf16>(w.wo2); d.b0 = u->at<float>(w.bo2);
      d.h_out = h3.p; d.ldh = h3.ld; d.c1 = u->at<float>(w.c1_f1); d.c2 = u->at<float>(w.c2_f1);
      d.wf1 = u->at<bf16>(w.wf1); d.wf2 = u->at<bf16>(w.wf2); d.bf2 = u->at<float>(w.bf2);
      d.wpo = u->at<bf16>(w.wpo); d.bpo = u->at<float>(w.bpo); d.xres = x.p; d.ldxres = x.ld;
      d.y = y.p; d.ldy = y.ld;
      ex.chain_stats(d, y);                            // (statistics records of y for the GroupNorm -> conv launch that reads it next)
      ex.xf_chain(1, d);
      ex.drop(a); ex.drop(h2); ex.drop(h3);
      return y;
    }
    Tn h2 = ex.linear(a, u->at<bf16>(w.wo1), C, u->at<float>(w.bo1), &h, false, &st2);
    ex.drop(a); ex.drop(h);
    // ---- cross attention over the cached glyph-context K / V
    Exec::LnIn ln2; ln2.stats = st2.buf; ln2.tiles = st2.tiles; ln2.c1 = u->at<float>(w.c1_q2); ln2.c2 = u->at<float>(w.c2_q2);
    Tn q = ex.linear(h2, u->at<bf16>(w.wq2), C, nullptr, nullptr, false, nullptr, &ln2);
    ex.drop(st2.buf);
    a = ex.make(x.B, x.H, x.W, C);
    ex.attention(q.p, C, kvc, 2 * C, kvc + C, 2 * C, sp, a.p, C, x.B, w.heads, S, ctx_len, 0.125f, true);
    ex.drop(q);
    Tn h3 = ex.linear(a, u->at<bf16>(w.wo2), C, u->at<float>(w.bo2), &h2, false, &st3);
    ex.drop(a); ex.drop(h2);
    // ---- GEGLU feed-forward
    Exec::LnIn ln3; ln3.stats = st3.buf; ln3.tiles = st3.tiles; ln3.c1 = u->at<float>(w.c1_f1); ln3.c2 = u->at<float>(w.c2_f1);
    Tn g = ex.linear(h3, u->at<bf16>(w.wf1), 8 * C, nullptr, nullptr, true, nullptr, &ln3);
    ex.drop(st3.buf);
    Tn h4 = ex.linear(g, u->at<bf16>(w.wf2), C, u->at<float>(w.bf2), &h3, false);
    ex.drop(g); ex.drop(h3);
    Tn y = ex.linear(h4, u->at<bf16>(w.wpo), C, u->at<float>(w.bpo), &x, false, nullptr, nullptr, true);   // (+ GroupNorm statistics for the next block)
    ex.drop(h4);
    return y;
  }
};

// row *step of the precomputed time-embedding projection table -> the buffer every resnet's conv1 reads its row bias from
__global__ __launch_bounds__(256) void dmx_temb_row_kernel(const float* table, const int* step, float* out, int n) {
  const float* src = table + (size_t)(*step) * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = src[i];
}

// the time-embedding MLP + the stacked time_emb_proj of every resnet for `rows` timesteps (product path: fp32 math, bf16 weights;
// each row is computed exactly as the per-step path computes its single row)
static int temb_rows(dmx_unet* u, Exec& ex, const long long* timesteps, int rows, float* tproj) {
  const int c0 = u->cfg.block_out_channels[0], temb = u->temb_dim;
  float* sinus = (float*)ex.raw((size_t)rows * c0 * 4);
  float* e1 = (float*)ex.raw((size_t)rows * temb * 4);
  float* emb = (float*)ex.raw((size_t)rows * temb * 4);
  if (!ex.dry && !ex.rc) {
    ex.rc = dmx_timestep_embedding_launch(timesteps, rows, u->at<float>(u->freq), rows, c0, sinus, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(sinus, c0, u->at<bf16>(u->te_w1), c0, u->at<float>(u->te_b1), e1, temb, rows, temb, c0, 0, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(e1, temb, u->at<bf16>(u->te_w2), temb, u->at<float>(u->te_b2), emb, temb, rows, temb, temb, 1, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(emb, temb, u->at<bf16>(u->tp_w), temb, u->at<float>(u->tp_b), tproj, u->tproj_total, rows, u->tproj_total, temb, 1, ex.stream);
  }
  ex.drop(sinus); ex.drop(e1); ex.drop(emb);
  return ex.rc;
}
extern "C" size_t dmx_unet_temb_table_floats(dmx_unet* u, int T) { return u ? (size_t)T * u->tproj_total : 0; }
extern "C" size_t dmx_unet_temb_table_workspace_bytes(dmx_unet* u, int T) {
  if (!u) return 0;
  Exec ex; ex.dry = true; ex.ws.reset(nullptr, 0, true);
  temb_rows(u, ex, nullptr, T, nullptr);
  return ex.ws.peak() + 4096;
}
extern "C" int dmx_unet_temb_table(dmx_unet* u, const int64_t* timesteps, int T, float* table, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_temb_table: weights not finalized");
  DMX_REQUIRE(timesteps && table && workspace && T > 0, "unet_temb_table: null argument");
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  return temb_rows(u, ex, (const long long*)timesteps, T, table);
}
extern "C" int dmx_unet_use_temb_table(dmx_unet* u, const float* table, const int* step_index) {
  DMX_REQUIRE(u != nullptr, "unet_use_temb_table: null handle");
  DMX_REQUIRE((table == nullptr) == (step_index == nullptr), "unet_use_temb_table: table and step index go together");
  u->temb_table = table; u->temb_step = step_index;
  return DMX_OK;
}

int unet_run(dmx_unet* u, Exec& ex, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
             const long long* timesteps, int t_count, const void* cache, int ctx_len, float* out, int B, int H, int W) {
  const dmx_unet_config& cfg = u->cfg;
  const int* boc = cfg.block_out_channels; const int L = cfg.layers_per_block; const int temb = u->temb_dim;
  // ---- time embedding (fp32, bf16 weights).  A scalar timestep (the denoise loop) is embedded once and every image reads
  // row 0 of the projections (row stride 0); per-sample timesteps (training) get one row each.
  const int Bt = (t_count == 1) ? 1 : B;
  const int tp_ld = (t_count == 1) ? 0 : u->tproj_total;
  float* sinus = (float*)ex.raw((size_t)B * boc[0] * 4);              // sized for B rows either way (workspace query)
  float* e1 = (float*)ex.raw((size_t)B * temb * 4);
  float* emb = (float*)ex.raw((size_t)B * temb * 4);
  float* tproj = (float*)ex.raw((size_t)B * u->tproj_total * 4);
  const char* wbase = ex.f32 ? (const char*)u->masters_f32 : u->arena;
  const int wmul = ex.f32 ? 2 : 1;
  Fwd f{u, ex, B, tproj, tp_ld, cache, ctx_len, wbase, wmul};
  Tn h;
  if (ex.f32) {
    // fp32 validation mode: same layers on the fp32 masters through the generic fp32 GEMM (SiLU as its own tiny pass)
    auto small = [&](float* x, int K, size_t w, size_t b, int N, float* y) {
      Tn xi; xi.p = (bf16*)x; xi.B = Bt; xi.H = xi.W = 1; xi.C = K; xi.ld = K;
      if (ex.dry || ex.rc) return;
      GemmF32Args a{}; a.x0 = a.x1 = x; a.ldx0 = a.ldx1 = K; a.cx0 = a.Cin = K; a.direct = 1; a.ksize = 1; a.stride = 1; a.Ktaps = a.K = K;
      a.w = f.W<float>(w); a.ldw = K; a.M = Bt; a.N = N; a.bias = f.W<float>(b); a.rows_per_group = 1; a.out = y; a.ldo = N;
      ex.rc = dmx_gemm_f32_launch(a, ex.stream);
    };
    if (!ex.dry && !ex.rc) ex.rc = dmx_timestep_embedding_launch(timesteps, t_count, (const float*)(u->arena + u->freq), Bt, boc[0], sinus, ex.stream);
    small(sinus, boc[0], u->te_w1, u->te_b1, temb, e1);
    if (!ex.dry && !ex.rc) ex.rc = dmx_silu_f32_launch(e1, (size_t)Bt * temb, ex.stream);
    small(e1, temb, u->te_w2, u->te_b2, temb, emb);
    if (!ex.dry && !ex.rc) ex.rc = dmx_silu_f32_launch(emb, (size_t)Bt * temb, ex.stream);
    // the 22 time_emb_proj biases are separate fp32 entries: in the master arena entry k sits at byte 2*dst_k, so (unlike the
    // stacked bf16 weight rows) they are not one contiguous vector there - gather them into one
    float* tpb = (float*)ex.raw((size_t)u->tproj_total * 4);
    if (!ex.dry && !ex.rc) {
      auto put = [&](const ResW& r) {
        if (r.temb_off >= 0 && !ex.rc &&
            hipMemcpyAsync(tpb + r.temb_off, f.W<float>(u->tp_b + (size_t)r.temb_off * 4), (size_t)r.cout * 4, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess) {
          dmx_set_error("unet_forward_f32: bias gather failed"); ex.rc = DMX_ERR_HIP;
        }
      };
      for (int i = 0; i < 4; ++i) { for (auto& r : u->down_res[i]) put(r); for (auto& r : u->up_res[i]) put(r); }
      put(u->mid_res[0]); put(u->mid_res[1]);
    }
    {
      Tn xi; (void)xi;
      if (!ex.dry && !ex.rc) {
        GemmF32Args a{}; a.x0 = a.x1 = emb; a.ldx0 = a.ldx1 = temb; a.cx0 = a.Cin = temb; a.direct = 1; a.ksize = 1; a.stride = 1; a.Ktaps = a.K = temb;
        a.w = f.W<float>(u->tp_w); a.ldw = temb; a.M = Bt; a.N = u->tproj_total; a.bias = tpb; a.rows_per_group = 1; a.out = tproj; a.ldo = u->tproj_total;
        ex.rc = dmx_gemm_f32_launch(a, ex.stream);
      }
    }
    ex.drop(tpb);
    ex.drop(sinus); ex.drop(e1); ex.drop(emb);
    Tn x9 = ex.make(B, H, W, cfg.in_channels);
    if (!ex.dry && !ex.rc) ex.rc = dmx_concat_nchw_to_nhwc_f32_launch(f0, c0, f1, c1, f2, c2, (float*)x9.p, B, H * W, ex.stream);
    ConvOpts oi; oi.bias = f.W<float>(u->ci_b); oi.ldw = u->ci_kpad;
    h = ex.conv(x9, nullptr, f.W<bf16>(u->ci_w), boc[0], oi);
    ex.drop(x9);
  } else {
    if (!ex.dry && !ex.rc) {
      if (u->temb_table && t_count == 1) {
        // the loop computed the projections of all its timesteps in one batched pass (dmx_unet_temb_table): fetch this step's row
        hipLaunchKernelGGL(dmx_temb_row_kernel, dim3(cdiv(u->tproj_total, 1024)), dim3(256), 0, ex.stream, u->temb_table, u->temb_step, tproj, u->tproj_total);
        ex.rc = dmx_check_launch("dmx_temb_row_kernel");
      } else {
        ex.rc = dmx_timestep_embedding_launch(timesteps, t_count, u->at<float>(u->freq), Bt, boc[0], sinus, ex.stream);
        if (!ex.rc) ex.rc = dmx_linear_small_launch(sinus, boc[0], u->at<bf16>(u->te_w1), boc[0], u->at<float>(u->te_b1), e1, temb, Bt, temb, boc[0], 0, ex.stream);
        if (!ex.rc) ex.rc = dmx_linear_small_launch(e1, temb, u->at<bf16>(u->te_w2), temb, u->at<float>(u->te_b2), emb, temb, Bt, temb, temb, 1, ex.stream);
        if (!ex.rc) ex.rc = dmx_linear_small_launch(emb, temb, u->at<bf16>(u->tp_w), temb, u->at<float>(u->tp_b), tproj, u->tproj_total, Bt, u->tproj_total, temb, 1, ex.stream);
      }
    }
    ex.drop(sinus); ex.drop(e1); ex.drop(emb);
    // ---- conv_in: cat + layout + im2col, then GEMM
    Tn col = ex.make(B, H, W, u->ci_kpad);
    if (!ex.dry && !ex.rc) {
      Im2colArgs a{}; a.f0 = f0; a.c0 = c0; a.f1 = f1; a.c1 = c1; a.f2 = f2; a.c2 = c2; a.C = cfg.in_channels;
      a.B = B; a.IH = a.OH = H; a.IW = a.OW = W; a.ksize = 3; a.stride = 1; a.pad = 1; a.out = col.p; a.Kpad = u->ci_kpad;
      ex.rc = dmx_im2col_small_launch(a, ex.stream);
    }
    h = ex.linear(col, u->at<bf16>(u->ci_w), boc[0], u->at<float>(u->ci_b), nullptr, false, nullptr, nullptr, true);
    ex.drop(col);
  }
  ex.tap(h);                                           // "conv_in"
  ex.ensure_stats(h);
  std::vector<Tn> skips; skips.push_back(h);
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < L; ++j) {
#ifdef DMX_PROBES
      static const bool fine = getenv("DMX_TAPS_FINE") != nullptr;      // debugging aid: also tap every resnet / transformer output of the down path
#else
      constexpr bool fine = false;
#endif
      Tn y = f.resnet(u->down_res[i][j], h, nullptr);
      if (fine) ex.tap(y);
      if (cfg.down_has_attn[i]) { Tn z = f.xformer(u->down_xf[i][j], y); ex.drop(y); y = z; if (fine) ex.tap(y); }
      ex.ensure_stats(y);                             // (two GroupNorms read it: the next block's and the up path's concat)
      h = y; skips.push_back(h);                      // previous h stays alive as a skip
    }
    if (i < 3) {
      ConvOpts o; o.stride = 2; o.pad = 1; o.bias = f.W<float>(u->down_ds[i].b); o.stats = 1;
      h = ex.conv(h, nullptr, f.W<bf16>(u->down_ds[i].w), boc[i], o);
      ex.ensure_stats(h);
      skips.push_back(h);
    }
    ex.tap(h);                                         // "down{i}"
  }
  Tn mid_in;
  { Tn y = f.resnet(u->mid_res[0], h, nullptr);          // h is also skips.back(): keep it
    Tn z = f.xformer(u->mid_xf, y); ex.drop(y);
    ex.ensure_stats(z);
    h = f.resnet(u->mid_res[1], z, nullptr); mid_in = z; }
  ex.tap(h);                                           // "mid"
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < L + 1; ++j) {
      Tn s = skips.back(); skips.pop_back();
      Tn y = f.resnet(u->up_res[i][j], h, &s);
      // (z is the residual of mid_res[1]'s conv2: where that conv left its split-K reduce to the GroupNorm that has just run, z had to live until here)
      if (i == 0 && j == 0) ex.drop(mid_in);
      ex.drop(h); ex.drop(s);
      if (cfg.up_has_attn[i]) { Tn z = f.xformer(u->up_xf[i][j], y); ex.drop(y); y = z; }
      ex.ensure_stats(y);
      h = y;
    }
    if (i < 3) {
      // nearest x2 + conv3x3 as four 2x2 phase convolutions on the source grid (pre-summed taps): 4/9 of the multiply-adds
      const bool direct = ex.f32;                                          // (the phase weights are derived data of the bf16 path)
      ConvOpts o; o.ups = 1; o.ups2 = direct ? 0 : 1; o.bias = f.W<float>(u->up_us[i].b); o.stats = 1;
      Tn y = ex.conv(h, nullptr, direct ? f.W<bf16>(u->up_us[i].w) : u->at<bf16>(u->up_us[i].wp), boc[3 - i], o);
      ex.ensure_stats(y);
      ex.drop(h); h = y;
    }
    ex.tap(h);                                         // "up{i}"
  }
  Tn t = ex.groupnorm(h, nullptr, f.W<float>(u->cno_g), f.W<float>(u->cno_b), cfg.norm_num_groups, 1e-5f, true);
  ex.drop(h);
  float* eps_nhwc = (float*)ex.raw((size_t)B * H * W * cfg.out_channels * 4);
  ConvOpts oo; oo.bias = f.W<float>(u->co_b); oo.out_f32 = 1;
  ex.conv(t, nullptr, f.W<bf16>(u->co_w), cfg.out_channels, oo, eps_nhwc);
  ex.drop(t);
  if (!ex.dry && !ex.rc) ex.rc = dmx_nhwc_to_nchw_f32_launch(eps_nhwc, cfg.out_channels, out, B, cfg.out_channels, H * W, ex.stream);
  ex.drop(eps_nhwc); ex.drop(tproj);
  return ex.rc;
}

// the product walk with the weight prefetch plan (Exec::note / peek): a dry walk of the same graph lists the weight ranges in launch
// order, the real walk hands every launch the ranges of the launches that follow it
int unet_run_planned(dmx_unet* u, Exec& ex, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                     const long long* timesteps, int t_count, const void* cache, int ctx_len, float* out, int B, int H, int W) {
  Exec::PfPlan plan;
  {
    Exec dr; dr.dry = true; dr.ws.reset(nullptr, 0, true); dr.plan = &plan; dr.plan_rec = true;
    unet_run(u, dr, nullptr, c0, nullptr, c1, nullptr, c2, nullptr, t_count, cache, ctx_len, nullptr, B, H, W);
  }
  ex.plan = &plan; ex.plan_rec = false; ex.plan_bad = false; ex.plan_i = 0;
  int rc = unet_run(u, ex, f0, c0, f1, c1, f2, c2, timesteps, t_count, cache, ctx_len, out, B, H, W);
  if (!rc && (ex.plan_bad || ex.plan_i != (int)plan.w.size())) { dmx_set_error("unet: the prefetch plan of the dry walk (%d launches) does not match the real walk (%d)", (int)plan.w.size(), ex.plan_i); rc = DMX_ERR_ARG; }
  ex.plan = nullptr;
  return rc;
}

}  // namespace

extern "C" size_t dmx_unet_workspace_bytes(dmx_unet* u, int B, int H, int W, int ctx_len) {
  if (!u) return 0;
  Exec ex; ex.dry = true; ex.ws.reset(nullptr, 0, true);
  unet_run(u, ex, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr, ctx_len, nullptr, B, H, W);
  size_t need = ex.ws.peak();
  // set_context needs the padded context + split-K scratch
  Exec e2; e2.dry = true; e2.ws.reset(nullptr, 0, true);
  const int D = u->cfg.cross_attention_dim, sp = ctx_pad(ctx_len);
  void* cp = e2.raw((size_t)B * sp * D * 2);
  for (const XfW* x : u->xf_all)
    e2.gemm_raw((const bf16*)cp, D, B * sp, nullptr, D, 2 * x->C, D, nullptr, nullptr, 2 * x->C, 0);
  if (e2.ws.peak() > need) need = e2.ws.peak();
  return need + 4096;
}

extern "C" int dmx_unet_forward(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                const int64_t* timesteps, int t_count, const void* cache, int ctx_len,
                                float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_forward: weights not finalized (bind_arena, load_param*, finalize)");
  DMX_REQUIRE(f0 && out && timesteps && cache && workspace, "unet_forward: null argument");
  DMX_REQUIRE(c0 + c1 + c2 == u->cfg.in_channels, "unet_forward: c0+c1+c2=%d != in_channels=%d", c0 + c1 + c2, u->cfg.in_channels);
  DMX_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "unet_forward: H=%d W=%d must be positive multiples of 8", H, W);
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  return unet_run_planned(u, ex, f0, c0, f1, c1, f2, c2, (const long long*)timesteps, t_count, cache, ctx_len, out, B, H, W);
}

// dmx_unet_forward + debug taps: the block outputs conv_in, down0..3, mid, up0..3 (the oracle's tap points) are copied out as
// NCHW fp32, back to back, into `taps`; shapes (B, C, H, W) land in tap_shapes[i*4..], the count in *n_taps.
extern "C" int dmx_unet_forward_taps(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                     const int64_t* timesteps, int t_count, const void* cache, int ctx_len, float* out, int B, int H, int W,
                                     void* workspace, size_t workspace_bytes, float* taps, size_t tap_floats, int* tap_shapes, int* n_taps, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_forward_taps: weights not finalized");
  DMX_REQUIRE(f0 && out && timesteps && cache && workspace && taps && tap_shapes && n_taps, "unet_forward_taps: null argument");
  DMX_REQUIRE(c0 + c1 + c2 == u->cfg.in_channels && B > 0 && H % 8 == 0 && W % 8 == 0, "unet_forward_taps: bad shapes");
  TapSink sink; sink.buf = taps; sink.cap = tap_floats;
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false); ex.taps = &sink;
  const int rc = unet_run(u, ex, f0, c0, f1, c1, f2, c2, (const long long*)timesteps, t_count, cache, ctx_len, out, B, H, W);
  *n_taps = sink.n;
  for (int i = 0; i < sink.n; ++i) for (int k = 0; k < 4; ++k) tap_shapes[4 * i + k] = sink.shape[i][k];
  return rc;
}

// fp32 VALIDATION forward (tests): the same graph walker on fp32 activations, the fp32 master copy of the parameters
// (`masters`: dmx_unet_grad_bytes(u) bytes filled by dmx_unet_master_import for every parameter) and the plain fp32 kernels of
// ref_f32.hip.  `context` is the raw glyph context [B][ctx_len][cross_attention_dim] fp32 (its K / V are projected in the call).
// taps / tap_shapes / n_taps may be NULL.  Never used by the product path.
extern "C" size_t dmx_unet_workspace_bytes_f32(dmx_unet* u, int B, int H, int W, int ctx_len) {
  if (!u) return 0;
  Exec ex; ex.dry = true; ex.f32 = true; ex.ws.reset(nullptr, 0, true);
  unet_run(u, ex, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, B, nullptr, ctx_len, nullptr, B, H, W);
  return ex.ws.peak() + 4096;
}
extern "C" int dmx_unet_forward_f32(dmx_unet* u, const void* masters, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                    const int64_t* timesteps, int t_count, const float* context, int ctx_len, float* out, int B, int H, int W,
                                    void* workspace, size_t workspace_bytes, float* taps, size_t tap_floats, int* tap_shapes, int* n_taps, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized && masters, "unet_forward_f32: weights not finalized / no master arena");
  DMX_REQUIRE(f0 && out && timesteps && context && workspace, "unet_forward_f32: null argument");
  DMX_REQUIRE(c0 + c1 + c2 == u->cfg.in_channels && B > 0 && H % 8 == 0 && W % 8 == 0, "unet_forward_f32: bad shapes");
  TapSink sink; sink.buf = taps; sink.cap = tap_floats;
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false); ex.f32 = true;
  if (taps) ex.taps = &sink;
  u->masters_f32 = masters;
  const int rc = unet_run(u, ex, f0, c0, f1, c1, f2, c2, (const long long*)timesteps, t_count, context, ctx_len, out, B, H, W);
  u->masters_f32 = nullptr;
  if (n_taps) { *n_taps = sink.n; for (int i = 0; i < sink.n; ++i) for (int k = 0; k < 4; ++k) tap_shapes[4 * i + k] = sink.shape[i][k]; }
  return rc;
}

// Same contract as dmx_unet_forward, but the launch sequence (~600 kernels) is captured into a hipGraph the second
// time an identical argument tuple is seen and replayed afterwards (one hipGraphLaunch per UNet step).  Needs a
// non-default stream (the legacy NULL stream cannot be captured); falls back to eager launches otherwise or while
// the profiler is recording.
bool dmx_profile_active();
extern "C" int dmx_unet_forward_graph(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                      const int64_t* timesteps, int t_count, const void* cache, int ctx_len,
                                      float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_forward_graph: weights not finalized");
  if (stream == nullptr || dmx_profile_active())
    return dmx_unet_forward(u, f0, c0, f1, c1, f2, c2, timesteps, t_count, cache, ctx_len, out, B, H, W, workspace, workspace_bytes, stream);
  dmx_unet::GraphKey key(f0, f1, f2, timesteps, cache, out, workspace, c0, c1, c2, t_count, ctx_len, B, H, W, u->temb_table, u->temb_step, dmx_plan_epoch());
  dmx_unet::GraphEntry& e = u->graphs[key];
  hipStream_t s = (hipStream_t)stream;
  if (e.exec) { DMX_HIP(hipGraphLaunch(e.exec, s)); return dmx_poll_device_error(); }      // (what an earlier replay raised: common.h)
  if (e.seen++ == 0)        // first sight: eager (also runs every one-time hipFuncSetAttribute outside a capture)
    return dmx_unet_forward(u, f0, c0, f1, c1, f2, c2, timesteps, t_count, cache, ctx_len, out, B, H, W, workspace, workspace_bytes, stream);
  if (u->graphs.size() > 64) { u->graphs.erase(key); u->drop_graphs(); }
  DMX_REQUIRE(f0 && out && timesteps && cache && workspace, "unet_forward_graph: null argument");
  DMX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  Exec ex; ex.stream = s; ex.ws.reset(workspace, workspace_bytes, false);
  const int rc = unet_run_planned(u, ex, f0, c0, f1, c1, f2, c2, (const long long*)timesteps, t_count, cache, ctx_len, out, B, H, W);
  hipGraph_t g = nullptr;
  const hipError_t ce = hipStreamEndCapture(s, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (ce != hipSuccess || !g) { dmx_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return DMX_ERR_HIP; }
  hipGraphExec_t exec = nullptr;
  const hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) { dmx_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ie)); return DMX_ERR_HIP; }
  u->graphs[key].exec = exec;
  DMX_HIP(hipGraphLaunch(exec, s));
  return DMX_OK;
}
