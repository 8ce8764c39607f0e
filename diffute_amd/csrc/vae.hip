// AutoencoderKL executor (SD VAE layout): the graphs behind `vae.encode(x)` (reference call
// sites app.ipynb:781,793; train_diffute_v1.py:875,886) and `vae.decode(z).sample`
// (app.ipynb:819); module structure SURVEY.md Appendix A.2.  Same kernels as the UNet:
// >95% of the FLOPs are the 3x3 convolutions (implicit GEMM on MFMA), GroupNorm+SiLU is the
// HBM-bound remainder.  The asymmetric (0,1,0,1) pad of the encoder's stride-2 convs and the
// decoder's nearest x2 upsamples are folded into the conv gather.  The single-head d=512
// mid-block attention is one fused flash-style kernel (attention_wide.hip).
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include "vae_model.h"

namespace {

void small_conv(ParamTable& pt, CW& c, const std::string& p, int cout, int cin, int ks) {
  c.cin = cin; c.cout = cout; c.kpad = (int)align_up((size_t)ks * ks * cin, 64);
  c.w = pt.reserve((size_t)cout * c.kpad * 2);
  pt.conv_at(p + "weight", cout, cin, ks, c.w, c.kpad, 0);
  c.b = pt.f32(p + "bias", cout);
}
void big_conv(ParamTable& pt, CW& c, const std::string& p, int cout, int cin) {
  c.cin = cin; c.cout = cout; c.kpad = 9 * cin;
  c.w = pt.reserve((size_t)cout * 9 * cin * 2);
  pt.conv_at(p + "weight", cout, cin, 3, c.w, 9 * cin, 0);
  c.b = pt.f32(p + "bias", cout);
}
void attn_build(ParamTable& pt, AttnW& a, const std::string& p, int C) {
  a.C = C;
  a.gg = pt.f32(p + "group_norm.weight", C); a.gb = pt.f32(p + "group_norm.bias", C);
  // to_q | to_k | to_v stacked (one GEMM with N = 3C; the fused attention kernel reads the three column ranges in place)
  a.wq = pt.reserve((size_t)3 * C * C * 2); a.wk = a.wq + (size_t)C * C * 2; a.wv = a.wk + (size_t)C * C * 2;
  a.bq = pt.reserve((size_t)3 * C * 4); a.bk = a.bq + (size_t)C * 4; a.bv = a.bk + (size_t)C * 4;
  pt.linear_at(p + "to_q.weight", C, C, a.wq, C); pt.f32_at(p + "to_q.bias", C, a.bq);
  pt.linear_at(p + "to_k.weight", C, C, a.wk, C); pt.f32_at(p + "to_k.bias", C, a.bk);
  pt.linear_at(p + "to_v.weight", C, C, a.wv, C); pt.f32_at(p + "to_v.bias", C, a.bv);
  a.wo = pt.linear(p + "to_out.0.weight", C, C); a.bo = pt.f32(p + "to_out.0.bias", C);
}

// single-head attention over H*W tokens with d = C: GroupNorm -> q|k|v (one GEMM) -> fused flash-style attention
// (attention_wide.hip: nothing of size S x S is materialised) -> to_out + residual
Tn attn_run(Exec& ex, const dmx_vae* v, const AttnW& w, const Tn& x, int G) {
  const int C = w.C, S = x.H * x.W;
  Tn n = ex.groupnorm(x, nullptr, v->W<float>(w.gg), v->W<float>(w.gb), G, 1e-6f, false);
  const float* bqkv = v->W<float>(w.bq);
  float* btmp = nullptr;
  if (ex.f32) {                                        // the three bias vectors are not adjacent in the fp32 master arena
    btmp = (float*)ex.raw((size_t)3 * C * 4);
    if (!ex.dry && !ex.rc) {
      const size_t offs[3] = {w.bq, w.bk, w.bv};
      for (int i = 0; i < 3 && !ex.rc; ++i)
        if (hipMemcpyAsync(btmp + (size_t)i * C, v->W<float>(offs[i]), (size_t)C * 4, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess) {
          dmx_set_error("hipMemcpyAsync failed (attention biases)"); ex.rc = DMX_ERR_HIP;
        }
    }
    bqkv = btmp;
  }
  Tn qkv = ex.linear(n, v->W<bf16>(w.wq), 3 * C, bqkv, nullptr, false);
  ex.drop(n);
  if (btmp) ex.drop(btmp);
  Tn a = ex.make(x.B, x.H, x.W, C);
  if (ex.f32) {
    if (!ex.dry && !ex.rc) {
      const float* qp = (const float*)qkv.p;
      ex.rc = dmx_attention_f32_launch(qp, 3 * C, qp + C, 3 * C, qp + 2 * C, 3 * C, S, (float*)a.p, C, x.B, 1, S, S, 1.0f / sqrtf((float)C), ex.stream, C);
    }
  } else if (!ex.dry && !ex.rc) {
    AttnWideArgs aa{};
    aa.q = qkv.p; aa.k = qkv.p + C; aa.v = qkv.p + 2 * C; aa.ldq = aa.ldk = aa.ldv = 3 * C; aa.kv_rows = S;
    aa.o = a.p; aa.ldo = C; aa.B = x.B; aa.Sq = S; aa.Skv = S; aa.D = C; aa.scale = 1.0f / sqrtf((float)C);
    char tag[96]; snprintf(tag, sizeof(tag), "B=%d H=1 Sq=%d Skv=%d d=%d", x.B, S, S, C);
    ProfScope ps(PROF_ATTN, ex.stream, 4.0 * x.B * (double)S * S * C, 2.0 * C * x.B * (4.0 * S), tag);
    ex.rc = dmx_attention_wide_launch(aa, ex.stream);
  }
  ex.drop(qkv);
  Tn y = ex.linear(a, v->W<bf16>(w.wo), C, v->W<float>(w.bo), &x, false, nullptr, nullptr, true);
  ex.drop(a);
  return y;
}

int vae_encode_run(dmx_vae* v, Exec& ex, const float* x, float* moments, int B, int H, int W) {
  const dmx_vae_config& c = v->cfg; const int G = c.norm_num_groups; const int L = c.layers_per_block;
  Tn h;
  if (ex.f32) {                                        // NCHW -> NHWC, then the generic conv on the K-padded filter matrix
    Tn xn = ex.make(B, H, W, c.in_channels);
    if (!ex.dry && !ex.rc) ex.rc = dmx_concat_nchw_to_nhwc_f32_launch(x, c.in_channels, nullptr, 0, nullptr, 0, (float*)xn.p, B, H * W, ex.stream);
    ConvOpts oi; oi.bias = v->W<float>(v->e_in.b); oi.ldw = v->e_in.kpad;
    h = ex.conv(xn, nullptr, v->W<bf16>(v->e_in.w), c.block_out_channels[0], oi);
    ex.drop(xn);
  } else {
    Tn col = ex.make(B, H, W, v->e_in.kpad);
    if (!ex.dry && !ex.rc) {
      Im2colArgs a{}; a.f0 = x; a.c0 = c.in_channels; a.C = c.in_channels; a.B = B; a.IH = a.OH = H; a.IW = a.OW = W;
      a.ksize = 3; a.stride = 1; a.pad = 1; a.out = col.p; a.Kpad = v->e_in.kpad;
      ex.rc = dmx_im2col_small_launch(a, ex.stream);
    }
    h = ex.linear(col, v->W<bf16>(v->e_in.w), c.block_out_channels[0], v->W<float>(v->e_in.b), nullptr, false, nullptr, nullptr, true);
    ex.drop(col);
  }
  ex.ensure_stats(h);                                  // (statistics records for the first resnet's fused GroupNorm -> conv launch)
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < L; ++j) {
      Tn y = resnet_run(ex, v->wbase(), v->e_res[i][j], h, nullptr, G, 1e-6f, nullptr, 0, v->wmul());
      ex.drop(h); h = y;
    }
    if (i < 3) {                       // F.pad(h,(0,1,0,1)) + conv s2 p0: the gather's range check is the pad
      ConvOpts o; o.stride = 2; o.pad = 0; o.bias = v->W<float>(v->e_ds[i].b); o.stats = 1;
      Tn y = ex.conv(h, nullptr, v->W<bf16>(v->e_ds[i].w), c.block_out_channels[i], o);
      ex.ensure_stats(y);
      ex.drop(h); h = y;
    }
  }
  { Tn y = resnet_run(ex, v->wbase(), v->e_mid[0], h, nullptr, G, 1e-6f, nullptr, 0, v->wmul()); ex.drop(h);
    Tn z = attn_run(ex, v, v->e_attn, y, G); ex.drop(y);
    ex.ensure_stats(z);
    h = resnet_run(ex, v->wbase(), v->e_mid[1], z, nullptr, G, 1e-6f, nullptr, 0, v->wmul()); ex.drop(z); }
  Tn t = ex.groupnorm(h, nullptr, v->W<float>(v->e_ng), v->W<float>(v->e_nb), G, 1e-6f, true);
  ex.drop(h);
  ConvOpts oo; oo.bias = v->W<float>(v->e_out.b);
  Tn m8 = ex.conv(t, nullptr, v->W<bf16>(v->e_out.w), 2 * c.latent_channels, oo);   // [M][8] bf16
  ex.drop(t);
  if (ex.f32) {
    ConvOpts oq; oq.ksize = 1; oq.pad = 0; oq.bias = v->W<float>(v->quant.b); oq.ldw = v->quant.kpad;
    Tn mo = ex.conv(m8, nullptr, v->W<bf16>(v->quant.w), 2 * c.latent_channels, oq);
    if (!ex.dry && !ex.rc) ex.rc = dmx_nhwc_to_nchw_f32_launch((const float*)mo.p, mo.ld, moments, B, mo.C, m8.H * m8.W, ex.stream);
    ex.drop(m8); ex.drop(mo);
    return ex.rc;
  }
  // quant_conv 1x1 (8 -> 8): pad K to 64 and reuse the GEMM; fp32 out, then NCHW
  Tn qc = ex.make(B, m8.H, m8.W, v->quant.kpad);
  if (!ex.dry && !ex.rc) {
    Im2colArgs a{}; a.h = m8.p; a.ldh = m8.ld; a.C = m8.C; a.B = B; a.IH = a.OH = m8.H; a.IW = a.OW = m8.W;
    a.ksize = 1; a.stride = 1; a.pad = 0; a.out = qc.p; a.Kpad = v->quant.kpad;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }
  const int Mo = B * m8.H * m8.W, C2 = 2 * c.latent_channels;
  float* mo = (float*)ex.raw((size_t)Mo * C2 * 4);
  ex.gemm_raw(qc.p, qc.ld, Mo, v->W<bf16>(v->quant.w), v->quant.kpad, C2, v->quant.kpad, v->W<float>(v->quant.b), mo, C2, 1);
  if (!ex.dry && !ex.rc) ex.rc = dmx_nhwc_to_nchw_f32_launch(mo, C2, moments, B, C2, m8.H * m8.W, ex.stream);
  ex.drop(qc); ex.drop(m8); ex.drop(mo);
  return ex.rc;
}

int vae_decode_run(dmx_vae* v, Exec& ex, const float* z, float* image, int B, int h0, int w0) {
  const dmx_vae_config& c = v->cfg; const int G = c.norm_num_groups; const int L = c.layers_per_block;
  const int lc = c.latent_channels;
  Tn h;
  if (ex.f32) {
    Tn zn = ex.make(B, h0, w0, lc);
    if (!ex.dry && !ex.rc) ex.rc = dmx_concat_nchw_to_nhwc_f32_launch(z, lc, nullptr, 0, nullptr, 0, (float*)zn.p, B, h0 * w0, ex.stream);
    ConvOpts op; op.ksize = 1; op.pad = 0; op.bias = v->W<float>(v->pquant.b); op.ldw = v->pquant.kpad;
    Tn z2 = ex.conv(zn, nullptr, v->W<bf16>(v->pquant.w), lc, op);
    ex.drop(zn);
    ConvOpts oi; oi.bias = v->W<float>(v->d_in.b); oi.ldw = v->d_in.kpad;
    h = ex.conv(z2, nullptr, v->W<bf16>(v->d_in.w), c.block_out_channels[3], oi);
    ex.drop(z2);
  } else {
  // post_quant_conv 1x1 on the NCHW fp32 latents
  Tn pc = ex.make(B, h0, w0, v->pquant.kpad);
  if (!ex.dry && !ex.rc) {
    Im2colArgs a{}; a.f0 = z; a.c0 = lc; a.C = lc; a.B = B; a.IH = a.OH = h0; a.IW = a.OW = w0;
    a.ksize = 1; a.stride = 1; a.pad = 0; a.out = pc.p; a.Kpad = v->pquant.kpad;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }
  Tn z2 = ex.make(B, h0, w0, lc);
  ex.gemm_raw(pc.p, pc.ld, B * h0 * w0, v->W<bf16>(v->pquant.w), v->pquant.kpad, lc, v->pquant.kpad, v->W<float>(v->pquant.b), z2.p, lc, 0);
  ex.drop(pc);
  Tn col = ex.make(B, h0, w0, v->d_in.kpad);
  if (!ex.dry && !ex.rc) {
    Im2colArgs a{}; a.h = z2.p; a.ldh = z2.ld; a.C = lc; a.B = B; a.IH = a.OH = h0; a.IW = a.OW = w0;
    a.ksize = 3; a.stride = 1; a.pad = 1; a.out = col.p; a.Kpad = v->d_in.kpad;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }
  ex.drop(z2);
  h = ex.linear(col, v->W<bf16>(v->d_in.w), c.block_out_channels[3], v->W<float>(v->d_in.b), nullptr, false, nullptr, nullptr, true);
  ex.drop(col);
  }
  ex.ensure_stats(h);
  { Tn y = resnet_run(ex, v->wbase(), v->d_mid[0], h, nullptr, G, 1e-6f, nullptr, 0, v->wmul()); ex.drop(h);
    Tn zz = attn_run(ex, v, v->d_attn, y, G); ex.drop(y);
    ex.ensure_stats(zz);
    h = resnet_run(ex, v->wbase(), v->d_mid[1], zz, nullptr, G, 1e-6f, nullptr, 0, v->wmul()); ex.drop(zz); }
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < L + 1; ++j) {
      Tn y = resnet_run(ex, v->wbase(), v->d_res[i][j], h, nullptr, G, 1e-6f, nullptr, 0, v->wmul());
      ex.drop(h); h = y;
    }
    if (i < 3) {
      // nearest x2 + conv3x3 as four 2x2 phase convolutions on the source grid (GemmArgs.ups2)
      const bool direct = ex.f32;                                          // (the phase weights are derived data of the bf16 path)
      ConvOpts o; o.ups = 1; o.ups2 = direct ? 0 : 1; o.bias = v->W<float>(v->d_us[i].b); o.stats = 1;
      Tn y = ex.conv(h, nullptr, v->W<bf16>(direct ? v->d_us[i].w : v->d_us[i].wp), c.block_out_channels[3 - i], o);
      ex.ensure_stats(y);
      ex.drop(h); h = y;
    }
  }
  Tn t = ex.groupnorm(h, nullptr, v->W<float>(v->d_ng), v->W<float>(v->d_nb), G, 1e-6f, true);
  ex.drop(h);
  const int Mo = B * t.H * t.W;
  float* im = (float*)ex.raw((size_t)Mo * c.out_channels * 4);
  ConvOpts oo; oo.bias = v->W<float>(v->d_out.b); oo.out_f32 = 1;
  ex.conv(t, nullptr, v->W<bf16>(v->d_out.w), c.out_channels, oo, im);
  if (!ex.dry && !ex.rc) ex.rc = dmx_nhwc_to_nchw_f32_launch(im, c.out_channels, image, B, c.out_channels, t.H * t.W, ex.stream);
  ex.drop(t); ex.drop(im);
  return ex.rc;
}

}  // namespace

extern "C" dmx_vae* dmx_vae_create(const dmx_vae_config* cfg) {
  if (!cfg) { dmx_set_error("vae_create: null config"); return nullptr; }
  for (int i = 0; i < 4; ++i)
    if (cfg->block_out_channels[i] % 64 != 0) { dmx_set_error("vae_create: block_out_channels must be multiples of 64"); return nullptr; }
  if (!dmx_attention_wide_supported(cfg->block_out_channels[3])) {
    dmx_set_error("vae_create: the mid-block attention width block_out_channels[3]=%d must be 128, 256 or 512", cfg->block_out_channels[3]); return nullptr;
  }
  auto v = std::make_unique<dmx_vae>();
  v->cfg = *cfg;
  ParamTable& pt = v->pt;
  const int* boc = cfg->block_out_channels; const int L = cfg->layers_per_block; const int lc = cfg->latent_channels;
  const std::string e = "encoder.", d = "decoder.";
  small_conv(pt, v->e_in, e + "conv_in.", boc[0], cfg->in_channels, 3);
  int cprev = boc[0];
  for (int i = 0; i < 4; ++i) {
    v->e_res[i].resize(L);
    for (int j = 0; j < L; ++j) {
      resnet_build(pt, v->e_res[i][j], e + "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", cprev, boc[i]);
      cprev = boc[i];
    }
    if (i < 3) big_conv(pt, v->e_ds[i], e + "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.", boc[i], boc[i]);
  }
  resnet_build(pt, v->e_mid[0], e + "mid_block.resnets.0.", cprev, cprev);
  attn_build(pt, v->e_attn, e + "mid_block.attentions.0.", cprev);
  resnet_build(pt, v->e_mid[1], e + "mid_block.resnets.1.", cprev, cprev);
  v->e_ng = pt.f32(e + "conv_norm_out.weight", cprev); v->e_nb = pt.f32(e + "conv_norm_out.bias", cprev);
  big_conv(pt, v->e_out, e + "conv_out.", 2 * lc, cprev);
  small_conv(pt, v->quant, "quant_conv.", 2 * lc, 2 * lc, 1);
  small_conv(pt, v->pquant, "post_quant_conv.", lc, lc, 1);
  cprev = boc[3];
  small_conv(pt, v->d_in, d + "conv_in.", cprev, lc, 3);
  resnet_build(pt, v->d_mid[0], d + "mid_block.resnets.0.", cprev, cprev);
  attn_build(pt, v->d_attn, d + "mid_block.attentions.0.", cprev);
  resnet_build(pt, v->d_mid[1], d + "mid_block.resnets.1.", cprev, cprev);
  for (int i = 0; i < 4; ++i) {
    const int c = boc[3 - i];
    v->d_res[i].resize(L + 1);
    for (int j = 0; j < L + 1; ++j) {
      resnet_build(pt, v->d_res[i][j], d + "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", cprev, c);
      cprev = c;
    }
    if (i < 3) {
      big_conv(pt, v->d_us[i], d + "up_blocks." + std::to_string(i) + ".upsamplers.0.conv.", c, c);
      v->d_us[i].wp = pt.reserve((size_t)4 * c * 4 * c * 2);        // derived: phase weights of the upsample conv (dmx_vae_finalize)
    }
  }
  v->d_ng = pt.f32(d + "conv_norm_out.weight", cprev); v->d_nb = pt.f32(d + "conv_norm_out.bias", cprev);
  big_conv(pt, v->d_out, d + "conv_out.", cfg->out_channels, cprev);
  return v.release();
}

extern "C" void dmx_vae_destroy(dmx_vae* v) { delete v; }
extern "C" int dmx_vae_param_count(const dmx_vae* v) { return v ? (int)v->pt.entries().size() : 0; }
extern "C" int dmx_vae_param_info(const dmx_vae* v, int index, const char** name, int shape[4]) {
  DMX_REQUIRE(v && index >= 0 && index < (int)v->pt.entries().size(), "vae_param_info: bad index %d", index);
  const ParamEntry& e = v->pt.entries()[index];
  if (name) *name = e.name.c_str();
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = e.shape[k];
  return DMX_OK;
}
extern "C" size_t dmx_vae_arena_bytes(const dmx_vae* v) { return v ? v->pt.total() : 0; }
extern "C" int dmx_vae_bind_arena(dmx_vae* v, void* arena, size_t bytes) {
  DMX_REQUIRE(v && arena && bytes >= v->pt.total(), "vae_bind_arena: need %zu bytes", v ? v->pt.total() : (size_t)0);
  v->arena = (char*)arena; v->finalized = false;
  DMX_HIP(hipMemset(arena, 0, v->pt.total()));
  return DMX_OK;
}
extern "C" int dmx_vae_load_param(dmx_vae* v, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(v != nullptr, "vae_load_param: null handle");
  v->finalized = false;
  return v->pt.load(v->arena, name, src, (hipStream_t)stream);
}
extern "C" int dmx_vae_finalize(dmx_vae* v, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->arena, "vae_finalize: arena not bound");
  hipStream_t s = (hipStream_t)stream;
  int rc = 0;
  for (int i = 0; i < 4 && !rc; ++i) {
    for (auto& r : v->e_res[i]) if (!rc) rc = resnet_finalize(r, v->arena, s);
    for (auto& r : v->d_res[i]) if (!rc) rc = resnet_finalize(r, v->arena, s);
  }
  for (int k = 0; k < 2 && !rc; ++k) { rc = resnet_finalize(v->e_mid[k], v->arena, s); if (!rc) rc = resnet_finalize(v->d_mid[k], v->arena, s); }
  for (int i = 0; i < 3 && !rc; ++i)
    rc = dmx_ups_phase_weights_launch(v->at<bf16>(v->d_us[i].w), v->d_us[i].kpad, v->at<bf16>(v->d_us[i].wp), v->d_us[i].cout, v->d_us[i].cin, s);
  DMX_HIP(hipStreamSynchronize(s));
  v->finalized = (rc == 0);
  return rc;
}
extern "C" size_t dmx_vae_workspace_bytes(dmx_vae* v, int B, int H, int W, int decode) {
  if (!v) return 0;
  Exec ex; ex.dry = true; ex.ws.reset(nullptr, 0, true);
  if (decode) vae_decode_run(v, ex, nullptr, nullptr, B, H, W); else vae_encode_run(v, ex, nullptr, nullptr, B, H, W);
  return ex.ws.peak() + 4096;
}
extern "C" int dmx_vae_encode(dmx_vae* v, const float* x, float* moments, int B, int H, int W,
                              void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->finalized, "vae_encode: weights not finalized");
  DMX_REQUIRE(x && moments && workspace, "vae_encode: null argument");
  DMX_REQUIRE(B > 0 && H % 8 == 0 && W % 8 == 0 && H > 0 && W > 0, "vae_encode: H=%d W=%d must be positive multiples of 8", H, W);
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  return vae_encode_run(v, ex, x, moments, B, H, W);
}
// ---- fp32 VALIDATION instantiation (tests only): the same walkers on fp32 activations, the caller's fp32 master copy of the
// parameters and the plain fp32 kernels of ref_f32.hip
extern "C" size_t dmx_vae_workspace_bytes_f32(dmx_vae* v, int B, int H, int W, int decode) {
  if (!v) return 0;
  Exec ex; ex.dry = true; ex.f32 = true; ex.ws.reset(nullptr, 0, true);
  v->masters_f32 = (const char*)4096;                  // dry run: pointers are never dereferenced
  if (decode) vae_decode_run(v, ex, nullptr, nullptr, B, H, W); else vae_encode_run(v, ex, nullptr, nullptr, B, H, W);
  v->masters_f32 = nullptr;
  return ex.ws.peak() + 4096;
}
extern "C" int dmx_vae_encode_f32(dmx_vae* v, const void* masters, const float* x, float* moments, int B, int H, int W,
                                  void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && masters && x && moments && workspace, "vae_encode_f32: null argument");
  DMX_REQUIRE(B > 0 && H % 8 == 0 && W % 8 == 0 && H > 0 && W > 0, "vae_encode_f32: H=%d W=%d must be positive multiples of 8", H, W);
  Exec ex; ex.stream = (hipStream_t)stream; ex.f32 = true; ex.ws.reset(workspace, workspace_bytes, false);
  v->masters_f32 = (const char*)masters;
  const int rc = vae_encode_run(v, ex, x, moments, B, H, W);
  v->masters_f32 = nullptr;
  return rc;
}
extern "C" int dmx_vae_decode_f32(dmx_vae* v, const void* masters, const float* z, float* image, int B, int h, int w,
                                  void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && masters && z && image && workspace, "vae_decode_f32: null argument");
  DMX_REQUIRE(B > 0 && h > 0 && w > 0, "vae_decode_f32: empty problem");
  Exec ex; ex.stream = (hipStream_t)stream; ex.f32 = true; ex.ws.reset(workspace, workspace_bytes, false);
  v->masters_f32 = (const char*)masters;
  const int rc = vae_decode_run(v, ex, z, image, B, h, w);
  v->masters_f32 = nullptr;
  return rc;
}
extern "C" int dmx_vae_decode(dmx_vae* v, const float* z, float* image, int B, int h, int w,
                              void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->finalized, "vae_decode: weights not finalized");
  DMX_REQUIRE(z && image && workspace, "vae_decode: null argument");
  DMX_REQUIRE(B > 0 && h > 0 && w > 0, "vae_decode: empty problem");
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  return vae_decode_run(v, ex, z, image, B, h, w);
}
