// Training graph of the UNet (reference train_diffute_v1.py:913-925): forward that keeps what the
// backward needs, MSE loss, and the hand-written backward through every block down to fp32 parameter gradients.
//
//   forward  : same function as the inference graph, but unfused where the backward needs an intermediate: explicit
//              LayerNorm (no folding), FF1 without the GEGLU epilogue (+ an elementwise GEGLU on the packed column
//              order), GroupNorm keeps (mean, rstd), attention keeps the row log-sum-exp; the context K/V are projected
//              inside the step (their weights train).  Nothing is recycled: activations stay in the workspace.
//   backward : per GEMM  dX = dY W  through the forward GEMM kernel with W^T from the transposed arena
//              (dmx_unet_train_prepare), dW = dY^T X with the wgrad kernel, db = column sums; GroupNorm / LayerNorm /
//              GEGLU / attention backward kernels; a tensor with several consumers gets its gradients summed through the
//              `res` input of whichever backward kernel runs last for it (no in-place accumulation, deterministic).
//   gradients: fp32, in the PACKED layout of the weights arena: the gradient of the element at arena byte offset o lives
//              at byte offset 2*o of the gradient arena (bf16 weight -> fp32, same row strides).  dmx_unet_grad_export
//              unpacks one parameter into its torch layout.
#include <math.h>
#include "unet_model.h"
#include "train_common.h"

namespace {

struct XfSave {
  Tn x, t, h, n1, qkv, a1, h2, n2, q, kv, a2, h3, n3, ffh, g, h4;
  float* stg = nullptr; float* lse1 = nullptr; float* lse2 = nullptr;
};
struct ConvSave { Tn x; };

struct Train : TrainOps {
  dmx_unet* u; int ctx_len;
  Tn ctxp;                                       // padded context [B*sp][D]
  Train(dmx_unet* u_, Exec& ex_, char* wt_, char* gr_, int B_, int ctx_len_)
      : TrainOps(ex_, u_ ? u_->arena : nullptr, wt_, gr_, u_->cfg.norm_num_groups, B_), u(u_), ctx_len(ctx_len_) { tproj_total = u_->tproj_total; }

  // ------------------------------------------------------------------ Transformer2DModel (one BasicTransformerBlock)
  Tn xf_fwd(const XfW& w, const Tn& x, XfSave& s) {
    const int C = w.C, S = x.H * x.W, sp = dmx_ctx_pad(ctx_len), D = u->cfg.cross_attention_dim;
    s.x = x;
    s.t = gn(x, nullptr, w.ng, w.nb, 1e-6f, false, &s.stg);
    s.h = ex.linear(s.t, W(w.wpi), C, F(w.bpi), nullptr, false);
    s.n1 = ex.layernorm(s.h, F(w.l1g), F(w.l1b), 1e-5f);
    s.qkv = ex.linear(s.n1, W(w.wqkv_raw), 3 * C, nullptr, nullptr, false);
    s.a1 = ex.make(x.B, x.H, x.W, C);
    attn(s.qkv.p, 3 * C, s.qkv.p + C, 3 * C, s.qkv.p + 2 * C, 3 * C, S, s.a1, &s.lse1, w.heads, S, S);
    s.h2 = ex.linear(s.a1, W(w.wo1), C, F(w.bo1), &s.h, false);
    s.n2 = ex.layernorm(s.h2, F(w.l2g), F(w.l2b), 1e-5f);
    s.q = ex.linear(s.n2, W(w.wq2_raw), C, nullptr, nullptr, false);
    s.kv = ex.make(1, 1, x.B * sp, 2 * C);
    ex.gemm_raw(ctxp.p, D, x.B * sp, W(w.wkv2), D, 2 * C, D, nullptr, s.kv.p, 2 * C, 0);
    s.a2 = ex.make(x.B, x.H, x.W, C);
    attn(s.q.p, C, s.kv.p, 2 * C, s.kv.p + C, 2 * C, sp, s.a2, &s.lse2, w.heads, S, ctx_len);
    s.h3 = ex.linear(s.a2, W(w.wo2), C, F(w.bo2), &s.h2, false);
    s.n3 = ex.layernorm(s.h3, F(w.l3g), F(w.l3b), 1e-5f);
    s.ffh = ex.linear(s.n3, W(w.wf1_raw), 8 * C, F(w.bf1), nullptr, false);       // packed column order, no GEGLU
    s.g = ex.make(x.B, x.H, x.W, 4 * C);
    if (live()) ex.rc = dmx_geglu_fwd_launch(s.ffh.p, s.ffh.ld, s.g.p, s.g.ld, s.g.rows(), 4 * C, 1, ex.stream);
    s.h4 = ex.linear(s.g, W(w.wf2), C, F(w.bf2), &s.h3, false);
    return ex.linear(s.h4, W(w.wpo), C, F(w.bpo), &x, false);
  }
  void attn_bwd(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* v, int ldv, int kv_rows, const Tn& o, const Tn& dout,
                const float* lse, bf16* dq, int lddq, bf16* dk, int lddk, bf16* dv, int lddv, int H, int Sq, int Skv) {
    const size_t wsb = dmx_attn_bwd_ws_bytes(o.B, H, Sq);
    void* ws = ex.raw(wsb);
    if (live()) {
      AttnBwdArgs a{};
      a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_rows = kv_rows;
      a.o = o.p; a.dout = dout.p; a.ldo = o.ld; a.lse = lse; a.delta = (float*)ws;
      a.dq = dq; a.lddq = lddq; a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
      a.B = o.B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = 0.125f;
      ex.rc = dmx_attention_bwd_launch(a, ex.stream);
    }
    ex.drop(ws);
  }
  Tn xf_bwd(const XfW& w, XfSave& s, const Tn& dy) {
    const int C = w.C, S = s.x.H * s.x.W, sp = dmx_ctx_pad(ctx_len);
    // proj_out (+x residual: its gradient dy is added in the GroupNorm backward at the end)
    Tn dh4 = linear_bwd(s.h4, dy, w.wpo, w.bpo, true, nullptr);
    // feed-forward: h4 = FF2(geglu(FF1(LN3(h3)))) + h3
    Tn dg = linear_bwd(s.g, dh4, w.wf2, w.bf2, true, nullptr);
    Tn dffh = ex.make(s.ffh.B, s.ffh.H, s.ffh.W, 8 * C);
    if (live()) ex.rc = dmx_geglu_bwd_launch(s.ffh.p, s.ffh.ld, dg.p, dg.ld, dffh.p, dffh.ld, dg.rows(), 4 * C, 1, ex.stream);
    ex.drop(dg);
    Tn dn3 = linear_bwd(s.n3, dffh, w.wf1_raw, w.bf1, true, nullptr);
    ex.drop(dffh);
    Tn dh3 = ln_bwd(s.h3, w.l3g, w.l3b, dn3, &dh4);
    ex.drop(dn3); ex.drop(dh4);
    // cross attention: h3 = to_out(attn(to_q(LN2(h2)), K, V)) + h2
    Tn da2 = linear_bwd(s.a2, dh3, w.wo2, w.bo2, true, nullptr);
    Tn dq = ex.make(s.q.B, s.q.H, s.q.W, C);
    Tn dkv = ex.make(1, 1, s.kv.rows(), 2 * C);
    if (live()) ex.rc = (int)hipMemsetAsync(dkv.p, 0, (size_t)dkv.rows() * 2 * C * 2, ex.stream) ? DMX_ERR_HIP : 0;   // padded context rows
    attn_bwd(s.q.p, C, s.kv.p, 2 * C, s.kv.p + C, 2 * C, sp, s.a2, da2, s.lse2, dq.p, C, dkv.p, 2 * C, dkv.p + C, 2 * C, w.heads, S, ctx_len);
    ex.drop(da2);
    wgrad(dkv, ctxp, nullptr, 1, 1, 0, G(w.wkv2), u->cfg.cross_attention_dim);
    ex.drop(dkv);
    Tn dn2 = linear_bwd(s.n2, dq, w.wq2_raw, 0, false, nullptr);
    ex.drop(dq);
    Tn dh2 = ln_bwd(s.h2, w.l2g, w.l2b, dn2, &dh3);
    ex.drop(dn2); ex.drop(dh3);
    // self attention: h2 = to_out(attn(qkv(LN1(h)))) + h
    Tn da1 = linear_bwd(s.a1, dh2, w.wo1, w.bo1, true, nullptr);
    Tn dqkv = ex.make(s.qkv.B, s.qkv.H, s.qkv.W, 3 * C);
    attn_bwd(s.qkv.p, 3 * C, s.qkv.p + C, 3 * C, s.qkv.p + 2 * C, 3 * C, S, s.a1, da1, s.lse1,
             dqkv.p, 3 * C, dqkv.p + C, 3 * C, dqkv.p + 2 * C, 3 * C, w.heads, S, S);
    ex.drop(da1);
    Tn dn1 = linear_bwd(s.n1, dqkv, w.wqkv_raw, 0, false, nullptr);
    ex.drop(dqkv);
    Tn dh = ln_bwd(s.h, w.l1g, w.l1b, dn1, &dh2);
    ex.drop(dn1); ex.drop(dh2);
    // proj_in and the GroupNorm; the residual gradient dy joins here
    Tn dt = linear_bwd(s.t, dh, w.wpi, w.bpi, true, nullptr);
    ex.drop(dh);
    Tn dx = gn_bwd(s.x, nullptr, w.ng, w.nb, false, s.stg, dt, &dy, nullptr, nullptr);
    ex.drop(dt);
    return dx;
  }
};

static size_t co_wt_off(const dmx_unet* u) { return align_up(u->pt.total(), 256); }
static size_t tr_table_off(const dmx_unet* u) { return align_up(co_wt_off(u) + (size_t)u->cfg.block_out_channels[0] * 64 * 2, 256); }   // transpose job table

struct TrainState {
  std::vector<ResSave> down_res[4], up_res[4]; std::vector<XfSave> down_xf[4], up_xf[4];
  ResSave mid_res[2]; XfSave mid_xf;
  ConvSave down_ds[4], up_us[4];
  float *sinus = nullptr, *e1 = nullptr, *emb = nullptr;
  Tn col, h0, hlast, tout; float* st_out = nullptr;
  std::vector<Tn> skips;                                // in push order (forward)
};

// One training pass lives in a session: the forward leaves its saved tensors in the caller's workspace (and the
// allocator state here), the backward continues in the same workspace.  One session per handle at a time.
struct TrainSession {
  dmx_unet* u; Exec ex; Train T; TrainState st;
  int B = 0, H = 0, W = 0; bool forward_done = false;
  std::vector<hipEvent_t>* events = nullptr; size_t ev_next = 0;
  TrainSession(dmx_unet* u_, char* wt, int B_, int ctx_len) : u(u_), T(u_, ex, wt, nullptr, B_, ctx_len) {}

  // bucket boundary: everything the backward has produced so far is complete once this event fires
  void mark() {
    if (events && ev_next < events->size() && !ex.dry && !ex.rc) {
      if (hipEventRecord((*events)[ev_next], ex.stream) != hipSuccess) { dmx_set_error("hipEventRecord failed"); ex.rc = DMX_ERR_HIP; }
    }
    ++ev_next;
  }

  int forward(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2, const long long* timesteps, int t_count,
              const void* ctx, int ctx_is_bf16, float* pred, int B_, int H_, int W_) {
    B = B_; H = H_; W = W_;
    const int ctx_len = T.ctx_len;
  const dmx_unet_config& cfg = u->cfg;
  const int* boc = cfg.block_out_channels; const int L = cfg.layers_per_block; const int temb = u->temb_dim;
  const int D = cfg.cross_attention_dim, sp = dmx_ctx_pad(ctx_len);
  auto live = [&]() { return !ex.dry && !ex.rc; };
  // ---- time embedding (kept: the backward needs every stage)
  st.sinus = (float*)ex.raw((size_t)B * boc[0] * 4);
  st.e1 = (float*)ex.raw((size_t)B * temb * 4);
  st.emb = (float*)ex.raw((size_t)B * temb * 4);
  T.tproj = (float*)ex.raw((size_t)B * u->tproj_total * 4);
  T.dtproj = (float*)ex.raw((size_t)B * u->tproj_total * 4);
  if (live()) {
    ex.rc = dmx_timestep_embedding_launch(timesteps, t_count, u->at<float>(u->freq), B, boc[0], st.sinus, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(st.sinus, boc[0], u->at<bf16>(u->te_w1), boc[0], u->at<float>(u->te_b1), st.e1, temb, B, temb, boc[0], 0, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(st.e1, temb, u->at<bf16>(u->te_w2), temb, u->at<float>(u->te_b2), st.emb, temb, B, temb, temb, 1, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_launch(st.emb, temb, u->at<bf16>(u->tp_w), temb, u->at<float>(u->tp_b), T.tproj, u->tproj_total, B, u->tproj_total, temb, 1, ex.stream);
  }
  // ---- context rows, padded to a multiple of 64 per image (zero rows)
  T.ctxp = ex.make(1, 1, B * sp, D);
  if (live()) ex.rc = dmx_cast_pad_rows_launch(ctx, ctx_is_bf16, T.ctxp.p, B, ctx_len, sp, D, ex.stream);
  // ---- conv_in
  st.col = ex.make(B, H, W, u->ci_kpad);
  if (live()) {
    Im2colArgs a{}; a.f0 = f0; a.c0 = c0; a.f1 = f1; a.c1 = c1; a.f2 = f2; a.c2 = c2; a.C = cfg.in_channels;
    a.B = B; a.IH = a.OH = H; a.IW = a.OW = W; a.ksize = 3; a.stride = 1; a.pad = 1; a.out = st.col.p; a.Kpad = u->ci_kpad;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }
  Tn h = ex.linear(st.col, u->at<bf16>(u->ci_w), boc[0], u->at<float>(u->ci_b), nullptr, false);
  st.h0 = h;
  st.skips.push_back(h);
  for (int i = 0; i < 4; ++i) {
    st.down_res[i].resize(L); if (cfg.down_has_attn[i]) st.down_xf[i].resize(L);
    for (int j = 0; j < L; ++j) {
      Tn y = T.res_fwd(u->down_res[i][j], h, nullptr, st.down_res[i][j]);
      if (cfg.down_has_attn[i]) y = T.xf_fwd(u->down_xf[i][j], y, st.down_xf[i][j]);
      h = y; st.skips.push_back(h);
    }
    if (i < 3) {
      ConvOpts o; o.stride = 2; o.pad = 1; o.bias = u->at<float>(u->down_ds[i].b);
      st.down_ds[i].x = h;
      h = ex.conv(h, nullptr, u->at<bf16>(u->down_ds[i].w), boc[i], o);
      st.skips.push_back(h);
    }
  }
  { Tn y = T.res_fwd(u->mid_res[0], h, nullptr, st.mid_res[0]);
    Tn z = T.xf_fwd(u->mid_xf, y, st.mid_xf);
    h = T.res_fwd(u->mid_res[1], z, nullptr, st.mid_res[1]); }
  std::vector<Tn> stack = st.skips;
  for (int i = 0; i < 4; ++i) {
    st.up_res[i].resize(L + 1); if (cfg.up_has_attn[i]) st.up_xf[i].resize(L + 1);
    for (int j = 0; j < L + 1; ++j) {
      Tn s = stack.back(); stack.pop_back();
      Tn y = T.res_fwd(u->up_res[i][j], h, &s, st.up_res[i][j]);
      if (cfg.up_has_attn[i]) y = T.xf_fwd(u->up_xf[i][j], y, st.up_xf[i][j]);
      h = y;
    }
    if (i < 3) {
      ConvOpts o; o.ups = 1; o.bias = u->at<float>(u->up_us[i].b);
      st.up_us[i].x = h;
      h = ex.conv(h, nullptr, u->at<bf16>(u->up_us[i].w), boc[3 - i], o);
    }
  }
  st.hlast = h;
  st.tout = T.gn(h, nullptr, u->cno_g, u->cno_b, 1e-5f, true, &st.st_out);
  const int OC = cfg.out_channels;
  float* eps_nhwc = (float*)ex.raw((size_t)B * H * W * OC * 4);
  { ConvOpts oo; oo.bias = u->at<float>(u->co_b); oo.out_f32 = 1;
    ex.conv(st.tout, nullptr, u->at<bf16>(u->co_w), OC, oo, eps_nhwc); }
  if (live()) ex.rc = dmx_nhwc_to_nchw_f32_launch(eps_nhwc, OC, pred, B, OC, H * W, ex.stream);
  ex.drop(eps_nhwc);
    forward_done = (ex.rc == 0);
    return ex.rc;
  }

  // dpred: gradient of the loss w.r.t. the model output, fp32 NCHW [B][OC][H][W]
  int backward(char* gr, const float* dpred) {
    T.gr = gr;
    const dmx_unet_config& cfg = u->cfg;
    const int* boc = cfg.block_out_channels; const int L = cfg.layers_per_block; const int temb = u->temb_dim;
    const int OC = cfg.out_channels;
    char* wt = T.wt;
    auto live = [&]() { return !ex.dry && !ex.rc; };
    ev_next = 0;
  // conv_out: OC (=4) output channels -> pad dY to 8 columns for the wgrad kernel; the data gradient goes through an
  // im2col of dY (K = 9*OC padded to 64) and a [C0][64] transposed filter
  Tn dy8 = ex.make(B, H, W, 8);
  float* dw8 = (float*)ex.raw((size_t)8 * 9 * boc[0] * 4);
  if (live()) {
    ex.rc = (int)hipMemsetAsync(dy8.p, 0, (size_t)dy8.rows() * 8 * 2, ex.stream) ? DMX_ERR_HIP : 0;
    if (!ex.rc) ex.rc = dmx_nchw_f32_to_nhwc_bf16_launch(dpred, dy8.p, 8, B, OC, H * W, ex.stream);
  }
  T.wgrad(dy8, st.tout, nullptr, 3, 1, 0, dw8, 9 * boc[0]);
  if (live()) ex.rc = (int)hipMemcpyAsync(T.G(u->co_w), dw8, (size_t)OC * 9 * boc[0] * 4, hipMemcpyDeviceToDevice, ex.stream) ? DMX_ERR_HIP : 0;
  { float* db8 = (float*)ex.raw(8 * 4);
    T.colsum(dy8, 1, db8, 8);
    if (live()) ex.rc = (int)hipMemcpyAsync(T.G(u->co_b), db8, (size_t)OC * 4, hipMemcpyDeviceToDevice, ex.stream) ? DMX_ERR_HIP : 0;
    ex.drop(db8); }
  ex.drop(dw8); ex.drop(dy8);
  Tn dcol = ex.make(B, H, W, 64);
  if (live()) {
    Im2colArgs a{}; a.f0 = dpred; a.c0 = OC; a.C = OC; a.B = B; a.IH = a.OH = H; a.IW = a.OW = W; a.ksize = 3; a.stride = 1; a.pad = 1;
    a.out = dcol.p; a.Kpad = 64;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }
  Tn dtout = ex.linear(dcol, (const bf16*)(wt + co_wt_off(u)), boc[0], nullptr, nullptr, false);   // [C0][64], behind the mirrored arena
  ex.drop(dcol);
  Tn dh = T.gn_bwd(st.hlast, nullptr, u->cno_g, u->cno_b, true, st.st_out, dtout, nullptr, nullptr, nullptr);
  ex.drop(dtout);

  mark();                                                     // bucket: conv_out, conv_norm_out
  if (live()) ex.rc = (int)hipMemsetAsync(T.dtproj, 0, (size_t)B * u->tproj_total * 4, ex.stream) ? DMX_ERR_HIP : 0;
  std::map<const bf16*, Tn> sgrad;                            // gradients the up path sends to the skip tensors
  // ---- up blocks, reversed
  size_t sp_idx = 0;                                          // skips consumed by the up path so far (forward order: from the back)
  std::vector<std::pair<int, int>> up_order;                  // (i, j) in forward order
  for (int i = 0; i < 4; ++i) for (int j = 0; j < L + 1; ++j) up_order.push_back({i, j});
  (void)sp_idx;
  for (int i = 3; i >= 0; --i) {
    if (i < 3) {
      // upsample conv: y = conv(up2(x)) + b
      const ConvW& cw = u->up_us[i];
      T.wgrad(dh, st.up_us[i].x, nullptr, 3, 1, 1, T.G(cw.w), 9 * cw.c);
      T.colsum(dh, 1, T.G(cw.b), cw.c);
      float* du = (float*)ex.raw((size_t)dh.rows() * cw.c * 4);
      { ConvOpts o; o.out_f32 = 1; ex.conv(dh, nullptr, T.WT(cw.w), cw.c, o, du); }
      Tn dx = ex.make(B, st.up_us[i].x.H, st.up_us[i].x.W, cw.c);
      if (live()) ex.rc = dmx_sumpool2_launch(du, cw.c, 1, dx.p, dx.ld, B, dx.H, dx.W, cw.c, 0, ex.stream);
      ex.drop(du); ex.drop(dh); dh = dx;
    }
    for (int j = L; j >= 0; --j) {
      if (cfg.up_has_attn[i]) { Tn d = T.xf_bwd(u->up_xf[i][j], st.up_xf[i][j], dh); ex.drop(dh); dh = d; }
      Tn dx1;
      Tn dx0 = T.res_bwd(u->up_res[i][j], st.up_res[i][j], dh, nullptr, &dx1);
      ex.drop(dh); dh = dx0;
      sgrad[st.up_res[i][j].x1.p] = dx1;
    }
    mark();                                                   // bucket: up_blocks[i]
  }
  auto take = [&](const Tn& x) -> Tn* { auto it = sgrad.find(x.p); return it == sgrad.end() ? nullptr : &it->second; };
  auto done = [&](const Tn& x) { auto it = sgrad.find(x.p); if (it != sgrad.end()) { ex.drop(it->second); sgrad.erase(it); } };
  // ---- mid block
  { Tn d = T.res_bwd(u->mid_res[1], st.mid_res[1], dh, nullptr, nullptr); ex.drop(dh);
    Tn e = T.xf_bwd(u->mid_xf, st.mid_xf, d); ex.drop(d);
    Tn* g0 = take(st.mid_res[0].x0);
    dh = T.res_bwd(u->mid_res[0], st.mid_res[0], e, g0, nullptr); ex.drop(e);
    done(st.mid_res[0].x0); }
  mark();                                                     // bucket: mid_block
  // ---- down blocks, reversed.  dh is now the TOTAL gradient of the last down-path tensor.
  for (int i = 3; i >= 0; --i) {
    if (i < 3) {
      // stride-2 conv: its output (a skip) already has its total gradient in dh; input = level i's last tensor (a skip too)
      const ConvW& cw = u->down_ds[i];
      const Tn& x = st.down_ds[i].x;
      T.wgrad(dh, x, nullptr, 3, 2, 0, T.G(cw.w), 9 * cw.c);
      T.colsum(dh, 1, T.G(cw.b), cw.c);
      Tn z = ex.make(B, x.H, x.W, cw.c);
      if (live()) ex.rc = dmx_zero_insert2_launch(dh.p, dh.ld, z.p, B, dh.H, dh.W, cw.c, ex.stream);
      ConvOpts o; o.res = take(x);
      Tn dx = ex.conv(z, nullptr, T.WT(cw.w), cw.c, o);
      ex.drop(z); ex.drop(dh); done(x); dh = dx;
    }
    for (int j = L - 1; j >= 0; --j) {
      if (cfg.down_has_attn[i]) { Tn d = T.xf_bwd(u->down_xf[i][j], st.down_xf[i][j], dh); ex.drop(dh); dh = d; }
      ResSave& rs = st.down_res[i][j];
      Tn* g0 = take(rs.x0);
      Tn dx0 = T.res_bwd(u->down_res[i][j], rs, dh, g0, nullptr);
      ex.drop(dh); done(rs.x0); dh = dx0;
    }
    mark();                                                   // bucket: down_blocks[i]
  }
  // ---- conv_in (im2col GEMM): dW = dh0^T col, db; no data gradient (the latents are inputs)
  T.wgrad(dh, st.col, nullptr, 1, 1, 0, T.G(u->ci_w), u->ci_kpad);
  T.colsum(dh, 1, T.G(u->ci_b), boc[0]);
  ex.drop(dh);
  // ---- time embedding: tproj = Wp silu(emb) + bp ; emb = W2 silu(e1) + b2 ; e1 = W1 sinus + b1
  float* demb = (float*)ex.raw((size_t)B * temb * 4);
  float* de1 = (float*)ex.raw((size_t)B * temb * 4);
  float* dbtp = (float*)ex.raw((size_t)u->tproj_total * 4);
  if (live()) {
    ex.rc = dmx_linear_small_bwd_launch(st.emb, temb, T.dtproj, u->tproj_total, u->at<bf16>(u->tp_w), temb, T.G(u->tp_w), temb, dbtp, 1,
                                        demb, temb, B, u->tproj_total, temb, 1, 0, ex.stream);
    // the per-resnet bias entries live at arena offset tp_b + 4*temb_off, i.e. gradient offset 2*tp_b + 8*temb_off
    auto put = [&](const ResW& r) {
      if (!ex.rc && hipMemcpyAsync(T.G(u->tp_b + (size_t)r.temb_off * 4), dbtp + r.temb_off, (size_t)r.cout * 4, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess) {
        dmx_set_error("hipMemcpyAsync failed (time_emb_proj bias gradient)"); ex.rc = DMX_ERR_HIP;
      }
    };
    for (int i = 0; i < 4; ++i) { for (auto& r : u->down_res[i]) put(r); for (auto& r : u->up_res[i]) put(r); }
    put(u->mid_res[0]); put(u->mid_res[1]);
    if (!ex.rc) ex.rc = dmx_linear_small_bwd_launch(st.e1, temb, demb, temb, u->at<bf16>(u->te_w2), temb, T.G(u->te_w2), temb, T.G(u->te_b2), 1,
                                                    de1, temb, B, temb, temb, 1, 0, ex.stream);
    if (!ex.rc) ex.rc = dmx_linear_small_bwd_launch(st.sinus, boc[0], de1, temb, u->at<bf16>(u->te_w1), boc[0], T.G(u->te_w1), boc[0], T.G(u->te_b1), 1,
                                                    nullptr, 0, B, temb, boc[0], 0, 0, ex.stream);
  }
  ex.drop(dbtp);
  ex.drop(demb); ex.drop(de1);
    mark();
    return ex.rc;
  }
};

// ------------------------------------------------------------------ transposed weights (data-gradient operands)
int transpose_linear(const dmx_unet* u, char* wt, size_t off, int N, int K, hipStream_t s) {
  return dmx_transpose_bf16_launch((const bf16*)(u->arena + off), K, (bf16*)(wt + off), N, N, K, s);
}
// conv [N][ld: tap*Cin + ci] -> [Cin][flip(tap)*N + n]
int transpose_conv(const dmx_unet* u, char* wt, size_t off, int N, int Cin, int ld, int ldt, hipStream_t s) {
  for (int tap = 0; tap < 9; ++tap) {
    const int rc = dmx_transpose_bf16_launch((const bf16*)(u->arena + off) + (size_t)tap * Cin, ld,
                                             (bf16*)(wt + off) + (size_t)(8 - tap) * N, ldt, N, Cin, s);
    if (rc) return rc;
  }
  return DMX_OK;
}
int transpose_resnet(const dmx_unet* u, char* wt, const ResW& r, hipStream_t s) {
  int rc = transpose_conv(u, wt, r.w1, r.cout, r.cin, 9 * r.cin, 9 * r.cout, s);
  const int k2 = 9 * r.cout + (r.shortcut ? r.cin : 0);
  if (!rc) rc = transpose_conv(u, wt, r.w2, r.cout, r.cout, k2, 9 * r.cout, s);
  if (!rc && r.shortcut)      // Wsc^T [cin][cout] behind the 9*cout*cout elements of the main filter
    rc = dmx_transpose_bf16_launch((const bf16*)(u->arena + r.w2) + 9 * r.cout, k2, (bf16*)(wt + r.w2) + (size_t)9 * r.cout * r.cout, r.cout, r.cout, r.cin, s);
  return rc;
}
int transpose_xf(const dmx_unet* u, char* wt, const XfW& x, hipStream_t s) {
  const int C = x.C;
  int rc = transpose_linear(u, wt, x.wpi, C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wqkv_raw, 3 * C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wo1, C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wq2_raw, C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wo2, C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wf1_raw, 8 * C, C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wf2, C, 4 * C, s);
  if (!rc) rc = transpose_linear(u, wt, x.wpo, C, C, s);
  return rc;
}

// ------------------------------------------------------------------ gradient export (packed fp32 -> torch layout)
__global__ __launch_bounds__(256) void dmx_grad_unpack_kernel(const float* g, float* out, int kind, int rows, int cols, int ks, int ld, int koff) {
  const size_t total = (kind == 1) ? (size_t)rows * cols * ks * ks : (size_t)rows * (cols > 0 ? cols : 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (kind == 0) {                                   // fp32 vector
      out[i] = g[i];
    } else if (kind == 1) {                            // conv: out[n][ci][tap] <- g[n][koff + tap*Cin + ci]
      const int kk = ks * ks;
      const int tap = (int)(i % kk); const size_t r = i / kk;
      const int ci = (int)(r % cols); const int n = (int)(r / cols);
      out[i] = g[(size_t)n * ld + koff + (size_t)tap * cols + ci];
    } else if (kind == 2) {                            // linear [rows][cols], row stride ld
      const int c = (int)(i % cols); const size_t r = i / cols;
      out[i] = g[r * ld + c];
    } else {                                           // GEGLU-packed rows (kind 3: matrix, kind 4: vector)
      const int c = (kind == 3) ? (int)(i % cols) : 0;
      const int r = (kind == 3) ? (int)(i / cols) : (int)i;
      const int J = r >> 6, w = r & 63;
      const int src = (w < 32) ? (32 * J + w) : (rows / 2 + 32 * J + (w - 32));       // packed row r holds torch row src
      if (kind == 3) out[(size_t)src * cols + c] = g[(size_t)r * ld + c];
      else out[src] = g[r];
    }
  }
}

}  // namespace

extern "C" size_t dmx_unet_train_workspace_bytes(dmx_unet* u, int B, int H, int W, int ctx_len) {
  if (!u) return 0;
  TrainSession ts(u, nullptr, B, ctx_len);
  ts.ex.dry = true; ts.ex.ws.reset(nullptr, 0, true);
  ts.forward(nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 1, nullptr, 0, nullptr, B, H, W);
  ts.backward(nullptr, nullptr);
  return ts.ex.ws.peak() + 4096;
}

// W^T of every GEMM weight into `wt` (same offsets as the weights arena, + the padded conv_out filter behind it);
// call after the weights change.
extern "C" size_t dmx_unet_train_wt_bytes(const dmx_unet* u) {
  return u ? tr_table_off(u) + DMX_TR_TABLE_BYTES : 0;
}
extern "C" int dmx_unet_train_prepare(dmx_unet* u, void* wt_arena, size_t wt_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_train_prepare: weights not finalized");
  DMX_REQUIRE(wt_arena && wt_bytes >= dmx_unet_train_wt_bytes(u), "unet_train_prepare: need %zu bytes", dmx_unet_train_wt_bytes(u));
  hipStream_t s = (hipStream_t)stream; char* wt = (char*)wt_arena;
  const int* boc = u->cfg.block_out_channels;
  int rc = 0;
  TrBatch batch;                                       // the ~600 transposes below are recorded and run as one launch
  for (int i = 0; i < 4 && !rc; ++i) {
    for (auto& r : u->down_res[i]) if (!rc) rc = transpose_resnet(u, wt, r, s);
    for (auto& r : u->up_res[i]) if (!rc) rc = transpose_resnet(u, wt, r, s);
    for (auto& x : u->down_xf[i]) if (!rc) rc = transpose_xf(u, wt, x, s);
    for (auto& x : u->up_xf[i]) if (!rc) rc = transpose_xf(u, wt, x, s);
    if (i < 3 && !rc) rc = transpose_conv(u, wt, u->down_ds[i].w, u->down_ds[i].c, u->down_ds[i].c, 9 * u->down_ds[i].c, 9 * u->down_ds[i].c, s);
    if (i < 3 && !rc) rc = transpose_conv(u, wt, u->up_us[i].w, u->up_us[i].c, u->up_us[i].c, 9 * u->up_us[i].c, 9 * u->up_us[i].c, s);
  }
  if (!rc) rc = transpose_resnet(u, wt, u->mid_res[0], s);
  if (!rc) rc = transpose_resnet(u, wt, u->mid_res[1], s);
  if (!rc) rc = transpose_xf(u, wt, u->mid_xf, s);
  // conv_out [OC][9*C0] -> [C0][64]: column flip(tap)*OC + n, zero padded
  if (!rc) {
    const int OC = u->cfg.out_channels;
    DMX_REQUIRE(9 * OC <= 64, "unet_train_prepare: out_channels=%d too large for the padded conv_out data gradient", OC);
    DMX_HIP(hipMemsetAsync(wt + co_wt_off(u), 0, (size_t)boc[0] * 64 * 2, s));
    for (int tap = 0; tap < 9 && !rc; ++tap)
      rc = dmx_transpose_bf16_launch((const bf16*)(u->arena + u->co_w) + (size_t)tap * boc[0], 9 * boc[0],
                                     (bf16*)(wt + co_wt_off(u)) + (size_t)(8 - tap) * OC, 64, OC, boc[0], s);
  }
  if (!rc) rc = batch.run(wt + tr_table_off(u), DMX_TR_TABLE_BYTES, u->tr_cache, s);
  return rc;
}

extern "C" size_t dmx_unet_grad_bytes(const dmx_unet* u) { return u ? 2 * u->pt.total() : 0; }

// Training forward (train_diffute_v1.py:913): pred = unet(cat(f0,f1,f2), t, ctx), fp32 NCHW.  What the backward needs
// stays in `workspace`, which must be left untouched until dmx_unet_train_backward (or the next forward) has run.
extern "C" int dmx_unet_train_forward(dmx_unet* u, const void* wt_arena,
                                      const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                      const int64_t* timesteps, int t_count, const void* ctx, int ctx_is_bf16, int ctx_len,
                                      float* pred, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && u->finalized, "unet_train_forward: weights not finalized");
  DMX_REQUIRE(wt_arena && f0 && timesteps && ctx && pred && workspace, "unet_train_forward: null argument");
  DMX_REQUIRE(c0 + c1 + c2 == u->cfg.in_channels, "unet_train_forward: c0+c1+c2=%d != in_channels=%d", c0 + c1 + c2, u->cfg.in_channels);
  DMX_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "unet_train_forward: H=%d W=%d must be positive multiples of 8", H, W);
  DMX_REQUIRE(t_count == 1 || t_count == B, "unet_train_forward: t_count=%d must be 1 or B", t_count);
  auto ts = std::make_shared<TrainSession>(u, (char*)wt_arena, B, ctx_len);
  ts->ex.stream = (hipStream_t)stream; ts->ex.ws.reset(workspace, workspace_bytes, false);
  u->train_state = ts;
  const int rc = ts->forward(f0, c0, f1, c1, f2, c2, (const long long*)timesteps, t_count, ctx, ctx_is_bf16, pred, B, H, W);
  if (rc) u->train_state.reset();
  return rc;
}

// Number of gradient buckets the backward completes in order (conv_out+norm, up_blocks 3..0, mid, down_blocks 3..0,
// conv_in + time embedding), and the [begin, end) byte range of bucket i inside the GRADIENT arena.
extern "C" int dmx_unet_train_bucket_count(const dmx_unet* u) { return u ? 11 : 0; }
extern "C" int dmx_unet_train_bucket_range(const dmx_unet* u, int i, size_t* begin, size_t* end) {
  DMX_REQUIRE(u && begin && end && i >= 0 && i < 11, "unet_train_bucket_range: bad bucket %d", i);
  // parameters are laid out in registration order: time_embedding, conv_in, down_blocks 0..3, mid_block, up_blocks 0..3,
  // conv_norm_out, conv_out, time_emb_proj block; bucket edges = first offset of each top-level module
  auto first = [&](const char* prefix) -> size_t {
    size_t lo = (size_t)-1;
    for (const ParamEntry& e : u->pt.entries()) if (e.name.rfind(prefix, 0) == 0 && e.rule.dst < lo) lo = e.rule.dst;
    return lo;
  };
  const size_t total = u->pt.total();
  const size_t e_down[4] = {first("down_blocks.0."), first("down_blocks.1."), first("down_blocks.2."), first("down_blocks.3.")};
  const size_t e_mid = first("mid_block."), e_up[4] = {first("up_blocks.0."), first("up_blocks.1."), first("up_blocks.2."), first("up_blocks.3.")};
  const size_t e_out = first("conv_norm_out."), e_tp = u->tp_w;
  size_t lo = 0, hi = 0;
  if (i == 0) { lo = e_out; hi = e_tp; }
  else if (i >= 1 && i <= 4) { const int lvl = 4 - i; lo = e_up[lvl]; hi = lvl == 3 ? e_out : e_up[lvl + 1]; }
  else if (i == 5) { lo = e_mid; hi = e_up[0]; }
  else if (i >= 6 && i <= 9) { const int lvl = 9 - i; lo = e_down[lvl]; hi = lvl == 3 ? e_mid : e_down[lvl + 1]; }
  else { lo = 0; hi = e_down[0]; }            // bucket 10: time_embedding + conv_in ... and the time_emb_proj block below
  *begin = 2 * lo; *end = 2 * hi;
  (void)total;
  return DMX_OK;
}
// the batched time_emb_proj matrix sits behind conv_out in the arena; its gradient completes with bucket 10
extern "C" int dmx_unet_train_tail_range(const dmx_unet* u, size_t* begin, size_t* end) {
  DMX_REQUIRE(u && begin && end, "unet_train_tail_range: null argument");
  *begin = 2 * u->tp_w; *end = 2 * u->pt.total();
  return DMX_OK;
}

// Backward of the last dmx_unet_train_forward (train_diffute_v1.py:925): dpred = dLoss/dpred (fp32 NCHW); every
// parameter gradient is written (not accumulated) into `grads`.  events: optional hipEvent_t[n_events], event i is
// recorded on `stream` when bucket i of the gradient arena is complete (gradient exchange can start on another stream).
extern "C" int dmx_unet_train_backward(dmx_unet* u, void* grads, const float* dpred, void* const* events, int n_events, dmx_stream_t stream) {
  DMX_REQUIRE(u && grads && dpred, "unet_train_backward: null argument");
  auto ts = std::static_pointer_cast<TrainSession>(u->train_state);
  DMX_REQUIRE(ts && ts->forward_done, "unet_train_backward: no forward pass to differentiate");
  DMX_REQUIRE((hipStream_t)stream == ts->ex.stream, "unet_train_backward: must run on the forward's stream");
  std::vector<hipEvent_t> evs;
  for (int i = 0; i < n_events; ++i) evs.push_back((hipEvent_t)events[i]);
  ts->events = n_events > 0 ? &evs : nullptr;
  const int rc = ts->backward((char*)grads, dpred);
  ts->events = nullptr;
  u->train_state.reset();
  return rc;
}

// MSE loss (train_diffute_v1.py:918) and its gradient: loss = mean((pred - target)^2), dpred = 2 (pred - target) / n * grad_scale
extern "C" size_t dmx_mse_loss_workspace_bytes(void) { return dmx_mse_workspace_bytes(); }
extern "C" int dmx_mse_loss(const float* pred, const float* target, size_t n, float* loss, float* dpred, float grad_scale,
                            void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  return dmx_mse_loss_launch(pred, target, n, loss, dpred, grad_scale, workspace, workspace_bytes, (hipStream_t)stream);
}

// [begin, end) bytes of one parameter's gradient inside the gradient arena (end covers the row stride of packed matrices)
extern "C" int dmx_unet_grad_range(const dmx_unet* u, const char* name, size_t* begin, size_t* end) {
  DMX_REQUIRE(u && name && begin && end, "unet_grad_range: null argument");
  const ParamEntry* e = u->pt.find(name);
  DMX_REQUIRE(e != nullptr, "unet_grad_range: unknown parameter %s", name);
  const PackRule& r = e->rule;
  size_t lo = 2 * r.dst, n = 0;
  switch (r.kind) {
    case PackRule::COPY_F32: case PackRule::GEGLU_B: n = (size_t)r.rows * 4; break;
    case PackRule::LINEAR: case PackRule::GEGLU_W: n = ((size_t)(r.rows - 1) * r.ld + r.cols) * 4; break;
    case PackRule::CONV: lo += (size_t)r.koff * 4; n = ((size_t)(r.rows - 1) * r.ld + (size_t)r.ks * r.ks * r.cols) * 4; break;
  }
  *begin = lo; *end = lo + n;
  return DMX_OK;
}

// Gradient (or master copy) of one parameter of a packed fp32 arena, in its torch layout; shared with vae_train.hip.
int dmx_param_grad_export(const ParamTable& pt, const void* grads, const char* name, float* dst, hipStream_t stream) {
  const ParamEntry* e = pt.find(name);
  DMX_REQUIRE(e != nullptr, "grad_export: unknown parameter %s", name);
  const PackRule& r = e->rule;
  const float* g = (const float*)((const char*)grads + 2 * r.dst);
  int kind = 0, rows = r.rows, cols = r.cols;
  switch (r.kind) {
    case PackRule::COPY_F32: kind = 0; cols = 0; break;
    case PackRule::CONV: kind = 1; break;
    case PackRule::LINEAR: kind = 2; break;
    case PackRule::GEGLU_W: kind = 3; break;
    case PackRule::GEGLU_B: kind = 4; cols = 0; break;
  }
  const size_t total = (kind == 1) ? (size_t)rows * cols * r.ks * r.ks : (size_t)rows * (cols > 0 ? cols : 1);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_grad_unpack_kernel, dim3(blocks), dim3(256), 0, stream, g, dst, kind, rows, cols, r.ks, r.ld, r.koff);
  return dmx_check_launch("dmx_grad_unpack_kernel");
}

// Gradient of one parameter (diffusers state-dict key) in its torch layout, fp32.
extern "C" int dmx_unet_grad_export(const dmx_unet* u, const void* grads, const char* name, float* dst, dmx_stream_t stream) {
  DMX_REQUIRE(u && grads && name && dst, "unet_grad_export: null argument");
  return dmx_param_grad_export(u->pt, grads, name, dst, (hipStream_t)stream);
}
