// Training building blocks shared by the UNet and VAE training graphs (unet_train.hip, vae_train.hip): forward ops that
// keep what their backward needs, the backward helpers (dW through the wgrad kernel, dX through the forward GEMM with
// transposed weights, column sums, GroupNorm / LayerNorm backward) and the ResnetBlock2D forward / backward pair.
// Weight gradients go to the gradient arena at byte offset 2 * (offset of the weight in the weights arena).
#pragma once
#include "exec.h"

struct ResSave { Tn x0, x1; bool has1 = false; float* st1 = nullptr; Tn a1, h1; float* st2 = nullptr; Tn a2; };

struct TrainOps {
  Exec& ex; const char* arena; char* wt; char* gr; int groups; int B;
  float* tproj = nullptr; float* dtproj = nullptr; int tproj_total = 0;     // time-embedding projections (UNet only)
  float res_eps = 1e-5f;                                                      // GroupNorm eps of the resnets (VAE: 1e-6)

  TrainOps(Exec& ex_, const char* arena_, char* wt_, char* gr_, int groups_, int B_) : ex(ex_), arena(arena_), wt(wt_), gr(gr_), groups(groups_), B(B_) {}
  const float* F(size_t off) const { return (const float*)(arena + off); }
  const bf16* W(size_t off) const { return (const bf16*)(arena + off); }
  const bf16* WT(size_t off) const { return (const bf16*)(wt + off); }
  float* G(size_t off) const { return (float*)(gr + 2 * off); }
  bool live() const { return !ex.dry && !ex.rc; }

  // ------------------------------------------------------------------ forward ops that keep state
  Tn gn(const Tn& x0, const Tn* x1, size_t g, size_t b, float eps, bool silu, float** stats) {
    const int C = x0.C + (x1 ? x1->C : 0), G_ = groups;
    Tn y = ex.make(x0.B, x0.H, x0.W, C);
    *stats = (float*)ex.raw((size_t)x0.B * G_ * 2 * sizeof(float));
    void* part = ex.raw(dmx_gn_workspace_bytes(x0.B, x0.H * x0.W, G_));
    if (live()) {
      GroupNormArgs a{};
      a.x0 = x0.p; a.ldx0 = x0.ld; a.c0 = x0.C; a.x1 = x1 ? x1->p : nullptr; a.ldx1 = x1 ? x1->ld : 0;
      a.C = C; a.groups = G_; a.B = x0.B; a.HW = x0.H * x0.W; a.gamma = F(g); a.beta = F(b); a.eps = eps; a.silu = silu ? 1 : 0;
      a.y = y.p; a.ldy = y.ld; a.partial = (float*)part; a.stats_out = *stats;
      ex.rc = dmx_groupnorm_launch(a, ex.stream);
    }
    ex.drop(part);
    return y;
  }
  void attn(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* v, int ldv, int kv_rows, Tn& o, float** lse,
            int H, int Sq, int Skv) {
    *lse = (float*)ex.raw((size_t)o.B * H * Sq * sizeof(float));
    if (live()) {
      AttnArgs a{};
      a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_rows = kv_rows; a.o = o.p; a.ldo = o.ld;
      a.B = o.B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = 0.125f; a.lse = *lse;
      ex.rc = dmx_attention_launch(a, ex.stream);
    }
  }

  // ------------------------------------------------------------------ backward helpers
  // dW (+)= dY^T X for the conv / linear whose activation operand was (x0 | x1)
  void wgrad(const Tn& dy, const Tn& x0, const Tn* x1, int ksize, int stride, int ups, float* out, int ldout, int pad = -1) {
    WgradArgs a{};
    a.dy = dy.p; a.lddy = dy.ld;
    a.x0 = x0.p; a.ldx0 = x0.ld; a.cx0 = x0.C; a.x1 = x1 ? x1->p : x0.p; a.ldx1 = x1 ? x1->ld : x0.ld;
    a.Cin = x0.C + (x1 ? x1->C : 0);
    a.direct = (ksize == 1) ? 1 : 0; a.ksize = ksize; a.stride = stride; a.pad = pad >= 0 ? pad : ksize / 2; a.ups = ups;
    a.IH = x0.H; a.IW = x0.W; a.OH = dy.H; a.OW = dy.W;
    a.M = dy.rows(); a.N = dy.C; a.K = ksize * ksize * a.Cin; a.out = out; a.ldout = ldout; a.accumulate = 0;
    const size_t wsb = dmx_wgrad_workspace_bytes(a);
    void* ws = wsb ? ex.raw(wsb) : nullptr;
    if (live()) ex.rc = dmx_wgrad_launch(a, ws, wsb, ex.stream);
    if (ws) ex.drop(ws);
  }
  void colsum(const Tn& dy, int groups, float* out, int ldo) {
    const int rpg = dy.rows() / groups;
    const size_t wsb = dmx_colsum_ws_bytes(groups, rpg, dy.C);
    void* ws = ex.raw(wsb);
    if (live()) ex.rc = dmx_colsum_launch(dy.p, dy.ld, groups, rpg, dy.C, out, ldo, 0, ws, wsb, ex.stream);
    ex.drop(ws);
  }
  Tn add(const Tn& a, const Tn& b) {
    Tn o = ex.make(a.B, a.H, a.W, a.C);
    if (live()) ex.rc = dmx_add_bf16_launch(a.p, a.ld, b.p, b.ld, o.p, o.ld, a.rows(), a.C, ex.stream);
    return o;
  }
  Tn gn_bwd(const Tn& x0, const Tn* x1, size_t g, size_t b, bool silu, const float* stats, const Tn& dy,
            const Tn* res0, const Tn* res1, Tn* dx1_out) {
    const int C = x0.C + (x1 ? x1->C : 0), G_ = groups;
    Tn dx0 = ex.make(x0.B, x0.H, x0.W, x0.C);
    Tn dx1; if (x1) dx1 = ex.make(x1->B, x1->H, x1->W, x1->C);
    void* ws = ex.raw(dmx_gn_bwd_workspace_bytes(x0.B, x0.H * x0.W, C));
    if (live()) {
      GroupNormBwdArgs a{};
      a.x0 = x0.p; a.ldx0 = x0.ld; a.c0 = x0.C; a.x1 = x1 ? x1->p : nullptr; a.ldx1 = x1 ? x1->ld : 0;
      a.C = C; a.groups = G_; a.B = x0.B; a.HW = x0.H * x0.W; a.gamma = F(g); a.beta = F(b); a.silu = silu ? 1 : 0; a.stats = stats;
      a.dy = dy.p; a.lddy = dy.ld; a.dx0 = dx0.p; a.lddx0 = dx0.ld; a.dx1 = x1 ? dx1.p : nullptr; a.lddx1 = x1 ? dx1.ld : 0;
      if (res0) { a.res0 = res0->p; a.ldres0 = res0->ld; }
      if (res1) { a.res1 = res1->p; a.ldres1 = res1->ld; }
      a.dgamma = G(g); a.dbeta = G(b); a.accumulate = 0; a.part = (float*)ws;
      ex.rc = dmx_groupnorm_bwd_launch(a, ex.stream);
    }
    ex.drop(ws);
    if (dx1_out) *dx1_out = dx1;
    return dx0;
  }
  Tn ln_bwd(const Tn& x, size_t g, size_t b, const Tn& dy, const Tn* res) {
    Tn dx = ex.make(x.B, x.H, x.W, x.C);
    const size_t wsb = dmx_ln_bwd_workspace_bytes(x.rows(), x.C);
    void* ws = ex.raw(wsb);
    if (live())
      ex.rc = dmx_layernorm_bwd_launch(x.p, x.ld, dy.p, dy.ld, F(g), dx.p, dx.ld, res ? res->p : nullptr, res ? res->ld : 0,
                                       G(g), G(b), 0, x.rows(), x.C, 1e-5f, ws, wsb, ex.stream);
    ex.drop(ws);
    return dx;
  }
  // backward of y = x W^T + b (+res): returns dX (+ gres if given); dW, db into the gradient arena
  Tn linear_bwd(const Tn& x, const Tn& dy, size_t w, size_t bias, bool has_bias, const Tn* gres, bool need_dx = true) {
    wgrad(dy, x, nullptr, 1, 1, 0, G(w), x.C);
    if (has_bias) colsum(dy, 1, G(bias), dy.C);
    if (!need_dx) return Tn();
    return ex.linear(dy, WT(w), x.C, nullptr, gres, false);
  }

  // ------------------------------------------------------------------ ResnetBlock2D
  Tn res_fwd(const ResW& r, const Tn& x0, const Tn* x1, ResSave& s) {
    s.x0 = x0; s.has1 = x1 != nullptr; if (x1) s.x1 = *x1;
    s.a1 = gn(x0, x1, r.n1g, r.n1b, res_eps, true, &s.st1);
    ConvOpts o1; o1.bias = F(r.b1);
    if (tproj && r.temb_off >= 0) { o1.rowbias = tproj + r.temb_off; o1.ldrb = tproj_total; }
    s.h1 = ex.conv(s.a1, nullptr, W(r.w1), r.cout, o1);
    s.a2 = gn(s.h1, nullptr, r.n2g, r.n2b, res_eps, true, &s.st2);
    ConvOpts o2; o2.bias = F(r.b2);
    if (r.shortcut) { o2.sc0 = &s.x0; o2.sc1 = x1 ? &s.x1 : nullptr; } else { o2.res = &s.x0; }
    return ex.conv(s.a2, nullptr, W(r.w2), r.cout, o2);
  }
  // dy: gradient of the block output; gx0: gradient x0 already received from another consumer (or null).
  // Returns the total gradient of x0 (and of x1 through dx1).
  Tn res_bwd(const ResW& r, ResSave& s, const Tn& dy, const Tn* gx0, Tn* dx1) {
    const int k2 = 9 * r.cout + (r.shortcut ? r.cin : 0);
    // conv2 (+ fused shortcut / residual)
    wgrad(dy, s.a2, nullptr, 3, 1, 0, G(r.w2), k2);
    colsum(dy, 1, G(r.b2raw), r.cout);
    if (r.shortcut && live()) {
      ex.rc = (int)hipMemcpyAsync(G(r.bscraw), G(r.b2raw), (size_t)r.cout * 4, hipMemcpyDeviceToDevice, ex.stream) ? DMX_ERR_HIP : 0;
    }
    ConvOpts od; Tn da2 = ex.conv(dy, nullptr, WT(r.w2), r.cout, od);
    Tn dxs; bool own_dxs = false;                    // gradient of x through the shortcut / residual path (+ gx0)
    if (r.shortcut) {
      wgrad(dy, s.x0, s.has1 ? &s.x1 : nullptr, 1, 1, 0, G(r.w2) + 9 * r.cout, k2);
      dxs = ex.linear(dy, WT(r.w2) + (size_t)9 * r.cout * r.cout, r.cin, nullptr, (gx0 && !s.has1) ? gx0 : nullptr, false);
      own_dxs = true;
    } else if (gx0) { dxs = add(dy, *gx0); own_dxs = true; }
    else dxs = dy;
    // norm2 + SiLU
    Tn dh1 = gn_bwd(s.h1, nullptr, r.n2g, r.n2b, true, s.st2, da2, nullptr, nullptr, nullptr);
    ex.drop(da2);
    // conv1 (+ bias + time-embedding row bias)
    wgrad(dh1, s.a1, nullptr, 3, 1, 0, G(r.w1), 9 * r.cin);
    colsum(dh1, 1, G(r.b1), r.cout);
    if (dtproj && r.temb_off >= 0) colsum(dh1, B, dtproj + r.temb_off, tproj_total);
    Tn da1 = ex.conv(dh1, nullptr, WT(r.w1), r.cin, od);
    ex.drop(dh1);
    // norm1 + SiLU over (x0 | x1); the shortcut-path gradient is added per source
    Tn r0 = dxs, r1;
    if (s.has1) { r0.C = s.x0.C; r1 = dxs; r1.p = dxs.p + s.x0.C; r1.C = s.x1.C; }
    Tn dx0 = gn_bwd(s.x0, s.has1 ? &s.x1 : nullptr, r.n1g, r.n1b, true, s.st1, da1, &r0, s.has1 ? &r1 : nullptr, dx1);
    ex.drop(da1);
    if (own_dxs) ex.drop(dxs);
    return dx0;
  }
};
