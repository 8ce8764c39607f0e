// extern "C" operator-level entry points declared in include/diffute_hip.h.
#include "kernels.h"
#include "../../include/diffute_hip.h"

const char* dmx_get_error();

extern "C" int dmx_version(void) { return 100; }
extern "C" const char* dmx_last_error(void) { return dmx_get_error(); }
extern "C" const char* dmx_element_type(void) { return DMX_ELEM_NAME; }

static GemmArgs to_args(const dmx_gemm_desc* d) {
  GemmArgs a{};
  a.x0 = (const bf16*)d->x0; a.x1 = d->x1 ? (const bf16*)d->x1 : (const bf16*)d->x0;
  a.ldx0 = d->ldx0; a.ldx1 = d->x1 ? d->ldx1 : d->ldx0; a.cx0 = d->cx0; a.direct = d->direct;
  a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.pad = d->pad; a.ups = d->ups; a.ksize = d->ksize; a.Cin = d->Cin; a.Ktaps = d->Ktaps;
  a.s0 = (const bf16*)d->s0; a.s1 = d->s1 ? (const bf16*)d->s1 : (const bf16*)d->s0;
  a.lds0 = d->lds0; a.lds1 = d->s1 ? d->lds1 : d->lds0; a.cs0 = d->cs0;
  a.w = (const bf16*)d->w; a.ldw = d->ldw; a.M = d->M; a.N = d->N; a.K = d->K;
  a.bias = d->bias; a.rowbias = d->rowbias; a.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1; a.ldrb = d->ldrb;
  a.res = (const bf16*)d->res; a.ldres = d->ldres; a.out = d->out; a.ldo = d->ldo; a.out_f32 = d->out_f32; a.geglu = d->geglu;
  a.force_tn = d->force_tn; a.force_splitk = d->force_splitk; a.group_m = d->group_m; a.timing = d->timing; a.dbg = d->dbg; a.act = d->act;
  a.rowstats_out = d->rowstats_out; a.ln_stats = d->ln_stats; a.ln_tiles = d->ln_tiles; a.ln_c1 = d->ln_c1; a.ln_c2 = d->ln_c2;
  a.ln_C = d->ln_C; a.ln_eps = d->ln_eps;
  a.colstats = d->colstats; a.cs_rows = d->cs_rows;
  return a;
}
extern "C" int dmx_conv_gemm_colstats_ok(const dmx_gemm_desc* d) { return d && dmx_gemm_colstats_ok(to_args(d)) ? 1 : 0; }
extern "C" int dmx_conv_gemm_rowstats_tiles(const dmx_gemm_desc* d) {
  if (!d) return 0;
  GemmArgs a = to_args(d);
  if (!a.rowstats_out) a.rowstats_out = (float*)8;      // the plan depends on whether statistics are emitted
  return dmx_gemm_tiles_n(a);
}
extern "C" size_t dmx_conv_gemm_workspace_bytes(const dmx_gemm_desc* d) { return d ? dmx_gemm_workspace_bytes(to_args(d)) : 0; }
extern "C" int dmx_conv_gemm(const dmx_gemm_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(d && d->x0 && d->w && d->out, "conv_gemm: null argument");
  return dmx_gemm_launch(to_args(d), workspace, workspace_bytes, (hipStream_t)stream);
}

// nearest x2 upsample + conv3x3 as four 2x2 phase convolutions (GemmArgs.ups2)
static GemmArgs ups2x_args(const void* x, int ldx, int B, int IH, int IW, int Cin, const void* wp, int N, const float* bias,
                           void* out, int ldo, int force_tn, int force_splitk) {
  GemmArgs a{};
  a.x0 = a.x1 = (const bf16*)x; a.ldx0 = a.ldx1 = ldx; a.cx0 = Cin; a.Cin = Cin;
  a.ksize = 2; a.stride = 1; a.ups2 = 1; a.IH = a.OH = IH; a.IW = a.OW = IW;
  a.M4 = B * IH * IW; a.M = 4 * a.M4; a.N = N; a.K = a.Ktaps = 4 * Cin;
  a.w = (const bf16*)wp; a.ldw = 4 * Cin; a.w_phase_stride = (long long)N * 4 * Cin;
  a.bias = bias; a.rows_per_group = 1; a.out = out; a.ldo = ldo;
  a.force_tn = force_tn; a.force_splitk = force_splitk;
  return a;
}
extern "C" size_t dmx_conv_ups2x_workspace_bytes(int B, int IH, int IW, int Cin, int N, int force_tn, int force_splitk) {
  return dmx_gemm_workspace_bytes(ups2x_args(nullptr, Cin, B, IH, IW, Cin, nullptr, N, nullptr, nullptr, N, force_tn, force_splitk));
}
extern "C" int dmx_conv_ups2x(const void* x, int ldx, int B, int IH, int IW, int Cin, const void* phase_weights, int N, const float* bias,
                              void* out, int ldo, int force_tn, int force_splitk, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(x && phase_weights && out, "conv_ups2x: null argument");
  DMX_REQUIRE(Cin % 32 == 0 && N % 8 == 0 && ldo % 8 == 0, "conv_ups2x: Cin %% 32, N %% 8 and ldo %% 8 must be 0");
  return dmx_gemm_launch(ups2x_args(x, ldx, B, IH, IW, Cin, phase_weights, N, bias, out, ldo, force_tn, force_splitk), workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int dmx_pack_ups_phase_weights(const void* w3, int ldw3, void* phase_weights, int N, int Cin, dmx_stream_t stream) {
  DMX_REQUIRE(w3 && phase_weights, "pack_ups_phase_weights: null argument");
  return dmx_ups_phase_weights_launch((const bf16*)w3, ldw3, (bf16*)phase_weights, N, Cin, (hipStream_t)stream);
}

// tuning aid: time candidate GEMM plans inside a real pass (cfg = template instance id, < 0 clears every override)
extern "C" int dmx_gemm_plan_override(int M, int N, int K, int stride, int ups, int cfg, int splitk) {
  dmx_gemm_plan_override_set(M, N, K, stride, ups, cfg, splitk);
  return DMX_OK;
}

static WgradArgs to_wgrad(const dmx_gemm_desc* d, const void* dy, int lddy, float* dw, int accumulate) {
  WgradArgs a{};
  a.dy = (const bf16*)dy; a.lddy = lddy;
  a.x0 = (const bf16*)d->x0; a.x1 = d->x1 ? (const bf16*)d->x1 : (const bf16*)d->x0;
  a.ldx0 = d->ldx0; a.ldx1 = d->x1 ? d->ldx1 : d->ldx0; a.cx0 = d->cx0; a.direct = d->direct;
  a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.pad = d->pad; a.ups = d->ups; a.ksize = d->ksize > 0 ? d->ksize : 1; a.Cin = d->Cin;
  a.M = d->M; a.N = d->N; a.K = d->Ktaps; a.out = dw; a.accumulate = accumulate;
  return a;
}
extern "C" size_t dmx_conv_wgrad_workspace_bytes(const dmx_gemm_desc* d, int accumulate) {
  return d ? dmx_wgrad_workspace_bytes(to_wgrad(d, nullptr, 0, nullptr, accumulate)) : 0;
}
extern "C" int dmx_conv_wgrad(const dmx_gemm_desc* d, const void* dy, int lddy, float* dw, int accumulate,
                              void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(d && d->x0 && dy && dw, "conv_wgrad: null argument");
  return dmx_wgrad_launch(to_wgrad(d, dy, lddy, dw, accumulate), workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" size_t dmx_colsum_workspace_bytes(int groups, int rows_per_group, int N) { return dmx_colsum_ws_bytes(groups, rows_per_group, N); }
extern "C" int dmx_colsum(const void* dy, int lddy, int groups, int rows_per_group, int N, float* out, int ldo, int accumulate,
                          void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(dy && out, "colsum: null argument");
  return dmx_colsum_launch((const bf16*)dy, lddy, groups, rows_per_group, N, out, ldo, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int dmx_groupnorm(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                             int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                             void* y, int ldy, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && y && gamma && beta, "groupnorm: null argument");
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_gn_workspace_bytes(B, HW, groups), "groupnorm: workspace too small");
  GroupNormArgs a{};
  a.x0 = (const bf16*)x0; a.ldx0 = ldx0; a.x1 = (const bf16*)x1; a.ldx1 = ldx1; a.c0 = x1 ? c0 : C;
  a.C = C; a.groups = groups; a.B = B; a.HW = HW; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
  a.y = (bf16*)y; a.ldy = ldy; a.partial = (float*)workspace;
  return dmx_groupnorm_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_groupnorm_from_stats(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                                        int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                                        const long long* st0, const long long* st1, void* y, int ldy, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && y && gamma && beta && st0 && (!x1 || st1), "groupnorm_from_stats: null argument");
  GroupNormArgs a{};
  a.x0 = (const bf16*)x0; a.ldx0 = ldx0; a.x1 = (const bf16*)x1; a.ldx1 = ldx1; a.c0 = x1 ? c0 : C;
  a.C = C; a.groups = groups; a.B = B; a.HW = HW; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
  a.y = (bf16*)y; a.ldy = ldy; a.st0 = st0; a.st1 = x1 ? st1 : st0;
  return dmx_groupnorm_sums_launch(a, (hipStream_t)stream);
}
static HaloConvArgs halo_args(const dmx_halo_conv_desc* d) {
  HaloConvArgs a{};
  a.x0 = (const bf16*)d->x0; a.x1 = (const bf16*)d->x1; a.ldx0 = d->ldx0; a.ldx1 = d->ldx1; a.cx0 = d->x1 ? d->cx0 : d->Cin; a.Cin = d->Cin;
  a.B = d->B; a.H = d->H; a.W = d->W; a.gn = d->gn; a.silu = d->silu; a.groups = d->groups; a.eps = d->eps;
  a.st0 = d->st0; a.st1 = d->st1; a.gamma = d->gamma; a.beta = d->beta;
  a.s0 = (const bf16*)d->s0; a.s1 = (const bf16*)d->s1; a.lds0 = d->lds0; a.lds1 = d->lds1; a.cs0 = d->s1 ? d->cs0 : d->Csc; a.Csc = d->s0 ? d->Csc : 0;
  a.w = (const bf16*)d->w; a.ldw = d->ldw; a.N = d->N; a.bias = d->bias; a.rowbias = d->rowbias; a.ldrb = d->ldrb;
  a.res = (const bf16*)d->res; a.ldres = d->ldres; a.out = (bf16*)d->out; a.ldo = d->ldo; a.colstats = d->colstats;
  a.force_split = d->force_split; a.force_bn = d->force_bn; a.force_waves = d->force_waves; a.dbg = d->dbg; a.timing = d->timing;
  return a;
}
extern "C" int dmx_conv3x3_gn_supported(const dmx_halo_conv_desc* d) { return d && dmx_conv_halo_supported(halo_args(d)) ? 1 : 0; }
extern "C" size_t dmx_conv3x3_gn_workspace_bytes(const dmx_halo_conv_desc* d) { return d ? dmx_conv_halo_workspace_bytes(halo_args(d)) : 0; }
extern "C" int dmx_conv3x3_gn(const dmx_halo_conv_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(d && d->x0 && d->w && d->out, "conv3x3_gn: null argument");
  return dmx_conv_halo_launch(halo_args(d), workspace, workspace_bytes, (hipStream_t)stream);
}
static SkinnyArgs skinny_args(const dmx_skinny_desc* d) {
  SkinnyArgs a{};
  a.nseg = d->nseg;
  for (int k = 0; k < 4 && k < d->nseg; ++k) {
    a.seg[k].x = (const bf16*)d->seg[k].x; a.seg[k].ld = d->seg[k].ld; a.seg[k].C = d->seg[k].C; a.seg[k].taps = d->seg[k].taps;
    a.seg[k].st = d->seg[k].st; a.seg[k].gamma = d->seg[k].gamma; a.seg[k].beta = d->seg[k].beta; a.seg[k].gn_c0 = d->seg[k].gn_c0;
  }
  a.B = d->B; a.H = d->H; a.W = d->W; a.gn_groups = d->gn_groups; a.gn_Ctot = d->gn_Ctot; a.gn_eps = d->gn_eps; a.silu = d->silu;
  a.wp = (const bf16*)d->wp; a.N = d->N; a.bias = d->bias; a.rowbias = d->rowbias; a.ldrb = d->ldrb;
  a.res = (const bf16*)d->res; a.ldres = d->ldres; a.out = (bf16*)d->out; a.ldo = d->ldo; a.colstats = d->colstats; a.force_S = d->force_S; a.timing = d->timing; a.dbg = d->dbg;
  return a;
}
extern "C" int dmx_skinny_conv_supported(const dmx_skinny_desc* d) { return d && d->nseg >= 1 && d->nseg <= 4 && dmx_skinny_supported(skinny_args(d)) ? 1 : 0; }
extern "C" size_t dmx_skinny_conv_workspace_bytes(const dmx_skinny_desc* d) { return d && d->nseg >= 1 && d->nseg <= 4 ? dmx_skinny_workspace_bytes(skinny_args(d)) : 0; }
extern "C" int dmx_skinny_conv(const dmx_skinny_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(d && d->nseg >= 1 && d->nseg <= 4 && d->wp && d->out, "skinny_conv: null argument");
  return dmx_skinny_launch(skinny_args(d), workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int dmx_skinny_pack(const void* w, int ldw, void* wp, int N, int nseg, const int* C, const int* taps, const int* tap_stride, const int* koff, dmx_stream_t stream) {
  DMX_REQUIRE(w && wp && C && taps && tap_stride && koff && nseg >= 1 && nseg <= 4, "skinny_pack: null argument");
  SkinnyPackDesc d{}; d.nseg = nseg; int f = 0;
  for (int k = 0; k < nseg; ++k) {
    DMX_REQUIRE(C[k] > 0 && C[k] % 16 == 0 && (taps[k] == 9 || taps[k] == 1), "skinny_pack: segment %d: C=%d taps=%d", k, C[k], taps[k]);
    d.frag0[k] = f; d.taps[k] = taps[k]; d.tap_stride[k] = tap_stride[k]; d.koff[k] = koff[k]; f += (C[k] / 16) * taps[k];
  }
  d.frags_per_nb = f;
  return dmx_skinny_pack_launch((const bf16*)w, ldw, (bf16*)wp, N, d, (hipStream_t)stream);
}
extern "C" int dmx_colstats(const void* x, int ldx, int B, int HW, int C, long long* st, dmx_stream_t stream) {
  DMX_REQUIRE(x && st, "colstats: null argument");
  return dmx_colstats_launch((const bf16*)x, ldx, B, HW, C, st, (hipStream_t)stream);
}
extern "C" int dmx_xf_chain_ok(int M, int C) { return dmx_xf_chain_supported(M, C) ? 1 : 0; }
extern "C" int dmx_xf_chain(const dmx_xf_chain_desc* d, int mode, dmx_stream_t stream) {
  DMX_REQUIRE(d != nullptr, "xf_chain: null descriptor");
  XfChainArgs a{};
  a.M = d->M; a.C = d->C; a.x = (const bf16*)d->x; a.ldx = d->ldx; a.res = (const bf16*)d->res; a.ldres = d->ldres;
  a.w0 = (const bf16*)d->w0; a.b0 = d->b0; a.h_out = (bf16*)d->h_out; a.ldh = d->ldh; a.w1 = (const bf16*)d->w1;
  a.c1 = d->c1; a.c2 = d->c2; a.y = (bf16*)d->y; a.ldy = d->ldy; a.wf1 = (const bf16*)d->wf1; a.wf2 = (const bf16*)d->wf2; a.bf2 = d->bf2;
  a.wpo = (const bf16*)d->wpo; a.bpo = d->bpo; a.xres = (const bf16*)d->xres; a.ldxres = d->ldxres; a.eps = d->eps; a.dbg = d->dbg; a.timing = d->timing;
  a.colstats = d->colstats; a.cs_rows = d->cs_rows;
  a.gn_st = d->gn_st; a.gn_gamma = d->gn_gamma; a.gn_beta = d->gn_beta; a.gn_groups = d->gn_groups; a.gn_rows = d->gn_rows; a.gn_eps = d->gn_eps;
  return dmx_xf_chain_launch(a, mode, (hipStream_t)stream);
}
extern "C" int dmx_groupnorm_train(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                                   int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                                   void* y, int ldy, float* stats, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && y && gamma && beta && stats, "groupnorm_train: null argument");
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_gn_workspace_bytes(B, HW, groups), "groupnorm_train: workspace too small");
  GroupNormArgs a{};
  a.x0 = (const bf16*)x0; a.ldx0 = ldx0; a.x1 = (const bf16*)x1; a.ldx1 = ldx1; a.c0 = x1 ? c0 : C;
  a.C = C; a.groups = groups; a.B = B; a.HW = HW; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
  a.y = (bf16*)y; a.ldy = ldy; a.partial = (float*)workspace; a.stats_out = stats;
  return dmx_groupnorm_launch(a, (hipStream_t)stream);
}
extern "C" size_t dmx_groupnorm_bwd_workspace_bytes(int B, int HW, int C) { return dmx_gn_bwd_workspace_bytes(B, HW, C); }
extern "C" int dmx_groupnorm_bwd(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups, int B, int HW,
                                 const float* gamma, const float* beta, int silu, const float* stats,
                                 const void* dy, int lddy, void* dx0, int lddx0, void* dx1, int lddx1,
                                 const void* res0, int ldres0, const void* res1, int ldres1,
                                 float* dgamma, float* dbeta, int accumulate,
                                 void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && dy && dx0 && gamma && beta && stats, "groupnorm_bwd: null argument");
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_gn_bwd_workspace_bytes(B, HW, C), "groupnorm_bwd: workspace too small");
  GroupNormBwdArgs a{};
  a.x0 = (const bf16*)x0; a.ldx0 = ldx0; a.x1 = (const bf16*)x1; a.ldx1 = ldx1; a.c0 = x1 ? c0 : C;
  a.C = C; a.groups = groups; a.B = B; a.HW = HW; a.gamma = gamma; a.beta = beta; a.silu = silu; a.stats = stats;
  a.dy = (const bf16*)dy; a.lddy = lddy; a.dx0 = (bf16*)dx0; a.lddx0 = lddx0; a.dx1 = (bf16*)dx1; a.lddx1 = lddx1;
  a.res0 = (const bf16*)res0; a.ldres0 = ldres0; a.res1 = (const bf16*)res1; a.ldres1 = ldres1;
  a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate; a.part = (float*)workspace;
  return dmx_groupnorm_bwd_launch(a, (hipStream_t)stream);
}
extern "C" size_t dmx_layernorm_bwd_workspace_bytes(int rows, int C) { return dmx_ln_bwd_workspace_bytes(rows, C); }
extern "C" int dmx_layernorm_bwd(const void* x, int ldx, const void* dy, int lddy, const float* gamma, void* dx, int lddx,
                                 const void* res, int ldres, float* dgamma, float* dbeta, int accumulate,
                                 int rows, int C, float eps, void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(x && dy && gamma && dx, "layernorm_bwd: null argument");
  return dmx_layernorm_bwd_launch((const bf16*)x, ldx, (const bf16*)dy, lddy, gamma, (bf16*)dx, lddx, (const bf16*)res, ldres,
                                  dgamma, dbeta, accumulate, rows, C, eps, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int dmx_geglu_fwd(const void* h, int ldh, void* y, int ldy, int rows, int C2, dmx_stream_t stream) {
  DMX_REQUIRE(h && y, "geglu_fwd: null argument");
  return dmx_geglu_fwd_launch((const bf16*)h, ldh, (bf16*)y, ldy, rows, C2, 0, (hipStream_t)stream);
}
extern "C" int dmx_geglu_bwd(const void* h, int ldh, const void* dy, int lddy, void* dh, int lddh, int rows, int C2, dmx_stream_t stream) {
  DMX_REQUIRE(h && dy && dh, "geglu_bwd: null argument");
  return dmx_geglu_bwd_launch((const bf16*)h, ldh, (const bf16*)dy, lddy, (bf16*)dh, lddh, rows, C2, 0, (hipStream_t)stream);
}
extern "C" int dmx_layernorm(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                             int rows, int C, float eps, dmx_stream_t stream) {
  DMX_REQUIRE(x && y && gamma && beta, "layernorm: null argument");
  return dmx_layernorm_launch((const bf16*)x, ldx, (bf16*)y, ldy, gamma, beta, rows, C, eps, (hipStream_t)stream);
}
extern "C" int dmx_attention_fwd(const void* q, int ldq, const void* k, int ldk, int kv_rows, const void* vt, int ldvt, int skv_stride,
                                 void* o, int ldo, int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream) {
  DMX_REQUIRE(q && k && vt && o, "attention: null argument");
  AttnArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.kv_rows = kv_rows;
  a.vt = (const bf16*)vt; a.ldvt = ldvt; a.skv_stride = skv_stride; a.o = (bf16*)o; a.ldo = ldo;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  return dmx_attention_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_attention_wide(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                  void* o, int ldo, int B, int Sq, int Skv, int D, float scale, dmx_stream_t stream) {
  DMX_REQUIRE(q && k && v && o, "attention_wide: null argument");
  AttnWideArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.v = (const bf16*)v; a.ldv = ldv; a.kv_rows = kv_rows;
  a.o = (bf16*)o; a.ldo = ldo; a.B = B; a.Sq = Sq; a.Skv = Skv; a.D = D; a.scale = scale;
  return dmx_attention_wide_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_attention_fwd_v(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                   void* o, int ldo, int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream) {
  DMX_REQUIRE(q && k && v && o, "attention: null argument");
  AttnArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.kv_rows = kv_rows;
  a.v = (const bf16*)v; a.ldv = ldv; a.o = (bf16*)o; a.ldo = ldo;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  return dmx_attention_launch(a, (hipStream_t)stream);
}
// K5 with the balanced schedule (attention_sk.hip): workspace = dmx_attention_fwd_v_balanced_workspace_bytes (0: the plan keeps the plain grid for this
// problem - dmx_set_attn_balanced(2) plans it wherever the kernel takes the problem); flags zeroed here
static size_t attn_bal_flag_bytes(int slots) { return align_up((size_t)slots * sizeof(int), 256); }
extern "C" size_t dmx_attention_fwd_v_balanced_workspace_bytes(int B, int H, int Sq, int Skv) {
  AttnArgs a{};
  a.v = (const bf16*)8; a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv;
  const int ns = dmx_attention_balanced_slots(a);
  return ns ? attn_bal_flag_bytes(ns) + dmx_attention_balanced_part_bytes(a) : 0;
}
extern "C" int dmx_attention_fwd_v_balanced(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                            void* o, int ldo, int B, int H, int Sq, int Skv, float scale,
                                            void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(q && k && v && o && workspace, "attention (balanced): null argument");
  AttnArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.kv_rows = kv_rows;
  a.v = (const bf16*)v; a.ldv = ldv; a.o = (bf16*)o; a.ldo = ldo;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  const int ns = dmx_attention_balanced_slots(a);
  DMX_REQUIRE(ns > 0, "attention (balanced): the plan keeps the plain grid for B=%d H=%d Sq=%d Skv=%d (dmx_attention_fwd_v)", B, H, Sq, Skv);
  const size_t fb = attn_bal_flag_bytes(ns), need = fb + dmx_attention_balanced_part_bytes(a);
  if (workspace_bytes < need) { dmx_set_error("attention (balanced): workspace %zu < %zu bytes", workspace_bytes, need); return DMX_ERR_WORKSPACE; }
  a.sk_flags = (int*)workspace; a.sk_part = (float*)((char*)workspace + fb);
  if (const int zr = dmx_zero16_launch(a.sk_flags, fb, (hipStream_t)stream)) return zr;
  return dmx_attention_balanced_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_attention_fwd_train(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                       void* o, int ldo, float* lse, int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream) {
  DMX_REQUIRE(q && k && v && o && lse, "attention_fwd_train: null argument");
  AttnArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.kv_rows = kv_rows;
  a.v = (const bf16*)v; a.ldv = ldv; a.o = (bf16*)o; a.ldo = ldo; a.lse = lse;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  return dmx_attention_launch(a, (hipStream_t)stream);
}
extern "C" size_t dmx_attention_bwd_workspace_bytes(int B, int H, int Sq) { return dmx_attn_bwd_ws_bytes(B, H, Sq); }
extern "C" int dmx_attention_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                 const void* o, const void* d_o, int ldo, const float* lse,
                                 void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                                 int B, int H, int Sq, int Skv, float scale,
                                 void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_attn_bwd_ws_bytes(B, H, Sq), "attention_bwd: workspace too small");
  AttnBwdArgs a{};
  a.q = (const bf16*)q; a.ldq = ldq; a.k = (const bf16*)k; a.ldk = ldk; a.v = (const bf16*)v; a.ldv = ldv; a.kv_rows = kv_rows;
  a.o = (const bf16*)o; a.dout = (const bf16*)d_o; a.ldo = ldo; a.lse = lse; a.delta = (float*)workspace;
  a.dq = (bf16*)dq; a.lddq = lddq; a.dk = (bf16*)dk; a.lddk = lddk; a.dv = (bf16*)dv; a.lddv = lddv;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  return dmx_attention_bwd_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_timestep_embedding(const int64_t* t, int t_count, const float* freq, int B, int dim, float* out, dmx_stream_t stream) {
  DMX_REQUIRE(t && freq && out, "timestep_embedding: null argument");
  return dmx_timestep_embedding_launch((const long long*)t, t_count, freq, B, dim, out, (hipStream_t)stream);
}
extern "C" int dmx_linear_small(const float* x, int ldx, const void* w, int ldw, const float* bias, float* y, int ldy,
                                int B, int N, int K, int silu_in, dmx_stream_t stream) {
  DMX_REQUIRE(x && w && y, "linear_small: null argument");
  return dmx_linear_small_launch(x, ldx, (const bf16*)w, ldw, bias, y, ldy, B, N, K, silu_in, (hipStream_t)stream);
}
extern "C" int dmx_im2col_small(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                                const void* h_nhwc, int ldh, int C, int B, int IH, int IW, int OH, int OW,
                                int ksize, int stride, int pad, void* out, int Kpad, dmx_stream_t stream) {
  DMX_REQUIRE(out != nullptr, "im2col_small: null output");
  Im2colArgs a{};
  a.f0 = f0; a.c0 = c0; a.f1 = f1; a.c1 = c1; a.f2 = f2; a.c2 = c2; a.h = (const bf16*)h_nhwc; a.ldh = ldh; a.C = C;
  a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.ksize = ksize; a.stride = stride; a.pad = pad; a.out = (bf16*)out; a.Kpad = Kpad;
  return dmx_im2col_small_launch(a, (hipStream_t)stream);
}
extern "C" int dmx_pack_conv_weight(const float* w, void* out, int Cout, int Cin, int ksize, int ldk, int koff, dmx_stream_t stream) {
  DMX_REQUIRE(w && out, "pack_conv_weight: null argument");
  return dmx_pack_conv_weight_launch(w, (bf16*)out, Cout, Cin, ksize, ldk, koff, (hipStream_t)stream);
}
extern "C" int dmx_pack_linear_weight(const float* w, void* out, int rows, int cols, int ldo, int geglu, dmx_stream_t stream) {
  DMX_REQUIRE(w && out, "pack_linear_weight: null argument");
  return dmx_pack_rows_launch(w, (bf16*)out, rows, cols, ldo, geglu, (hipStream_t)stream);
}
extern "C" int dmx_pack_conv_weight_t(const float* w, void* out, int Cout, int Cin, int ksize, int ldk, int koff, dmx_stream_t stream) {
  DMX_REQUIRE(w && out, "pack_conv_weight_t: null argument");
  return dmx_pack_conv_weight_t_launch(w, (bf16*)out, Cout, Cin, ksize, ldk, koff, (hipStream_t)stream);
}
extern "C" int dmx_pack_linear_weight_t(const float* w, void* out, int rows, int cols, int ldo, dmx_stream_t stream) {
  DMX_REQUIRE(w && out, "pack_linear_weight_t: null argument");
  return dmx_pack_rows_t_launch(w, (bf16*)out, rows, cols, ldo, (hipStream_t)stream);
}
extern "C" int dmx_zero_insert2(const void* dy, int lddy, void* z, int B, int OH, int OW, int C, dmx_stream_t stream) {
  DMX_REQUIRE(dy && z, "zero_insert2: null argument");
  return dmx_zero_insert2_launch((const bf16*)dy, lddy, (bf16*)z, B, OH, OW, C, (hipStream_t)stream);
}
extern "C" int dmx_sumpool2(const void* du, int lddu, int du_f32, void* dx, int lddx, int B, int H, int W, int C, int accumulate, dmx_stream_t stream) {
  DMX_REQUIRE(du && dx, "sumpool2: null argument");
  return dmx_sumpool2_launch(du, lddu, du_f32, (bf16*)dx, lddx, B, H, W, C, accumulate, (hipStream_t)stream);
}
extern "C" int dmx_pack_geglu_bias(const float* b, float* out, int n, dmx_stream_t stream) {
  DMX_REQUIRE(b && out, "pack_geglu_bias: null argument");
  return dmx_pack_geglu_bias_launch(b, out, n, (hipStream_t)stream);
}
extern "C" int dmx_cast_f32_to_bf16(const float* in, void* out, size_t n, dmx_stream_t stream) {
  DMX_REQUIRE(in && out, "cast: null argument");
  return dmx_cast_f32_to_bf16_launch(in, (bf16*)out, n, (hipStream_t)stream);
}
extern "C" int dmx_nhwc_bf16_to_nchw_f32(const void* in, int ldin, float* out, int B, int C, int HW, dmx_stream_t stream) {
  DMX_REQUIRE(in && out, "nhwc_bf16_to_nchw_f32: null argument");
  return dmx_nhwc_bf16_to_nchw_f32_launch((const bf16*)in, ldin, out, B, C, HW, (hipStream_t)stream);
}
extern "C" int dmx_nhwc_f32_to_nchw_f32(const float* in, int ldin, float* out, int B, int C, int HW, dmx_stream_t stream) {
  DMX_REQUIRE(in && out, "nhwc_f32_to_nchw_f32: null argument");
  return dmx_nhwc_to_nchw_f32_launch(in, ldin, out, B, C, HW, (hipStream_t)stream);
}
extern "C" int dmx_nchw_f32_to_nhwc_bf16(const float* in, void* out, int ldo, int B, int C, int HW, dmx_stream_t stream) {
  DMX_REQUIRE(in && out, "nchw_f32_to_nhwc_bf16: null argument");
  return dmx_nchw_f32_to_nhwc_bf16_launch(in, (bf16*)out, ldo, B, C, HW, (hipStream_t)stream);
}
extern "C" int dmx_sched_step_ddim(const float* sample, const float* model_output, const float* noise, float* prev_sample, size_t n,
                                   float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float sqrt_alpha_prod_prev,
                                   float dir_coef, float std_dev, int v_prediction, dmx_stream_t stream) {
  DMX_REQUIRE(sample && model_output && prev_sample, "sched_step_ddim: null argument");
  return dmx_sched_ddim_launch(sample, model_output, noise, prev_sample, n, sqrt_beta_prod_t, sqrt_alpha_prod_t, sqrt_alpha_prod_prev,
                               dir_coef, std_dev, v_prediction, (hipStream_t)stream);
}
extern "C" int dmx_sched_step_ddpm(const float* sample, const float* model_output, const float* noise, float* prev_sample, size_t n,
                                   float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float coef_x0, float coef_xt,
                                   float sigma, int v_prediction, dmx_stream_t stream) {
  DMX_REQUIRE(sample && model_output && prev_sample, "sched_step_ddpm: null argument");
  return dmx_sched_ddpm_launch(sample, model_output, noise, prev_sample, n, sqrt_beta_prod_t, sqrt_alpha_prod_t, coef_x0, coef_xt,
                               sigma, v_prediction, (hipStream_t)stream);
}
extern "C" int dmx_sched_add_noise(const float* x0, const float* noise, const float* sa, const float* sb, float* out, int B, size_t per, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && noise && sa && sb && out, "sched_add_noise: null argument");
  return dmx_add_noise_launch(x0, noise, sa, sb, out, B, per, 0, (hipStream_t)stream);
}
extern "C" int dmx_sched_get_velocity(const float* x0, const float* noise, const float* sa, const float* sb, float* out, int B, size_t per, dmx_stream_t stream) {
  DMX_REQUIRE(x0 && noise && sa && sb && out, "sched_get_velocity: null argument");
  return dmx_add_noise_launch(x0, noise, sa, sb, out, B, per, 1, (hipStream_t)stream);
}
extern "C" int dmx_gaussian_sample(const float* moments, const float* noise, float* out, int B, int C, int HW, float scale, dmx_stream_t stream) {
  DMX_REQUIRE(moments && out, "gaussian_sample: null argument");
  return dmx_gaussian_sample_launch(moments, noise, out, B, C, HW, scale, (hipStream_t)stream);
}
extern "C" size_t dmx_groupnorm_workspace_bytes(int B, int HW, int groups) { return dmx_gn_workspace_bytes(B, HW, groups); }

// ---- test support (tests/test_ops_gpu.py: the device -> host error channel).  dmx_test_raise_device_error launches one thread that raises
// `code` like a kernel that gave up on a wait; dmx_test_occupy_cus keeps `blocks` CUs busy for `ticks` x 10 ns with a block that fills the CU's
// LDS (nothing else becomes resident next to it) - the "another stream holds the CUs" situation in which the peers of a K-split tile are not
// co-resident.
__global__ void dmx_test_raise_kernel(int* err, int code, int d0, int d1, int d2) { dmx_dev_raise(err, code, (int)blockIdx.x, d0, d1, d2); }
extern "C" int dmx_test_raise_device_error(int code, dmx_stream_t stream) {
  int* e = dmx_dev_err_words();
  DMX_REQUIRE(e != nullptr, "test_raise_device_error: no pinned error words");
  hipLaunchKernelGGL(dmx_test_raise_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, e, code, 11, 22, 33);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { dmx_set_error("launch of dmx_test_raise_kernel failed: %s", hipGetErrorString(le)); return DMX_ERR_HIP; }
  return DMX_OK;
}
__global__ __launch_bounds__(64) void dmx_test_occupy_kernel(long long ticks, int* sink) {
  extern __shared__ char hog[];
  hog[threadIdx.x] = (char)threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (hog[(threadIdx.x + 1) & 63] == 77 && sink) *sink = 1;
}
extern "C" int dmx_test_occupy_cus(int blocks, long long ticks, dmx_stream_t stream) {
  constexpr int LDS = 160 * 1024;
  DMX_REQUIRE(blocks > 0 && ticks > 0 && ticks <= 50000000, "test_occupy_cus: blocks > 0, 0 < ticks <= 5e7 (0.5 s)");
  DMX_LDS_OPT_IN(dmx_test_occupy_kernel, LDS);
  hipLaunchKernelGGL(dmx_test_occupy_kernel, dim3(blocks), dim3(64), LDS, (hipStream_t)stream, ticks, (int*)nullptr);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { dmx_set_error("launch of dmx_test_occupy_kernel failed: %s", hipGetErrorString(le)); return DMX_ERR_HIP; }
  return DMX_OK;
}
