// Training graph of the AutoencoderKL (reference train_vae.py:716-736:
// `pred = vae(x)["sample"]` = decode(encode(x).latent_dist.mode()), `loss = F.mse_loss(pred, target)`, backward).
// Same construction as the UNet training graph (unet_train.hip): a forward that keeps what the backward reads, then the
// backward through the shared building blocks of train_common.h.  VAE specifics:
//   - GroupNorm eps 1e-6, resnets without time embedding;
//   - encoder downsampling conv = pad (0,1,0,1) + stride 2: dW through the wgrad gather with pad 0 / stride 2, dX = conv of
//     the zero-inserted dY with the flipped filter and pad 2;
//   - the channel-starved ends (3, 4, 8 channels): conv_in layers keep their im2col matrix (dW = plain TN GEMM on it);
//     conv_out layers get their data gradient through an im2col of dY and a transposed filter padded to K = 64 / 128;
//     quant_conv / post_quant_conv (1x1 between <= 8 channels) run as per-pixel kernels forward and backward;
//   - single-head d = C mid attention: P (bf16, [S][S] per image) is kept; dV = P^T dA and dK = dS^T Q through the wgrad
//     kernel, dP = dA V^T and dQ = dS K through the forward GEMM, dS by a row kernel.
// Gradients use the packed fp32 arena convention (byte offset 2 * arena offset).
#include <math.h>
#include "vae_model.h"
#include "train_common.h"

int dmx_param_grad_export(const ParamTable& pt, const void* grads, const char* name, float* dst, hipStream_t stream);   // unet_train.hip

namespace {

struct AttnSave { Tn x, n, q, k, v, a; float* stg = nullptr; bf16* P = nullptr; };
struct ConvSave { Tn x; };

__global__ __launch_bounds__(256) void dmx_slice_cast_kernel(const float* in, int ldin, bf16* out, int ldo, int M, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int m = i / C, c = i - m * C;
  ((unsigned short*)out)[(size_t)m * ldo + c] = f2bf_bits(in[(size_t)m * ldin + c]);
}
// dmom[m][0:C] = bf16 dz[m][0:C] ; dmom[m][C:2C] = 0   (the mode of the posterior ignores the log-variance half)
__global__ __launch_bounds__(256) void dmx_mode_bwd_kernel(const bf16* dz, int lddz, float* dmom, int M, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * 2 * C) return;
  const int m = i / (2 * C), c = i - m * 2 * C;
  dmom[i] = c < C ? bf_bits2f(((const unsigned short*)dz)[(size_t)m * lddz + c]) : 0.f;
}
__global__ __launch_bounds__(256) void dmx_bf16_to_f32_rows_kernel(const bf16* in, int ldin, float* out, int M, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int m = i / C, c = i - m * C;
  out[i] = bf_bits2f(((const unsigned short*)in)[(size_t)m * ldin + c]);
}

size_t wt_extra_eout(const dmx_vae* v) { return align_up(v->pt.total(), 256); }                       // [C_last][128]
size_t wt_extra_dout(const dmx_vae* v) { return wt_extra_eout(v) + align_up((size_t)v->cfg.block_out_channels[3] * 128 * 2, 256); }   // [C_0][64]
size_t wt_total(const dmx_vae* v) { return wt_extra_dout(v) + (size_t)v->cfg.block_out_channels[0] * 64 * 2; }

struct VaeTrain {
  dmx_vae* v; Exec ex; TrainOps T;
  int B = 0, H = 0, W = 0; bool forward_done = false;
  // saved
  Tn col_in; std::vector<ResSave> e_res[4]; ConvSave e_ds[4]; ResSave e_mid[2]; AttnSave e_attn;
  Tn e_hlast, e_tout, m8, z, z2, dcol; float* e_st = nullptr;
  ResSave d_mid[2]; AttnSave d_attn; std::vector<ResSave> d_res[4]; ConvSave d_us[4];
  Tn d_hlast, d_tout; float* d_st = nullptr;

  VaeTrain(dmx_vae* v_, char* wt, int B_) : v(v_), T(ex, v_->arena, wt, nullptr, v_->cfg.norm_num_groups, B_) { T.res_eps = 1e-6f; }
  bool live() const { return !ex.dry && !ex.rc; }

  // ---------------------------------------------------------------- single-head attention, d = C
  Tn attn_fwd(const AttnW& w, const Tn& x, AttnSave& s) {
    const int C = w.C, S = x.H * x.W, Bn = x.B;
    s.x = x;
    s.n = T.gn(x, nullptr, w.gg, w.gb, 1e-6f, false, &s.stg);
    s.q = ex.linear(s.n, T.W(w.wq), C, T.F(w.bq), nullptr, false);
    s.k = ex.linear(s.n, T.W(w.wk), C, T.F(w.bk), nullptr, false);
    s.v = ex.linear(s.n, T.W(w.wv), C, T.F(w.bv), nullptr, false);
    s.a = ex.make(x.B, x.H, x.W, C);
    s.P = (bf16*)ex.raw((size_t)Bn * S * S * 2);
    bf16* vt = (bf16*)ex.raw((size_t)C * S * 2);
    float* sc = (float*)ex.raw((size_t)S * S * 4);
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < Bn; ++b) {
      bf16* Pb = s.P + (size_t)b * S * S;
      if (live()) ex.rc = dmx_transpose_bf16_launch(s.v.p + (size_t)b * S * C, C, vt, S, S, C, ex.stream);
      ex.gemm_raw(s.q.p + (size_t)b * S * C, C, S, s.k.p + (size_t)b * S * C, C, S, C, nullptr, sc, S, 1);
      if (live()) ex.rc = dmx_softmax_rows_launch(sc, S, Pb, S, S, S, scale, ex.stream);
      ex.gemm_raw(Pb, S, S, vt, S, C, S, nullptr, s.a.p + (size_t)b * S * C, C, 0);
    }
    ex.drop(sc); ex.drop(vt);
    return ex.linear(s.a, T.W(w.wo), C, T.F(w.bo), &x, false);
  }
  Tn attn_bwd(const AttnW& w, AttnSave& s, const Tn& dy) {
    const int C = w.C, S = s.x.H * s.x.W, Bn = s.x.B;
    const float scale = 1.0f / sqrtf((float)C);
    Tn da = T.linear_bwd(s.a, dy, w.wo, w.bo, true, nullptr);
    Tn dq = ex.make(s.x.B, s.x.H, s.x.W, C), dk = ex.make(s.x.B, s.x.H, s.x.W, C), dv = ex.make(s.x.B, s.x.H, s.x.W, C);
    float* tmp = (float*)ex.raw((size_t)S * C * 4);
    float* dP = (float*)ex.raw((size_t)S * S * 4);
    bf16* dS = (bf16*)ex.raw((size_t)S * S * 2);
    bf16* kt = (bf16*)ex.raw((size_t)C * S * 2);
    for (int b = 0; b < Bn; ++b) {
      const size_t ro = (size_t)b * S * C;
      Tn Pb; Pb.p = s.P + (size_t)b * S * S; Pb.B = 1; Pb.H = 1; Pb.W = S; Pb.C = S; Pb.ld = S;
      Tn dab; dab.p = da.p + ro; dab.B = 1; dab.H = 1; dab.W = S; dab.C = C; dab.ld = C;
      Tn qb = dab; qb.p = s.q.p + ro;
      // dV = P^T dA
      T.wgrad(Pb, dab, nullptr, 1, 1, 0, tmp, C);
      if (live()) ex.rc = dmx_cast_f32_to_bf16_launch(tmp, dv.p + ro, (size_t)S * C, ex.stream);
      // dP = dA V^T ; dS = scale * P o (dP - rowsum(dP o P))
      ex.gemm_raw(dab.p, C, S, s.v.p + ro, C, S, C, nullptr, dP, S, 1);
      if (live()) ex.rc = dmx_softmax_bwd_rows_launch(Pb.p, S, dP, S, dS, S, S, S, scale, ex.stream);
      // dQ = dS K
      if (live()) ex.rc = dmx_transpose_bf16_launch(s.k.p + ro, C, kt, S, S, C, ex.stream);
      ex.gemm_raw(dS, S, S, kt, S, C, S, nullptr, dq.p + ro, C, 0);
      // dK = dS^T Q
      Tn dSb = Pb; dSb.p = dS;
      T.wgrad(dSb, qb, nullptr, 1, 1, 0, tmp, C);
      if (live()) ex.rc = dmx_cast_f32_to_bf16_launch(tmp, dk.p + ro, (size_t)S * C, ex.stream);
    }
    ex.drop(kt); ex.drop(dS); ex.drop(dP); ex.drop(tmp); ex.drop(da);
    Tn dn = T.linear_bwd(s.n, dq, w.wq, w.bq, true, nullptr);
    Tn dn2 = T.linear_bwd(s.n, dk, w.wk, w.bk, true, &dn); ex.drop(dn);
    Tn dn3 = T.linear_bwd(s.n, dv, w.wv, w.bv, true, &dn2); ex.drop(dn2);
    ex.drop(dq); ex.drop(dk); ex.drop(dv);
    Tn dx = T.gn_bwd(s.x, nullptr, w.gg, w.gb, false, s.stg, dn3, &dy, nullptr, nullptr);
    ex.drop(dn3);
    return dx;
  }

  void im2col(const float* f, const Tn* h, int C, int Bn, int IH, int IW, int ks, int pad, Tn& out) {
    if (!live()) return;
    Im2colArgs a{};
    if (f) { a.f0 = f; a.c0 = C; } else { a.h = h->p; a.ldh = h->ld; }
    a.C = C; a.B = Bn; a.IH = a.OH = IH; a.IW = a.OW = IW; a.ksize = ks; a.stride = 1; a.pad = pad; a.out = out.p; a.Kpad = out.C;
    ex.rc = dmx_im2col_small_launch(a, ex.stream);
  }

  // ---------------------------------------------------------------- forward: recon = decode(mode(encode(x)))
  int forward(const float* x, float* recon, int B_, int H_, int W_) {
    B = B_; H = H_; W = W_;
    const dmx_vae_config& c = v->cfg; const int L = c.layers_per_block; const int lc = c.latent_channels;
    const int* boc = c.block_out_channels;
    col_in = ex.make(B, H, W, v->e_in.kpad);
    im2col(x, nullptr, c.in_channels, B, H, W, 3, 1, col_in);
    Tn h = ex.linear(col_in, T.W(v->e_in.w), boc[0], T.F(v->e_in.b), nullptr, false);
    for (int i = 0; i < 4; ++i) {
      e_res[i].resize(L);
      for (int j = 0; j < L; ++j) h = T.res_fwd(v->e_res[i][j], h, nullptr, e_res[i][j]);
      if (i < 3) {
        ConvOpts o; o.stride = 2; o.pad = 0; o.bias = T.F(v->e_ds[i].b);
        e_ds[i].x = h;
        h = ex.conv(h, nullptr, T.W(v->e_ds[i].w), boc[i], o);
      }
    }
    { Tn y = T.res_fwd(v->e_mid[0], h, nullptr, e_mid[0]);
      Tn zz = attn_fwd(v->e_attn, y, e_attn);
      h = T.res_fwd(v->e_mid[1], zz, nullptr, e_mid[1]); }
    e_hlast = h;
    e_tout = T.gn(h, nullptr, v->e_ng, v->e_nb, 1e-6f, true, &e_st);
    { ConvOpts oo; oo.bias = T.F(v->e_out.b); m8 = ex.conv(e_tout, nullptr, T.W(v->e_out.w), 2 * lc, oo); }
    const int Ml = m8.rows(), lh = m8.H, lw = m8.W;
    float* mom = (float*)ex.raw((size_t)Ml * 2 * lc * 4);
    if (live()) ex.rc = dmx_pointwise_small_fwd_launch(m8.p, m8.ld, T.W(v->quant.w), v->quant.kpad, T.F(v->quant.b), mom, 2 * lc, Ml, 2 * lc, 2 * lc, 1, ex.stream);
    z = ex.make(B, lh, lw, lc);                      // posterior mode = mean, rounded as the decoder's input is
    if (live()) { hipLaunchKernelGGL(dmx_slice_cast_kernel, dim3(cdiv(Ml * lc, 256)), dim3(256), 0, ex.stream, mom, 2 * lc, z.p, lc, Ml, lc); ex.rc = dmx_check_launch("dmx_slice_cast_kernel"); }
    ex.drop(mom);
    z2 = ex.make(B, lh, lw, lc);
    if (live()) ex.rc = dmx_pointwise_small_fwd_launch(z.p, z.ld, T.W(v->pquant.w), v->pquant.kpad, T.F(v->pquant.b), z2.p, lc, Ml, lc, lc, 0, ex.stream);
    dcol = ex.make(B, lh, lw, v->d_in.kpad);
    im2col(nullptr, &z2, lc, B, lh, lw, 3, 1, dcol);
    h = ex.linear(dcol, T.W(v->d_in.w), boc[3], T.F(v->d_in.b), nullptr, false);
    { Tn y = T.res_fwd(v->d_mid[0], h, nullptr, d_mid[0]);
      Tn zz = attn_fwd(v->d_attn, y, d_attn);
      h = T.res_fwd(v->d_mid[1], zz, nullptr, d_mid[1]); }
    for (int i = 0; i < 4; ++i) {
      d_res[i].resize(L + 1);
      for (int j = 0; j < L + 1; ++j) h = T.res_fwd(v->d_res[i][j], h, nullptr, d_res[i][j]);
      if (i < 3) {
        ConvOpts o; o.ups = 1; o.bias = T.F(v->d_us[i].b);
        d_us[i].x = h;
        h = ex.conv(h, nullptr, T.W(v->d_us[i].w), boc[3 - i], o);
      }
    }
    d_hlast = h;
    d_tout = T.gn(h, nullptr, v->d_ng, v->d_nb, 1e-6f, true, &d_st);
    const int Mo = B * d_tout.H * d_tout.W, OC = c.out_channels;
    float* im = (float*)ex.raw((size_t)Mo * OC * 4);
    { ConvOpts oo; oo.bias = T.F(v->d_out.b); oo.out_f32 = 1; ex.conv(d_tout, nullptr, T.W(v->d_out.w), OC, oo, im); }
    if (live()) ex.rc = dmx_nhwc_to_nchw_f32_launch(im, OC, recon, B, OC, d_tout.H * d_tout.W, ex.stream);
    ex.drop(im);
    forward_done = (ex.rc == 0);
    return ex.rc;
  }

  // small-Cout conv (conv_out layers): dY (NHWC bf16, C <= 8 columns used) -> dW rows, db, and dX through an im2col of dY
  Tn convout_bwd(const CW& cw, const Tn& xin, const Tn& dy8, int cout, const float* dy_nchw, const Tn* dy_nhwc, size_t wt_off, int kpad_t) {
    const int cin = cw.cin;
    float* dw8 = (float*)ex.raw((size_t)8 * 9 * cin * 4);
    T.wgrad(dy8, xin, nullptr, 3, 1, 0, dw8, 9 * cin);
    if (live()) ex.rc = (int)hipMemcpyAsync(T.G(cw.w), dw8, (size_t)cout * 9 * cin * 4, hipMemcpyDeviceToDevice, ex.stream) ? DMX_ERR_HIP : 0;
    float* db8 = (float*)ex.raw(8 * 4);
    T.colsum(dy8, 1, db8, 8);
    if (live()) ex.rc = (int)hipMemcpyAsync(T.G(cw.b), db8, (size_t)cout * 4, hipMemcpyDeviceToDevice, ex.stream) ? DMX_ERR_HIP : 0;
    ex.drop(db8); ex.drop(dw8);
    Tn col = ex.make(xin.B, xin.H, xin.W, kpad_t);
    im2col(dy_nchw, dy_nhwc, cout, xin.B, xin.H, xin.W, 3, 1, col);
    Tn dx = ex.linear(col, (const bf16*)(T.wt + wt_off), cin, nullptr, nullptr, false);
    ex.drop(col);
    return dx;
  }

  // ---------------------------------------------------------------- backward from drecon (fp32 NCHW [B][OC][H][W])
  int backward(char* gr, const float* drecon) {
    T.gr = gr;
    const dmx_vae_config& c = v->cfg; const int L = c.layers_per_block; const int lc = c.latent_channels; const int OC = c.out_channels;
    ConvOpts od;
    // ---- decoder conv_out (OC = 3): dY padded to 8 columns
    Tn dy8 = ex.make(B, d_tout.H, d_tout.W, 8);
    if (live()) {
      ex.rc = (int)hipMemsetAsync(dy8.p, 0, (size_t)dy8.rows() * 8 * 2, ex.stream) ? DMX_ERR_HIP : 0;
      if (!ex.rc) ex.rc = dmx_nchw_f32_to_nhwc_bf16_launch(drecon, dy8.p, 8, B, OC, d_tout.H * d_tout.W, ex.stream);
    }
    Tn dt = convout_bwd(v->d_out, d_tout, dy8, OC, drecon, nullptr, wt_extra_dout(v), 64);
    ex.drop(dy8);
    Tn dh = T.gn_bwd(d_hlast, nullptr, v->d_ng, v->d_nb, true, d_st, dt, nullptr, nullptr, nullptr);
    ex.drop(dt);
    for (int i = 3; i >= 0; --i) {
      if (i < 3) {
        const CW& cw = v->d_us[i];
        T.wgrad(dh, d_us[i].x, nullptr, 3, 1, 1, T.G(cw.w), 9 * cw.cin);
        T.colsum(dh, 1, T.G(cw.b), cw.cout);
        float* du = (float*)ex.raw((size_t)dh.rows() * cw.cin * 4);
        { ConvOpts o; o.out_f32 = 1; ex.conv(dh, nullptr, T.WT(cw.w), cw.cin, o, du); }
        Tn dx = ex.make(B, d_us[i].x.H, d_us[i].x.W, cw.cin);
        if (live()) ex.rc = dmx_sumpool2_launch(du, cw.cin, 1, dx.p, dx.ld, B, dx.H, dx.W, cw.cin, 0, ex.stream);
        ex.drop(du); ex.drop(dh); dh = dx;
      }
      for (int j = L; j >= 0; --j) { Tn d = T.res_bwd(v->d_res[i][j], d_res[i][j], dh, nullptr, nullptr); ex.drop(dh); dh = d; }
    }
    { Tn d = T.res_bwd(v->d_mid[1], d_mid[1], dh, nullptr, nullptr); ex.drop(dh);
      Tn e = attn_bwd(v->d_attn, d_attn, d); ex.drop(d);
      dh = T.res_bwd(v->d_mid[0], d_mid[0], e, nullptr, nullptr); ex.drop(e); }
    // ---- decoder conv_in (4 -> C): dW on the kept im2col matrix; dX over the 4 latent channels through the forward conv
    T.wgrad(dh, dcol, nullptr, 1, 1, 0, T.G(v->d_in.w), v->d_in.kpad);
    T.colsum(dh, 1, T.G(v->d_in.b), v->d_in.cout);
    Tn dz2 = ex.conv(dh, nullptr, T.WT(v->d_in.w), lc, od);
    ex.drop(dh);
    const int Ml = z.rows();
    // ---- post_quant_conv, mode(), quant_conv
    float* dz2f = (float*)ex.raw((size_t)Ml * lc * 4);
    if (live()) { hipLaunchKernelGGL(dmx_bf16_to_f32_rows_kernel, dim3(cdiv(Ml * lc, 256)), dim3(256), 0, ex.stream, dz2.p, dz2.ld, dz2f, Ml, lc); ex.rc = dmx_check_launch("dmx_bf16_to_f32_rows_kernel"); }
    ex.drop(dz2);
    Tn dz = ex.make(z.B, z.H, z.W, lc);
    { const size_t wsb = dmx_pointwise_small_bwd_ws_bytes(Ml, 2 * lc, 2 * lc);
      void* ws = ex.raw(wsb);
      if (live()) ex.rc = dmx_pointwise_small_bwd_launch(z.p, z.ld, dz2f, lc, T.W(v->pquant.w), v->pquant.kpad, dz.p, dz.ld,
                                                         T.G(v->pquant.w), v->pquant.kpad, T.G(v->pquant.b), Ml, lc, lc, ws, wsb, ex.stream);
      float* dmom = (float*)ex.raw((size_t)Ml * 2 * lc * 4);
      if (live()) { hipLaunchKernelGGL(dmx_mode_bwd_kernel, dim3(cdiv(Ml * 2 * lc, 256)), dim3(256), 0, ex.stream, dz.p, dz.ld, dmom, Ml, lc); ex.rc = dmx_check_launch("dmx_mode_bwd_kernel"); }
      Tn dm8 = ex.make(m8.B, m8.H, m8.W, 2 * lc);
      if (live()) ex.rc = dmx_pointwise_small_bwd_launch(m8.p, m8.ld, dmom, 2 * lc, T.W(v->quant.w), v->quant.kpad, dm8.p, dm8.ld,
                                                         T.G(v->quant.w), v->quant.kpad, T.G(v->quant.b), Ml, 2 * lc, 2 * lc, ws, wsb, ex.stream);
      ex.drop(dmom); ex.drop(ws); ex.drop(dz); ex.drop(dz2f);
      // ---- encoder conv_out (C -> 8)
      Tn dt2 = convout_bwd(v->e_out, e_tout, dm8, 2 * lc, nullptr, &dm8, wt_extra_eout(v), 128);
      ex.drop(dm8);
      dh = T.gn_bwd(e_hlast, nullptr, v->e_ng, v->e_nb, true, e_st, dt2, nullptr, nullptr, nullptr);
      ex.drop(dt2); }
    { Tn d = T.res_bwd(v->e_mid[1], e_mid[1], dh, nullptr, nullptr); ex.drop(dh);
      Tn e = attn_bwd(v->e_attn, e_attn, d); ex.drop(d);
      dh = T.res_bwd(v->e_mid[0], e_mid[0], e, nullptr, nullptr); ex.drop(e); }
    for (int i = 3; i >= 0; --i) {
      if (i < 3) {
        const CW& cw = v->e_ds[i];
        const Tn& x = e_ds[i].x;
        T.wgrad(dh, x, nullptr, 3, 2, 0, T.G(cw.w), 9 * cw.cin, 0);
        T.colsum(dh, 1, T.G(cw.b), cw.cout);
        Tn zi = ex.make(B, x.H, x.W, cw.cout);
        if (live()) ex.rc = dmx_zero_insert2_launch(dh.p, dh.ld, zi.p, B, dh.H, dh.W, cw.cout, ex.stream);
        ConvOpts o; o.pad = 2;
        Tn dx = ex.conv(zi, nullptr, T.WT(cw.w), cw.cin, o);
        ex.drop(zi); ex.drop(dh); dh = dx;
      }
      for (int j = L - 1; j >= 0; --j) { Tn d = T.res_bwd(v->e_res[i][j], e_res[i][j], dh, nullptr, nullptr); ex.drop(dh); dh = d; }
    }
    // ---- encoder conv_in (3 -> C): dW on the kept im2col matrix
    T.wgrad(dh, col_in, nullptr, 1, 1, 0, T.G(v->e_in.w), v->e_in.kpad);
    T.colsum(dh, 1, T.G(v->e_in.b), v->e_in.cout);
    ex.drop(dh);
    return ex.rc;
  }
};

int t_linear(const dmx_vae* v, char* wt, size_t off, int N, int K, hipStream_t s) {
  return dmx_transpose_bf16_launch((const bf16*)(v->arena + off), K, (bf16*)(wt + off), N, N, K, s);
}
int t_conv(const dmx_vae* v, char* wt, size_t off, int N, int Cin, int ld, int ldt, hipStream_t s) {
  for (int tap = 0; tap < 9; ++tap) {
    const int rc = dmx_transpose_bf16_launch((const bf16*)(v->arena + off) + (size_t)tap * Cin, ld, (bf16*)(wt + off) + (size_t)(8 - tap) * N, ldt, N, Cin, s);
    if (rc) return rc;
  }
  return DMX_OK;
}
int t_resnet(const dmx_vae* v, char* wt, const ResW& r, hipStream_t s) {
  int rc = t_conv(v, wt, r.w1, r.cout, r.cin, 9 * r.cin, 9 * r.cout, s);
  const int k2 = 9 * r.cout + (r.shortcut ? r.cin : 0);
  if (!rc) rc = t_conv(v, wt, r.w2, r.cout, r.cout, k2, 9 * r.cout, s);
  if (!rc && r.shortcut)
    rc = dmx_transpose_bf16_launch((const bf16*)(v->arena + r.w2) + 9 * r.cout, k2, (bf16*)(wt + r.w2) + (size_t)9 * r.cout * r.cout, r.cout, r.cout, r.cin, s);
  return rc;
}
int t_attn(const dmx_vae* v, char* wt, const AttnW& a, hipStream_t s) {
  int rc = t_linear(v, wt, a.wq, a.C, a.C, s);
  if (!rc) rc = t_linear(v, wt, a.wk, a.C, a.C, s);
  if (!rc) rc = t_linear(v, wt, a.wv, a.C, a.C, s);
  if (!rc) rc = t_linear(v, wt, a.wo, a.C, a.C, s);
  return rc;
}
}  // namespace

extern "C" size_t dmx_vae_train_wt_bytes(const dmx_vae* v) { return v ? wt_total(v) : 0; }
extern "C" size_t dmx_vae_grad_bytes(const dmx_vae* v) { return v ? 2 * v->pt.total() : 0; }

extern "C" int dmx_vae_train_prepare(dmx_vae* v, void* wt_arena, size_t wt_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->finalized, "vae_train_prepare: weights not finalized");
  DMX_REQUIRE(wt_arena && wt_bytes >= wt_total(v), "vae_train_prepare: need %zu bytes", wt_total(v));
  hipStream_t s = (hipStream_t)stream; char* wt = (char*)wt_arena;
  const int* boc = v->cfg.block_out_channels; const int lc = v->cfg.latent_channels, OC = v->cfg.out_channels;
  DMX_REQUIRE(9 * 2 * lc <= 128 && 9 * OC <= 64, "vae_train_prepare: latent/out channel counts too large for the padded conv_out data gradients");
  int rc = 0;
  for (int i = 0; i < 4 && !rc; ++i) {
    for (auto& r : v->e_res[i]) if (!rc) rc = t_resnet(v, wt, r, s);
    for (auto& r : v->d_res[i]) if (!rc) rc = t_resnet(v, wt, r, s);
    if (i < 3 && !rc) rc = t_conv(v, wt, v->e_ds[i].w, v->e_ds[i].cout, v->e_ds[i].cin, 9 * v->e_ds[i].cin, 9 * v->e_ds[i].cout, s);
    if (i < 3 && !rc) rc = t_conv(v, wt, v->d_us[i].w, v->d_us[i].cout, v->d_us[i].cin, 9 * v->d_us[i].cin, 9 * v->d_us[i].cout, s);
  }
  for (int k = 0; k < 2 && !rc; ++k) { rc = t_resnet(v, wt, v->e_mid[k], s); if (!rc) rc = t_resnet(v, wt, v->d_mid[k], s); }
  if (!rc) rc = t_attn(v, wt, v->e_attn, s);
  if (!rc) rc = t_attn(v, wt, v->d_attn, s);
  // decoder conv_in [C][kpad: tap*lc + c] -> [lc][flip(tap)*C + n] (plain conv data gradient with N = lc)
  for (int tap = 0; tap < 9 && !rc; ++tap)
    rc = dmx_transpose_bf16_launch((const bf16*)(v->arena + v->d_in.w) + (size_t)tap * lc, v->d_in.kpad,
                                   (bf16*)(wt + v->d_in.w) + (size_t)(8 - tap) * v->d_in.cout, 9 * v->d_in.cout, v->d_in.cout, lc, s);
  // encoder conv_out [2lc][tap*C + ci] -> [C][128]: column flip(tap)*2lc + n (im2col of dY), zero padded
  if (!rc) {
    const int C = v->e_out.cin, N = 2 * lc;
    DMX_HIP(hipMemsetAsync(wt + wt_extra_eout(v), 0, (size_t)C * 128 * 2, s));
    for (int tap = 0; tap < 9 && !rc; ++tap)
      rc = dmx_transpose_bf16_launch((const bf16*)(v->arena + v->e_out.w) + (size_t)tap * C, 9 * C, (bf16*)(wt + wt_extra_eout(v)) + (size_t)(8 - tap) * N, 128, N, C, s);
  }
  // decoder conv_out [OC][tap*C0 + ci] -> [C0][64]
  if (!rc) {
    const int C = v->d_out.cin;
    DMX_HIP(hipMemsetAsync(wt + wt_extra_dout(v), 0, (size_t)C * 64 * 2, s));
    for (int tap = 0; tap < 9 && !rc; ++tap)
      rc = dmx_transpose_bf16_launch((const bf16*)(v->arena + v->d_out.w) + (size_t)tap * C, 9 * C, (bf16*)(wt + wt_extra_dout(v)) + (size_t)(8 - tap) * OC, 64, OC, C, s);
  }
  (void)boc;
  return rc;
}

extern "C" size_t dmx_vae_train_workspace_bytes(dmx_vae* v, int B, int H, int W) {
  if (!v) return 0;
  VaeTrain ts(v, nullptr, B);
  ts.ex.dry = true; ts.ex.ws.reset(nullptr, 0, true);
  ts.forward(nullptr, nullptr, B, H, W);
  ts.backward(nullptr, nullptr);
  return ts.ex.ws.peak() + 4096;
}

// recon = decode(encode(x).mode()) (train_vae.py:721), keeping what the backward needs inside `workspace`
extern "C" int dmx_vae_train_forward(dmx_vae* v, const void* wt_arena, const float* x, float* recon, int B, int H, int W,
                                     void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->finalized, "vae_train_forward: weights not finalized");
  DMX_REQUIRE(wt_arena && x && recon && workspace, "vae_train_forward: null argument");
  DMX_REQUIRE(B > 0 && H > 0 && W > 0 && H % 64 == 0 && W % 64 == 0, "vae_train_forward: H=%d W=%d must be positive multiples of 64", H, W);
  auto ts = std::make_shared<VaeTrain>(v, (char*)wt_arena, B);
  ts->ex.stream = (hipStream_t)stream; ts->ex.ws.reset(workspace, workspace_bytes, false);
  v->train_state = ts;
  const int rc = ts->forward(x, recon, B, H, W);
  if (rc) v->train_state.reset();
  return rc;
}
// gradients of every parameter for dLoss/drecon (fp32 NCHW), written into `grads` (packed fp32 arena)
extern "C" int dmx_vae_train_backward(dmx_vae* v, void* grads, const float* drecon, dmx_stream_t stream) {
  DMX_REQUIRE(v && grads && drecon, "vae_train_backward: null argument");
  auto ts = std::static_pointer_cast<VaeTrain>(v->train_state);
  DMX_REQUIRE(ts && ts->forward_done, "vae_train_backward: no forward pass to differentiate");
  DMX_REQUIRE((hipStream_t)stream == ts->ex.stream, "vae_train_backward: must run on the forward's stream");
  const int rc = ts->backward((char*)grads, drecon);
  v->train_state.reset();
  return rc;
}
extern "C" int dmx_vae_grad_export(const dmx_vae* v, const void* grads, const char* name, float* dst, dmx_stream_t stream) {
  DMX_REQUIRE(v && grads && name && dst, "vae_grad_export: null argument");
  return dmx_param_grad_export(v->pt, grads, name, dst, (hipStream_t)stream);
}
