"""Operator-level Python wrappers over the C-ABI (include/diffute_hip.h).

Activations are NHWC bf16 torch tensors [B,H,W,C] (last dim contiguous; a row stride larger than C is
allowed, e.g. a channel slice of a wider tensor).  These wrappers exist for the parity tests and for
users who want single kernels; the model executors call the same kernels from C++.
"""
import ctypes

import torch

import contextlib

from . import _cabi
from ._cabi import GemmDesc, HaloConvDesc, check, current_stream, ptr

_ELEM = "bf16"


def lib():
    """the build of the library the operator wrappers call: bf16, or fp16 inside `with element_type("fp16")`"""
    return _cabi.lib(_ELEM)


def h16():
    """torch dtype of the 16-bit activation / weight element of the active build"""
    return _cabi.torch_elem(_ELEM)


@contextlib.contextmanager
def element_type(elem):
    """run the operator wrappers on the fp16 build (libdiffute_hip_f16.so): activations / packed weights are torch.float16"""
    global _ELEM
    old, _ELEM = _ELEM, elem
    try:
        yield
    finally:
        _ELEM = old


def _ld(x):
    assert x.stride(-1) == 1, "channel dim must be contiguous"
    return x.stride(-2)


def nchw_to_nhwc_bf16(x):
    """fp32 NCHW -> bf16 NHWC through dmx_nchw_f32_to_nhwc_bf16."""
    x = x.to(torch.float32).contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, C, dtype=h16(), device=x.device)
    check(lib().dmx_nchw_f32_to_nhwc_bf16(ptr(x), ptr(out), C, B, C, H * W, current_stream()), "nchw_f32_to_nhwc_bf16")
    return out


def nhwc_bf16_to_nchw(x):
    B, H, W, C = x.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    check(lib().dmx_nhwc_bf16_to_nchw_f32(ptr(x), _ld(x), ptr(out), B, C, H * W, current_stream()), "nhwc_bf16_to_nchw_f32")
    return out


def pack_conv_weight(w, shortcut_w=None):
    """[Cout,Cin,k,k] fp32 -> bf16 [Cout][k*k*Cin (+ Csc)] ; optional 1x1 shortcut appended as extra K."""
    w = w.to(torch.float32).contiguous()
    Cout, Cin, k, _ = w.shape
    K = k * k * Cin + (shortcut_w.shape[1] if shortcut_w is not None else 0)
    Kp = (K + 63) // 64 * 64
    out = torch.zeros(Cout, Kp, dtype=h16(), device=w.device)
    check(lib().dmx_pack_conv_weight(ptr(w), ptr(out), Cout, Cin, k, Kp, 0, current_stream()), "pack_conv_weight")
    if shortcut_w is not None:
        s = shortcut_w.to(torch.float32).contiguous()
        check(lib().dmx_pack_conv_weight(ptr(s), ptr(out), Cout, s.shape[1], 1, Kp, k * k * Cin, current_stream()), "pack_conv_weight(sc)")
    return out


def pack_linear_weight(w, geglu=False):
    w = w.to(torch.float32).contiguous()
    out = torch.empty(w.shape, dtype=h16(), device=w.device)
    check(lib().dmx_pack_linear_weight(ptr(w), ptr(out), w.shape[0], w.shape[1], w.shape[1], int(geglu), current_stream()), "pack_linear_weight")
    return out


def pack_conv_weight_t(w):
    """[Cout,Cin,k,k] fp32 -> bf16 [Cin][k*k*Cout]: flipped taps, channel roles swapped (data-gradient filter)."""
    w = w.to(torch.float32).contiguous()
    Cout, Cin, k, _ = w.shape
    out = torch.empty(Cin, k * k * Cout, dtype=h16(), device=w.device)
    check(lib().dmx_pack_conv_weight_t(ptr(w), ptr(out), Cout, Cin, k, k * k * Cout, 0, current_stream()), "pack_conv_weight_t")
    return out


def pack_linear_weight_t(w):
    """[N,K] fp32 -> bf16 [K][N]"""
    w = w.to(torch.float32).contiguous()
    out = torch.empty(w.shape[1], w.shape[0], dtype=h16(), device=w.device)
    check(lib().dmx_pack_linear_weight_t(ptr(w), ptr(out), w.shape[0], w.shape[1], w.shape[0], current_stream()), "pack_linear_weight_t")
    return out


def conv_dgrad(dy, wt, Cin, *, ksize=3, stride=1, ups=False, res=None):
    """dX [B,IH,IW,Cin] bf16 of a pad-(k//2) conv from dy [B,OH,OW,Cout] and the transposed pack of its filter.
    stride 2: dY is zero-inserted onto the input grid first; ups: the gradient on the upsampled grid is sum-pooled.
    `res` (bf16, input-shaped) is added: the gradient arriving over a residual / second consumer."""
    B, OH, OW, Cout = dy.shape
    g = dy
    if stride == 2:
        g = torch.empty(B, 2 * OH, 2 * OW, Cout, dtype=h16(), device=dy.device)
        check(lib().dmx_zero_insert2(ptr(dy), _ld(dy), ptr(g), B, OH, OW, Cout, current_stream()), "zero_insert2")
    du = conv_gemm(g, wt, Cin, ksize=ksize, stride=1, pad=ksize // 2, res=None if ups else res, out_f32=ups)
    if not ups:
        return du
    H, W = du.shape[1] // 2, du.shape[2] // 2
    dx = res.clone() if res is not None else torch.empty(B, H, W, Cin, dtype=h16(), device=dy.device)
    check(lib().dmx_sumpool2(ptr(du), _ld(du), 1, ptr(dx), _ld(dx), B, H, W, Cin, int(res is not None), current_stream()), "sumpool2")
    return dx


def pack_geglu_bias(b):
    b = b.to(torch.float32).contiguous()
    out = torch.empty_like(b)
    check(lib().dmx_pack_geglu_bias(ptr(b), ptr(out), b.numel(), current_stream()), "pack_geglu_bias")
    return out


def pack_ups_phase_weights(w3, N, Cin):
    """taps-major packed 3x3 weights [N][9*Cin] (pack_conv_weight) -> [4][N][4*Cin] phase weights of conv_ups2x"""
    wp = torch.empty(4, N, 4 * Cin, dtype=h16(), device=w3.device)
    check(lib().dmx_pack_ups_phase_weights(ptr(w3), w3.stride(0), ptr(wp), N, Cin, current_stream()), "pack_ups_phase_weights")
    return wp


def conv_ups2x(x, wp, N, bias=None, force_tn=0, force_splitk=0):
    """conv3x3(nearest_x2(x)) through four 2x2 phase convolutions on the source grid.  x NHWC bf16 -> NHWC bf16 [B,2H,2W,N]."""
    B, H, W, Cin = x.shape
    out = torch.empty(B, 2 * H, 2 * W, N, dtype=h16(), device=x.device)
    wsb = lib().dmx_conv_ups2x_workspace_bytes(B, H, W, Cin, N, force_tn, force_splitk)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x.device)
    check(lib().dmx_conv_ups2x(ptr(x), _ld(x), B, H, W, Cin, ptr(wp), N, ptr(bias) if bias is not None else None, ptr(out), N,
                               force_tn, force_splitk, ptr(ws), wsb, current_stream()), "conv_ups2x")
    return out


def conv_gemm(x0, w, N, *, x1=None, ksize=3, stride=1, pad=1, ups=False, bias=None, rowbias=None,
              res=None, sc0=None, sc1=None, out_f32=False, geglu=False, direct=None, force_tn=0, force_splitk=0, timing=None, group_m=0, dbg=0, act=0,
              rowstats=False, ln=None, gn_stats=False):
    """Fused conv / linear.  x0 (and x1) NHWC bf16; w packed bf16 [N][K].  Returns NHWC (bf16 or fp32).
    gn_stats=True: also returns the [B][N][2] int64 fixed-point (sum * 2^20, sumsq * 2^32) GroupNorm statistics of the output
    (or None when the plan this problem runs on cannot emit them: split-K / tiles straddling samples).
    rowstats=True: also returns the per-row (sum, sumsq) partials [tiles][M][2] of the rounded output (folded-LayerNorm
    producer); ln=(stats, c1, c2, eps): folded-LayerNorm consumer (w must already be W*diag(gamma))."""
    B, H, W, C0 = x0.shape
    Cin = C0 + (x1.shape[-1] if x1 is not None else 0)
    OH, OW = H, W
    if ups:
        OH, OW = 2 * OH, 2 * OW
    if stride == 2:
        OH, OW = OH // 2, OW // 2
    d = GemmDesc()
    d.x0 = x0.data_ptr(); d.ldx0 = _ld(x0); d.cx0 = C0
    if x1 is not None:
        d.x1 = x1.data_ptr(); d.ldx1 = _ld(x1)
    d.direct = int(ksize == 1 and stride == 1 and not ups) if direct is None else int(direct)
    d.IH, d.IW, d.OH, d.OW = H, W, OH, OW
    d.stride, d.pad, d.ups, d.ksize, d.Cin = stride, pad, int(ups), ksize, Cin
    d.Ktaps = ksize * ksize * Cin
    K = d.Ktaps
    if sc0 is not None:
        d.s0 = sc0.data_ptr(); d.lds0 = _ld(sc0); d.cs0 = sc0.shape[-1]; K += sc0.shape[-1]
        if sc1 is not None:
            d.s1 = sc1.data_ptr(); d.lds1 = _ld(sc1); K += sc1.shape[-1]
    d.w = w.data_ptr(); d.ldw = w.stride(0)
    d.M = B * OH * OW; d.N = N; d.K = K
    if bias is not None:
        d.bias = bias.data_ptr()
    if rowbias is not None:
        d.rowbias = rowbias.data_ptr(); d.ldrb = rowbias.stride(0)
    d.rows_per_group = OH * OW
    if res is not None:
        d.res = res.data_ptr(); d.ldres = _ld(res)
    Nout = N // 2 if geglu else N
    out = torch.empty(B, OH, OW, Nout, dtype=torch.float32 if out_f32 else h16(), device=x0.device)
    d.out = out.data_ptr(); d.ldo = Nout; d.out_f32 = int(out_f32); d.geglu = int(geglu)
    d.force_tn = force_tn; d.force_splitk = force_splitk; d.group_m = group_m; d.dbg = dbg; d.act = act
    if timing is not None:
        d.timing = timing.data_ptr()
    if ln is not None:
        st, c1, c2, eps = ln
        d.ln_stats = st.data_ptr(); d.ln_tiles = st.shape[0]; d.ln_c1 = c1.data_ptr(); d.ln_c2 = c2.data_ptr(); d.ln_C = K; d.ln_eps = eps
    stats = None
    if rowstats:
        stats = torch.empty(lib().dmx_conv_gemm_rowstats_tiles(ctypes.byref(d)), d.M, 2, dtype=torch.float32, device=x0.device)   # (every entry is written)
        d.rowstats_out = stats.data_ptr()
    cst = None
    if gn_stats:
        d.cs_rows = OH * OW
        if lib().dmx_conv_gemm_colstats_ok(ctypes.byref(d)):
            cst = torch.zeros(B, N, 4, dtype=torch.int64, device=x0.device)
            d.colstats = cst.data_ptr()
        else:
            d.cs_rows = 0
    wsb = lib().dmx_conv_gemm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x0.device)
    check(lib().dmx_conv_gemm(ctypes.byref(d), ptr(ws), wsb, current_stream()), "conv_gemm")
    if gn_stats:
        return out, cst
    return (out, stats) if rowstats else out


def _gather_desc(x0, x1, ksize, stride, pad, ups, direct):
    B, H, W, C0 = x0.shape
    Cin = C0 + (x1.shape[-1] if x1 is not None else 0)
    OH, OW = (2 * H, 2 * W) if ups else (H, W)
    if stride == 2:
        OH, OW = OH // 2, OW // 2
    d = GemmDesc()
    d.x0 = x0.data_ptr(); d.ldx0 = _ld(x0); d.cx0 = C0
    if x1 is not None:
        d.x1 = x1.data_ptr(); d.ldx1 = _ld(x1)
    d.direct = int(ksize == 1 and stride == 1 and not ups) if direct is None else int(direct)
    d.IH, d.IW, d.OH, d.OW = H, W, OH, OW
    d.stride, d.pad, d.ups, d.ksize, d.Cin = stride, pad, int(ups), ksize, Cin
    d.Ktaps = ksize * ksize * Cin
    d.M = B * OH * OW; d.K = d.Ktaps
    return d


def conv_wgrad(x0, dy, *, x1=None, ksize=3, stride=1, pad=1, ups=False, direct=None, into=None):
    """dW[N][K] fp32 (packed tap-major k) = sum over output pixels of dy[m][n] * gathered x[m][k]; `into` accumulates."""
    d = _gather_desc(x0, x1, ksize, stride, pad, ups, direct)
    N = dy.shape[-1]
    d.N = N
    assert dy.numel() == d.M * N, "dy must be [B, OH, OW, N]"
    out = into if into is not None else torch.empty(N, d.K, dtype=torch.float32, device=x0.device)
    acc = int(into is not None)
    wsb = lib().dmx_conv_wgrad_workspace_bytes(ctypes.byref(d), acc)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x0.device)
    check(lib().dmx_conv_wgrad(ctypes.byref(d), ptr(dy), _ld(dy), ptr(out), acc, ptr(ws), wsb, current_stream()), "conv_wgrad")
    return out


def colsum(dy, groups=1, into=None):
    """column sums of dy viewed as [groups][rows/groups][N] -> [groups][N] fp32 (bias / row-bias gradients)."""
    N = dy.shape[-1]
    rows = dy.numel() // N
    rpg = rows // groups
    out = into if into is not None else torch.empty(groups, N, dtype=torch.float32, device=dy.device)
    wsb = lib().dmx_colsum_workspace_bytes(groups, rpg, N)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dy.device)
    check(lib().dmx_colsum(ptr(dy), _ld(dy), groups, rpg, N, ptr(out), N, int(into is not None), ptr(ws), wsb, current_stream()), "colsum")
    return out


def linear(x, w, bias=None, res=None, geglu=False, out_f32=False, act=0, force_tn=0, rowstats=False, ln=None, timing=None, dbg=0):
    """x [..., K] bf16 (2-D view [rows][K]) @ w[N][K]^T."""
    K = x.shape[-1]
    x4 = x.reshape(1, 1, -1, K)
    r4 = None if res is None else res.reshape(1, 1, -1, res.shape[-1])
    y = conv_gemm(x4, w, w.shape[0], ksize=1, pad=0, bias=bias, res=r4, geglu=geglu, out_f32=out_f32, act=act, force_tn=force_tn,
                  rowstats=rowstats, ln=ln, timing=timing, dbg=dbg)
    if rowstats:
        y, st = y
        return y.reshape(*x.shape[:-1], y.shape[-1]), st
    return y.reshape(*x.shape[:-1], y.shape[-1])


def xf_chain(mode, x, res, w0, b0, c1, c2, w1=None, wf1=None, wf2=None, bf2=None, wpo=None, bpo=None, xres=None, eps=1e-5, dbg=0, timing=None,
             out_stats_rows=0, gn=None):
    """The row-local chains of a transformer block at the C = 320 levels, one launch each (include/diffute_hip.h dmx_xf_chain).
    x / res / xres [M][C]; returns (h_out, y).  out_stats_rows (mode 1): also return the statistics records of y, samples of that many
    rows.  gn = (st, gamma, beta, groups, rows_per_sample, eps) (mode 2): x is raw, GroupNorm from its records in the operand load."""
    M, C = x.shape
    d = _cabi.XfChainDesc()
    h = torch.empty(M, C, dtype=h16(), device=x.device)
    y = torch.empty(M, 3 * C if mode == 2 else C, dtype=h16(), device=x.device)
    d.M, d.C, d.eps, d.dbg = M, C, float(eps), int(dbg)
    if timing is not None:
        d.timing = timing.data_ptr()
    d.x, d.ldx, d.res, d.ldres = ptr(x), _ld(x), ptr(res), (_ld(res) if res is not None else 0)
    d.w0, d.b0, d.h_out, d.ldh, d.y, d.ldy = ptr(w0), ptr(b0), ptr(h), C, ptr(y), y.shape[1]
    d.c1, d.c2 = ptr(c1), ptr(c2)
    keep = [x, res, w0, b0, c1, c2, w1, wf1, wf2, bf2, wpo, bpo, xres]
    if mode != 1:
        d.w1 = ptr(w1)
    else:
        d.wf1, d.wf2, d.bf2, d.wpo, d.bpo, d.xres, d.ldxres = ptr(wf1), ptr(wf2), ptr(bf2), ptr(wpo), ptr(bpo), ptr(xres), _ld(xres)
    cst = None
    if out_stats_rows:
        cst = torch.zeros(M // out_stats_rows, C, 4, dtype=torch.int64, device=x.device)
        d.colstats, d.cs_rows = cst.data_ptr(), int(out_stats_rows)
    if gn is not None:
        st, gamma, beta, groups, rows, geps = gn
        keep += [st, gamma, beta]
        d.gn_st, d.gn_gamma, d.gn_beta, d.gn_groups, d.gn_rows, d.gn_eps = st.data_ptr(), ptr(gamma), ptr(beta), int(groups), int(rows), float(geps)
    check(lib().dmx_xf_chain(ctypes.byref(d), int(mode), current_stream()), "xf_chain")
    del keep
    return (h, y, cst) if out_stats_rows else (h, y)


def groupnorm(x0, gamma, beta, groups, eps, silu, x1=None):
    B, H, W, C0 = x0.shape
    C = C0 + (x1.shape[-1] if x1 is not None else 0)
    y = torch.empty(B, H, W, C, dtype=h16(), device=x0.device)
    wsb = lib().dmx_groupnorm_workspace_bytes(B, H * W, groups)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x0.device)
    check(lib().dmx_groupnorm(ptr(x0), _ld(x0), ptr(x1), _ld(x1) if x1 is not None else 0, C0, C, groups, B, H * W,
                              ptr(gamma), ptr(beta), float(eps), int(silu), ptr(y), C, ptr(ws), wsb, current_stream()), "groupnorm")
    return y


def stat_sums(st):
    """statistics records [B][C][4] int64 -> (sum, sum of squares) as float64 tensors [B][C]"""
    s = st.cpu().double()
    return s[..., 0] / 2 ** 20, s[..., 1] / 2 ** 8 + s[..., 2] / 2 ** 40


def colstats(x):
    """statistics records [B][C][4] of an NHWC tensor (dmx_colstats: for tensors whose producer emitted none)"""
    B, H, W, C = x.shape
    st = torch.zeros(B, C, 4, dtype=torch.int64, device=x.device)
    check(lib().dmx_colstats(ptr(x), _ld(x), B, H * W, C, ptr(st), current_stream()), "colstats")
    return st


def conv3x3_gn(x0, w, N, *, x1=None, gn=None, st0=None, st1=None, sc0=None, sc1=None, bias=None, rowbias=None, res=None,
               out_stats=False, force_split=0, force_bn=0, force_waves=0, timing=None, dbg=0):
    """conv3x3 (stride 1, pad 1) over a halo tile staged in LDS, with GroupNorm(+SiLU) applied to the staged tile in place
    (dmx_conv3x3_gn).  gn = (gamma, beta, groups, eps, silu) with st0 / st1 the statistics records of x0 / x1, or None for a plain
    conv.  Returns out, or (out, records of out) with out_stats."""
    B, H, W, C0 = x0.shape
    d = HaloConvDesc()
    d.x0 = x0.data_ptr(); d.ldx0 = _ld(x0); d.cx0 = C0; d.Cin = C0
    if x1 is not None:
        d.x1 = x1.data_ptr(); d.ldx1 = _ld(x1); d.Cin = C0 + x1.shape[-1]
    d.B, d.H, d.W = B, H, W
    if gn is not None:
        gamma, beta, groups, eps, silu = gn
        d.gn = 1; d.silu = int(silu); d.groups = groups; d.eps = float(eps); d.gamma = gamma.data_ptr(); d.beta = beta.data_ptr()
        d.st0 = st0.data_ptr()
        if st1 is not None:
            d.st1 = st1.data_ptr()
    if sc0 is not None:
        d.s0 = sc0.data_ptr(); d.lds0 = _ld(sc0); d.cs0 = sc0.shape[-1]; d.Csc = sc0.shape[-1]
        if sc1 is not None:
            d.s1 = sc1.data_ptr(); d.lds1 = _ld(sc1); d.Csc += sc1.shape[-1]
    d.w = w.data_ptr(); d.ldw = w.stride(0); d.N = N
    if bias is not None:
        d.bias = bias.data_ptr()
    if rowbias is not None:
        d.rowbias = rowbias.data_ptr(); d.ldrb = rowbias.stride(0)
    if res is not None:
        d.res = res.data_ptr(); d.ldres = _ld(res)
    out = torch.empty(B, H, W, N, dtype=h16(), device=x0.device)
    d.out = out.data_ptr(); d.ldo = N
    cst = None
    if out_stats:
        cst = torch.zeros(B, N, 4, dtype=torch.int64, device=x0.device)
        d.colstats = cst.data_ptr()
    d.force_split = force_split; d.force_bn = force_bn; d.force_waves = force_waves; d.dbg = dbg
    if timing is not None:
        d.timing = timing.data_ptr()
    if not lib().dmx_conv3x3_gn_supported(ctypes.byref(d)):
        raise RuntimeError("conv3x3_gn: the halo kernel does not take this problem")
    wsb = lib().dmx_conv3x3_gn_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x0.device)
    check(lib().dmx_conv3x3_gn(ctypes.byref(d), ptr(ws), wsb, current_stream()), "conv3x3_gn")
    return (out, cst) if out_stats else out


def skinny_pack(w, segs):
    """row-major packed weights [N][ldw] (pack_conv_weight / pack_linear_weight) -> fragment order for skinny_conv.  segs: list of
    (C, taps, tap_stride, koff): the source column of (tap t, channel c) of the segment is t * tap_stride + koff + c."""
    import ctypes as _c
    N = w.shape[0]
    n = len(segs)
    arr = lambda k: (_c.c_int * n)(*[int(sg[k]) for sg in segs])
    K = sum(sg[0] * sg[1] for sg in segs)
    wp = torch.empty(N * K, dtype=h16(), device=w.device)
    check(lib().dmx_skinny_pack(ptr(w), w.stride(0), ptr(wp), N, n, arr(0), arr(1), arr(2), arr(3), current_stream()), "skinny_pack")
    return wp


def skinny_conv(segs, wp, N, *, gn=None, bias=None, rowbias=None, res=None, out_stats=False, force_S=0, timing=None, dbg=0):
    """Weight-streaming conv / linear for M = B H W <= 256 rows (dmx_skinny_conv).  segs: list of dicts x=[B,H,W,C] tensor, taps=9|1, and for
    GroupNorm'ed sources st= statistics records, gamma=, beta= (this tensor's channels), gn_c0=; gn = (groups, Ctot, eps, silu) of the norm
    over the concatenation of those sources.  Returns out [B,H,W,N] (and the statistics records of out with out_stats)."""
    from ._cabi import SkinnyDesc
    x0 = segs[0]["x"]
    B, H, W, _ = x0.shape
    d = SkinnyDesc()
    d.nseg = len(segs)
    keep = []
    for k, sg in enumerate(segs):
        x = sg["x"]
        d.seg[k].x = x.data_ptr(); d.seg[k].ld = _ld(x); d.seg[k].C = x.shape[-1]; d.seg[k].taps = sg.get("taps", 9)
        if sg.get("st") is not None:
            d.seg[k].st = sg["st"].data_ptr(); d.seg[k].gamma = sg["gamma"].data_ptr(); d.seg[k].beta = sg["beta"].data_ptr(); d.seg[k].gn_c0 = sg.get("gn_c0", 0)
            keep += [sg["st"], sg["gamma"], sg["beta"]]
    d.B, d.H, d.W = B, H, W
    if gn is not None:
        d.gn_groups, d.gn_Ctot, d.gn_eps, d.silu = int(gn[0]), int(gn[1]), float(gn[2]), int(gn[3])
    d.wp = wp.data_ptr(); d.N = N
    if bias is not None:
        d.bias = bias.data_ptr()
    if rowbias is not None:
        d.rowbias = rowbias.data_ptr(); d.ldrb = rowbias.stride(0)
    if res is not None:
        d.res = res.data_ptr(); d.ldres = _ld(res)
    out = torch.empty(B, H, W, N, dtype=h16(), device=x0.device)
    d.out = out.data_ptr(); d.ldo = N
    cst = None
    if out_stats:
        cst = torch.zeros(B, N, 4, dtype=torch.int64, device=x0.device)
        d.colstats = cst.data_ptr()
    d.force_S = force_S; d.dbg = dbg
    if timing is not None:
        d.timing = timing.data_ptr()
    if not lib().dmx_skinny_conv_supported(ctypes.byref(d)):
        raise RuntimeError("skinny_conv: the kernel does not take this problem")
    wsb = lib().dmx_skinny_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x0.device)
    check(lib().dmx_skinny_conv(ctypes.byref(d), ptr(ws), wsb, current_stream()), "skinny_conv")
    return (out, cst) if out_stats else out


def groupnorm_from_stats(x0, st0, gamma, beta, groups, eps, silu, x1=None, st1=None):
    """GroupNorm (+SiLU) whose statistics were emitted by the GEMM(s) that produced x0 / x1 (conv_gemm(..., gn_stats=True))"""
    B, H, W, C0 = x0.shape
    C = C0 + (x1.shape[-1] if x1 is not None else 0)
    y = torch.empty(B, H, W, C, dtype=h16(), device=x0.device)
    check(lib().dmx_groupnorm_from_stats(ptr(x0), _ld(x0), ptr(x1), _ld(x1) if x1 is not None else 0, C0, C, groups, B, H * W,
                                         ptr(gamma), ptr(beta), float(eps), int(silu), ptr(st0), ptr(st1), ptr(y), C, current_stream()),
          "groupnorm_from_stats")
    return y


def groupnorm_train(x0, gamma, beta, groups, eps, silu, x1=None):
    """forward GroupNorm that also returns the saved (mean, rstd) [B][groups][2]"""
    B, H, W, C0 = x0.shape
    C = C0 + (x1.shape[-1] if x1 is not None else 0)
    y = torch.empty(B, H, W, C, dtype=h16(), device=x0.device)
    stats = torch.empty(B, groups, 2, dtype=torch.float32, device=x0.device)
    wsb = lib().dmx_groupnorm_workspace_bytes(B, H * W, groups)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x0.device)
    check(lib().dmx_groupnorm_train(ptr(x0), _ld(x0), ptr(x1), _ld(x1) if x1 is not None else 0, C0, C, groups, B, H * W,
                                    ptr(gamma), ptr(beta), float(eps), int(silu), ptr(y), C, ptr(stats), ptr(ws), wsb, current_stream()),
          "groupnorm_train")
    return y, stats


def groupnorm_bwd(x0, dy, gamma, beta, groups, silu, stats, x1=None, res0=None, res1=None):
    """-> (dx0, dx1 or None, dgamma, dbeta)"""
    B, H, W, C0 = x0.shape
    C = C0 + (x1.shape[-1] if x1 is not None else 0)
    dx0 = torch.empty_like(x0)
    dx1 = torch.empty_like(x1) if x1 is not None else None
    dg = torch.empty(C, dtype=torch.float32, device=x0.device); db = torch.empty_like(dg)
    wsb = lib().dmx_groupnorm_bwd_workspace_bytes(B, H * W, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x0.device)
    check(lib().dmx_groupnorm_bwd(ptr(x0), _ld(x0), ptr(x1), _ld(x1) if x1 is not None else 0, C0, C, groups, B, H * W,
                                  ptr(gamma), ptr(beta), int(silu), ptr(stats), ptr(dy), _ld(dy),
                                  ptr(dx0), _ld(dx0), ptr(dx1), _ld(dx1) if dx1 is not None else 0,
                                  ptr(res0), _ld(res0) if res0 is not None else 0, ptr(res1), _ld(res1) if res1 is not None else 0,
                                  ptr(dg), ptr(db), 0, ptr(ws), wsb, current_stream()), "groupnorm_bwd")
    return dx0, dx1, dg, db


def layernorm_bwd(x, dy, gamma, eps=1e-5, res=None):
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device); db = torch.empty_like(dg)
    wsb = lib().dmx_layernorm_bwd_workspace_bytes(rows, C)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x.device)
    check(lib().dmx_layernorm_bwd(ptr(x), C, ptr(dy), C, ptr(gamma), ptr(dx), C, ptr(res), C if res is not None else 0,
                                  ptr(dg), ptr(db), 0, rows, C, float(eps), ptr(ws), wsb, current_stream()), "layernorm_bwd")
    return dx, dg, db


def geglu_fwd(h):
    C2 = h.shape[-1] // 2
    rows = h.numel() // (2 * C2)
    y = torch.empty(*h.shape[:-1], C2, dtype=h16(), device=h.device)
    check(lib().dmx_geglu_fwd(ptr(h), 2 * C2, ptr(y), C2, rows, C2, current_stream()), "geglu_fwd")
    return y


def geglu_bwd(h, dy):
    C2 = h.shape[-1] // 2
    rows = h.numel() // (2 * C2)
    dh = torch.empty_like(h)
    check(lib().dmx_geglu_bwd(ptr(h), 2 * C2, ptr(dy), C2, ptr(dh), 2 * C2, rows, C2, current_stream()), "geglu_bwd")
    return dh


def layernorm(x, gamma, beta, eps=1e-5):
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    check(lib().dmx_layernorm(ptr(x), C, ptr(y), C, ptr(gamma), ptr(beta), rows, C, float(eps), current_stream()), "layernorm")
    return y


def attention(q, k, vt, B, H, Sq, Skv, scale, kv_rows=None, skv_stride=None):
    """q [B*Sq, >=H*64], k [B*kv_rows, >=H*64], vt [H*64, >= B*skv_stride] (2-D, row-major views)."""
    o = torch.empty(B * Sq, H * 64, dtype=h16(), device=q.device)
    kv_rows = Skv if kv_rows is None else kv_rows
    skv_stride = Skv if skv_stride is None else skv_stride
    check(lib().dmx_attention_fwd(ptr(q), q.stride(0), ptr(k), k.stride(0), kv_rows, ptr(vt), vt.stride(0), skv_stride,
                                  ptr(o), H * 64, B, H, Sq, Skv, float(scale), current_stream()), "attention_fwd")
    return o


def attention_v(q, k, v, B, H, Sq, Skv, scale, kv_rows=None):
    """q [B*Sq, >=H*64], k / v [B*kv_rows, >=H*64] row-major 2-D views (V read through LDS transpose reads)."""
    o = torch.empty(B * Sq, H * 64, dtype=h16(), device=q.device)
    kv_rows = Skv if kv_rows is None else kv_rows
    check(lib().dmx_attention_fwd_v(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), kv_rows,
                                    ptr(o), H * 64, B, H, Sq, Skv, float(scale), current_stream()), "attention_fwd_v")
    return o


def attention_v_balanced(q, k, v, B, H, Sq, Skv, scale, kv_rows=None):
    """attention_v on the balanced schedule (attention_sk.hip); None when the plan keeps the plain grid for this problem (dmx_set_attn_balanced)"""
    wsb = lib().dmx_attention_fwd_v_balanced_workspace_bytes(B, H, Sq, Skv)
    if not wsb:
        return None
    o = torch.empty(B * Sq, H * 64, dtype=h16(), device=q.device)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    kv_rows = Skv if kv_rows is None else kv_rows
    check(lib().dmx_attention_fwd_v_balanced(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), kv_rows,
                                             ptr(o), H * 64, B, H, Sq, Skv, float(scale), ptr(ws), wsb, current_stream()), "attention_fwd_v_balanced")
    return o


def attention_wide(q, k, v, B, Sq, Skv, D, scale, kv_rows=None):
    """single head of width D (128 / 256 / 512): q [B*Sq, >=D], k / v [B*kv_rows, >=D] row-major 2-D views -> [B*Sq, D]"""
    o = torch.empty(B * Sq, D, dtype=h16(), device=q.device)
    check(lib().dmx_attention_wide(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), Skv if kv_rows is None else kv_rows,
                                   ptr(o), D, B, Sq, Skv, D, float(scale), current_stream()), "attention_wide")
    return o


def attention_train(q, k, v, B, H, Sq, Skv, scale, kv_rows=None):
    """forward with row-major V that also returns lse [B,H,Sq] (log2 domain)"""
    o = torch.empty(B * Sq, H * 64, dtype=h16(), device=q.device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device)
    check(lib().dmx_attention_fwd_train(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), kv_rows or Skv,
                                        ptr(o), H * 64, ptr(lse), B, H, Sq, Skv, float(scale), current_stream()), "attention_fwd_train")
    return o, lse


def attention_bwd(q, k, v, o, do, lse, B, H, Sq, Skv, scale, kv_rows=None):
    dq = torch.empty_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    wsb = lib().dmx_attention_bwd_workspace_bytes(B, H, Sq)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    check(lib().dmx_attention_bwd(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v), v.stride(0), kv_rows or Skv,
                                  ptr(o), ptr(do), o.stride(0), ptr(lse), ptr(dq), dq.stride(0), ptr(dk), dk.stride(0), ptr(dv), dv.stride(0),
                                  B, H, Sq, Skv, float(scale), ptr(ws), wsb, current_stream()), "attention_bwd")
    return dq, dk, dv


def im2col_small(sources=None, nhwc=None, ksize=3, stride=1, pad=1, Kpad=64):
    if nhwc is not None:
        B, H, W, C = nhwc.shape
        args = [None, 0, None, 0, None, 0, ptr(nhwc), _ld(nhwc)]
    else:
        srcs = [s.to(torch.float32).contiguous() for s in sources]
        B, _, H, W = srcs[0].shape
        C = sum(s.shape[1] for s in srcs)
        args = []
        for i in range(3):
            args += [ptr(srcs[i]), srcs[i].shape[1]] if i < len(srcs) else [None, 0]
        args += [None, 0]
    OH, OW = (H // stride, W // stride)
    dev = nhwc.device if nhwc is not None else srcs[0].device
    out = torch.empty(B, OH, OW, Kpad, dtype=h16(), device=dev)
    check(lib().dmx_im2col_small(*args, C, B, H, W, OH, OW, ksize, stride, pad, ptr(out), Kpad, current_stream()), "im2col_small")
    return out


def timestep_embedding(t, freq, B, dim):
    out = torch.empty(B, dim, dtype=torch.float32, device=t.device)
    check(lib().dmx_timestep_embedding(ptr(t), t.numel(), ptr(freq), B, dim, ptr(out), current_stream()), "timestep_embedding")
    return out


def linear_small(x, w, bias=None, silu_in=False):
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(lib().dmx_linear_small(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(y), N, B, N, K, int(silu_in),
                                 current_stream()), "linear_small")
    return y
