"""GPU parity of the training (backward) kernels against torch autograd on the CPU oracle ops.

Inputs are rounded to bf16 first, so the only differences are fp32 accumulation order (and, where a kernel rounds an
intermediate to bf16, that rounding): tolerances are stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import assert_close, bf, seeded

pytestmark = pytest.mark.gpu
TOL_W = 2e-3      # fp32 accumulation of bf16 products over up to 16k rows


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def nhwc(x, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)


def unpack_dw(dw, Cout, Cin, ks):
    """packed [N][tap*Cin + c] -> [Cout][Cin][ks][ks]"""
    return dw.view(Cout, ks, ks, Cin).permute(0, 3, 1, 2).contiguous()


WGRAD_CASES = [
    # name, B, H, W, Cin, Cout, kwargs
    ("3x3_64_128", 2, 16, 16, 64, 128, {}),
    ("3x3_320_320", 1, 32, 32, 320, 320, {}),
    ("3x3_tailN_40", 2, 8, 8, 128, 40, {}),
    ("3x3_ragged_rows", 3, 5, 7, 64, 64, {}),
    ("3x3_s2", 2, 16, 16, 64, 64, dict(stride=2)),
    ("3x3_ups", 2, 8, 8, 64, 128, dict(ups=True)),
    ("3x3_concat", 2, 8, 8, 192, 64, dict(split=128)),
    ("1x1_direct", 2, 8, 8, 256, 192, dict(ksize=1, pad=0)),
    ("linear_big_rows", 1, 64, 64, 128, 320, dict(ksize=1, pad=0)),
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad(cuda, case):
    from diffute_amd import ops
    name, B, H, W, Cin, Cout, kw = case
    ks = kw.get("ksize", 3); st = kw.get("stride", 1); pad = kw.get("pad", 1); ups = kw.get("ups", False)
    x = bf(seeded((B, Cin, H, W), 1))
    w = bf(seeded((Cout, Cin, ks, ks), 2, 1.0 / math.sqrt(Cin * ks * ks))).requires_grad_(True)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xi, w, None, stride=st, padding=pad)
    dy = bf(seeded(tuple(y.shape), 3))
    y.backward(dy)
    split = kw.get("split")
    if split:
        x0, x1 = nhwc(x[:, :split], cuda), nhwc(x[:, split:], cuda)
    else:
        x0, x1 = nhwc(x, cuda), None
    dw = ops.conv_wgrad(x0, nhwc(dy, cuda), x1=x1, ksize=ks, stride=st, pad=pad, ups=ups)
    assert_close(unpack_dw(dw, Cout, Cin, ks), w.grad, TOL_W, name)
    # accumulate=1 adds onto an existing gradient (gradient accumulation, train_diffute_v1.py:874)
    dw2 = ops.conv_wgrad(x0, nhwc(dy, cuda), x1=x1, ksize=ks, stride=st, pad=pad, ups=ups, into=dw.clone())
    assert_close(dw2, 2.0 * dw.cpu(), 1e-6, name + " accumulate")


def test_colsum(cuda):
    from diffute_amd import ops
    dy = bf(seeded((4, 24, 24, 200), 5))
    got = ops.colsum(dy.to(cuda).to(torch.bfloat16), groups=1)
    assert_close(got, dy.reshape(-1, 200).sum(0, keepdim=True), 1e-5, "bias grad")
    got = ops.colsum(dy.to(cuda).to(torch.bfloat16), groups=4)
    assert_close(got, dy.reshape(4, -1, 200).sum(1), 1e-5, "row-bias grad")


TOL_D = 1e-3      # one bf16 rounding of the fp32-accumulated result
DGRAD_CASES = [
    ("3x3_64_128", 2, 16, 16, 64, 128, {}),
    ("3x3_320_640", 1, 16, 16, 320, 640, {}),
    ("3x3_s2", 2, 16, 16, 64, 128, dict(stride=2)),
    ("3x3_ups", 2, 8, 8, 64, 128, dict(ups=True)),
    ("1x1", 2, 8, 8, 256, 192, dict(ksize=1)),
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv_dgrad(cuda, case):
    from diffute_amd import ops
    name, B, H, W, Cin, Cout, kw = case
    ks = kw.get("ksize", 3); st = kw.get("stride", 1); ups = kw.get("ups", False)
    x = bf(seeded((B, Cin, H, W), 1)).requires_grad_(True)
    w = bf(seeded((Cout, Cin, ks, ks), 2, 1.0 / math.sqrt(Cin * ks * ks)))
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xi, w, None, stride=st, padding=ks // 2)
    dy = bf(seeded(tuple(y.shape), 3))
    y.backward(dy)
    wt = ops.pack_conv_weight_t(w.to(cuda))
    dx = ops.conv_dgrad(nhwc(dy, cuda), wt, Cin, ksize=ks, stride=st, ups=ups)
    assert_close(dx.permute(0, 3, 1, 2), bf(x.grad), TOL_D, name)
    r = bf(seeded((B, Cin, H, W), 4))
    dx2 = ops.conv_dgrad(nhwc(dy, cuda), wt, Cin, ksize=ks, stride=st, ups=ups, res=nhwc(r, cuda))
    assert_close(dx2.permute(0, 3, 1, 2), bf(x.grad + r), 2 * TOL_D, name + " +res")


def test_linear_dgrad_and_wgrad(cuda):
    from diffute_amd import ops
    M, K, N = 1000, 320, 1280
    x = bf(seeded((M, K), 1)).requires_grad_(True)
    w = bf(seeded((N, K), 2, 1.0 / math.sqrt(K))).requires_grad_(True)
    y = x @ w.t()
    dy = bf(seeded((M, N), 3))
    y.backward(dy)
    dyd = dy.to(cuda).to(torch.bfloat16)
    dx = ops.linear(dyd, ops.pack_linear_weight_t(w.detach().to(cuda)))
    assert_close(dx, bf(x.grad), TOL_D, "linear dgrad")
    dw = ops.conv_wgrad(x.detach().to(cuda).to(torch.bfloat16).view(1, 1, M, K), dyd.view(1, 1, M, N), ksize=1, pad=0)
    assert_close(dw, w.grad, TOL_W, "linear wgrad")


TOL_N = 4e-3      # dx is rounded to bf16 once; du / xhat are recomputed from bf16 x in fp32
GN_CASES = [
    # name, B, H, W, C, groups, silu, split (None or c0)
    ("320_silu", 2, 16, 16, 320, 32, True, None),
    ("960_concat_silu", 2, 8, 8, 960, 32, True, 640),
    ("1280_plain", 1, 8, 8, 1280, 32, False, None),
    ("2560_concat", 2, 4, 4, 2560, 32, True, 1280),
    ("64_rows_4096", 1, 64, 64, 64, 32, True, None),
]


@pytest.mark.parametrize("case", GN_CASES, ids=[c[0] for c in GN_CASES])
def test_groupnorm_bwd(cuda, case):
    from diffute_amd import ops
    name, B, H, W, C, G, silu, split = case
    x = bf(seeded((B, C, H, W), 1) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1.0 + 0.1 * seeded((C,), 2)).requires_grad_(True)
    beta = (0.1 * seeded((C,), 3)).requires_grad_(True)
    u = F.group_norm(x, G, gamma, beta, eps=1e-5)
    y = F.silu(u) if silu else u
    dy = bf(seeded((B, C, H, W), 4))
    r = bf(seeded((B, C, H, W), 5))
    y.backward(dy)
    xs = [nhwc(x.detach()[:, :split], cuda), nhwc(x.detach()[:, split:], cuda)] if split else [nhwc(x.detach(), cuda), None]
    yh, stats = ops.groupnorm_train(xs[0], gamma.detach().to(cuda), beta.detach().to(cuda), G, 1e-5, silu, x1=xs[1])
    assert_close(yh.permute(0, 3, 1, 2), bf(y.detach()), 4e-3, name + " fwd")
    rs = [nhwc(r[:, :split], cuda), nhwc(r[:, split:], cuda)] if split else [nhwc(r, cuda), None]
    dx0, dx1, dg, db = ops.groupnorm_bwd(xs[0], nhwc(dy, cuda), gamma.detach().to(cuda), beta.detach().to(cuda), G, silu, stats,
                                         x1=xs[1], res0=rs[0], res1=rs[1])
    dx = torch.cat([dx0, dx1], dim=-1) if split else dx0
    assert_close(dx.permute(0, 3, 1, 2), bf(x.grad + r), TOL_N, name + " dx")
    assert_close(dg, gamma.grad, TOL_N, name + " dgamma")
    assert_close(db, beta.grad, TOL_N, name + " dbeta")


@pytest.mark.parametrize("rows,C", [(1000, 320), (256, 1280), (70, 640), (33, 2048)])
def test_layernorm_bwd(cuda, rows, C):
    from diffute_amd import ops
    x = bf(seeded((rows, C), 1) * 2.0 + 0.5).requires_grad_(True)
    gamma = (1.0 + 0.1 * seeded((C,), 2)).requires_grad_(True)
    beta = (0.1 * seeded((C,), 3)).requires_grad_(True)
    y = F.layer_norm(x, (C,), gamma, beta, eps=1e-5)
    dy = bf(seeded((rows, C), 4)); r = bf(seeded((rows, C), 5))
    y.backward(dy)
    xd = x.detach().to(cuda).to(torch.bfloat16)
    dx, dg, db = ops.layernorm_bwd(xd, dy.to(cuda).to(torch.bfloat16), gamma.detach().to(cuda), res=r.to(cuda).to(torch.bfloat16))
    assert_close(dx, bf(x.grad + r), TOL_N, "ln dx")
    assert_close(dg, gamma.grad, TOL_N, "ln dgamma")
    assert_close(db, beta.grad, TOL_N, "ln dbeta")


def test_geglu_fwd_bwd(cuda):
    from diffute_amd import ops
    rows, C2 = 300, 1280
    h = bf(seeded((rows, 2 * C2), 1) * 1.5).requires_grad_(True)
    a, g = h.chunk(2, dim=-1)
    y = a * F.gelu(g)
    dy = bf(seeded((rows, C2), 2))
    y.backward(dy)
    hd = h.detach().to(cuda).to(torch.bfloat16)
    assert_close(ops.geglu_fwd(hd), bf(y.detach()), 4e-3, "geglu fwd")
    assert_close(ops.geglu_bwd(hd, dy.to(cuda).to(torch.bfloat16)), bf(h.grad), 4e-3, "geglu bwd")


TOL_A = 1.5e-2    # P and dS pass through bf16 (rel 2^-9 each) before the second MFMA, like the forward kernel
ATTN_CASES = [("self_256", 2, 5, 256, 256), ("self_1024", 1, 10, 1024, 1024), ("cross_577", 2, 5, 256, 577),
              ("tiny_64", 2, 20, 64, 64), ("ragged_200_150", 1, 3, 200, 150)]


@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention_bwd(cuda, case):
    from diffute_amd import ops
    name, B, H, Sq, Skv = case
    D = 64
    q = bf(seeded((B, Sq, H * D), 1)).requires_grad_(True)
    k = bf(seeded((B, Skv, H * D), 2)).requires_grad_(True)
    v = bf(seeded((B, Skv, H * D), 3)).requires_grad_(True)
    def heads(x, S):
        return x.view(B, S, H, D).transpose(1, 2)
    o = F.scaled_dot_product_attention(heads(q, Sq), heads(k, Skv), heads(v, Skv), scale=0.125)
    o = o.transpose(1, 2).reshape(B, Sq, H * D)
    do = bf(seeded((B, Sq, H * D), 4))
    o.backward(do)
    dev = lambda x, S: x.detach().reshape(B * S, H * D).to(cuda).to(torch.bfloat16)
    qd, kd, vd = dev(q, Sq), dev(k, Skv), dev(v, Skv)
    oh, lse = ops.attention_train(qd, kd, vd, B, H, Sq, Skv, 0.125)
    assert_close(oh.view(B, Sq, H * D), bf(o.detach()), 4e-3, name + " fwd")
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, oh, dev(do, Sq), lse, B, H, Sq, Skv, 0.125)
    assert_close(dq.view(B, Sq, H * D), q.grad, TOL_A, name + " dq")
    assert_close(dk.view(B, Skv, H * D), k.grad, TOL_A, name + " dk")
    assert_close(dv.view(B, Skv, H * D), v.grad, TOL_A, name + " dv")
