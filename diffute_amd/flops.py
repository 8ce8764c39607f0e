"""Algorithmic work of the hot path (2*MACs of every conv / linear / attention matmul; norms and
activations excluded, <0.1 %) - the figure roofline numbers are computed from."""


def unet_flops(cfg, B, h, w, ctx_len=577, cached_ctx_kv=True, phase_upsample=False):
    """phase_upsample: count the upsampler convs as executed by dmx_conv_ups2x (4 instead of 9 taps per output)."""
    boc = tuple(cfg.block_out_channels); L = cfg.layers_per_block; ctxd = cfg.cross_attention_dim
    down_attn = [t.startswith("CrossAttn") for t in cfg.down_block_types]
    up_attn = [t.startswith("CrossAttn") for t in cfg.up_block_types]

    def conv(hw, cin, cout, k=3): return 2 * B * hw * cin * cout * k * k
    def lin(m, cin, cout): return 2 * m * cin * cout

    def res(hw, cin, cout):
        x = conv(hw, cin, cout) + conv(hw, cout, cout) + lin(B, boc[0] * 4, cout)
        return x + (conv(hw, cin, cout, 1) if cin != cout else 0)

    def xf(hw, c):
        m = B * hw
        x = 2 * lin(m, c, c) + 4 * lin(m, c, c) + 4 * B * hw * hw * c
        x += 2 * lin(m, c, c) + 4 * B * hw * ctx_len * c
        if not cached_ctx_kv:
            x += 2 * lin(B * ctx_len, ctxd, c)
        return x + lin(m, c, 8 * c) + lin(m, 4 * c, c)

    f = lin(B, boc[0], boc[0] * 4) + lin(B, boc[0] * 4, boc[0] * 4)
    hw = h * w
    f += conv(hw, cfg.in_channels, boc[0])
    skips = [boc[0]]; cprev = boc[0]
    for i, c in enumerate(boc):
        for _ in range(L):
            f += res(hw, cprev, c) + (xf(hw, c) if down_attn[i] else 0)
            cprev = c; skips.append(c)
        if i < len(boc) - 1:
            hw //= 4; f += conv(hw, c, c); skips.append(c)
    f += 2 * res(hw, cprev, cprev) + xf(hw, cprev)
    for i, c in enumerate(reversed(boc)):
        for _ in range(L + 1):
            f += res(hw, cprev + skips.pop(), c) + (xf(hw, c) if up_attn[i] else 0)
            cprev = c
        if i < len(boc) - 1:
            hw *= 4; f += conv(hw, c, c) * 4 // 9 if phase_upsample else conv(hw, c, c)
    return f + conv(hw, boc[0], cfg.out_channels)


def context_kv_flops(cfg, B, ctx_len=577):
    """One-off cross-attention K/V projections of the glyph context (per image, not per step)."""
    boc = tuple(cfg.block_out_channels); L = cfg.layers_per_block
    down_attn = [t.startswith("CrossAttn") for t in cfg.down_block_types]
    up_attn = [t.startswith("CrossAttn") for t in cfg.up_block_types]
    f = 0
    for i, c in enumerate(boc):
        if down_attn[i]:
            f += L * 2 * 2 * B * ctx_len * cfg.cross_attention_dim * c
    f += 2 * 2 * B * ctx_len * cfg.cross_attention_dim * boc[-1]
    for i, c in enumerate(reversed(boc)):
        if up_attn[i]:
            f += (L + 1) * 2 * 2 * B * ctx_len * cfg.cross_attention_dim * c
    return f
