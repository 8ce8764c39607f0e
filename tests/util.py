"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
import torch


def bf(x):
    """round to bf16 and back (fp32 container)"""
    return x.to(torch.bfloat16).to(torch.float32)


def rel_l2(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def assert_close(hip, ref, tol, name=""):
    """relative L2 error of the HIP result vs the oracle <= tol, and everything finite."""
    h = hip.detach().float().cpu()
    assert h.shape == ref.shape, f"{name}: shape {tuple(h.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(h).all(), f"{name}: non-finite values in HIP output"
    e = rel_l2(h, ref)
    assert e <= tol, f"{name}: rel-L2 {e:.3e} > {tol:.1e} (max abs diff {float((h - ref.float()).abs().max()):.3e})"
    return e


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale
