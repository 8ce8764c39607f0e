"""Deterministic synthetic weights (no checkpoints exist offline: pretrained/.gitkeep only).

Counter-based murmur3-style PRNG evaluated with torch int64 ops masked to 32 bits, so the same
values come out on CPU and on the GPU and match the oracle's numpy restatement bit-for-bit.
Init scheme: conv/linear ~ U(-a,a), a=sqrt(3/fan_in); norm gamma 1+0.1u, beta 0.1u;
other biases 0.05u, u ~ U(-1,1).
"""
import math
import zlib

import torch

_M = 0xFFFFFFFF


def tensor_id(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & _M


def uniform01(seed: int, tid: int, n: int, device="cpu") -> torch.Tensor:
    i = torch.arange(n, dtype=torch.int64, device=device)
    key = ((seed * 0x9E3779B1) ^ (tid * 0x85EBCA77)) & _M
    x = (i * 0x9E3779B1 + key) & _M
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) * (1.0 / (1 << 24))


def normal(seed: int, tid: int, n: int, device="cpu") -> torch.Tensor:
    """Box-Muller N(0,1) (synthetic inputs)."""
    u1 = uniform01(seed, tid, n, device).to(torch.float64)
    u2 = uniform01(seed ^ 0x5BD1E995, tid, n, device).to(torch.float64)
    r = torch.sqrt(-2.0 * torch.log(1.0 - u1))
    return (r * torch.cos(2.0 * math.pi * u2)).to(torch.float32)


def init_param(name: str, shape, seed: int = 1234, device="cpu") -> torch.Tensor:
    n = 1
    for s in shape:
        n *= int(s)
    u = uniform01(seed, tensor_id(name), n, device) * 2.0 - 1.0
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= int(s)
        v = u * torch.tensor(math.sqrt(3.0 / fan_in), dtype=torch.float32, device=u.device)
    elif name.endswith("weight"):
        v = 1.0 + 0.1 * u
    elif "norm" in name.rsplit(".", 2)[-2]:
        v = 0.1 * u
    else:
        v = 0.05 * u
    return v.to(torch.float32).reshape(tuple(shape))
