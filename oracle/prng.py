"""Counter-based PRNG shared (by restatement) between oracle and product.

value(seed, tensor_id, i) is a murmur3-style 32-bit finaliser over the counter;
it is reproducible bit-for-bit in numpy, torch (int64 arithmetic masked to 32
bits, CPU or GPU) and C++.  Oracle copy: numpy.  The product has its own copy in
diffute_amd/init.py; tests assert the two agree.
"""
import zlib
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def tensor_id(name: str) -> int:
    """Stable 32-bit id of a parameter name (crc32 of the utf-8 key)."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def uniform01(seed: int, tid: int, n: int) -> np.ndarray:
    """n floats in [0,1) with 24 random bits each (exact in fp32)."""
    i = np.arange(n, dtype=np.uint64)
    key = np.uint64(((seed * 0x9E3779B1) ^ (tid * 0x85EBCA77)) & 0xFFFFFFFF)
    x = (i * np.uint64(0x9E3779B1) + key) & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M
    x ^= x >> np.uint64(16)
    return ((x >> np.uint64(8)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def normal(seed: int, tid: int, n: int) -> np.ndarray:
    """Box-Muller N(0,1) in fp32 from two uniform streams (synthetic inputs only)."""
    u1 = uniform01(seed, tid, n).astype(np.float64)
    u2 = uniform01(seed ^ 0x5BD1E995, tid, n).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)
