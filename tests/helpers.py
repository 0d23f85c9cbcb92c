"""Shared helpers of the parity tests (pure numpy; no reference code)."""
import numpy as np

M64 = (1 << 64) - 1


def mix64(z):
    z = np.asarray(z, np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def counter_u01(seed, n):
    with np.errstate(over="ignore"):
        idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & M64)
        z = mix64(idx)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def binary_planes(seed, shape):
    """Same input generator as tests/golden/gen_nn_golden.py."""
    n = int(np.prod(shape))
    return (counter_u01(seed, n) < 0.3).astype(np.float32).reshape(shape)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def frac_bit_equal(a, b):
    return float(np.mean(bits(a) == bits(b)))
