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


# round 5: network shapes beyond BASELINE.json, goldens from the reference's Python (tests/golden/gen_nn_golden.py): the reference's default 1 block x 256 channels
# (config/configuration.cpp:70-72), wider towers, 7x7 .. 19x19 Go (go_unit.h:11), channel counts that are no multiple of 16 (run-time-shaped kernels)
WIDE_NN_CFG = {
    "w_go9_1bx256_az": ("go_9x9", 18, 9, 9, 256, 9, 9, 1, 1, 82, 256, 1, "alphazero"),
    "w_go9_6bx128_az": ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "w_go19_6bx64_az": ("go_19x19", 18, 19, 19, 64, 19, 19, 1, 6, 362, 256, 1, "alphazero"),
    "w_go7_2bx32_az": ("go_7x7", 18, 7, 7, 32, 7, 7, 1, 2, 50, 256, 1, "alphazero"),
    "w_go13_2bx96_az": ("go_13x13", 18, 13, 13, 96, 13, 13, 1, 2, 170, 64, 1, "alphazero"),
    "w_go5_3bx24_az": ("go_5x5", 18, 5, 5, 24, 5, 5, 1, 3, 26, 20, 1, "alphazero"),
    "w_go9_2bx128_mz": ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 2, 82, 256, 1, "muzero"),
    "w_go7_1bx40_mz": ("go_7x7", 18, 7, 7, 40, 7, 7, 1, 1, 50, 32, 1, "muzero"),
    # the reference's default network (1 block x 256 channels) on the other two games with a device leaf: Othello, TicTacToe (docs/Training.md:23)
    "w_oth8_1bx256_az": ("othello_8x8", 4, 8, 8, 256, 8, 8, 1, 1, 65, 256, 1, "alphazero"),
    "w_ttt_1bx256_az": ("tictactoe", 4, 3, 3, 256, 3, 3, 1, 1, 9, 256, 1, "alphazero"),
}
