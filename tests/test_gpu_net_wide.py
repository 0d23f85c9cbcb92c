"""GPU parity of the inference path on shapes beyond BASELINE.json (round 5): the reference's default 1 block x 256 channels (config/configuration.cpp:70-72),
128-channel towers, 7x7 / 13x13 / 19x19 Go (go_unit.h:11) on the one-tile tower (net_wide.hip tower_wide), channel counts that are no multiple of 16 and boards
without an instance on the run-time-shaped conv3x3_band — against the CPU oracle (bit-exact: same k-ordered chains) and against goldens generated from the
reference's own Python modules (tests/golden/gen_nn_golden.py; <= 1e-4, north star 1e-3).  Math: network/py/network_unit.py:6-87, alphazero_network.py:90-113,
muzero_network.py:137-164; loadModel accepts any of these shapes (network/network.cpp:14-42)."""
import os

import numpy as np
import pytest

from helpers import WIDE_NN_CFG, binary_planes, counter_u01, same_bits, frac_bit_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
AZ = sorted(k for k, v in WIDE_NN_CFG.items() if v[12] == "alphazero")
MZ = sorted(k for k, v in WIDE_NN_CFG.items() if v[12] == "muzero")


def _descs(mz, oracle, args):
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    return mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)


@pytest.mark.parametrize("name", AZ)
def test_wide_az_forward_matches_oracle_and_reference_golden(mz, oracle, name):
    args = WIDE_NN_CFG[name]
    d, od = _descs(mz, oracle, args)
    g = np.load(os.path.join(GOLD, f"nn_{name}.npz"))
    w = mz.generate_weights(d, int(g["weight_seed"]))
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    for B in (1, 3):
        x = binary_planes(int(g[f"b{B}_input_seed"]), (B, args[1] * args[2] * args[3]))
        p, l, v = net.forward(x)
        assert np.abs(p - g[f"b{B}_policy"]).max() <= 1e-4 and np.abs(l - g[f"b{B}_policy_logit"]).max() <= 1e-4 and np.abs(v - g[f"b{B}_value"]).max() <= 1e-4
        op, ol, ov = onet.forward_az(x)
        assert same_bits(l, ol), f"logits not bit-exact: {frac_bit_equal(l, ol):.4f} equal, max diff {np.abs(l - ol).max():.3g}"
        assert same_bits(p, op) and same_bits(v, ov)


@pytest.mark.parametrize("name", AZ)
def test_wide_az_batches_are_sample_independent(mz, oracle, name):
    """A batch of 37 (more workgroups than one wave of CUs for none of them, but uneven) == its samples alone; float (non 0/1) planes against the oracle."""
    args = WIDE_NN_CFG[name]
    d, od = _descs(mz, oracle, args)
    w = mz.generate_weights(d, 5)
    net = mz.Net(d, w)
    n = args[1] * args[2] * args[3]
    x = binary_planes(3, (37, n))
    p, l, v = net.forward(x)
    idx = [0, 13, 36]
    p1, l1, v1 = net.forward(x[idx])
    assert same_bits(p[idx], p1) and same_bits(l[idx], l1) and same_bits(v[idx], v1)
    xf = (counter_u01(8, 2 * n) * 2 - 1).reshape(2, n).astype(np.float32)
    a, b = net.forward(xf), oracle.OracleNet(od, w).forward_az(xf)
    assert same_bits(a[1], b[1]) and same_bits(a[0], b[0]) and same_bits(a[2], b[2])


@pytest.mark.parametrize("name", MZ)
def test_wide_muzero_initial_and_recurrent(mz, oracle, name):
    args = WIDE_NN_CFG[name]
    d, od = _descs(mz, oracle, args)
    g = np.load(os.path.join(GOLD, f"nn_{name}.npz"))
    w = mz.generate_weights(d, int(g["weight_seed"]))
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    P = args[5] * args[6]
    for B in (1, 3):
        x = binary_planes(int(g[f"b{B}_input_seed"]), (B, args[1] * args[2] * args[3]))
        p, l, v, h = net.initial_inference(x)
        op, ol, ov, oh = onet.initial(x)
        assert same_bits(h, oh), f"hidden not bit-exact: {frac_bit_equal(h, oh):.4f}"
        assert same_bits(l, ol) and same_bits(p, op) and same_bits(v, ov)
        assert np.abs(h - g[f"b{B}_init_hidden_state"]).max() <= 1e-4 and np.abs(p - g[f"b{B}_init_policy"]).max() <= 1e-4 and np.abs(v - g[f"b{B}_init_value"]).max() <= 1e-4
        act = np.zeros((B, P), np.float32)
        for b in range(B):
            act[b, (7 * b + 3) % P] = 1.0
        hin = g[f"b{B}_init_hidden_state"]
        p2, l2, v2, r2, h2 = net.recurrent_inference(hin, act)
        op2, ol2, ov2, or2, oh2 = onet.recurrent(hin, act)
        assert same_bits(h2, oh2) and same_bits(l2, ol2) and same_bits(p2, op2) and same_bits(v2, ov2) and np.all(r2 == 0.0)
        assert np.abs(h2 - g[f"b{B}_rec_hidden_state"]).max() <= 1e-4 and np.abs(p2 - g[f"b{B}_rec_policy"]).max() <= 1e-4 and np.abs(v2 - g[f"b{B}_rec_value"]).max() <= 1e-4


@pytest.mark.parametrize("args", [
    ("go_19x19", 18, 19, 19, 128, 19, 19, 1, 1, 362, 64, 1, "alphazero"),   # one tile would be 237 KB: per-layer run-time-shaped kernels, heads read the activations in global memory
    ("go_11x11", 18, 11, 11, 48, 11, 11, 1, 2, 122, 32, 1, "alphazero"),
    ("othello_10x10", 4, 10, 10, 20, 10, 10, 1, 2, 101, 24, 1, "alphazero"),
    ("go_6x6", 18, 6, 6, 72, 6, 6, 1, 1, 37, 16, 1, "muzero"),
    ("go_2x2", 18, 2, 2, 4, 2, 2, 1, 1, 5, 3, 1, "alphazero"),
    # conv3x3_band's shapes: five bands of 4 / 3 rows with the 147-KB patch (19x19 x 256), bands of 12 + 1 rows (13x13 x 96), a layer whose k-steps are no multiple of the
    # chunk of 8 with three jobs on eight waves (7x7 x 36: 9 channel groups), a board wider than a workgroup has threads for one patch row pass (25x25: 27 x 27 positions)
    ("go_19x19", 18, 19, 19, 256, 19, 19, 1, 1, 362, 64, 1, "alphazero"),
    ("go_13x13", 18, 13, 13, 96, 13, 13, 1, 1, 170, 32, 1, "alphazero"),
    ("go_7x7", 18, 7, 7, 36, 7, 7, 1, 1, 50, 16, 1, "muzero"),
    ("go_25x25", 18, 25, 25, 40, 25, 25, 1, 1, 626, 16, 1, "alphazero"),
], ids=lambda a: f"{a[0]}_{a[8]}bx{a[4]}_{a[12]}")
def test_any_shape_is_served(mz, oracle, args):
    """No network create_network.py can build for a board game ends in "no kernel instance": shapes with neither a fused nor a one-tile tower instance run on
    conv3x3_band (net_wide.hip: a band of output rows staged in LDS), bit-exact against the oracle."""
    d, od = _descs(mz, oracle, args)
    w = mz.generate_weights(d, 7)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    x = binary_planes(11, (2, args[1] * args[2] * args[3]))
    if args[12] == "alphazero":
        a, b = net.forward(x), onet.forward_az(x)
        assert same_bits(a[1], b[1]) and same_bits(a[0], b[0]) and same_bits(a[2], b[2])
    else:
        p, l, v, h = net.initial_inference(x)
        op, ol, ov, oh = onet.initial(x)
        assert same_bits(h, oh) and same_bits(l, ol) and same_bits(v, ov)
        act = np.zeros((2, args[5] * args[6]), np.float32)
        act[0, 3] = 1.0  # the second sample passes (all-zero plane, ref go.cpp:310-315)
        a, b = net.recurrent_inference(h, act), onet.recurrent(h, act)
        assert same_bits(a[4], b[4]) and same_bits(a[1], b[1]) and same_bits(a[2], b[2])


def test_muzero_atari_network_of_another_width(mz, oracle):
    """muzero_atari with 96 hidden channels (no conv3x3_tiled / fused-tower instance: ref muzero_atari_network.py:21-70 builds any width): the strided
    run-time-shaped convolutions for the 96x96 representation, residual blocks layer by layer, the 601-bin heads — bit-exact against the oracle."""
    args = ("atari_ms_pacman", 32, 96, 96, 96, 6, 6, 18, 1, 18, 48, 601, "muzero_atari")
    d, od = _descs(mz, oracle, args)
    w = mz.generate_weights(d, 3)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    x = counter_u01(21, 2 * 32 * 96 * 96).reshape(2, -1).astype(np.float32)
    p, l, v, h = net.initial_inference(x)
    op, ol, ov, oh = onet.initial(x)
    assert same_bits(h, oh), f"hidden not bit-exact: {frac_bit_equal(h, oh):.4f}"
    assert same_bits(l, ol) and same_bits(p, op) and same_bits(v, ov)
    act = np.zeros((2, 18, 36), np.float32)
    act[0, 5] = 1.0
    act[1, 11] = 1.0
    a, b = net.recurrent_inference(h, act.reshape(2, -1)), onet.recurrent(h, act.reshape(2, -1))
    for u, t in zip(a, b):
        assert same_bits(u, t)
