"""GPU parity of the inference path (HIP, through the C ABI) against the CPU oracle and the reference goldens.

Tolerances: north_star asks for 1e-3 (fp32 reference).  The HIP kernels follow the oracle's arithmetic
order (DESIGN.md "Network numerics"), so we assert 1e-5 against the oracle and 1e-4 against the
reference-generated goldens, and additionally require bit-exactness where the order contract applies.
"""
import numpy as np
import pytest

from helpers import binary_planes, counter_u01, same_bits, frac_bit_equal

pytestmark = pytest.mark.gpu

GOLDEN_CFG = {
    "c1_tictactoe_az": ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"),
    "c2_go_az": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "c3_othello_az": ("othello_8x8", 4, 8, 8, 64, 8, 8, 1, 6, 65, 256, 1, "alphazero"),
    "c4_go_mz": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "muzero"),
    "small_go_az": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"),
}


def _descs(mz, oracle, args):
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    return mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)


@pytest.mark.parametrize("name", ["c1_tictactoe_az", "c2_go_az", "c3_othello_az", "small_go_az"])
@pytest.mark.parametrize("batch", [1, 5, 64])
def test_az_forward_matches_oracle(mz, oracle, name, batch):
    args = GOLDEN_CFG[name]
    d, od = _descs(mz, oracle, args)
    w = mz.generate_weights(d, 0)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    x = binary_planes(77 + batch, (batch, args[1] * args[2] * args[3]))
    p, l, v = net.forward(x)
    op, ol, ov = onet.forward_az(x)
    assert np.abs(l - ol).max() <= 1e-5 and np.abs(p - op).max() <= 1e-5 and np.abs(v - ov).max() <= 1e-5
    # order contract: the f32 MFMA chain == the oracle's fmaf chain
    assert same_bits(l, ol), f"logits not bit-exact: {frac_bit_equal(l, ol):.4f} equal"
    assert same_bits(p, op), f"policy not bit-exact: {frac_bit_equal(p, op):.4f} equal"
    assert same_bits(v, ov), f"value not bit-exact: {frac_bit_equal(v, ov):.4f} equal"


@pytest.mark.parametrize("name", ["c1_tictactoe_az", "c2_go_az", "c3_othello_az", "small_go_az"])
def test_az_forward_matches_reference_golden(mz, name):
    import os
    args = GOLDEN_CFG[name]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"nn_{name}.npz"))
    d = mz.make_desc(*args[:10], vh=args[10], dv=args[11], type_name=args[12])
    net = mz.Net(d, mz.generate_weights(d, int(g["weight_seed"])))
    for B in (1, 3):
        x = binary_planes(int(g[f"b{B}_input_seed"]), (B, args[1] * args[2] * args[3]))
        p, l, v = net.forward(x)
        assert np.abs(p - g[f"b{B}_policy"]).max() <= 1e-4
        assert np.abs(l - g[f"b{B}_policy_logit"]).max() <= 1e-4
        assert np.abs(v - g[f"b{B}_value"]).max() <= 1e-4


def test_muzero_initial_and_recurrent(mz, oracle):
    import os
    args = GOLDEN_CFG["c4_go_mz"]
    d, od = _descs(mz, oracle, args)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nn_c4_go_mz.npz"))
    w = mz.generate_weights(d, 0)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    for B in (1, 3):
        x = binary_planes(int(g[f"b{B}_input_seed"]), (B, 18 * 81))
        p, l, v, h = net.initial_inference(x)
        op, ol, ov, oh = onet.initial(x)
        for a, b in ((p, op), (l, ol), (v, ov), (h, oh)):
            assert np.abs(a - b).max() <= 1e-5
        assert same_bits(h, oh) and same_bits(l, ol) and same_bits(p, op) and same_bits(v, ov)
        assert np.abs(h - g[f"b{B}_init_hidden_state"]).max() <= 1e-4
        assert np.abs(p - g[f"b{B}_init_policy"]).max() <= 1e-4 and np.abs(v - g[f"b{B}_init_value"]).max() <= 1e-4
        act = np.zeros((B, 81), np.float32)
        for b in range(B):
            act[b, (7 * b + 3) % 81] = 1.0
        hin = g[f"b{B}_init_hidden_state"]
        p2, l2, v2, r2, h2 = net.recurrent_inference(hin, act)
        op2, ol2, ov2, or2, oh2 = onet.recurrent(hin, act)
        assert same_bits(h2, oh2) and same_bits(l2, ol2) and same_bits(p2, op2) and same_bits(v2, ov2)
        assert np.all(r2 == 0.0)
        assert np.abs(h2 - g[f"b{B}_rec_hidden_state"]).max() <= 1e-4
        assert np.abs(p2 - g[f"b{B}_rec_policy"]).max() <= 1e-4 and np.abs(v2 - g[f"b{B}_rec_value"]).max() <= 1e-4


def test_full_batch_c2_properties(mz, oracle):
    """BASELINE size (256 x Go 6bx64): batch-independence (sample b of a 256-batch == the same sample alone)
    and agreement with the oracle on a subset the CPU finishes in seconds."""
    d, od = mz.DESCS["c2"](), oracle.desc_c2()
    w = mz.generate_weights(d, 3)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    x = binary_planes(9, (256, 18 * 81))
    p, l, v = net.forward(x)
    assert np.all(np.isfinite(p)) and np.all(np.abs(p.sum(1) - 1) < 1e-5) and np.all(np.abs(v) <= 1)
    idx = [0, 17, 128, 255]
    p1, l1, v1 = net.forward(x[idx])
    assert same_bits(p[idx], p1) and same_bits(l[idx], l1) and same_bits(v[idx], v1)
    op, ol, ov = onet.forward_az(x[idx])
    assert same_bits(l1, ol) and same_bits(p1, op) and same_bits(v1, ov)


def test_float_inputs_and_reload(mz, oracle):
    """non-binary inputs (MuZero-like planes) and load_model on a live network"""
    d, od = mz.DESCS["c3"](), oracle.desc_c3()
    w0, w1 = mz.generate_weights(d, 0), mz.generate_weights(d, 1)
    net = mz.Net(d, w0)
    x = (counter_u01(5, 3 * 4 * 64) * 2 - 1).reshape(3, -1).astype(np.float32)
    a = net.forward(x)
    net.reload(w1)
    b = net.forward(x)
    ob = oracle.OracleNet(od, w1).forward_az(x)
    assert not same_bits(a[1], b[1])
    assert same_bits(b[1], ob[1]) and same_bits(b[0], ob[0]) and same_bits(b[2], ob[2])


def test_errors_are_loud(mz):
    d = mz.DESCS["c1"]()
    w = mz.generate_weights(d, 0)
    with pytest.raises(mz.MzError):
        mz.Net(d, w[:-1])  # wrong blob size
    with pytest.raises(mz.MzError):
        mz.Net(d, w, device=99)
    with pytest.raises(mz.MzError):
        mz.Net(d, w).initial_inference(np.zeros((1, 36), np.float32))  # MuZero call on an AlphaZero net


ATARI = {
    "c5_atari_mz": ("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 6, 18, 256, 601, "muzero_atari"),
    "small_atari_mz": ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"),
}


@pytest.mark.parametrize("name", sorted(ATARI))
def test_muzero_atari_network(mz, oracle, name):
    """BASELINE configs[4] network: tiled strided convs, average pools, fused 6x6 towers, 601-bin value / reward heads"""
    import os
    args = ATARI[name]
    d, od = _descs(mz, oracle, args)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"nn_{name}.npz"))
    w = mz.generate_weights(d, 0)
    net, onet = mz.Net(d, w), oracle.OracleNet(od, w)
    for B in (1, 2):
        x = counter_u01(int(g[f"b{B}_input_seed"]), B * 32 * 96 * 96).reshape(B, -1).astype(np.float32)
        p, l, v, h = net.initial_inference(x)
        op, ol, ov, oh = onet.initial(x)
        assert same_bits(h, oh), f"hidden not bit-exact: {frac_bit_equal(h, oh):.4f}"
        assert same_bits(l, ol) and same_bits(p, op) and same_bits(v, ov)
        assert np.abs(h - g[f"b{B}_init_hidden_state"]).max() <= 1e-4 and np.abs(p - g[f"b{B}_init_policy"]).max() <= 1e-4
        assert np.abs(v - onet.invert(g[f"b{B}_init_value"])).max() <= 1e-3
        act = np.zeros((B, 18, 36), np.float32)
        for b in range(B):
            act[b, (7 * b + 3) % 18] = 1.0
        hin = g[f"b{B}_init_hidden_state"]
        p2, l2, v2, r2, h2 = net.recurrent_inference(hin, act.reshape(B, -1))
        op2, ol2, ov2, or2, oh2 = onet.recurrent(hin, act.reshape(B, -1))
        assert same_bits(h2, oh2) and same_bits(l2, ol2) and same_bits(p2, op2) and same_bits(v2, ov2) and same_bits(r2, or2)
        assert np.abs(h2 - g[f"b{B}_rec_hidden_state"]).max() <= 1e-4 and np.abs(p2 - g[f"b{B}_rec_policy"]).max() <= 1e-4
        assert np.abs(v2 - onet.invert(g[f"b{B}_rec_value"])).max() <= 1e-3 and np.abs(r2 - onet.invert(g[f"b{B}_rec_reward"])).max() <= 1e-3


def test_invert_value_on_device(mz):
    """The device twin of invertValue (simulation kernel, muzero_atari) against the host function, which calls the same libm as the
    reference (utils.h:102-108): bit-exact on a dense sweep of the 601-bin range and on random values."""
    rng = np.random.default_rng(5)
    v = np.concatenate([np.linspace(-300, 300, 200001, dtype=np.float32), rng.normal(0, 3, 200000).astype(np.float32),
                        rng.uniform(-300, 300, 200000).astype(np.float32), np.array([0.0, -0.0, 1e-30, -1e-30, 300.0, -300.0], np.float32)])
    dev = mz.invert_values_device(v)
    L = mz.lib.load()
    L.mz_invert_value.restype = __import__("ctypes").c_float
    L.mz_invert_value.argtypes = [__import__("ctypes").c_float]
    host = np.array([L.mz_invert_value(float(x)) for x in v[::7]], np.float32)
    assert np.array_equal(dev[::7].view(np.uint32), host.view(np.uint32))


def test_invert_value_on_device_matches_the_closed_form(mz):
    """... and against the closed form of utils.h:102-108 evaluated in numpy with the reference's promotions (tests/test_value_transform.py): every
    one of 600 k values, so the device function is pinned to the formula itself, not to a transcription of it."""
    from test_value_transform import closed_form_invert, vectors
    rng = np.random.default_rng(6)
    v = np.concatenate([vectors(), np.linspace(-300, 300, 200001, dtype=np.float32), rng.normal(0, 3, 200000).astype(np.float32),
                        rng.uniform(-300, 300, 200000).astype(np.float32)])
    dev = mz.invert_values_device(v)
    assert np.array_equal(dev.view(np.uint32), closed_form_invert(v).view(np.uint32))
