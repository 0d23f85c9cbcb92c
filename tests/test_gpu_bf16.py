"""GPU tests of the opt-in `bf16x3` residual tower (net_bf16_body.h): split-bf16 operands on v_mfma_f32_16x16x32_bf16 with f32 accumulation.
Not bit-exact by design (the summation order inside a K = 32 MFMA cannot be mirrored on the CPU): what is asserted is the north star's
tolerance — policy / value within 1e-3 of the f32 path AND of the CPU oracle — on the BASELINE networks, and that the default stays f32."""
import numpy as np
import pytest

from helpers import binary_planes

pytestmark = pytest.mark.gpu
TOL = 1e-3  # BASELINE.json north_star: "network outputs within 1e-3 fp32"


@pytest.mark.parametrize("key,seed", [("c2", 0), ("c2", 7), ("c3", 0), ("c3", 3)])
def test_bf16x3_forward_within_tolerance(mz, oracle, key, seed):
    d, od = mz.DESCS[key](), getattr(oracle, "desc_" + key)()
    w = mz.generate_weights(d, seed)
    B = 24
    x = binary_planes(100 + seed, (B, d.num_input_channels * d.input_channel_height * d.input_channel_width))
    net = mz.Net(d, w)
    p32, l32, v32 = net.forward(x)
    net.set_precision("bf16x3")
    p16, l16, v16 = net.forward(x)
    net.set_precision("f32")
    p32b, l32b, v32b = net.forward(x)
    assert np.array_equal(l32, l32b) and np.array_equal(v32, v32b), "switching back restores the bit-exact f32 path"
    op, ol, ov = oracle.OracleNet(od, w).forward_az(x)
    for name, a, b in (("policy vs f32", p16, p32), ("value vs f32", v16, v32), ("policy vs oracle", p16, op), ("value vs oracle", v16, ov)):
        err = float(np.max(np.abs(a - b)))
        assert err <= TOL, f"{key} seed {seed}: {name} differs by {err:.3e}"
    assert float(np.max(np.abs(l16 - l32))) <= 5e-3  # logits (unnormalised): same decimal places as their magnitude allows
    assert not np.array_equal(l16, l32), "the bf16x3 path is a different arithmetic: identical bits mean it did not run"
    print(f"{key} seed {seed}: max |dpolicy| {np.max(np.abs(p16 - p32)):.2e}, max |dvalue| {np.max(np.abs(v16 - v32)):.2e}, max |dlogit| {np.max(np.abs(l16 - l32)):.2e}")


def test_bf16x3_is_refused_where_it_is_not_built(mz):
    for key in ("c1", "c4"):
        d = mz.DESCS[key]()
        net = mz.Net(d, mz.generate_weights(d, 0))
        with pytest.raises(mz.MzError):
            net.set_precision("bf16x3")
