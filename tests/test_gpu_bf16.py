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


def _records(mz, key, games, cycles, prec, seed=1):
    d = mz.DESCS[key]()
    conf = mz.CONFIGS[key].replace(mz.CONFIGS[key].split("zero_num_parallel_games=")[1].split(":")[0], str(games))
    wk = mz.Worker(f"{conf}:program_seed={seed}:nn_file_name=s.pt:zero_num_threads=2:mz_nn_precision={prec}", d, mz.generate_weights(d, 0))
    wk.command("start")
    assert wk.run_cycles(cycles) == cycles
    st = wk.stats()
    assert st["sim_launches"] > 0
    return wk.peek_records(games), wk.pop_lines()


@pytest.mark.parametrize("key,games,cycles", [("c2", 16, 401 * 3 + 1), ("c3", 64, 17 * 20 + 1)])
def test_bf16x3_search_in_the_simulation_kernel(mz, key, games, cycles):
    """The opt-in tower inside sim_kernel<..., BF = true>: the search code is the f32 kernel's (template), only the leaf evaluation differs by
    ~1e-6, so the visit distributions of the first moves are the f32 mode's in (nearly) every game; records stay well-formed."""
    import re
    r32, _ = _records(mz, key, games, cycles, "f32")
    r16, _ = _records(mz, key, games, cycles, "bf16x3")
    same = total = 0
    for a, b in zip(r32, r16):
        pa, pb = re.findall(r";[BW]\[(\d+)\]P\[([^\]]*)\]", a), re.findall(r";[BW]\[(\d+)\]P\[([^\]]*)\]", b)
        assert len(pa) == len(pb) >= 1
        total += 1
        # the first move: identical RNG stream, identical position — only the network arithmetic differs.  PUCT roots print integer visit
        # counts (compared exactly), Gumbel roots print the improved policy as 6-digit floats (compared to 1e-3)
        da, db = (dict((x.split(":")[0], float(x.split(":")[1])) for x in pp[0][1].split(",")) for pp in (pa, pb))
        same += pa[0][0] == pb[0][0] and da.keys() == db.keys() and all(abs(da[k] - db[k]) <= (1e-3 if "." in pa[0][1] else 0) for k in da)
        for mv, p in pb:
            counts = [float(x.split(":")[1]) for x in p.split(",")]
            assert all(c > 0 for c in counts)
    assert same >= 0.9 * total, f"{key}: only {same} of {total} first-move visit distributions equal those of the f32 mode"
    print(f"{key}: first-move visit distributions identical to the f32 mode in {same} of {total} games")


def test_bf16x3_worker_refuses_unsupported_networks(mz):
    d = mz.DESCS["c4"]()
    with pytest.raises(mz.MzError):
        mz.Worker(mz.CONFIGS["c4"] + ":program_seed=1:mz_nn_precision=bf16x3", d, mz.generate_weights(d, 0))
    d2 = mz.DESCS["c2"]()
    with pytest.raises(mz.MzError):
        mz.Worker(mz.CONFIGS["c2"] + ":program_seed=1:mz_nn_precision=fp8", d2, mz.generate_weights(d2, 0))
