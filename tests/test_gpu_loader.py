"""GPU tests of the learner-side sampler (ref learner/data_loader.cpp:15-255): records produced by the self-play side are
loaded into the product's DataLoader (features replayed ON the device, targets on the host) and into the oracle's restatement of the
reference's DataLoader; with the same seed both must sample the same (game, position, rotation) sequence and return bit-identical features,
action features, policy / value / reward targets and loss scales.  The record parser and tag map of the oracle side are pinned to the
reference's own sgf_loader.cpp / vector_map.h (tests/test_oracle_pinning.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(oracle, conf, dargs, cycles, wseed=2):
    kw = dict(vh=dargs[10], dv=dargs[11], type_name=dargs[12])
    od = oracle.make_desc(*dargs[:10], **kw)
    og = oracle.OracleGroup(conf + ":program_seed=7:nn_file_name=weight_iter_5.pt:zero_num_threads=1", od, oracle.gen_weights(od, wseed))
    og.cycles(cycles)
    return og.lines()


def _compare(mz, oracle, lconf, lines, tmp_path, batches=3, as_file=True):
    path = str(tmp_path / "5.sgf")
    with open(path, "w") as f:  # the server's sgf file holds the bare records (zero_server.cpp)
        f.write("\n".join(l.split(" ", 5)[5][:-2] for l in lines) + "\n")
    dl = mz.DataLoader(lconf)
    dl.initialize()
    ol = oracle.OracleLoader(lconf)
    if as_file:
        assert dl.load_data_from_file(path) == len(lines)
    else:
        for l in lines:
            assert dl.add_record(l) == 1  # `SelfPlay ... #` lines are accepted as they come off the worker
    ol.load_data_from_file(path)
    assert dl.num_games() == ol.num_games() == len(lines) and dl.num_data() == ol.num_data() > 0
    B, nf, na, npol, nv, nr = dl.shapes()
    out = []
    for it in range(batches):
        mine = [np.zeros((B, max(n, 1)), np.float32) for n in (nf, na, npol, nv, nr)] + [np.zeros(B, np.float32), np.zeros((B, 2), np.int32)]
        theirs = [np.zeros_like(a) for a in mine]
        dl.sample_data(*mine)
        ol.sample_data(*theirs)
        names = ["features", "action_features", "policy", "value", "reward", "loss_scale", "sampled_index"]
        for name, a, b in zip(names, mine, theirs):
            if name == "sampled_index":
                assert np.array_equal(a, b), f"batch {it}: sampled (game, position) differ"
            else:
                same = a.view(np.uint32) == b.view(np.uint32)
                assert same.all(), f"batch {it}: {name} differs in {int((~same).sum())} of {same.size} values (first sample {int(np.argwhere(~same)[0][0])})"
        out.append((mine, theirs))
    return dl, ol, out


GO = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1)


def test_alphazero_go_batches_match_the_oracle(mz, oracle, tmp_path):
    conf = "env_game=go:env_board_size=9:actor_num_simulation=6:zero_num_parallel_games=6"
    lines = _records(oracle, conf, GO + ("alphazero",), 7 * 175)
    assert len(lines) >= 6
    lconf = conf + ":nn_type_name=alphazero:learner_batch_size=96:program_seed=13"
    dl, ol, out = _compare(mz, oracle, lconf, lines, tmp_path)
    feats = out[0][0][0].reshape(96, 18, 81)
    assert set(np.unique(feats)) <= {0.0, 1.0} and feats[:, 16:18].sum() == 96 * 81  # one of the two colour planes is all ones
    pol = out[0][0][2]
    assert np.allclose(pol.sum(1), 1.0, atol=1e-5)


def test_alphazero_go_selfplay_lines_and_situational_ko(mz, oracle, tmp_path):
    conf = "env_game=go:env_board_size=9:env_go_ko_rule=situational:actor_num_simulation=4:zero_num_parallel_games=4"
    lines = _records(oracle, conf, GO + ("alphazero",), 5 * 175, wseed=5)
    _compare(mz, oracle, conf + ":learner_batch_size=40:program_seed=2", lines, tmp_path, batches=2, as_file=False)


def test_alphazero_othello_and_tictactoe(mz, oracle, tmp_path):
    conf = "env_game=othello:env_board_size=8:actor_num_simulation=5:zero_num_parallel_games=8"
    lines = _records(oracle, conf, ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero"), 6 * 70)
    _compare(mz, oracle, conf + ":learner_batch_size=128:program_seed=4", lines, tmp_path)
    conf = "env_game=tictactoe:actor_num_simulation=8:zero_num_parallel_games=8"
    lines = _records(oracle, conf, ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"), 9 * 40)
    _compare(mz, oracle, conf + ":learner_batch_size=64:program_seed=9", lines, tmp_path)


def test_muzero_go_unrolled_targets(mz, oracle, tmp_path):
    """MuZero: 5 unrolled steps of action features / policy / value / reward, absorbing states past the end of a game (uniform policy, random
    action planes drawn from the sampler's RNG stream, ref go.cpp:725-737)"""
    conf = "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=5:zero_num_parallel_games=5"
    lines = _records(oracle, conf, GO + ("muzero",), 6 * 175)
    lconf = conf + ":learner_batch_size=200:learner_muzero_unrolling_step=5:program_seed=21"
    dl, ol, out = _compare(mz, oracle, lconf, lines, tmp_path)
    B, nf, na, npol, nv, nr = dl.shapes()
    assert (na, npol, nv, nr) == (5 * 81, 6 * 82, 6, 5)


def test_atari_shaped_per_targets_and_priority_update(mz, oracle, tmp_path):
    """Atari-shaped games: features from the OBS screens, n-step values with the L (life lost) cut, 601-bin value / reward targets,
    prioritised sampling with importance weights, then update_priority and another batch"""
    conf = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=4:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
            "actor_gumbel_sample_size=4:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
            "zero_actor_intermediate_sequence_length=10:learner_n_step_return=3:learner_muzero_unrolling_step=2:env_atari_episode_length=45:zero_num_parallel_games=3:actor_resign_threshold=-2")
    # (no resignations: a resigned Atari game re-emits positions whose V / R tags an earlier sequence already cleared — the reference's
    #  loader dies on std::stof("") there, the product skips such a record with an error)
    lines = _records(oracle, conf, ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"), 5 * 50)
    assert len(lines) >= 6 and any("L[" in l for l in lines)
    lconf = conf + ":learner_batch_size=24:learner_use_per=true:learner_per_alpha=0.8:learner_per_init_beta=0.4:program_seed=5"
    dl, ol, out = _compare(mz, oracle, lconf, lines, tmp_path, batches=2)
    mine = out[-1][0]
    assert not np.allclose(mine[5], 1.0), "importance weights of prioritised replay"
    si = mine[6].copy()
    vals = np.linspace(-2.0, 3.0, 3 * 24).astype(np.float32).reshape(3, 24)  # [unroll + 1][batch], transformed scale
    dl.update_priority(si, vals)
    ol.update_priority(si, vals)
    B, nf, na, npol, nv, nr = dl.shapes()
    a = [np.zeros((B, max(n, 1)), np.float32) for n in (nf, na, npol, nv, nr)] + [np.zeros(B, np.float32), np.zeros((B, 2), np.int32)]
    b = [np.zeros_like(x) for x in a]
    dl.sample_data(*a)
    ol.sample_data(*b)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_device_buffers_and_errors(mz, oracle, tmp_path):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")  # the runtime libmzgpu itself is linked against
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    conf = "env_game=othello:env_board_size=8:actor_num_simulation=4:zero_num_parallel_games=4"
    lines = _records(oracle, conf, ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero"), 5 * 70)
    lconf = conf + ":learner_batch_size=32:program_seed=1"
    d1, d2 = mz.DataLoader(lconf), mz.DataLoader(lconf)
    for l in lines:
        assert d1.add_record(l) == 1 and d2.add_record(l) == 1
    B, nf, na, npol, nv, nr = d1.shapes()
    host = [np.zeros((B, nf), np.float32), None, np.zeros((B, npol), np.float32), np.zeros((B, nv), np.float32), None, np.zeros(B, np.float32), np.zeros((B, 2), np.int32)]
    d1.sample_data(*host)

    class DevArray:  # what a GPU trainer hands over: a device pointer (duck-types a CUDA tensor for DataLoader.sample_data)
        is_cuda = True

        def __init__(self, like):
            self.nbytes = like.nbytes
            self.p = C.c_void_p()
            assert hip.hipMalloc(C.byref(self.p), self.nbytes) == 0

        def is_contiguous(self): return True
        def data_ptr(self): return self.p.value

        def to_host(self, like):
            out = np.empty_like(like)
            assert hip.hipMemcpy(out.ctypes.data, self.p, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
            return out

    dev = [DevArray(h) if h is not None else None for h in host]
    d2.sample_data(*dev)  # the batch is written in place on the device
    for h, t in zip(host, dev):
        if h is not None:
            assert np.array_equal(h, t.to_host(h))
            hip.hipFree(t.p)
    assert d1.add_record("(;GM[othello]SZ[8];B[99])") == 0  # action out of range: the record is skipped
    assert d1.add_record("(;GM[othello]SZ[8];B[19]") == 0   # no closing parenthesis
    with pytest.raises(mz.MzError):
        mz.DataLoader("env_game=othello:nn_type_name=gumbel")
    empty = mz.DataLoader(lconf)
    with pytest.raises(mz.MzError):
        empty.sample_data(*host)


def _seeds(default_n, var):
    import os
    lo, hi = (int(x) for x in os.environ.get(var, f"0:{default_n}").split(":"))
    return range(lo, hi)


def _loader_case(seed):
    """A seeded random sampler configuration: game, network type, search size, unrolling / n-step / discount, prioritised replay, batch size."""
    rng = np.random.default_rng(4000 + seed)
    game = str(rng.choice(["go", "othello", "tictactoe", "atari"]))
    n = int(rng.choice([2, 4, 6]))
    games = int(rng.integers(2, 7))
    unroll = int(rng.integers(1, 6))
    nstep = int(rng.integers(1, 6))
    per = bool(rng.random() < 0.5)
    batch = int(rng.choice([8, 24, 50, 128]))
    lseed = int(rng.integers(1, 1000))
    common = f"actor_num_simulation={n}:zero_num_parallel_games={games}"
    if game == "atari":
        seq = int(rng.choice([0, 5, 10]))
        # (an intermediate sequence must be longer than unrolling + n-step: otherwise the self-play side's first emission is the range 0-0, cleared and then
        #  emitted again without its V / R tags — the reference's loader dies on std::stof("") there, the product refuses the batch with an error)
        while seq > 0 and unroll + nstep >= seq:
            unroll, nstep = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        ep = int(rng.integers(12, 40))
        conf = (f"env_game=atari:nn_type_name=muzero:{common}:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=4:"
                f"actor_mcts_value_rescale=true:actor_mcts_reward_discount={float(rng.choice([0.997, 0.9]))}:atari_init_q=true:zero_actor_intermediate_sequence_length={seq}:"
                f"learner_n_step_return={nstep}:learner_muzero_unrolling_step={unroll}:env_atari_episode_length={ep}:actor_resign_threshold=-2")
        dargs = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
        cycles = (n + 1) * (ep + 8)
        per = per or bool(rng.random() < 0.5)
    else:
        typ = "muzero" if game != "othello" and rng.random() < 0.5 else "alphazero"
        base, shape, glen = {"go": ("env_game=go:env_board_size=9", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1), 170),
                             "othello": ("env_game=othello:env_board_size=8", ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1), 64),
                             "tictactoe": ("env_game=tictactoe", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1), 10)}[game]
        conf = f"{base}:{common}" + (":nn_type_name=muzero" if typ == "muzero" else "")
        if game == "go" and rng.random() < 0.3:
            conf += ":env_go_ko_rule=situational"
        if typ == "muzero":
            conf += f":learner_muzero_unrolling_step={unroll}:learner_n_step_return={nstep}"
        dargs = shape + (typ,)
        cycles = (n + 1) * (glen + 6)
    lconf = conf + f":learner_batch_size={batch}:program_seed={lseed}"
    if per:
        lconf += f":learner_use_per=true:learner_per_alpha={float(rng.choice([0.5, 0.8, 1.0]))}:learner_per_init_beta={float(rng.choice([0.4, 1.0]))}"
    return conf, dargs, cycles, lconf, int(rng.integers(0, 20)), bool(rng.random() < 0.5)


@pytest.mark.parametrize("seed", _seeds(16, "MZ_FUZZ_LOADER_SEEDS"))
def test_random_sampler_configuration_matches_oracle(mz, oracle, tmp_path, seed):
    """Seeded random sampler configurations (MZ_FUZZ_LOADER_SEEDS=lo:hi for longer sweeps): records of whole games from the oracle's self-play side, loaded into the
    product's and the oracle's DataLoader, three batches each, every array bit for bit."""
    conf, dargs, cycles, lconf, wseed, as_file = _loader_case(seed)
    lines = _records(oracle, conf, dargs, cycles, wseed=wseed)
    if not lines:
        pytest.skip("no game finished within the cycle budget")
    _compare(mz, oracle, lconf, lines, tmp_path, batches=3, as_file=as_file)
