"""GPU tests of the C++ facades (include/minizero/{network,actor,actor_group}.h) through tests/csrc/facade_check.cpp (built by
__graft_entry__.build() into tests/_bin/): the reference's class / method names executed for real —
  * createNetwork -> pushBack -> forward and pushBackInitialData / pushBackRecurrentData -> initial / recurrentInference
    (ref network/alphazero_network.h:48-104, muzero_network.h:65-178, create_network.h:11-30): bit-equal to the ctypes path;
  * createActor -> think() / beforeNNEvaluation() + afterNNEvaluation() / isSearchDone() / getSearchAction() / isResign() / act() /
    reset() / getRecord(tags) (ref base_actor.h:16-55, zero_actor.h:24-70, create_actor.h:10-19) in ActorGroup's handleSearchDone
    loop (actor_group.cpp:116-134): `SelfPlay` lines identical to the oracle's one-actor group."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_bin", "facade_check")


def _pattern(n):
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (((z >> np.uint64(40)) % np.uint64(10)) < 3).astype(np.float32)


def _write(mz, tmp_path, d, w, name="weight_iter_7.pt"):
    from minizero_amd.export_weights import write_mzw
    pt = str(tmp_path / name)
    write_mzw(pt[:-3] + ".mzw", d, w)  # the facade is given the .pt name and opens the sibling .mzw
    return pt


def _run(args, timeout=300):
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    p = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"facade_check {args[:1]} failed ({p.returncode}): {p.stderr[-2000:]}"
    return p


@pytest.mark.parametrize("key", ["c2", "c3"])
def test_alphazero_network_facade(mz, tmp_path, key):
    d = mz.DESCS[key]()
    w = mz.generate_weights(d, 2)
    pt = _write(mz, tmp_path, d, w)
    B, A = 5, d.action_size
    out = str(tmp_path / "out.bin")
    p = _run(["net", pt, out, str(B)])
    assert f"Number of blocks: {d.num_blocks}" in p.stderr and "Network type name: alphazero" in p.stderr and "weight_iter_7.pt" in p.stderr
    fs = d.num_input_channels * d.input_channel_height * d.input_channel_width
    x = _pattern(B * fs).reshape(B, fs)
    pol, lg, v = mz.Net(d, w).forward(x)
    got = np.fromfile(out, np.float32).reshape(B, 2 * A + 1)
    assert np.array_equal(got[:, :A].view(np.uint32), pol.view(np.uint32))
    assert np.array_equal(got[:, A:2 * A].view(np.uint32), lg.view(np.uint32))
    assert np.array_equal(got[:, 2 * A].view(np.uint32), v.view(np.uint32))


@pytest.mark.parametrize("key", ["c4", "small_atari"])
def test_muzero_network_facade(mz, tmp_path, key):
    d = mz.DESCS[key]() if key in mz.DESCS else mz.make_desc("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
    w = mz.generate_weights(d, 3)
    pt = _write(mz, tmp_path, d, w)
    B, A = 3, d.action_size
    out = str(tmp_path / "out.bin")
    p = _run(["net", pt, out, str(B)])
    assert "Number of action feature channels" in p.stderr
    fs = d.num_input_channels * d.input_channel_height * d.input_channel_width
    P = d.hidden_channel_height * d.hidden_channel_width
    HS = d.num_hidden_channels * P
    x = _pattern(B * fs).reshape(B, fs)
    net = mz.Net(d, w)
    pol, lg, v, h = net.initial_inference(x)
    ac = d.num_action_feature_channels
    act = np.zeros((B, ac * P), np.float32)
    for b in range(B):
        a = b % A
        if ac == 1:
            if a < P:
                act[b, a] = 1.0
        else:
            act[b, (a % ac) * P:(a % ac + 1) * P] = 1.0
    rp, rl, rv, rr, rh = net.recurrent_inference(h, act)
    got = np.fromfile(out, np.float32)
    n_init = B * (2 * A + 1 + HS)
    gi = got[:n_init].reshape(B, 2 * A + 1 + HS)
    gr = got[n_init:].reshape(B, 2 * A + 2 + HS)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)  # noqa: E731
    assert np.array_equal(bits(gi[:, :A]), bits(pol)) and np.array_equal(bits(gi[:, A:2 * A]), bits(lg)) and np.array_equal(bits(gi[:, 2 * A]), bits(v))
    assert np.array_equal(bits(gi[:, 2 * A + 1:]), bits(h))
    assert np.array_equal(bits(gr[:, :A]), bits(rp)) and np.array_equal(bits(gr[:, A:2 * A]), bits(rl))
    assert np.array_equal(bits(gr[:, 2 * A]), bits(rv)) and np.array_equal(bits(gr[:, 2 * A + 1]), bits(rr)) and np.array_equal(bits(gr[:, 2 * A + 2:]), bits(rh))


CASES = [
    ("c1 tictactoe", "env_game=tictactoe:actor_num_simulation=16", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"), 70, 8),
    ("go 9x9 (simulation kernel)", "env_game=go:env_board_size=9:actor_num_simulation=10", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"), 24, 0),
    ("othello gumbel", "env_game=othello:env_board_size=8:actor_num_simulation=16:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
     "actor_use_gumbel_noise=true:actor_gumbel_sample_size=8", ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero"), 130, 2),
    ("go muzero", "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=9", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero"), 12, 0),
    # the Atari-shaped game through the per-actor surface: Gumbel rounds in manual stepping (think() = whole moves), act() by action NAME, 601-bin heads, OBS / L tags
    ("atari gumbel muzero", "env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
     "actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
     "env_atari_episode_length=14", ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"), 37, 2),
]


@pytest.mark.parametrize("name,conf,dargs,moves,min_lines", CASES)
def test_actor_facade_against_the_oracle(mz, oracle, tmp_path, name, conf, dargs, moves, min_lines):
    kw = dict(vh=dargs[10], dv=dargs[11], type_name=dargs[12])
    d, od = mz.make_desc(*dargs[:10], **kw), oracle.make_desc(*dargs[:10], **kw)
    w = mz.generate_weights(d, 1)
    pt = _write(mz, tmp_path, d, w)
    conf = f"{conf}:zero_num_parallel_games=1:program_seed=3:nn_file_name={pt}"
    feat_file = str(tmp_path / "features.bin")
    p = _run(["actor", conf, str(moves), feat_file])
    out = p.stdout.strip().split("\n")
    lines, rec = [l for l in out if l.startswith("SelfPlay ")], [l for l in out if l.startswith("RECORD ")]
    n = int(conf.split("actor_num_simulation=")[1].split(":")[0])
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(moves * (n + 1) + 1)
    olines = og.lines()
    assert len(olines) >= min_lines
    assert lines == olines
    assert len(rec) == 1 and "XX[tag]" in rec[0]
    record = rec[0][len("RECORD "):].replace("XX[tag]", "")
    assert record == og.peek_records(1)[0]
    # ---- the rest of the BaseActor / Environment surface (ref base_actor.h:16-55, base_env.h:74-114) against the oracle's environment ----
    import json
    import re
    k = record.find(";", 2)
    moves_part = record[k:-1] if k >= 0 else ""  # ";B[id]P[..]V[..]R[..];W[..]..." ("" right after a reset)
    played = re.findall(r";([BW])\[(\d+)\]((?:[A-Z]+\[[^\]]*\])*)", moves_part)
    hist = [l for l in out if l.startswith("HIST")]
    assert len(hist) == 1
    # getActionInfoHistory(): exactly the per-move tags of the record, move by move
    assert hist[0][len("HIST"):] == "".join(" ;" + tags for _, _, tags in played)
    envl = [l for l in out if l.startswith("ENV ")][0]
    kv = dict(t.split("=") for t in envl.split()[1:])
    legal = [int(a) for a in [l for l in out if l.startswith("LEGAL")][0].split()[1:]]
    got = np.fromfile(feat_file, np.float32)
    board = dargs[2]
    if dargs[12] == "muzero_atari":  # (the synthetic screens depend on the game's seed, which only the record carries: shapes and the rule "every action is legal")
        assert int(kv["turn"]) == 1 and int(kv["actions"]) == len(played) and int(kv["terminal"]) == 0
        assert int(kv["rot5"]) == 5  # the Atari environment has no rotations (ref atari.h:77-78)
        assert legal == list(range(18)) and got.size == 32 * 96 * 96 and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
        assert "OBS[1f8b08" in lines[0] and "SD[" in lines[0]  # finished episodes carry their observations (gzip member as hex) and their seed
    else:
        oe = oracle.OracleEnv(conf)
        for colour, aid, _ in played:
            assert oe.act(int(aid), 1 if colour == "B" else 2)
        assert int(kv["turn"]) == oe.turn() and int(kv["actions"]) == len(played) and int(kv["terminal"]) == int(oe.is_terminal())
        rot_tab = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_rng_rotation_config.json")))["rotation"][str(board)]
        assert int(kv["rot5"]) == rot_tab[3][5]  # getRotateAction(5, kRotation270): the reference's own table (pinned golden)
        assert float(kv["reward"]) == oe.reward()
        assert legal == [int(a) for a in np.flatnonzero(oe.legal_mask())]
        assert np.array_equal(got, oe.features(3).ravel())  # getFeatures(kRotation270)
    assert "model file name:" in p.stderr and "move number:" in p.stderr  # getSearchInfo()


@pytest.mark.parametrize("gumbel", [False, True])
def test_think_time_limit_decides_from_the_simulations_run_so_far(mz, oracle, tmp_path, gumbel):
    """actor_mcts_think_time_limit (ref zero_actor.cpp:36-49): think() looks at the clock after every step and, when the limit has passed, decides with
    the simulations run so far.  With a limit of a microsecond every move stops after its first library call of 2 cycles = the root + 1 simulation
    (include/minizero/actor.h kThinkFirst), so the record must be exactly that of a 1-simulation search (the oracle with actor_num_simulation=1: same tree,
    same RNG stream) although 2000 are configured."""
    dargs = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    kw = dict(vh=16, dv=1, type_name="alphazero")
    d, od = mz.make_desc(*dargs[:10], **kw), oracle.make_desc(*dargs[:10], **kw)
    w = mz.generate_weights(d, 2)
    pt = _write(mz, tmp_path, d, w)
    base = "env_game=go:env_board_size=9:zero_num_parallel_games=1:program_seed=5"
    if gumbel:
        base += ":actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8"
    moves = 6
    p = _run(["think", f"{base}:actor_num_simulation=2000:actor_mcts_think_time_limit=0.000001:nn_file_name={pt}", str(moves)])
    rec = [l for l in p.stdout.strip().split("\n") if l.startswith("RECORD ")][0][len("RECORD "):]
    assert rec.count(";B[") + rec.count(";W[") == moves and rec.count("]P[") == moves and rec.count("]V[") == moves
    if not gumbel:  # (a Gumbel root's visiting schedule and policy string depend on the CONFIGURED number of simulations: only the shape of the record is checked there)
        og = oracle.OracleGroup(f"{base}:actor_num_simulation=1:nn_file_name={pt}:zero_num_threads=1", od, w)
        og.cycles(2 * moves + 1)
        assert rec == og.peek_records(1)[0]


def test_think_time_limit_ends_a_long_search_on_time(mz, tmp_path):
    """The granularity of the limit: the clock is looked at between library calls of a quarter of the cycles that still fit (not every 32 cycles), so a search
    configured far beyond the limit ends within a fraction of it: 12 moves of a 100 000-simulation search under 50 ms each."""
    import time
    dargs = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    d = mz.make_desc(*dargs[:10], vh=16, dv=1, type_name="alphazero")
    pt = _write(mz, tmp_path, d, mz.generate_weights(d, 2))
    conf = f"env_game=go:env_board_size=9:zero_num_parallel_games=1:program_seed=5:actor_num_simulation=3000:actor_mcts_think_time_limit=0.05:nn_file_name={pt}"
    t0 = time.perf_counter()
    p = _run(["think", conf, "12"])
    dt = time.perf_counter() - t0
    rec = [l for l in p.stdout.strip().split("\n") if l.startswith("RECORD ")][0]
    assert rec.count("]P[") == 12
    assert dt < 12 * 0.05 * 2 + 20  # (process start, network load and context creation are in dt: the bound only catches a search that ignores its limit)


def test_two_actors_on_one_network_do_not_take_each_others_leaves(mz, oracle, tmp_path):
    """setNetwork(shared_ptr<Network>) puts several actors on ONE network (ref actor_group.cpp:179-187).  The leaves a Gumbel round evaluates ahead live in
    entries of the network, tagged with a per-move serial number — which therefore comes from the network: two actors that both counted their moves from 1 took
    each other's stale entries (same parent slot, same action, same number) for their own.  Two muzero_atari actors, different seeds, moves interleaved: each
    record equals the oracle's one-actor group of its seed, and the rounds were really on (leaves evaluated ahead and found)."""
    dargs = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
    kw = dict(vh=32, dv=601, type_name="muzero_atari")
    d, od = mz.make_desc(*dargs[:10], **kw), oracle.make_desc(*dargs[:10], **kw)
    w = mz.generate_weights(d, 4)
    pt = _write(mz, tmp_path, d, w)
    conf = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=20:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
            "actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
            f"env_atari_episode_length=30:zero_num_parallel_games=1:nn_file_name={pt}")
    moves, seeds = 24, (3, 4)
    p = _run(["two", conf, str(seeds[0]), str(seeds[1]), str(moves)])
    out = p.stdout.strip().split("\n")
    for k, seed in enumerate(seeds):
        og = oracle.OracleGroup(f"{conf}:program_seed={seed}:zero_num_threads=1", od, w)
        og.cycles(moves * 21 + 1)
        rec = [l for l in out if l.startswith(f"RECORD{k} ")][0][len(f"RECORD{k} "):]
        assert rec == og.peek_records(1)[0], f"actor {k}"
        assert [l[len(f"LINE{k} "):] for l in out if l.startswith(f"LINE{k} ")] == og.lines(), f"actor {k}"
        st = dict(t.split("=") for t in [l for l in out if l.startswith(f"STATS{k} ")][0].split()[1:])
        assert int(st["pre_evals"]) > 0 and int(st["pre_hits"]) > 0


def _facade_seeds():
    lo, hi = (int(x) for x in os.environ.get("MZ_FUZZ_FACADE_SEEDS", "0:10").split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("seed", _facade_seeds())
def test_random_configuration_through_the_actor_facade(mz, oracle, tmp_path, seed):
    """Seeded random search configurations (the generators of tests/test_gpu_fuzz.py, one game) played through createActor / think() / act() / getRecord() in
    ActorGroup's handleSearchDone loop (tests/csrc/facade_check.cpp `actor`): every `SelfPlay` line and the record of the game in progress against the oracle's
    one-actor group.  MZ_FUZZ_FACADE_SEEDS=lo:hi for longer sweeps."""
    import re
    import test_gpu_fuzz as F
    rng = np.random.default_rng(9000 + seed)
    if rng.random() < 0.3:
        conf, cycles, _, games, wseed, pseed, _ = F._atari_case(int(rng.integers(0, 10000)))
        dargs = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
    else:
        conf, shape, typ, cycles, _, wseed, pseed, _ = F._case(int(rng.integers(0, 10000)))
        dargs = shape + (16, 1, typ)
    conf = re.sub(r"zero_num_parallel_games=\d+", "zero_num_parallel_games=1", conf)
    conf = re.sub(r"zero_actor_intermediate_sequence_length=\d+", "zero_actor_intermediate_sequence_length=0", conf)  # (emitting sequences is ActorGroup's part, not the actor's)
    n = int(conf.split("actor_num_simulation=")[1].split(":")[0])
    moves = min(cycles // (n + 1), 40)
    kw = dict(vh=dargs[10], dv=dargs[11], type_name=dargs[12])
    d, od = mz.make_desc(*dargs[:10], **kw), oracle.make_desc(*dargs[:10], **kw)
    w = mz.generate_weights(d, wseed)
    pt = _write(mz, tmp_path, d, w)
    conf = f"{conf}:program_seed={pseed}:nn_file_name={pt}"
    p = _run(["actor", conf, str(moves), str(tmp_path / "features.bin")])
    out = p.stdout.strip().split("\n")
    lines, rec = [l for l in out if l.startswith("SelfPlay ")], [l for l in out if l.startswith("RECORD ")]
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(moves * (n + 1) + 1)
    assert lines == og.lines(), conf
    assert len(rec) == 1 and rec[0][len("RECORD "):].replace("XX[tag]", "") == og.peek_records(1)[0], conf
