"""The self-play worker on network shapes beyond BASELINE.json (round 5) — through the per-game simulation kernel on the one-tile tower
(sim_wide.inc sim_kernel_wide: walk, Go leaf, tower, heads, candidates, expand + backup of whole simulations in one launch), records against OracleGroup.
Shapes: the reference's DEFAULT network 1 block x 256 channels (config/configuration.cpp:70-72), 6 blocks x 128 channels, 19x19 Go with 6 blocks x 64 channels
(go_unit.h:11), 7x7 and 13x13 Go; a shape with no instance at all (13x13 x 96 channels: run-time-shaped kernels, lock-step mode).  Every simulation-kernel test
asserts through the worker's counters that the kernel is what ran.  ref actor/zero_actor.cpp:51-252, actor/mcts.cpp:20-228, network/network.cpp:14-42."""
import pytest

pytestmark = pytest.mark.gpu


def _run(mz, oracle, conf, args, chunks, seed=3, wseed=1, threads=2, wextra="", expect_sim=True):
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, wseed)
    conf = f"{conf}:program_seed={seed}:nn_file_name=/tmp/w/wide_{args[0]}_{args[4]}.pt"
    games = int(conf.split("zero_num_parallel_games=")[1].split(":")[0])
    total = sum(chunks)
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(total)
    wk = mz.Worker(conf + wextra + f":zero_num_threads={threads}", d, w)
    wk.command("start")
    for c in chunks:
        assert wk.run_cycles(c) == c
    st = wk.stats()
    assert st["cycles"] == total and st["leaf_evals"] == og.leaf_evals() == total * games
    if expect_sim:
        assert st["sim_launches"] > 0 and st["sim_cycles"] >= total - 2 * (st["moves"] // games + 2), "the per-game simulation kernel did not run"
    else:
        assert st["sim_launches"] == 0
    lines, olines = wk.pop_lines(), og.lines()
    for i, (a, b) in enumerate(zip(lines, olines)):
        assert a == b, f"line {i} differs:\n  hip   : {a[:400]}\n  oracle: {b[:400]}"
    assert len(lines) == len(olines)
    recs, orecs = wk.peek_records(games), og.peek_records(games)
    for g, (a, b) in enumerate(zip(recs, orecs)):
        assert a == b, f"game {g}: records as they stand differ:\n  hip   : {a[:600]}\n  oracle: {b[:600]}"
    return lines, recs, st


GO = "env_game=go:env_board_size={n}:actor_num_simulation={sims}:zero_num_parallel_games={games}"


def test_go9_default_network_1bx256(mz, oracle):
    """The reference's default network (nn_num_blocks=1, nn_num_hidden_channels=256): sim_kernel_wide<9,9,32,256,2>, 6 games, moves of 33 simulations, calls cut inside a move."""
    args = ("go_9x9", 18, 9, 9, 256, 9, 9, 1, 1, 82, 256, 1, "alphazero")
    lines, recs, st = _run(mz, oracle, GO.format(n=9, sims=32, games=6), args, [33 * 3 + 7, 33 * 4 - 7 + 2])
    assert st["moves"] == 6 * 7


def test_go9_6bx128(mz, oracle):
    """sim_kernel_wide<9,9,32,128,2> (all optional LDS blocks fit: path speculation, superko table, the leaf's second half beside the heads)."""
    args = ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 6, 82, 256, 1, "alphazero")
    lines, recs, st = _run(mz, oracle, GO.format(n=9, sims=40, games=5), args, [41 * 2 + 11, 41 * 2 - 11 + 1])
    assert st["moves"] == 5 * 4


def test_go9_6bx128_whole_games_small_search(mz, oracle):
    """... and whole games (finished records with results) on the same kernel."""
    args = ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 1, 82, 64, 1, "alphazero")
    lines, recs, st = _run(mz, oracle, GO.format(n=9, sims=4, games=4), args, [5 * 170], wseed=4)
    assert len(lines) >= 4


def test_go19_6bx64(mz, oracle):
    """19x19 Go (go_unit.h:11), 6 blocks x 64 channels: sim_kernel_wide<19,19,32,64,6>; the tile is 119 KB, the walk runs without its speculation memory."""
    args = ("go_19x19", 18, 19, 19, 64, 19, 19, 1, 6, 362, 256, 1, "alphazero")
    lines, recs, st = _run(mz, oracle, GO.format(n=19, sims=20, games=4), args, [21 * 2 + 5, 21 - 5 + 3])
    assert st["moves"] == 4 * 3


def test_go19_n400_plan_fits(mz):
    """The LDS plan of the 19x19 kernel at the reference's search size (n = 400): the worker takes the simulation kernel, one move of 3 games."""
    args = ("go_19x19", 18, 19, 19, 64, 19, 19, 1, 1, 362, 256, 1, "alphazero")
    d = mz.make_desc(*args[:10], vh=args[10], dv=args[11], type_name=args[12])
    wk = mz.Worker(GO.format(n=19, sims=400, games=3) + ":program_seed=1:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 0))
    wk.command("start")
    assert wk.run_cycles(401 + 3) == 404
    st = wk.stats()
    assert st["sim_launches"] > 0 and st["moves"] == 3


@pytest.mark.parametrize("n,c,blocks,sims,games,moves", [(7, 32, 2, 16, 9, 6), (7, 64, 1, 12, 5, 5), (7, 128, 1, 10, 4, 4), (7, 256, 1, 8, 3, 3), (9, 32, 2, 20, 7, 4),
                                                          (13, 64, 2, 14, 4, 3), (13, 128, 1, 10, 3, 3), (19, 32, 1, 10, 3, 3)])
def test_other_wide_instances(mz, oracle, n, c, blocks, sims, games, moves):
    """Every other instance of sim_kernel_wide (7x7 / 9x9 / 13x13 / 19x19 x 32 .. 256 channels)."""
    args = (f"go_{n}x{n}", 18, n, n, c, n, n, 1, blocks, n * n + 1, 64, 1, "alphazero")
    lines, recs, st = _run(mz, oracle, GO.format(n=n, sims=sims, games=games), args, [(sims + 1) * moves + 3], wseed=2)
    assert st["moves"] == games * moves


@pytest.mark.parametrize("game,n,planes,c,blocks,sims,games,cycles,extra", [
    ("othello", 8, 4, 256, 1, 12, 5, 13 * 8 + 5, ""),      # the reference's default network (1 block x 256 channels, configuration.cpp:70-72) on Othello: sim_kernel_wide<8,8,16,256,0>
    ("othello", 8, 4, 128, 2, 10, 6, 11 * 70, ""),         # whole games (forced passes, two-pass ends) on <8,8,16,128,0>
    ("othello", 8, 4, 256, 1, 16, 4, 17 * 6 + 2, ":actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8"),  # BASELINE configs[2]'s root
    ("tictactoe", 3, 4, 256, 1, 16, 8, 17 * 24, ""),       # docs/Training.md's first example: `train tictactoe az` = TicTacToe with the default 1 block x 256: <3,3,16,256,-1>, whole games
    ("tictactoe", 3, 4, 128, 2, 8, 5, 9 * 20 + 4, ""),
])
def test_othello_and_tictactoe_on_the_one_tile_tower(mz, oracle, game, n, planes, c, blocks, sims, games, cycles, extra):
    """The wide simulation kernel is not Go's alone: Othello (two bitboards per node) and TicTacToe leaves on the one-tile tower at 128 / 256 channels — the shapes the
    reference's default configuration gives these games.  Records against OracleGroup; the counters say that the simulation kernel ran."""
    name = {"othello": "othello_8x8", "tictactoe": "tictactoe"}[game]
    args = (name, planes, n, n, c, n, n, 1, blocks, n * n + (1 if game == "othello" else 0), 64, 1, "alphazero")
    conf = f"env_game={game}:" + (f"env_board_size={n}:" if game == "othello" else "") + f"actor_num_simulation={sims}:zero_num_parallel_games={games}" + extra
    lines, recs, st = _run(mz, oracle, conf, args, [cycles], wseed=5)
    if cycles >= 9 * 20:
        assert len(lines) >= 1


def test_go9_1bx256_gumbel_and_count_selection(mz, oracle):
    """The wide kernel with a Gumbel root (device-side sequential halving) and, separately, without noise and with count selection."""
    args = ("go_9x9", 18, 9, 9, 256, 9, 9, 1, 1, 82, 32, 1, "alphazero")
    gum = ":actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8"
    _run(mz, oracle, GO.format(n=9, sims=24, games=5) + gum, args, [25 * 4 + 3], seed=9)
    det = ":actor_use_dirichlet_noise=false:actor_select_action_by_count=true:actor_select_action_by_softmax_count=false"
    _run(mz, oracle, GO.format(n=9, sims=16, games=3) + det, args, [17 * 5], seed=4)


def test_wide_modes_are_equivalent(mz, oracle):
    """The same games on the simulation kernel, on the lock-step kernels with the device rules (tower_wide stand-alone) and with the host rules."""
    args = ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 2, 82, 64, 1, "alphazero")
    conf = GO.format(n=9, sims=12, games=5)
    a = _run(mz, oracle, conf, args, [13 * 6])
    b = _run(mz, oracle, conf, args, [13 * 6], wextra=":mz_sim_kernel=false", expect_sim=False)
    c = _run(mz, oracle, conf, args, [13 * 6], wextra=":mz_device_env=false", expect_sim=False)
    assert a[1] == b[1] == c[1]


def test_shape_without_any_instance_runs_lock_step(mz, oracle):
    """13x13 Go with 96 channels: neither a fused nor a one-tile tower (96 is no power-of-two multiple of 16): the worker falls back to the lock-step kernels on
    conv3x3_band — no error, same records as the oracle."""
    args = ("go_13x13", 18, 13, 13, 96, 13, 13, 1, 1, 170, 32, 1, "alphazero")
    _run(mz, oracle, GO.format(n=13, sims=6, games=3), args, [7 * 4 + 2], expect_sim=False)


@pytest.mark.parametrize("game,n,c,blocks,sims,games,moves", [("go", 13, 96, 2, 8, 5, 6), ("go", 19, 128, 1, 6, 3, 3), ("go", 9, 80, 2, 10, 6, 8), ("go", 9, 192, 1, 8, 4, 5),
                                                              ("othello", 8, 96, 2, 10, 6, 12), ("tictactoe", 3, 48, 1, 16, 8, 30)])
def test_lock_step_shapes_play_on_the_device_rules(mz, oracle, game, n, c, blocks, sims, games, moves):
    """Shapes with no simulation-kernel instance (19x19 x 128 channels; channel counts that are no 16 * 2^k: 48, 80, 96, 192) run the lock-step cycle — and since
    round 6 with the leaf environment ON THE DEVICE (go_dev.hip: rules, planes, candidate sort), no host hop inside a move: records equal to the oracle's and to
    the host-rules mode's, and no host environment time in the stats."""
    name = {"go": f"go_{n}x{n}", "othello": "othello_8x8", "tictactoe": "tictactoe"}[game]
    planes, A = (18, n * n + 1) if game == "go" else (4, 65 if game == "othello" else 9)
    args = (name, planes, n, n, c, n, n, 1, blocks, A, 32, 1, "alphazero")
    conf = f"env_game={game}:env_board_size={n}:actor_num_simulation={sims}:zero_num_parallel_games={games}"
    cycles = (sims + 1) * moves + 3
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 2)
    conf += ":program_seed=11:nn_file_name=x.pt"
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(cycles)
    out = {}
    for mode in ("", ":mz_device_env=false"):
        wk = mz.Worker(conf + mode, d, w)
        wk.command("start")
        assert wk.run_cycles(cycles) == cycles
        st = wk.stats()
        assert st["sim_launches"] == 0 and st["leaf_evals"] == og.leaf_evals()
        assert (st["ms_env"] == 0) == (mode == ""), (mode, st["ms_env"])
        out[mode] = (wk.pop_lines(), wk.peek_records(games))
    assert out[""] == out[":mz_device_env=false"]
    assert out[""][0] == og.lines() and out[""][1] == og.peek_records(games)


def test_heavy_lock_step_pool_runs_as_two_lanes(mz, oracle):
    """mz_pipeline_lanes = 0 (default): a lock-step pool whose cycle is long — 19x19 Go, 6 blocks x 128 channels, 160 games: 206 GFLOP of convolutions per cycle —
    is cut into two lanes on two streams (one lane's single-wave tree kernels under the other's convolutions); lighter pools and every simulation-kernel
    shape stay on one.  The lane count never shows in a record."""
    args = ("go_19x19", 18, 19, 19, 128, 19, 19, 1, 6, 362, 32, 1, "alphazero")
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 1)
    conf = "env_game=go:env_board_size=19:actor_num_simulation=2:zero_num_parallel_games=160:program_seed=5:nn_file_name=x.pt"
    wk = mz.Worker(conf + ":zero_num_threads=4", d, w)
    assert wk.lanes() == 2
    light = mz.Worker(conf.replace("zero_num_parallel_games=160", "zero_num_parallel_games=64") + ":zero_num_threads=4", d, w)
    assert light.lanes() == 1
    light.close()
    one = mz.Worker(conf + ":mz_pipeline_lanes=1:zero_num_threads=4", d, w)
    assert one.lanes() == 1
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    cycles = 3 * 2 + 1
    og.cycles(cycles)
    for x in (wk, one):
        x.command("start")
        assert x.run_cycles(cycles) == cycles
        assert x.stats()["sim_launches"] == 0 and x.pop_lines() == og.lines() and x.peek_records(160) == og.peek_records(160)
    sim = mz.Worker("env_game=go:env_board_size=9:actor_num_simulation=4:zero_num_parallel_games=256:program_seed=5:nn_file_name=x.pt", mz.DESCS["c2"](), mz.generate_weights(mz.DESCS["c2"](), 0))
    assert sim.lanes() == 1


# ---- MuZero board games on the one-tile tower (sim_wide_mz.hip sim_kernel_mz_wide) ----
MZGO = "env_game=go:env_board_size={n}:nn_type_name=muzero:actor_num_simulation={sims}:zero_num_parallel_games={games}"


@pytest.mark.parametrize("n,c,blocks,sims,games,moves", [(9, 128, 2, 16, 5, 4), (9, 256, 1, 10, 3, 3), (7, 64, 2, 12, 6, 5), (13, 64, 1, 10, 3, 3), (19, 64, 1, 8, 2, 2)])
def test_wide_muzero_instances(mz, oracle, n, c, blocks, sims, games, moves):
    """Representation trunk at the root, dynamics trunk (hidden state of the parent's slab slot + the move's plane) at every other leaf, rescaled hidden states
    into the slab: records against the oracle, the simulation kernel is what ran."""
    args = (f"go_{n}x{n}", 18, n, n, c, n, n, 1, blocks, n * n + 1, 64, 1, "muzero")
    lines, recs, st = _run(mz, oracle, MZGO.format(n=n, sims=sims, games=games), args, [(sims + 1) * moves - 4, 4 + 2], wseed=2)
    assert st["moves"] == games * moves


@pytest.mark.parametrize("game,n,c,blocks,sims,games,cycles", [
    ("othello", 8, 256, 1, 12, 5, 13 * 6 + 4),   # MuZero has no leaf environment: only the board differs — sim_kernel_mz_wide<8,8,16,272,256>, the default network
    ("othello", 8, 128, 2, 10, 4, 11 * 66),      # whole games on <8,8,16,144,128>
    ("tictactoe", 3, 256, 1, 12, 6, 13 * 22),    # <3,3,16,272,256>, whole games
])
def test_wide_muzero_on_othello_and_tictactoe(mz, oracle, game, n, c, blocks, sims, games, cycles):
    name = {"othello": "othello_8x8", "tictactoe": "tictactoe"}[game]
    args = (name, 4, n, n, c, n, n, 1, blocks, n * n + (1 if game == "othello" else 0), 64, 1, "muzero")
    conf = f"env_game={game}:" + (f"env_board_size={n}:" if game == "othello" else "") + f"nn_type_name=muzero:actor_num_simulation={sims}:zero_num_parallel_games={games}"
    lines, recs, st = _run(mz, oracle, conf, args, [cycles], wseed=6)
    if cycles > 13 * 20:
        assert len(lines) >= 1


def test_wide_muzero_gumbel_and_modes(mz, oracle):
    """A Gumbel root on the wide MuZero kernel (device-side sequential halving; the rounds' leaves are NOT evaluated ahead for these shapes), and the same games on
    the lock-step kernels."""
    args = ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 1, 82, 32, 1, "muzero")
    gum = ":actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8"
    a = _run(mz, oracle, MZGO.format(n=9, sims=24, games=5) + gum, args, [25 * 3 + 3], seed=9)
    assert a[2]["pre_evals"] == 0
    b = _run(mz, oracle, MZGO.format(n=9, sims=24, games=5) + gum, args, [25 * 3 + 3], seed=9, wextra=":mz_sim_kernel=false", expect_sim=False)
    assert a[1] == b[1]


# ---- the shapes run_configs.py times (w9x128, w9x256, w19x64), at its pool size: 256 games = one game per CU on the whole chip, 16 RNG streams; the search is cut to 40-80
# simulations per move so that the CPU oracle finishes in tens of seconds (the kernels, the LDS plans and the launch split are those of n = 400 only in their table sizes) ----
@pytest.mark.slow
@pytest.mark.parametrize("n,c,blocks,sims", [(9, 128, 6, 60), (9, 256, 1, 80), (19, 64, 6, 40)])
def test_wide_full_pool_256_games_one_move(mz, oracle, n, c, blocks, sims):
    args = (f"go_{n}x{n}", 18, n, n, c, n, n, 1, blocks, n * n + 1, 256, 1, "alphazero")
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 0)
    conf = GO.format(n=n, sims=sims, games=256) + ":program_seed=1:nn_file_name=/tmp/w/wide_full.pt"
    total = sims + 1 + 5
    og = oracle.OracleGroup(conf + ":zero_num_threads=1:oracle_throughput_threads=16", od, w)
    og.cycles(total)
    wk = mz.Worker(conf + f":mz_rng_streams=16:zero_num_threads={max(2, mz.usable_cpus() - 1)}", d, w)
    wk.command("start")
    assert wk.run_cycles(sims + 1) == sims + 1 and wk.run_cycles(5) == 5
    st = wk.stats()
    assert st["sim_launches"] >= 3 and st["moves"] == 256 and st["leaf_evals"] == og.leaf_evals() == total * 256
    assert wk.pop_lines() == og.lines()
    recs, orecs = wk.peek_records(256), og.peek_records(256)
    for g, (a, b) in enumerate(zip(recs, orecs)):
        assert a == b, f"game {g}: records as they stand differ:\n  hip   : {a[:600]}\n  oracle: {b[:600]}"
