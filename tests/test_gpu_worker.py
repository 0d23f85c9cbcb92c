"""End-to-end parity of the self-play worker: the HIP worker (through the C ABI) must emit exactly the same
`SelfPlay ... #` lines as the CPU oracle's ActorGroup loop under the same seed (1 host thread == the
reference's deterministic contract).  Bit-identical records need bit-identical network
outputs, which the order contract of DESIGN.md provides (f32 MFMA chain == fmaf chain)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_both(mz, oracle, conf, desc_args, cycles, threads=1, seed=1, wseed=0):
    kw = dict(vh=desc_args[10], dv=desc_args[11], type_name=desc_args[12])
    d, od = mz.make_desc(*desc_args[:10], **kw), oracle.make_desc(*desc_args[:10], **kw)
    w = mz.generate_weights(d, wseed)
    conf = f"{conf}:program_seed={seed}:nn_file_name=/tmp/weights/synthetic_{wseed}.pt"
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(cycles)
    wk = mz.Worker(conf + f":zero_num_threads={threads}", d, w)
    wk.command("start")
    assert wk.run_cycles(cycles) == cycles
    st = wk.stats()
    lines = wk.pop_lines()
    assert st["leaf_evals"] == og.leaf_evals()
    return lines, og.lines(), st


def check(lines, olines, min_lines):
    assert len(olines) >= min_lines, "test too short to finish games"
    for i, (a, b) in enumerate(zip(lines, olines)):
        assert a == b, f"line {i} differs:\n  hip   : {a}\n  oracle: {b}"
    assert len(lines) == len(olines)
    for l in lines:
        assert l.startswith("SelfPlay ") and l.endswith(" #") and l.count("SelfPlay") == 1 and " " not in l.split(" ", 5)[5][:-2]


C1 = ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero")


def test_c1_tictactoe_alphazero(mz, oracle):
    """BASELINE.json configs[0]: TicTacToe AZ n=16, 2b x 16ch, 8 games"""
    lines, olines, st = run_both(mz, oracle, mz.CONFIGS["c1"], C1, 17 * 120)
    check(lines, olines, 100)
    assert st["games"] == len(lines)


def test_c1_multithreaded_host_is_still_deterministic(mz, oracle):
    lines, olines, _ = run_both(mz, oracle, mz.CONFIGS["c1"], C1, 17 * 40, threads=4, seed=7)
    check(lines, olines, 30)


def test_small_go_alphazero(mz, oracle):
    conf = "env_game=go:env_board_size=9:actor_num_simulation=8:zero_num_parallel_games=6"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 9 * 400, threads=3)
    check(lines, olines, 6)


def test_go_resign_and_count_selection(mz, oracle):
    conf = ("env_game=go:env_board_size=9:actor_num_simulation=6:zero_num_parallel_games=4:actor_select_action_by_count=true:"
            "actor_select_action_by_softmax_count=false:actor_resign_threshold=0.5:zero_disable_resign_ratio=0.5:"
            "actor_use_random_rotation_features=false")
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 7 * 300, seed=3)
    check(lines, olines, 4)


def test_go_situational_superko(mz, oracle):
    """env_go_ko_rule=situational through the whole worker: per-game simulation kernel with the device rules, and the lock-step mode with the
    host engine (the rule itself is exercised on small boards by tests/test_env_parity.py and tests/test_gpu_godev.py)"""
    conf9 = "env_game=go:env_board_size=9:env_go_ko_rule=situational:actor_num_simulation=8:zero_num_parallel_games=5"
    args9 = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    lines9, olines9, st9 = run_both(mz, oracle, conf9, args9, 9 * 380, threads=2, seed=6)
    check(lines9, olines9, 5)
    assert st9["sim_launches"] > 0
    d9 = mz.make_desc(*args9[:10], vh=16, dv=1)
    wk = mz.Worker(conf9 + ":mz_device_env=false:mz_sim_kernel=false:program_seed=6:nn_file_name=/tmp/weights/synthetic_0.pt:zero_num_threads=2", d9, mz.generate_weights(d9, 0))
    wk.command("start")
    assert wk.run_cycles(9 * 380) == 9 * 380
    assert wk.stats()["sim_launches"] == 0 and wk.pop_lines() == lines9
    with pytest.raises(mz.MzError):
        mz.Worker(conf9.replace("situational", "natural") + ":program_seed=1", d9, mz.generate_weights(d9, 0))


def test_small_othello_gumbel_alphazero(mz, oracle):
    conf = ("env_game=othello:env_board_size=8:actor_num_simulation=16:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
            "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:actor_gumbel_sigma_visit_c=50:actor_gumbel_sigma_scale_c=1:"
            "zero_num_parallel_games=12")
    args = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 17 * 150, threads=2)
    check(lines, olines, 12)


def test_gumbel_sequential_halving_n50(mz, oracle):
    conf = ("env_game=othello:env_board_size=8:actor_num_simulation=50:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
            "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:zero_num_parallel_games=4")
    args = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 51 * 140)
    check(lines, olines, 4)


@pytest.mark.parametrize("m,n,chunks", [(12, 50, [51 * 140]), (6, 50, [7, 30, 51 * 140 - 37]), (3, 24, [25 * 140]), (12, 120, [13, 121 * 70 - 13]), (5, 16, [17 * 150])])
def test_gumbel_sample_size_not_a_power_of_two(mz, oracle, m, n, chunks):
    """Sequential halving with m = 12 -> 6 -> 3 ...: the next budget floor(n / (log2(m) * size / 2)) depends on the CURRENT size
    (ref gumbel_zero.cpp:110); the device step evaluates it in the reference's double operations, whatever the run_cycles chunking."""
    conf = (f"env_game=othello:env_board_size=8:actor_num_simulation={n}:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
            f"actor_use_gumbel_noise=true:actor_gumbel_sample_size={m}:zero_num_parallel_games=3")
    args = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")
    kw = dict(vh=16, dv=1, type_name="alphazero")
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 4)
    conf += ":program_seed=9:nn_file_name=g.pt"
    total = sum(chunks)
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles(total)
    wk = mz.Worker(conf, d, w)
    wk.command("start")
    for c in chunks:
        assert wk.run_cycles(c) == c
    assert wk.stats()["sim_launches"] > 0
    check(wk.pop_lines(), og.lines(), 3)


def test_small_go_muzero(mz, oracle):
    conf = "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=10:zero_num_parallel_games=4"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 11 * 340, threads=2)
    check(lines, olines, 4)


def test_tictactoe_muzero(mz, oracle):
    conf = "env_game=tictactoe:nn_type_name=muzero:actor_num_simulation=12:zero_num_parallel_games=8"
    args = ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 1, 9, 32, 1, "muzero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 13 * 100)
    check(lines, olines, 50)


def test_commands(mz):
    d = mz.DESCS["c1"]()
    w = mz.generate_weights(d, 0)
    wk = mz.Worker(mz.CONFIGS["c1"] + ":program_seed=1", d, w)
    assert wk.run_cycles(5) == 0  # not started
    wk.command("keep_alive")
    wk.command("reset_actors")  # ignored by default (ref configuration.cpp:47)
    wk.command("start")
    assert wk.run_cycles(17) == 17
    wk.command("stop")
    assert wk.run_cycles(17) == 0
    wk.command("update_config actor_num_simulation=16")
    with pytest.raises(mz.MzError):
        wk.command("update_config no_such_key=1")
    wk.set_weights(mz.generate_weights(d, 5))
    wk.command("load_model /x/y/weight_iter_100.pt")
    wk.command("start")
    assert wk.run_cycles(17 * 12) == 17 * 12
    assert any("EV[weight_iter_100.pt]" in l for l in wk.pop_lines())
    assert wk.command("quit") == 1


def test_small_atari_gumbel_muzero(mz, oracle):
    """BASELINE configs[4] shape at test size: synthetic Atari-shaped env, Gumbel MuZero with value rescale, discount,
    ATARI init-Q, 601-bin value / reward heads, intermediate-sequence emission (SelfPlay false ... DLEN[a-b])."""
    conf = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
            "actor_use_gumbel_noise=true:actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:"
            "actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=6:learner_n_step_return=2:"
            "learner_muzero_unrolling_step=1:env_atari_episode_length=20:zero_num_parallel_games=4")
    args = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
    lines, olines, st = run_both(mz, oracle, conf, args, 9 * 45, threads=2)
    assert len(olines) >= 12 and any(l.startswith("SelfPlay false") for l in olines) and any(l.startswith("SelfPlay true") for l in olines)
    for i, (a, b) in enumerate(zip(lines, olines)):
        assert a == b, f"line {i} differs:\n  hip   : {a[:300]}\n  oracle: {b[:300]}"
    assert len(lines) == len(olines)


def _lines_of(mz, conf, desc_args, chunks, total):
    kw = dict(vh=desc_args[10], dv=desc_args[11], type_name=desc_args[12])
    d = mz.make_desc(*desc_args[:10], **kw)
    wk = mz.Worker(conf + ":program_seed=11:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 3))
    wk.command("start")
    done, k = 0, 0
    while done < total:
        c = min(chunks[k % len(chunks)], total - done)
        assert wk.run_cycles(c) == c
        done += c
        k += 1
    st = wk.stats()
    assert st["cycles"] == total and st["leaf_evals"] == total * 5
    return wk.pop_lines()


@pytest.mark.parametrize("noise", ["true", "false"])
def test_go_execution_modes_are_equivalent(mz, monkeypatch, noise):
    """Host leaf environment (lock-step, host hops), device-resident lock-step kernels, and the per-game simulation kernel
    (whole runs of cycles in one launch, batches cut at arbitrary run_cycles boundaries) must emit identical records.  The simulation kernel runs the
    leaf in two halves — what the network needs before the tower, the rest (path hashes, liberties, legal mask, the slot's store, a terminal leaf's
    score) on two idle waves beside the heads (go_body.h goLeafBody PART 1 / 2) — or, with MZ_NO_SPEC=8, in one piece: both against the other modes."""
    conf = f"env_game=go:env_board_size=9:actor_num_simulation=12:zero_num_parallel_games=5:actor_use_dirichlet_noise={noise}"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    total = 13 * 200
    host = _lines_of(mz, conf + ":mz_device_env=false", args, [total], total)
    resident = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=false", args, [7, 1, 30], total)
    sim_whole = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [1, 2, 5, 13, 3, 40, 12, 14], total)
    monkeypatch.setenv("MZ_NO_SPEC", "8")
    sim_one_piece = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [total], total)
    assert len(host) >= 5
    assert host == resident
    assert host == sim_whole
    assert host == sim_chunks
    assert host == sim_one_piece


@pytest.mark.parametrize("variant", ["dirichlet", "no_noise", "gumbel", "muzero"])
def test_split_launches_are_equivalent(mz, monkeypatch, variant):
    """mz_sim_split (default): a move's launch goes out in up to three parts so that the host's noise and rotation draws overlap the parts already
    running; the records must be those of the single launch (mz_sim_split=false).  MZ_SIM_SPLIT_MIN_DRAWS=0 makes the small pool take all three parts."""
    monkeypatch.setenv("MZ_SIM_SPLIT_MIN_DRAWS", "0")
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero" if variant == "muzero" else "alphazero")
    n = 20
    conf = {"dirichlet": "env_game=go:env_board_size=9:zero_num_parallel_games=5",
            "no_noise": "env_game=go:env_board_size=9:zero_num_parallel_games=5:actor_use_dirichlet_noise=false",
            "gumbel": GO_GUMBEL,
            "muzero": "env_game=go:env_board_size=9:nn_type_name=muzero:zero_num_parallel_games=5"}[variant] + f":actor_num_simulation={n}"
    total = (n + 1) * 200
    one = _lines_of(mz, conf + ":mz_sim_split=false", args, [total], total)
    split = _lines_of(mz, conf, args, [total], total)
    split_chunks = _lines_of(mz, conf, args, [n + 1, 2 * (n + 1), 5, 40, n], total)
    assert len(one) >= 2
    assert one == split
    assert one == split_chunks


def test_go_muzero_execution_modes_are_equivalent(mz):
    """MuZero: lock-step kernels with host candidate lists vs the per-game simulation kernel (initial + recurrent inference, hidden-state
    slab, candidate sort and expand + backup in one launch per run of cycles)."""
    conf = "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=10:zero_num_parallel_games=5"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero")
    total = 11 * 180
    lockstep = _lines_of(mz, conf + ":mz_sim_kernel=false", args, [total], total)
    sim_whole = _lines_of(mz, conf + ":mz_sim_kernel=true", args, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_sim_kernel=true", args, [1, 2, 5, 11, 3, 40, 12, 10], total)
    assert len(lockstep) >= 5
    assert lockstep == sim_whole
    assert lockstep == sim_chunks


def test_go_full_games_to_the_end(mz, oracle):
    """Whole 9x9 games (captures, ko and superko late in the game, passes, the 2 * 81 move cap, Tromp-Taylor results in the records) through
    the default execution mode — device rules engine + per-game simulation kernel — against the oracle's records."""
    conf = "env_game=go:env_board_size=9:actor_num_simulation=4:zero_num_parallel_games=6"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    lines, olines, st = run_both(mz, oracle, conf, args, 5 * 164 * 3, threads=2, seed=5)
    check(lines, olines, 12)
    assert st["games"] >= 12


GO_GUMBEL = ("env_game=go:env_board_size=9:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
             "actor_gumbel_sample_size=16:actor_gumbel_sigma_visit_c=50:actor_gumbel_sigma_scale_c=1:zero_num_parallel_games=5")


@pytest.mark.parametrize("n", [16, 50])
def test_go_gumbel_on_the_device_matches_oracle(mz, oracle, n):
    """Gumbel root logic (noise on the logits, top-m sampling, sequential halving, start node of every simulation) inside the simulation
    kernel, records against the oracle."""
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, GO_GUMBEL + f":actor_num_simulation={n}", args, (n + 1) * 200, threads=2, seed=9)
    check(lines, olines, 3)


def test_go_gumbel_execution_modes_are_equivalent(mz):
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    conf = GO_GUMBEL + ":actor_num_simulation=20:actor_select_action_by_count=true:actor_select_action_by_softmax_count=false"
    total = 21 * 200
    host = _lines_of(mz, conf + ":mz_device_env=false", args, [total], total)
    resident = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=false", args, [9, 1, 30], total)
    sim_whole = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [1, 2, 5, 21, 3, 40, 20, 22, 7], total)
    assert len(host) >= 3
    assert host == resident
    assert host == sim_whole
    assert host == sim_chunks


OTHELLO_ARGS = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")


@pytest.mark.parametrize("gumbel", ["true", "false"])
def test_othello_execution_modes_are_equivalent(mz, gumbel):
    """Othello's device rules engine (go_body.h othLeafBody: bitboard flips, forced passes, two-pass end, disc-count result) in the
    lock-step kernels and inside the simulation kernel, against the host engine."""
    conf = ("env_game=othello:env_board_size=8:actor_num_simulation=10:zero_num_parallel_games=5:"
            f"actor_use_gumbel={gumbel}:actor_use_gumbel_noise={gumbel}:actor_use_dirichlet_noise={'false' if gumbel == 'true' else 'true'}:"
            "actor_gumbel_sample_size=8")
    total = 11 * 150
    host = _lines_of(mz, conf + ":mz_device_env=false", OTHELLO_ARGS, [total], total)
    resident = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=false", OTHELLO_ARGS, [7, 1, 30], total)
    sim_whole = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", OTHELLO_ARGS, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", OTHELLO_ARGS, [1, 2, 5, 11, 3, 40, 12, 10], total)
    assert len(host) >= 5
    assert host == resident
    assert host == sim_whole
    assert host == sim_chunks


def test_othello_full_games_to_the_end(mz, oracle):
    """Whole 8x8 Othello games through the default mode (device rules + simulation kernel): end-game passes and results vs the oracle."""
    conf = "env_game=othello:env_board_size=8:actor_num_simulation=4:zero_num_parallel_games=6"
    lines, olines, st = run_both(mz, oracle, conf, OTHELLO_ARGS, 5 * 62 * 3, threads=2, seed=11)
    check(lines, olines, 12)
    assert st["games"] >= 12


ATARI_SMALL = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
               "actor_use_gumbel_noise=true:actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:"
               "actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=6:learner_n_step_return=2:"
               "learner_muzero_unrolling_step=1:env_atari_episode_length=20:zero_num_parallel_games=5")
ATARI_ARGS = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")


def test_atari_execution_modes_are_equivalent(mz):
    """muzero_atari on the simulation kernel (root by a lock-step cycle with the stand-alone 96x96 representation kernels, simulations
    1..n in one launch: dynamics trunk with 18 action planes, 601-bin value / reward heads + invertValue on the device, value-rescaled
    PUCT, Gumbel root logic, serial backup with the value-bound multiset) against the lock-step kernels with host hops."""
    total = 9 * 60
    lockstep = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=false", ATARI_ARGS, [total], total)
    sim_whole = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=true", ATARI_ARGS, [total], total)
    sim_chunks = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=true", ATARI_ARGS, [1, 2, 5, 9, 3, 40, 10, 8], total)
    # the default evaluates the leaves of whole Gumbel rounds ahead (a call that covers whole moves takes that path, a call that ends inside a
    # move the ordinary one); mz_sim_rounds=false = the cluster mode (sim_cluster.h: four workgroups per game, tower split by output-channel tile, one
    # 601-bin head per workgroup); mz_sim_cluster=false = one workgroup per game
    sim_single = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=true:mz_sim_cluster=false", ATARI_ARGS, [3, 40, 9], total)
    sim_cluster = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=true:mz_sim_rounds=false", ATARI_ARGS, [1, 2, 5, 9, 3, 40, 10, 8], total)  # four workgroups per game, no rounds
    assert lockstep == sim_cluster
    # mz_sim_split (default): the root is expanded on the device from the stand-alone kernels' outputs and simulations 1..n follow without a host
    # round trip (device-side Gumbel noise and first halving step); false = the root's candidate list, noise and first Gumbel step on the host
    sim_host_root = _lines_of(mz, ATARI_SMALL + ":mz_sim_kernel=true:mz_sim_split=false", ATARI_ARGS, [total], total)
    assert len(lockstep) >= 10
    assert lockstep == sim_whole
    assert lockstep == sim_chunks
    assert lockstep == sim_single
    assert lockstep == sim_host_root


@pytest.mark.parametrize("games", [13, 64])
def test_atari_cluster_pools_are_equivalent(mz, games):
    """Cluster mode on pools that span several games per XCD.  64 games: full octets — the 8 games that share an XCD (game % 8) compute their 601-bin
    heads together (sim_cluster.h octetHead: the CU of a game takes one column slice of both FC layers for four games); 13 games: ragged octets,
    every game keeps its own heads.  Against one workgroup per game."""
    conf = ATARI_SMALL.replace("zero_num_parallel_games=5", f"zero_num_parallel_games={games}")
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d = mz.make_desc(*ATARI_ARGS[:10], **kw)

    def run(extra, chunks):
        wk = mz.Worker(conf + extra + ":program_seed=11:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 3))
        wk.command("start")
        for c in chunks:
            assert wk.run_cycles(c) == c
        return wk.pop_lines()

    total = 9 * (40 if games < 64 else 24)
    single = run(":mz_sim_cluster=false:mz_sim_rounds=false", [total])
    cluster = run(":mz_sim_rounds=false", [5, 9, 100, total - 114])  # (with Gumbel rounds the in-order part runs on one workgroup per game: §3.7)
    rounds_on_clusters = run(":mz_sim_round_min=4", [total])         # rounds of >= 4 evaluated ahead, the rest on the cluster kernel (which then skips hits)
    assert len(single) >= 20
    assert single == cluster
    assert single == rounds_on_clusters


@pytest.mark.parametrize("games,extra", [(5, ""), (13, ""), (64, ""), (64, ":mz_sim_cluster=false"), (16, ":mz_sim_round_min=4"), (64, ":mz_sim_round_min=4"),
                                         (13, ":mz_sim_round_alt=false"), (64, ":mz_sim_round_alt=false"), (13, ":mz_sim_round_batch=false"),
                                         (64, ":mz_sim_round_batch=false"), (13, ":mz_sim_round_leaves=4"), (64, ":mz_sim_round_leaves=2"), (64, ":mz_sim_round_leaves=4"),
                                         (7, ":mz_sim_round_leaves=2:mz_sim_round_alt=false"), (5, ":mz_sim_round_leaves=1"), (64, ":mz_sim_round_leaves=1"), (13, ":mz_sim_round_pairs=false"), (30, "")])
def test_atari_gumbel_rounds_are_equivalent(mz, games, extra):
    """mz_sim_rounds (default): the leaves of a whole Gumbel round — the simulations between two halvings visit different root children — are evaluated side
    by side ahead of the simulations, which then run in order and skip tower + heads when their leaf is the one evaluated for them (sim.hip
    sim_pre_kernel_mz / simPreProbe).  Only where an evaluation runs changes: the records must be those of mz_sim_rounds=false, on clusters of four
    workgroups per game and on one workgroup per game, and the counters must show that leaves were evaluated ahead and found.
    The evaluation itself is the batched pipeline of sim_rounds.hip by default (mz_sim_round_leaves = leaves per trunk workgroup, forced here so that small pools
    exercise the stacked-board trunks and the 64-sample FC tiles too) or one workgroup per leaf (mz_sim_round_batch=false): the same entries."""
    conf = ATARI_SMALL.replace("zero_num_parallel_games=5", f"zero_num_parallel_games={games}")
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d = mz.make_desc(*ATARI_ARGS[:10], **kw)

    def run(more, chunks):
        wk = mz.Worker(conf + extra + more + ":program_seed=17:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 4))
        wk.command("start")
        for c in chunks:
            assert wk.run_cycles(c) == c
        return wk.pop_lines(), wk.peek_records(games), wk.stats()

    moves = 30 if games < 64 else 12
    off, roff, soff = run(":mz_sim_rounds=false", [9] * moves)
    on, ron, son = run("", [9] * moves)  # whole moves: every call takes the rounds path
    assert soff["pre_evals"] == 0 and soff["pre_hits"] == 0
    assert son["pre_evals"] >= games * 4 * (moves - 1) and 0 < son["pre_hits"] <= son["pre_evals"]
    # round 1 (the root's children, visited once each) can never miss: at least those hits
    assert son["pre_hits"] >= games * 4 * (moves - 1)
    # rounds that fit the chip twice and stop using their second expected leaves run every trunk on a PAIR of workgroups (sim_pre_pair_kernel_mz) — from the
    # second move on, when the worker has seen a move's counters; not with the pipeline forced, and not where the key switches them off
    if "mz_sim_round_leaves" not in extra and "mz_sim_round_alt=false" not in extra and "mz_sim_round_pairs=false" not in extra and 2 * (-(-games * 4 // 8) * 8) <= 256:
        assert son["pre_pair_launches"] > 0, son
    if "mz_sim_round_pairs=false" in extra:
        assert son["pre_pair_launches"] == 0
    # the batched pipeline takes the rounds whose leaves outnumber the CUs (none in these pools) or every round when the leaves per workgroup are forced
    assert (son["pre_batch_launches"] > 0) == ("mz_sim_round_leaves" in extra) and son["pre_batch_launches"] <= son["pre_launches"]
    assert on == off and ron == roff and (len(on) >= 5 or games >= 64)
    # a call that ends inside a move takes the ordinary path for the broken move: same records again
    mixed, rmixed, _ = run("", [9, 4, 5, 9, 9 * (moves - 3)])
    assert mixed == off and rmixed == roff


@pytest.mark.parametrize("m,n", [(16, 50), (12, 50), (6, 33), (5, 20), (3, 7), (2, 9), (18, 40), (16, 16), (8, 100), (2, 60), (4, 90)])
def test_atari_gumbel_rounds_other_shapes_match_the_oracle(mz, oracle, m, n):
    """The rounds of other (n, m): m not a power of two (12 -> 6 -> 3: rounds of 12, 6, 3 ...), m = 18 > 16 (every root child sampled; the Gumbel step's
    sorts then fall back to the one-lane replay of libstdc++'s introsort), n < m (the first round is cut short), n much larger than the schedule's
    halvings.  Whole moves per call (the rounds path), records against the oracle and the leaves-found counter."""
    conf = (ATARI_SMALL.replace("actor_num_simulation=8", f"actor_num_simulation={n}").replace("actor_gumbel_sample_size=4", f"actor_gumbel_sample_size={m}")
            .replace("zero_num_parallel_games=5", "zero_num_parallel_games=6") + ":program_seed=23:nn_file_name=x.pt")
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d, od = mz.make_desc(*ATARI_ARGS[:10], **kw), oracle.make_desc(*ATARI_ARGS[:10], **kw)
    w = mz.generate_weights(d, 5)
    moves = 9
    og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w)
    og.cycles((n + 1) * moves)
    # (the deep searches below check that second expected leaves are consumed: without the pairs of workgroups per leaf a round that fits the chip twice always
    # evaluates them; with pairs — the default — it only does while the worker sees them used, Worker::adaptRounds)
    wk = mz.Worker(conf + ":zero_num_threads=2" + (":mz_sim_round_pairs=false" if (m, n) in ((4, 90), (8, 100)) else ""), d, w)
    wk.command("start")
    for _ in range(moves):
        assert wk.run_cycles(n + 1) == n + 1
    st = wk.stats()
    assert wk.pop_lines() == og.lines()
    assert wk.peek_records(6) == og.peek_records(6)
    assert st["pre_evals"] > 0 and st["pre_hits"] >= 6 * min(m, n) * (moves - 1)  # at least the first round of every move
    # six games leave most CUs idle: every round also evaluates each simulation's SECOND expected leaf (mz_sim_round_alt, the walk stopping one level
    # earlier); the deep searches of few root children do consume some of them — the node then remembers a slot of the slab's second bank
    print(f"m={m} n={n}: evaluated ahead {st['pre_evals']}, found {st['pre_hits']}, second expected leaf {st['pre_alt_hits']}")
    if (m, n) in ((4, 90), (8, 100)):
        assert st["pre_alt_hits"] > 0


@pytest.mark.parametrize("games,n,m", [(5, 12, 8), (9, 16, 16), (3, 50, 16), (70, 9, 4)])
def test_go_muzero_gumbel_rounds_are_equivalent(mz, oracle, games, n, m):
    """Round 4: the Gumbel rounds of a MuZero BOARD game (mz_sim_rounds_board, default): the leaves of a round are evaluated ahead by sim_pre_kernel_mz with the board
    heads (rescale + policy + tanh value, no reward), the root's initial inference is simulation 0 of the move's first launch and the noise goes out as its own
    launch before the first round.  Records of whole-move calls must equal those without rounds and the oracle's; the counters must show leaves found."""
    conf = (f"env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation={n}:zero_num_parallel_games={games}:actor_use_dirichlet_noise=false:"
            f"actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size={m}")
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero")
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 6)
    moves = 40 if games < 20 else 12

    def run(extra, chunks):
        wk = mz.Worker(conf + extra + ":program_seed=19:nn_file_name=x.pt:zero_num_threads=2", d, w)
        wk.command("start")
        for c in chunks:
            assert wk.run_cycles(c) == c
        return wk.pop_lines(), wk.peek_records(games), wk.stats()

    on, ron, son = run("", [n + 1] * moves)
    off, roff, soff = run(":mz_sim_rounds_board=false", [n + 1] * moves)
    assert soff["pre_evals"] == 0 and son["pre_evals"] > 0 and son["pre_hits"] >= games * min(m, n) * (moves - 1)  # at least the first round of every move
    assert on == off and ron == roff
    mixed, rmixed, _ = run("", [n + 1, 5, n - 4, 2 * (n + 1), (n + 1) * (moves - 4)])  # calls that end inside a move take the ordinary path for it
    assert mixed == off and rmixed == roff
    og = oracle.OracleGroup(conf + ":program_seed=19:nn_file_name=x.pt:zero_num_threads=1", od, w)
    og.cycles((n + 1) * moves)
    assert on == og.lines() and ron == og.peek_records(games)


def test_go_muzero_gumbel_execution_modes_are_equivalent(mz, oracle):
    """MuZero board game with a Gumbel root: device Gumbel step + noise on the logits inside sim_kernel_mz vs lock-step vs oracle."""
    conf = ("env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=12:zero_num_parallel_games=5:actor_use_dirichlet_noise=false:"
            "actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8")
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero")
    total = 13 * 170
    lockstep = _lines_of(mz, conf + ":mz_sim_kernel=false", args, [total], total)
    sim_whole = _lines_of(mz, conf + ":mz_sim_kernel=true", args, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_sim_kernel=true", args, [1, 2, 5, 13, 3, 40, 12, 14], total)
    assert len(lockstep) >= 5
    assert lockstep == sim_whole
    assert lockstep == sim_chunks
    lines, olines, _ = run_both(mz, oracle, conf, args, 13 * 170, threads=2, seed=4)
    check(lines, olines, 4)


def test_tictactoe_execution_modes_are_equivalent(mz):
    """TicTacToe's device rules (go_body.h tttLeafBody) in the lock-step kernels and in the simulation kernel against the host engine."""
    conf = "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=5"
    total = 17 * 60
    host = _lines_of(mz, conf + ":mz_device_env=false", C1, [total], total)
    resident = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=false", C1, [7, 1, 30], total)
    sim_whole = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", C1, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", C1, [1, 2, 5, 17, 3, 40, 16, 18], total)
    assert len(host) >= 30
    assert host == resident
    assert host == sim_whole
    assert host == sim_chunks


def test_go_deep_search_modes_are_equivalent(mz):
    """200-simulation searches: principal variations tens of levels deep, i.e. the walk's path speculation (pool_body.h: remembered paths, 16
    predicted levels per pass) takes most levels — records must still equal those of the lock-step kernels with the host engine."""
    conf = "env_game=go:env_board_size=9:actor_num_simulation=200:zero_num_parallel_games=5"
    args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
    total = 201 * 168  # whole games (the synthetic nets play to the 2 * 81 move cap): the records only appear at the end
    host = _lines_of(mz, conf + ":mz_device_env=false", args, [total], total)
    sim_whole = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [total], total)
    sim_chunks = _lines_of(mz, conf + ":mz_device_env=true:mz_sim_kernel=true", args, [1, 150, 5, 201, 3, 40, 777, 22], total)
    assert len(host) >= 4
    assert host == sim_whole
    assert host == sim_chunks


def test_go_19x19_lockstep_kernels_match_oracle(mz, oracle):
    """19x19 Go has no fused-tower / simulation-kernel instance: per-layer MFMA convolutions + lock-step search kernels with the host engine."""
    conf = "env_game=go:env_board_size=19:actor_num_simulation=4:zero_num_parallel_games=2"
    args = ("go_19x19", 18, 19, 19, 8, 19, 19, 1, 1, 362, 16, 1, "alphazero")
    lines, olines, _ = run_both(mz, oracle, conf, args, 5 * 800, threads=2, seed=2)
    check(lines, olines, 2)


def test_atari_raw_observation_path_is_equivalent(mz):
    """Root observations as bytes + planes expanded on the device (default) vs float planes built on the host: identical records."""
    total = 9 * 50
    raw = _lines_of(mz, ATARI_SMALL + ":mz_raw_observations=true", ATARI_ARGS, [total], total)
    host = _lines_of(mz, ATARI_SMALL + ":mz_raw_observations=false", ATARI_ARGS, [total], total)
    lock = _lines_of(mz, ATARI_SMALL + ":mz_raw_observations=true:mz_sim_kernel=false", ATARI_ARGS, [total], total)
    assert len(raw) >= 8
    assert raw == host
    assert raw == lock


def test_atari_manual_steps_with_extra_moves_keep_the_observation_block_right(mz):
    """BaseActor-style stepping (mz_manual_step) on the Atari-shaped env: after a search the caller plays the searched action and sometimes a second
    one without a search in between (console `play`), so the device's copy of the last 8 screens is NOT the previous block shifted by one screen — the
    worker must notice (GameEnv::rawSerial) and send the whole block again.  Byte observations vs float planes built on the host: same actions, same record."""
    import ctypes as C
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d = mz.make_desc(*ATARI_ARGS[:10], **kw)
    conf = ATARI_SMALL.replace("zero_num_parallel_games=5", "zero_num_parallel_games=1") + ":mz_manual_step=true:program_seed=3:nn_file_name=x.pt:zero_num_threads=1"

    def play(extra):
        wk = mz.Worker(conf + extra, d, mz.generate_weights(d, 3))
        L = wk.L
        L.mz_worker_search_action.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mz_worker_act.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        for f in (L.mz_worker_search_done, L.mz_worker_reset_search):
            f.argtypes = [C.c_void_p]
        L.mz_worker_emit_game.argtypes = [C.c_void_p, C.c_int]
        wk.command("start")
        actions = []
        for move in range(7):
            while not L.mz_worker_search_done(wk.h):
                assert wk.run_cycles(9) >= 0
            a, p, r = C.c_int(), C.c_int(), C.c_int()
            assert L.mz_worker_search_action(wk.h, 0, C.byref(a), C.byref(p), C.byref(r)) == 0
            actions.append(a.value)
            assert L.mz_worker_act(wk.h, 0, a.value, p.value) == 1
            if move % 3 == 1:  # a move the search did not choose, played without a search
                assert L.mz_worker_act(wk.h, 0, (a.value + 5) % 18, p.value) == 1
            assert L.mz_worker_reset_search(wk.h) == 0
        assert L.mz_worker_emit_game(wk.h, 0) == 0
        return actions, wk.pop_lines()

    raw = play(":mz_raw_observations=true")
    host = play(":mz_raw_observations=false")
    assert len(raw[1]) == 1 and raw == host


@pytest.mark.parametrize("games", [5, 64])
def test_cluster_member_going_missing_ends_in_an_error_not_a_hang(mz, monkeypatch, games):
    """Cluster mode (sim_cluster.h): the four workgroups of a game wait for each other in global memory; every wait is bounded.  With a helper of game 0
    made to leave the kernel (fault injection: MZ_NO_SPEC=4) the launch has to come back — within seconds — and the worker has to report the error."""
    import time
    monkeypatch.setenv("MZ_NO_SPEC", "4")
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d = mz.make_desc(*ATARI_ARGS[:10], **kw)
    conf = ATARI_SMALL.replace("zero_num_parallel_games=5", f"zero_num_parallel_games={games}")  # 64 games: the heads of an octet wait for each other too
    # (mz_sim_rounds=false: with the leaves of every Gumbel round evaluated ahead the in-order part runs on one workgroup per game, not on clusters)
    wk = mz.Worker(conf + ":mz_sim_rounds=false:program_seed=11:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 3))
    wk.command("start")
    t0 = time.time()
    with pytest.raises(mz.MzError):
        for _ in range(4):
            wk.run_cycles(9)
    assert time.time() - t0 < 60


@pytest.mark.parametrize("variant", ["puct_dirichlet", "no_rescale", "gumbel_m3_n21", "gumbel_m16_n40"])
def test_atari_cluster_other_search_settings(mz, variant):
    """The cluster kernel under search settings BASELINE configs[4] does not use: PUCT roots with Dirichlet noise instead of Gumbel, no value
    rescaling (the backup then runs on a second wave beside expand), and other Gumbel shapes (sample sizes 3 and 16, more simulations: the
    Gumbel step the owner computes ahead of the backup must fall back to the real step whenever a halving is due) — against the lock-step kernels
    and against one workgroup per game."""
    conf = ATARI_SMALL
    n = 8
    if variant.startswith("gumbel_"):
        m, n = int(variant.split("_")[1][1:]), int(variant.split("_")[2][1:])
        conf = conf.replace("actor_gumbel_sample_size=4", f"actor_gumbel_sample_size={m}").replace("actor_num_simulation=8", f"actor_num_simulation={n}")
    elif variant == "puct_dirichlet":
        conf = conf.replace("actor_use_dirichlet_noise=false", "actor_use_dirichlet_noise=true").replace("actor_use_gumbel=true", "actor_use_gumbel=false").replace(
            "actor_use_gumbel_noise=true", "actor_use_gumbel_noise=false")
    else:
        conf = conf.replace("actor_mcts_value_rescale=true", "actor_mcts_value_rescale=false")
    total = (n + 1) * 50
    lockstep = _lines_of(mz, conf + ":mz_sim_kernel=false", ATARI_ARGS, [total], total)
    cluster = _lines_of(mz, conf + ":mz_sim_rounds=false", ATARI_ARGS, [4, n + 1, 100, 7], total)  # (the cluster kernel is what runs without Gumbel rounds)
    single = _lines_of(mz, conf + ":mz_sim_cluster=false:mz_sim_rounds=false", ATARI_ARGS, [total], total)
    default = _lines_of(mz, conf, ATARI_ARGS, [n + 1], total)  # whole moves per call: Gumbel variants take the rounds path
    assert len(lockstep) >= 8
    assert lockstep == cluster
    assert lockstep == single
    assert lockstep == default


@pytest.mark.parametrize("games,n,extra", [(300, 50, ""), (300, 50, ":mz_sim_rounds=false"), (170, 100, ":mz_sim_round_batch=false"), (540, 24, "")])
def test_atari_pools_larger_than_the_chip(mz, oracle, games, n, extra):
    """More Atari-shaped games on ONE GPU than BASELINE's shard of 64 (a one-GPU deployment of the reference's 512-game configuration): pools beyond 256 workgroups,
    and — the regression this test pins — beyond the pool size at which a game's index times max_depth exceeds the LDS offset of the simulation's path block
    (260 games at n = 50, 137 at n = 100): the MuZero simulation kernels read that block through pointers biased by the game's offset, and where the compiler could
    see that they were LDS pointers the biased address left the LDS — a GPU memory fault from `run_configs.py c5x512` in round 5.  Records against the oracle, two moves."""
    conf = (ATARI_SMALL.replace("actor_num_simulation=8", f"actor_num_simulation={n}").replace("actor_gumbel_sample_size=4", "actor_gumbel_sample_size=8")
            .replace("zero_num_parallel_games=5", f"zero_num_parallel_games={games}") + ":program_seed=3:nn_file_name=x.pt")
    kw = dict(vh=ATARI_ARGS[10], dv=ATARI_ARGS[11], type_name=ATARI_ARGS[12])
    d, od = mz.make_desc(*ATARI_ARGS[:10], **kw), oracle.make_desc(*ATARI_ARGS[:10], **kw)
    w = mz.generate_weights(d, 5)
    cycles = 2 * (n + 1) + 3
    og = oracle.OracleGroup(conf + ":zero_num_threads=1:oracle_throughput_threads=8", od, w)
    og.cycles(cycles)
    wk = mz.Worker(conf + extra + ":zero_num_threads=4:mz_rng_streams=8", d, w)
    wk.command("start")
    assert wk.run_cycles(n + 1) == n + 1 and wk.run_cycles(n + 1) == n + 1 and wk.run_cycles(3) == 3
    st = wk.stats()
    assert st["sim_launches"] > 0 and st["moves"] == 2 * games
    assert wk.pop_lines() == og.lines()
    assert wk.peek_records(games) == og.peek_records(games)
