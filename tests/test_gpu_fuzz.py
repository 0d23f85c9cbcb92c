"""Randomised end-to-end parity: for seeded random search configurations (game, network type, PUCT / Gumbel root, noise, number of
simulations and games, weights, program seed, chunking of run_cycles) the HIP worker in its default mode (per-game simulation kernels,
device rules, path speculation) must emit exactly the `SelfPlay` lines of the CPU oracle.  Whole games, so every record is compared."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _seeds(default_n, var):
    """The seeds of a sweep: 0 .. default_n - 1, or `lo:hi` from the environment (an extended sweep: MZ_FUZZ_SEEDS=80:480 pytest tests/test_gpu_fuzz.py)."""
    lo, hi = (int(x) for x in os.environ.get(var, f"0:{default_n}").split(":"))
    return range(lo, hi)


GAMES = {
    "go": ("env_game=go:env_board_size=9", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82), 170),
    "othello": ("env_game=othello:env_board_size=8", ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65), 62),
    "tictactoe": ("env_game=tictactoe", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9), 10),
}


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    game = str(rng.choice(list(GAMES)))
    base, dargs, glen = GAMES[game]
    typ = "muzero" if game != "othello" and rng.random() < 0.35 else "alphazero"
    n = int(rng.choice([2, 5, 9, 16, 24]))
    if game == "go":
        n = min(n, 16)  # the oracle plays whole 9x9 games on the CPU
    gumbel = bool(rng.random() < 0.4)
    m = int(rng.choice([2, 4, 8]))
    games = int(rng.integers(1, 5))
    conf = (f"{base}:actor_num_simulation={n}:zero_num_parallel_games={games}:"
            f"actor_use_gumbel={'true' if gumbel else 'false'}:actor_use_gumbel_noise={'true' if gumbel else 'false'}:actor_gumbel_sample_size={m}:"
            f"actor_use_dirichlet_noise={'false' if gumbel or rng.random() < 0.3 else 'true'}:"
            f"actor_select_action_by_count={'true' if rng.random() < 0.3 else 'false'}")
    if typ == "muzero":
        conf += ":nn_type_name=muzero"
    cycles = (n + 1) * (glen + 4)
    chunks = [int(x) for x in rng.integers(1, 3 * (n + 1), 5)]
    wseed, pseed = int(rng.integers(0, 50)), int(rng.integers(1, 1000))
    wextra = ""  # worker-only keys (drawn last: the earlier draws of a seed stay what they were)
    if rng.random() < 0.3:
        wextra += f":mz_pipeline_lanes={int(rng.choice([2, 3]))}"
    if rng.random() < 0.3:
        wextra += f":zero_num_threads={int(rng.choice([1, 4]))}"
    for key, values, p in (("mz_sim_kernel", ["false"], 0.12), ("mz_device_env", ["false"], 0.12), ("mz_sim_split", ["false"], 0.15), ("mz_zero_copy", [0, 1, 2], 0.15),
                           ("mz_signal_wait", ["false"], 0.1)):
        if rng.random() < p:
            wextra += f":{key}={rng.choice(values)}"
    if game != "go" and rng.random() < 0.1:  # a pool larger than the chip has CUs / than one workgroup per CU (short games only: the oracle plays them on the CPU)
        big = int(rng.choice([40, 150, 300, 600]))
        conf = conf.replace(f"zero_num_parallel_games={games}:", f"zero_num_parallel_games={big}:")
    return conf, dargs, typ, cycles, chunks, wseed, pseed, wextra


@pytest.mark.parametrize("seed", _seeds(80, "MZ_FUZZ_SEEDS"))
def test_random_configuration_matches_oracle(mz, oracle, seed):
    conf, dargs, typ, cycles, chunks, wseed, pseed, wextra = _case(seed)
    kw = dict(vh=16, dv=1, type_name=typ)
    d, od = mz.make_desc(*dargs, **kw), oracle.make_desc(*dargs, **kw)
    w = mz.generate_weights(d, wseed)
    conf = f"{conf}:program_seed={pseed}:nn_file_name=/tmp/fuzz_{wseed}.pt"
    # (round 4, from a generator of its own so that a seed's configuration stays what it was) the number of host RNG streams — the oracle then runs as many slave
    # threads with the same static partition of the actors —, and for MuZero with a Gumbel root calls of whole moves: the Gumbel-round path of the board games
    rng4 = np.random.default_rng(4000 + seed)
    streams = int(rng4.choice([1, 1, 1, 2, 3, 4]))
    if typ == "muzero" and "actor_use_gumbel=true" in conf and rng4.random() < 0.6:
        n = int(conf.split("actor_num_simulation=")[1].split(":")[0])
        chunks = [n + 1, n + 1, 2 * (n + 1), int(rng4.integers(1, n + 1)), 3 * (n + 1)]
    # (round 5, again from a generator of its own) network shapes beyond the 8-channel test nets — Go on the one-tile tower's simulation kernel (32 channels on 7x7 / 9x9:
    # sim_kernel_wide, sim_kernel_mz_wide has no 32-channel instance: lock-step there) or on a shape with no instance at all (5x5 x 24 channels: conv3x3_band, lock-step) —,
    # up to 16 RNG streams (what bench.py times), and now and then a BASELINE-size pool of the short games
    rng5 = np.random.default_rng(5000 + seed)
    if dargs[0] == "go_9x9" and rng5.random() < 0.3:
        bn, ch = ((7, 32), (9, 32), (5, 24))[int(rng5.integers(0, 3))]
        dargs = (f"go_{bn}x{bn}", 18, bn, bn, ch, bn, bn, 1, 1, bn * bn + 1)
        conf = conf.replace("env_board_size=9", f"env_board_size={bn}")
        d, od = mz.make_desc(*dargs, **kw), oracle.make_desc(*dargs, **kw)
        w = mz.generate_weights(d, wseed)
    if rng5.random() < 0.25:
        streams = int(rng5.choice([6, 8, 16]))
    if dargs[0] != "go_9x9" and not dargs[0].startswith("go_") and rng5.random() < 0.04:
        big = int(rng5.choice([256, 1024]))
        gcur = int(conf.split("zero_num_parallel_games=")[1].split(":")[0])
        conf = conf.replace(f"zero_num_parallel_games={gcur}:", f"zero_num_parallel_games={big}:")
        cycles = min(cycles, 6 * (int(conf.split("actor_num_simulation=")[1].split(":")[0]) + 1))  # a few moves of the big pool
    og = oracle.OracleGroup(conf + ":zero_num_threads=1" + (f":oracle_throughput_threads={streams}" if streams > 1 else ""), od, w)
    og.cycles(cycles)
    wextra += f":mz_rng_streams={streams}"
    wk = mz.Worker(conf + ":zero_num_threads=2" + wextra, d, w)
    wk.command("start")
    done, k = 0, 0
    while done < cycles:
        c = min(chunks[k % len(chunks)], cycles - done)
        assert wk.run_cycles(c) == c
        done += c
        k += 1
    lines, olines = wk.pop_lines(), og.lines()
    assert lines == olines, conf + wextra
    # the games still in progress too (an Othello game with many passes can outlast the cycle budget: then this is all the case compares)
    games = int(conf.split("zero_num_parallel_games=")[1].split(":")[0])
    assert wk.peek_records(games) == og.peek_records(games), conf


def _atari_case(seed):
    rng = np.random.default_rng(7000 + seed)
    gumbel = bool(rng.random() < 0.75)
    n = int(rng.choice([4, 8, 13, 24, 50]))
    m = int(rng.choice([2, 3, 4, 6, 8, 16]))
    games = int(rng.choice([1, 3, 6, 9]))
    ep = int(rng.integers(5, 14))
    seq = int(rng.choice([0, 3, 5]))
    conf = (f"env_game=atari:nn_type_name=muzero:actor_num_simulation={n}:zero_num_parallel_games={games}:env_atari_episode_length={ep}:"
            f"actor_use_gumbel={'true' if gumbel else 'false'}:actor_use_gumbel_noise={'true' if gumbel else 'false'}:actor_gumbel_sample_size={m}:"
            f"actor_use_dirichlet_noise={'false' if gumbel else 'true'}:actor_gumbel_sigma_scale_c={float(rng.choice([0.1, 1.0]))}:"
            f"actor_mcts_value_rescale={'true' if rng.random() < 0.8 else 'false'}:actor_mcts_reward_discount={float(rng.choice([0.997, 1.0, 0.9]))}:"
            f"atari_init_q={'true' if rng.random() < 0.8 else 'false'}:zero_actor_intermediate_sequence_length={seq}:learner_n_step_return={int(rng.integers(1, 4))}:"
            f"learner_muzero_unrolling_step={int(rng.integers(1, 3))}:actor_select_action_by_count={'true' if rng.random() < 0.3 else 'false'}")
    moves = 2 * ep + 3
    # calls of whole moves (the Gumbel-round path) mixed with calls that end inside a move (the ordinary path)
    chunks = [n + 1, n + 1, int(rng.integers(1, n + 1)), 2 * (n + 1), int(rng.integers(1, 3 * (n + 1)))] if rng.random() < 0.6 else [n + 1]
    wseed, pseed = int(rng.integers(0, 50)), int(rng.integers(1, 1000))
    wextra = ""  # worker-only keys (drawn last: the earlier draws of a seed stay what they were): execution modes that must not change a record
    for key, values, p in (("mz_pipeline_lanes", [2, 3], 0.2), ("zero_num_threads", [1, 4], 0.3), ("mz_sim_round_alt", ["false"], 0.2), ("mz_sim_rounds", ["false"], 0.15),
                           ("mz_sim_cluster", ["false"], 0.2), ("mz_sim_round_min", [1, 4, 8], 0.2), ("mz_raw_observations", ["false"], 0.15)):
        if rng.random() < p:
            wextra += f":{key}={rng.choice(values)}"
    if rng.random() < 0.1:  # a pool that does not fit the small-pool kernels: more games than a cluster launch takes, rounds of more workgroups than CUs
        big = int(rng.choice([20, 33, 70]))
        conf = conf.replace(f"zero_num_parallel_games={games}:", f"zero_num_parallel_games={big}:")
        games = big
        if n > 13:
            conf = conf.replace(f"actor_num_simulation={n}:", "actor_num_simulation=13:")
            n = 13
            chunks = [n + 1, n + 1, 3, 2 * (n + 1), 17]
    # (round 4, drawn after everything else) how the rounds' leaves are evaluated: the batched pipeline with 1 / 2 / 4 leaves per trunk workgroup, or one workgroup per leaf
    for key, values, p in (("mz_sim_round_batch", ["false"], 0.2), ("mz_sim_round_leaves", [1, 2, 4], 0.5)):
        if rng.random() < p:
            wextra += f":{key}={rng.choice(values)}"
    return conf, (n + 1) * moves, chunks, games, wseed, pseed, wextra


@pytest.mark.parametrize("seed", _seeds(24, "MZ_FUZZ_ATARI_SEEDS"))
def test_random_atari_configuration_matches_oracle(mz, oracle, seed):
    """The same for the Atari-shaped game on the muzero_atari network (601-bin heads, value rescaling, discount, ATARI init-Q, intermediate sequences with
    their OBS / L tags): Gumbel roots of random sample sizes take the Gumbel-round path (leaves of a round evaluated ahead) whenever a call covers whole moves."""
    conf, cycles, chunks, games, wseed, pseed, wextra = _atari_case(seed)
    dargs = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18)
    kw = dict(vh=32, dv=601, type_name="muzero_atari")
    d, od = mz.make_desc(*dargs, **kw), oracle.make_desc(*dargs, **kw)
    w = mz.generate_weights(d, wseed)
    conf = f"{conf}:program_seed={pseed}:nn_file_name=/tmp/fuzz_atari_{wseed}.pt"
    streams = int(np.random.default_rng(4000 + seed).choice([1, 1, 1, 2, 3, 4]))  # (round 4) host RNG streams = the oracle's slave threads, same static partition
    rng5 = np.random.default_rng(5000 + seed)  # (round 5, from a generator of its own) up to 16 RNG streams: what bench.py / run_configs.py time
    if rng5.random() < 0.25:
        streams = int(rng5.choice([6, 8, 16]))
    og = oracle.OracleGroup(conf + ":zero_num_threads=1" + (f":oracle_throughput_threads={streams}" if streams > 1 else ""), od, w)
    og.cycles(cycles)
    wextra += f":mz_rng_streams={streams}"
    wk = mz.Worker(conf + ":zero_num_threads=2" + wextra, d, w)
    wk.command("start")
    done, k = 0, 0
    while done < cycles:
        c = min(chunks[k % len(chunks)], cycles - done)
        assert wk.run_cycles(c) == c
        done += c
        k += 1
    lines, olines = wk.pop_lines(), og.lines()
    conf += wextra
    assert len(olines) >= 1, conf
    assert lines == olines, conf
    assert wk.peek_records(games) == og.peek_records(games), conf
