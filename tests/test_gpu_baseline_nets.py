"""Oracle parity of the worker on the BASELINE.json networks (6 blocks x 64 channels) — the very `sim_kernel` / `sim_kernel_mz`
instantiations `bench.py` and `tools/run_configs.py` time (sim.hip: <9,9,20,64,2>, <8,8,4,64,0>, mz<9,9,20,68,64>, mz<6,6,64,84,64>).
The first group runs fewer games than BASELINE so that the CPU oracle finishes in seconds; simulations per move, network, search options =
BASELINE.  The `slow` group at the end runs the SHIPPED shapes — BASELINE's own game counts, i.e. every CU busy, the real launch-split
thresholds, two games per CU (C3), clusters of four workgroups per game with the octet heads (C5) — against the oracle for one or two moves.
Games that have not finished are compared through their records as they stand (`peek_record`: every move with its P[visit
distribution] V[root value] R[reward] tags), finished ones through their `SelfPlay` lines.  Every test asserts that the per-game
simulation kernel did run (worker stats: sim_launches / sim_cycles)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def _games(conf, games):
    head, tail = conf.split("zero_num_parallel_games=")
    return head + f"zero_num_parallel_games={games}" + (":" + tail.split(":", 1)[1] if ":" in tail else "")


def _run(mz, oracle, key, games, chunks, extra="", seed=1, wseed=0, threads=2, wextra="", streams=1):
    d, od = mz.DESCS[key](), getattr(oracle, "desc_" + key)()
    w = mz.generate_weights(d, wseed)
    conf = _games(mz.CONFIGS[key], games) + extra + f":program_seed={seed}:nn_file_name=/tmp/w/baseline_{key}_seed{wseed}.pt"
    total = sum(chunks)
    if streams > 1:  # S generators, the games statically partitioned over them (ref actor_group.cpp:18-22,66-70): what bench.py / run_configs.py time
        wextra += f":mz_rng_streams={streams}"
    og = oracle.OracleGroup(conf + ":zero_num_threads=1" + (f":oracle_throughput_threads={streams}" if streams > 1 else ""), od, w)
    og.cycles(total)
    wk = mz.Worker(conf + wextra + f":zero_num_threads={threads}", d, w)  # wextra: keys of the worker alone (execution modes)
    wk.command("start")
    for c in chunks:
        assert wk.run_cycles(c) == c
    st = wk.stats()
    assert st["cycles"] == total and st["leaf_evals"] == og.leaf_evals() == total * games
    assert st["sim_launches"] > 0 and st["sim_cycles"] >= total - 2 * (st["moves"] // games + 2), "the per-game simulation kernel did not run"
    lines, olines = wk.pop_lines(), og.lines()
    for i, (a, b) in enumerate(zip(lines, olines)):
        assert a == b, f"line {i} differs:\n  hip   : {a[:400]}\n  oracle: {b[:400]}"
    assert len(lines) == len(olines)
    recs, orecs = wk.peek_records(games), og.peek_records(games)
    for g, (a, b) in enumerate(zip(recs, orecs)):
        assert a == b, f"game {g}: records as they stand differ:\n  hip   : {a[:600]}\n  oracle: {b[:600]}"
    return lines, recs, st


def test_c2_go_alphazero_6bx64_n400(mz, oracle):
    """BASELINE configs[1] (sim_kernel<9,9,20,64,2>): 8 games, 2 complete moves + 37 simulations of the third, one run_cycles call cut
    in the middle of a move (ref zero_actor.cpp:51-98)."""
    lines, recs, st = _run(mz, oracle, "c2", 8, [150, 401 + 251, 37])
    assert st["moves"] == 16
    for r in recs:
        assert r.count(";B[") + r.count(";W[") == 2 and r.count("P[") == 2


def test_c2_go_alphazero_6bx64_no_noise_count_selection(mz, oracle):
    """Same kernel, the deterministic move rule and no root noise; other weights / seed."""
    extra = ":actor_use_dirichlet_noise=false:actor_select_action_by_count=true:actor_select_action_by_softmax_count=false"
    _run(mz, oracle, "c2", 4, [401, 401, 5], extra=extra, seed=5, wseed=3)


def test_c3_othello_gumbel_alphazero_6bx64_n16(mz, oracle):
    """BASELINE configs[2] (sim_kernel<8,8,4,64,0>, two games per CU): 64 games x 12 moves."""
    lines, recs, st = _run(mz, oracle, "c3", 64, [17 * 5 + 3, 17 * 7 - 3 + 1])
    assert st["moves"] == 64 * 12


def test_c3_othello_whole_games(mz, oracle):
    """... and whole games (records with results) on the same kernel: 8 games to the end."""
    lines, recs, st = _run(mz, oracle, "c3", 8, [17 * 70])
    assert len(lines) >= 8


def test_c4_go_muzero_6bx64_n50(mz, oracle):
    """BASELINE configs[3] (sim_kernel_mz<9,9,20,68,64>): 8 games x 4 moves."""
    lines, recs, st = _run(mz, oracle, "c4", 8, [51 * 2 + 20, 51 * 2 - 20 + 1])
    assert st["moves"] == 32


def test_c5_atari_gumbel_muzero_6bx64_n50(mz, oracle):
    """BASELINE configs[4] per-GPU shard (sim_kernel_mz<6,6,64,84,64>, 601-bin heads, value rescale, discount, ATARI init-Q): 4 games x
    8 moves with a short intermediate-sequence length, so that `SelfPlay false ... DLEN[a-b]` lines appear (ref actor_group.cpp:52-64)."""
    extra = ":zero_actor_intermediate_sequence_length=4:learner_n_step_return=1:learner_muzero_unrolling_step=1:env_atari_episode_length=7"
    lines, recs, st = _run(mz, oracle, "c5", 4, [51 * 3 + 9, 51 * 5 - 9 + 1], extra=extra, seed=2)
    assert any(l.startswith("SelfPlay false") for l in lines) and any(l.startswith("SelfPlay true") for l in lines)


# ---- the shipped shapes: BASELINE.json's own game counts against the oracle (tens of seconds of CPU oracle each) ----

@pytest.mark.slow
def test_c2_full_size_256_games_one_move(mz, oracle):
    """BASELINE configs[1] exactly as bench.py runs it: 256 games x 401 cycles = one whole move of every game (+ 17 cycles of the next, so
    that the move is decided and recorded and the second move's first two launch parts run), default mz_sim_split (a move = launches of
    1 + 16 + 384 simulations at this size), one game per CU on all 256 CUs.  ref actor_group.cpp:81-147, zero_actor.cpp:51-98."""
    lines, recs, st = _run(mz, oracle, "c2", 256, [401, 17], threads=max(2, mz.usable_cpus() - 1))
    assert st["moves"] == 256 and st["sim_launches"] >= 5  # 3 parts of move 1 + 2 parts of move 2
    for r in recs:
        assert r.count(";B[") == 1 and r.count("P[") == 1


@pytest.mark.slow
def test_c3_full_size_1024_games_two_moves(mz, oracle):
    """BASELINE configs[2] at its own size: 1024 games (four per CU slot pair: the 128-VGPR build with two resident games per CU), Gumbel
    roots with noise, 2 moves + 3 cycles."""
    lines, recs, st = _run(mz, oracle, "c3", 1024, [17, 17 + 3], threads=max(2, mz.usable_cpus() - 1))
    assert st["moves"] == 2048


@pytest.mark.slow
def test_c4_full_size_256_games_one_move(mz, oracle):
    """BASELINE configs[3] at its own size: 256 games of Go MuZero, one move + 9 cycles of the next."""
    lines, recs, st = _run(mz, oracle, "c4", 256, [51, 9], threads=max(2, mz.usable_cpus() - 1))
    assert st["moves"] == 256


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["gumbel_rounds", "gumbel_rounds_one_workgroup_per_leaf", "cluster_octet_heads"])
def test_c5_full_size_64_games(mz, oracle, mode):
    """BASELINE configs[4]'s per-GPU shard at its own size AND on the 6-block x 64-channel muzero_atari network, 2 moves + 5 cycles, sequence length as in
    the config (no line is due yet: records as they stand), on both shipped paths:
    gumbel_rounds (default): the leaves of every Gumbel round evaluated ahead by the batched pipeline of sim_rounds.hip (walks, trunks of
    4 / 2 / 1 leaves per workgroup, the FC layers of the 601-bin heads as MFMA GEMMs over all leaves, per-leaf tails), consumed in order by one workgroup
    per game — the first call is a whole move and takes that path, the counters say so;
    gumbel_rounds_one_workgroup_per_leaf (mz_sim_round_batch=false): the same rounds with one workgroup per leaf doing everything (sim.hip sim_pre_kernel_mz);
    cluster_octet_heads (mz_sim_rounds=false): 64 games = 256 workgroups in clusters of four with the 601-bin heads of the eight games of an
    XCD computed together (sim_cluster.h octetHead: only pools of full octets take that path)."""
    rounds = mode != "cluster_octet_heads"
    wextra = "" if mode == "gumbel_rounds" else ":mz_sim_round_batch=false" if rounds else ":mz_sim_rounds=false"
    lines, recs, st = _run(mz, oracle, "c5", 64, [51, 20, 51 - 20 + 5], wextra=wextra, threads=max(2, mz.usable_cpus() - 1))
    if rounds:
        assert st["pre_evals"] >= 64 * 50 and st["pre_hits"] >= 64 * 40  # (the second move is cut by the calls: only its first round is evaluated ahead ... or none)
    else:
        assert st["pre_evals"] == 0
    assert st["moves"] == 128
    for r in recs:
        assert r.count(";B[") == 2 and r.count("P[") == 2


@pytest.mark.slow
@pytest.mark.parametrize("key,games,chunks", [("c2", 256, [401, 17]), ("c3", 1024, [17, 17 + 3]), ("c4", 256, [51, 9]), ("c5", 64, [51, 20, 51 - 20 + 5])])
def test_full_size_bench_streams(mz, oracle, key, games, chunks):
    """The configuration bench.py (`--rng-streams`, default 16) and tools/run_configs.py (`RNG_STREAMS`) TIME — mz_rng_streams=16 on a pool of usable_cpus - 1 host
    threads — at BASELINE's own game counts against OracleGroup(oracle_throughput_threads=16): generator t = program_seed + t owns games [t * B / 16, (t + 1) * B / 16)
    (ref actor_group.cpp:18-22,66-70).  The round-4 verdict's gap: the timed mode had only been oracle-checked for T <= 4 on <= 10 games."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_configs
    assert run_configs.RNG_STREAMS == 16
    lines, recs, st = _run(mz, oracle, key, games, chunks, threads=max(2, mz.usable_cpus() - 1), streams=run_configs.RNG_STREAMS)
    assert st["moves"] >= games


@pytest.mark.parametrize("key,games,chunks,streams,threads", [("c2", 37, [401 + 9], 16, 3), ("c3", 100, [17 * 3 + 2], 16, 5), ("c4", 21, [51 * 2 + 1], 16, 2), ("c5", 19, [51 * 2, 9], 16, 7)])
def test_bench_streams_on_uneven_pools(mz, oracle, key, games, chunks, streams, threads):
    """The same 16 generators on pools that do not divide by 16 and on other thread counts than generators (the records must not depend on zero_num_threads)."""
    extra = ":env_atari_episode_length=12:zero_actor_intermediate_sequence_length=5" if key == "c5" else ""
    _run(mz, oracle, key, games, chunks, extra=extra, threads=threads, streams=streams)


@pytest.mark.parametrize("key,games,chunks,lanes", [("c2", 9, [401 + 30, 60], 2), ("c3", 70, [17 * 4 + 5, 17 * 3], 3), ("c4", 9, [51 + 20, 51], 2), ("c5", 9, [51 * 2, 51 + 7, 51], 3)])
def test_baseline_nets_on_several_pipeline_lanes(mz, oracle, key, games, chunks, lanes):
    """The BASELINE networks with the pool cut into pipeline lanes (mz_pipeline_lanes: own device pool, network instance and stream per lane; uneven lanes),
    calls that end inside a move and — C5 — right after a root cycle (the host-side root expansion of every lane): records against the oracle."""
    extra = ":env_atari_episode_length=12:zero_actor_intermediate_sequence_length=5" if key == "c5" else ""
    _run(mz, oracle, key, games, chunks, extra=extra, wextra=f":mz_pipeline_lanes={lanes}")
