"""The product's HOST code under ThreadSanitizer and AddressSanitizer + UBSan, on this machine, without a GPU.

tests/csrc/Makefile builds tests/_bin/host_check_{tsan,asan} from the product's host sources AS THEY ARE (worker.cpp, env.cpp, config.cpp, gzhex.cpp,
ptfile.cpp, loader.cpp, host_threads.h) + tests/csrc/fake_device.cpp, a test-only stand-in for the device seam backed by the ORACLE (its MCTS as the
node pool, its forward as the network) — so the worker's RNG-ordered per-move logic, record building, spin-wait thread pool, per-stream sinks and command
handling run for real, with T in {2, 8, 16} RNG streams on T host threads, and their records are compared with the oracle's ActorGroup loop on top.
Also: stress of the thread pool and of the OBS compressor, byte-mutation fuzzers of every parser (TorchScript archive, records, configuration strings,
action strings), compressToHex on random buffers, random play through the host rules engines.  A sanitizer report fails the test (exit code 66)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "_bin")
SAN_ENV = {"TSAN_OPTIONS": "exitcode=66 halt_on_error=0 second_deadlock_stack=1", "ASAN_OPTIONS": "exitcode=66 detect_leaks=1",
           "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1 exitcode=66"}

TTT = ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero")
GO_AZ = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
GO7_AZ = ("go_7x7", 18, 7, 7, 8, 7, 7, 1, 1, 50, 16, 1, "alphazero")
OTH_AZ = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")
GO_MZ = ("go_7x7", 18, 7, 7, 8, 7, 7, 1, 1, 50, 16, 1, "muzero")
TTT_MZ = ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 1, 9, 32, 1, "muzero")
GUMBEL = "actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=8:actor_gumbel_sigma_visit_c=50:actor_gumbel_sigma_scale_c=1"


@pytest.fixture(scope="session")
def binaries():
    r = subprocess.run(["make", "-s", "-j2", "-C", os.path.join(HERE, "csrc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return {k: os.path.join(BIN, "host_check_" + k) for k in ("tsan", "asan")}


def run(binary, *args, timeout=600):
    env = dict(os.environ)
    env.update(SAN_ENV)
    p = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=timeout)
    report = p.stderr[-6000:]
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, report
    assert p.returncode == 0, f"exit code {p.returncode}\n{report}\n{p.stdout[-2000:]}"
    return p.stdout


def worker_vs_oracle(oracle, binary, conf, args, T, steps, seed=3, wseed=0):
    """steps: ints (cycles) and protocol lines; ('load', name, weight_seed) swaps the network on both sides."""
    tn = {"alphazero": 0, "muzero": 1, "muzero_atari": 2}[args[12]]
    games = int(conf.split("zero_num_parallel_games=")[1].split(":")[0])
    wconf = f"{conf}:program_seed={seed}:nn_file_name=/m/weight_iter_0.pt:mz_device_env=false:zero_num_threads={T}:mz_rng_streams={T}"
    base = ":".join(kv for kv in conf.split(":") if not kv.startswith("mz_")) + f":program_seed={seed}:nn_file_name=/m/weight_iter_0.pt"  # (worker-only keys stay with the worker)
    chunks = []
    od = oracle.make_desc(*args[:10], vh=args[10], dv=args[11], type_name=args[12])
    og = oracle.OracleGroup(base + ":zero_num_threads=1" + (f":oracle_throughput_threads={T}" if T > 1 else ""), od, oracle.gen_weights(od, wseed))
    for s in steps:
        if isinstance(s, int):
            chunks.append(f"c{s}")
            og.cycles(s)
        elif isinstance(s, tuple):
            chunks += [f"w{s[2]}", "!load_model " + s[1]]
            og.command("load_model " + s[1], oracle.gen_weights(od, s[2]))
        else:
            chunks.append("!" + s)
            og.command(s)
    out = run(binary, "worker", wconf, args[0], *args[1:12], tn, wseed, *chunks)
    lines = [l[2:] for l in out.splitlines() if l.startswith("L ")]
    recs = [l[2:] for l in out.splitlines() if l.startswith("R ")]
    stats = dict(kv.split("=") for kv in out.splitlines()[-1].split()[1:])
    assert lines == og.lines(), (len(lines), len(og.lines()))
    assert recs == og.peek_records(games)
    assert int(stats["leaf_evals"]) == og.leaf_evals() and int(stats["sim_launches"]) == 0
    return lines


WORKER_CASES = {
    # name: (network, conf, steps, min records)
    "tictactoe": (TTT, "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=32", [17 * 6 + 5, 17 * 14 - 5], 40),
    "go_noise_resign": (GO7_AZ, "env_game=go:env_board_size=7:actor_num_simulation=6:zero_num_parallel_games=19:actor_resign_threshold=0.6:zero_disable_resign_ratio=0.5",
                        [7 * 40 + 3, 7 * 60 - 3], 10),
    "othello_gumbel": (OTH_AZ, f"env_game=othello:env_board_size=8:actor_num_simulation=12:{GUMBEL}:zero_num_parallel_games=17", [13 * 64], 10),
    "go_muzero": (GO_MZ, "env_game=go:env_board_size=7:nn_type_name=muzero:actor_num_simulation=6:zero_num_parallel_games=16", [7 * 30], 0),
    "tictactoe_muzero_gumbel": (TTT_MZ, f"env_game=tictactoe:nn_type_name=muzero:actor_num_simulation=8:{GUMBEL.replace('sample_size=8', 'sample_size=4')}:zero_num_parallel_games=24",
                                [9 * 12], 20),
    # the Atari-shaped game: OBS tags compressed by the background helpers while the worker plays on, intermediate sequences, value rescale, 601-bin heads (decoded by the oracle)
    "atari_gumbel": (("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"),
                     "env_game=atari:nn_type_name=muzero:actor_num_simulation=6:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
                     "actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
                     "zero_actor_intermediate_sequence_length=5:learner_n_step_return=2:learner_muzero_unrolling_step=1:env_atari_episode_length=16:mz_raw_observations=false:"
                     "zero_num_parallel_games=8", [7 * 30], 8),
    # the training iteration (tests/test_gpu_iteration.py runs the same protocol on the GPU): stop / update_config / load_model / reset_actors / start, mid-move
    "tictactoe_iteration": (TTT, "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=16:zero_actor_ignored_command=keep_alive",
                            [17 * 5 + 4, "stop", 3, "update_config actor_select_action_softmax_temperature=0.5", ("load", "/m/weight_iter_1.pt", 1), "reset_actors", "start",
                             17 * 6 + 2, "stop", ("load", "/m/weight_iter_2.pt", 2), "start", 17 * 6, "keep_alive"], 15),
}


@pytest.mark.parametrize("T", [2, 8, 16])
@pytest.mark.parametrize("name", sorted(WORKER_CASES))
def test_worker_host_half_under_tsan(oracle, binaries, name, T):
    """clean under ThreadSanitizer at T in {2, 8, 16} RNG streams / host threads, records equal to the oracle's"""
    args, conf, steps, min_lines = WORKER_CASES[name]
    if T == 16 and name not in ("tictactoe", "go_noise_resign", "tictactoe_iteration"):
        pytest.skip("16 streams: three cases are enough for the CPU suite's time budget")
    lines = worker_vs_oracle(oracle, binaries["tsan"], conf, args, T, steps)
    assert len(lines) >= min_lines


@pytest.mark.parametrize("name", sorted(WORKER_CASES))
def test_worker_host_half_under_asan(oracle, binaries, name):
    args, conf, steps, min_lines = WORKER_CASES[name]
    worker_vs_oracle(oracle, binaries["asan"], conf, args, 4, steps, seed=9)


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_thread_pool_and_obs_compressor_stress(binaries, kind):
    for T in (2, 8, 16):
        assert "pool ok" in run(binaries[kind], "pool", T, 1500)
    for helpers in (1, 4):
        assert "obs ok" in run(binaries[kind], "obs", helpers, 120)


def test_parsers_under_asan_ubsan(mz, oracle, binaries, tmp_path):
    """byte mutations: TorchScript archives (three network types), records of three games, configuration strings; random buffers through the gzip + hex
    writer; random play through the host rules engines"""
    import pt_writer
    b = binaries["asan"]
    for i, a in enumerate([GO_AZ, ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 2, 82, 16, 1, "muzero"), ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")]):
        d = mz.make_desc(*a[:10], vh=a[10], dv=a[11], type_name=a[12])
        path = pt_writer.write_pt(str(tmp_path / f"weight_iter_{i}.pt"), d, mz.generate_weights(d, i))
        assert "fuzz-pt ok" in run(b, "fuzz-pt", path, 400, i + 1)
    for game, args, conf, cycles in [("tictactoe", TTT, "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=8", 17 * 12),
                                      ("othello", OTH_AZ, "env_game=othello:env_board_size=8:actor_num_simulation=4:zero_num_parallel_games=6", 5 * 80),
                                      ("go", GO7_AZ, "env_game=go:env_board_size=7:actor_num_simulation=4:zero_num_parallel_games=8:actor_resign_threshold=0.2", 5 * 200)]:
        od = oracle.make_desc(*args[:10], vh=args[10], dv=args[11], type_name=args[12])
        og = oracle.OracleGroup(conf + ":program_seed=1:nn_file_name=x.pt:zero_num_threads=1", od, oracle.gen_weights(od, 0))
        og.cycles(cycles)
        recs = og.lines()
        assert len(recs) >= 4, (game, len(recs))
        f = tmp_path / f"{game}.sgf"
        f.write_text("\n".join(recs) + "\n")
        assert "fuzz-loader ok" in run(b, "fuzz-loader", conf + ":learner_batch_size=8:zero_replay_buffer=1000:zero_num_games_per_iteration=1000", str(f), 3000, 5)
    assert "fuzz-config ok" in run(b, "fuzz-config", 20000, 3)
    assert "fuzz-gz ok" in run(b, "fuzz-gz", 300, 4)
    for game, size in [("tictactoe", 3), ("othello", 8), ("go", 9), ("go", 19), ("atari", 0)]:
        assert "fuzz-env ok" in run(b, "fuzz-env", game, size, 20000 if game != "atari" else 300, 6)


@pytest.mark.parametrize("seed", range(10))
def test_random_iteration_schedules_on_the_host_half(oracle, binaries, seed):
    """tests/test_gpu_iteration.py::test_random_iteration_schedules without a GPU: random protocol lines (stop / start, live update_config keys, load_model of
    other weights, reset_actors) at random cycle counts through the worker's host half (ASan + UBSan build, T = 4 streams), records equal to the oracle's"""
    import numpy as np
    from test_gpu_iteration import _live_keys
    rng = np.random.default_rng(500 + seed)
    name = ["tictactoe", "go_noise_resign", "othello_gumbel", "go_muzero", "tictactoe_muzero_gumbel"][seed % 5]
    args, conf, _, _ = WORKER_CASES[name]
    cpm = int(conf.split("actor_num_simulation=")[1].split(":")[0]) + 1
    if rng.random() < 0.4:
        conf += ":zero_actor_ignored_command=keep_alive"
    steps, it = [], 0
    for _ in range(int(rng.integers(3, 7))):
        steps.append(cpm * int(rng.integers(1, 12)) + (int(rng.integers(0, cpm)) if rng.random() < 0.7 else 0))
        stopped = rng.random() < 0.6
        if stopped:
            steps.append("stop")
        for _k in range(int(rng.integers(0, 4))):
            what = rng.random()
            if what < 0.35:
                steps.append("update_config " + _live_keys(rng, "actor_use_gumbel=true" in conf, "nn_type_name=muzero" in conf))
            elif what < 0.7:
                it += 1
                steps.append(("load", f"/m/weight_iter_{it}.pt", it))
            elif what < 0.9:
                steps.append("reset_actors")
            else:
                steps.append("keep_alive")
        if stopped:
            steps.append("start")
    steps.append(cpm * int(rng.integers(1, 15)))
    worker_vs_oracle(oracle, binaries["asan"], conf, args, 4, steps, seed=int(rng.integers(1, 99)))


@pytest.mark.parametrize("name,G", [("tictactoe", 2), ("go_reset", 2), ("othello_gumbel_eight_devices", 8)])
def test_sp_executable_protocol_under_tsan(mz, oracle, binaries, tmp_path, name, G):
    """apps/mzgpu_sp.cpp itself — include/minizero/actor_group.h: the stdin thread, one host thread per logical device, the stdout / stderr mutexes, one read per
    weight file (the product's capi.cpp / weights.cpp / ptfile.cpp really read a TorchScript archive and an .mzw) — built over the device stand-in and run under
    ThreadSanitizer with G logical devices: the same two-iteration protocol as tests/test_gpu_iteration.py::test_iterations_through_the_sp_executable, every device's
    logged schedule replayed on an oracle of its own, every printed record accounted for, `weight files read: 3`, no sanitizer report."""
    import test_gpu_iteration as it
    case = [c for c in it.EXE_CASES if c[0] == name][0]
    _, game, args, extra, cpm, _, tail = case
    it.sp_executable_iterations(mz, oracle, tmp_path, game, args, extra, cpm, G, tail, exe=os.path.join(BIN, "mzgpu_sp_tsan"), worker_extra=":mz_device_env=false",
                                make_env=SAN_ENV)
