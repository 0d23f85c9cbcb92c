"""Several host RNG streams (mz_rng_streams; ref actor_group.cpp:66-70: slave thread `id` seeds its generator with program_seed + id, utils/paralleler.h).
The reference hands actors to its zero_num_threads slave threads first come first served, so which generator an actor draws from is a race there; the worker
and the oracle (oracle_throughput_threads = T) both take the static partition "actor i belongs to thread i * T / B" — one of the schedules that race can
produce — and must then agree on every record, for every T.  T = 1 is the deterministic contract every other test runs on."""
import pytest

from test_gpu_worker import check  # noqa: F401  (same line checks)

pytestmark = pytest.mark.gpu

TTT = ("env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=9", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"), 17 * 30, 20)
GO_AZ = ("env_game=go:env_board_size=9:actor_num_simulation=8:zero_num_parallel_games=7", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"), 9 * 340, 5)
OTH_GUMBEL = ("env_game=othello:env_board_size=8:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
              "actor_gumbel_sample_size=4:zero_num_parallel_games=10", ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero"), 9 * 130, 8)
GO_MZ = ("env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=6:zero_num_parallel_games=5", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero"), 7 * 340, 4)
ATARI = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
         "actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
         "zero_actor_intermediate_sequence_length=6:learner_n_step_return=2:learner_muzero_unrolling_step=1:env_atari_episode_length=20:zero_num_parallel_games=6",
         ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"), 9 * 45, 6)


@pytest.mark.parametrize("T", [1, 2, 4])
@pytest.mark.parametrize("case", [TTT, GO_AZ, OTH_GUMBEL, GO_MZ, ATARI], ids=["tictactoe", "go_alphazero", "othello_gumbel", "go_muzero", "atari_gumbel_muzero"])
def test_records_with_T_rng_streams_match_the_oracle(mz, oracle, case, T):
    conf, args, cycles, min_lines = case
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 2)
    conf = f"{conf}:program_seed=31:nn_file_name=/tmp/weights/s.pt"
    og = oracle.OracleGroup(conf + f":zero_num_threads=1:oracle_throughput_threads={T}", od, w)
    og.cycles(cycles)
    # mz_rng_streams = 0: as many generators as slave threads (zero_num_threads = T), the reference's own arrangement
    wk = mz.Worker(conf + f":zero_num_threads={T}:mz_rng_streams=0", d, w)
    wk.command("start")
    for c in (cycles // 3, 7, cycles - cycles // 3 - 7):  # calls that end inside a move
        assert wk.run_cycles(c) == c
    lines = wk.pop_lines()
    check(lines, og.lines(), min_lines)
    if T > 1:  # ... or an explicit number of generators on a pool of another size: the same records
        wk2 = mz.Worker(conf + f":zero_num_threads=3:mz_rng_streams={T}", d, w)
        wk2.command("start")
        assert wk2.run_cycles(cycles) == cycles
        assert wk2.pop_lines() == lines


def test_streams_change_the_records_and_one_stream_is_the_default(mz):
    """T streams are other draws than one stream (the games after the first block draw from program_seed + t), and without the key the worker plays every game
    from slave thread 0's generator whatever zero_num_threads says."""
    conf, args, cycles, _ = GO_AZ
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d = mz.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 2)

    def run(extra):
        wk = mz.Worker(f"{conf}:program_seed=31:nn_file_name=s.pt{extra}", d, w)
        wk.command("start")
        assert wk.run_cycles(cycles) == cycles
        return wk.pop_lines()

    one = run(":zero_num_threads=1")
    assert run(":zero_num_threads=4") == one and run(":zero_num_threads=4:mz_rng_streams=1") == one
    assert run(":zero_num_threads=4:mz_rng_streams=0") != one
