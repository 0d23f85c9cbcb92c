#!/usr/bin/env python3
"""Odd shapes (1 game, 300 games, 1 simulation, unsupported boards ...): default mode vs host-engine lock-step mode, or a clean error."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz
def lines(conf, desc_args, cycles, kw):
    d = mz.make_desc(*desc_args, **kw)
    wk = mz.Worker(conf + ":program_seed=5:nn_file_name=x.pt:zero_num_threads=3", d, mz.generate_weights(d, 4))
    wk.command("start")
    wk.run_cycles(cycles)
    return wk.pop_lines(), wk.stats()
cases = [
 ("go19", "env_game=go:env_board_size=19:actor_num_simulation=6:zero_num_parallel_games=3", ("go_19x19", 18, 19, 19, 8, 19, 19, 1, 1, 362), dict(vh=16, dv=1, type_name="alphazero"), 7 * 30),
 ("go9_300games", "env_game=go:env_board_size=9:actor_num_simulation=6:zero_num_parallel_games=300", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82), dict(vh=16, dv=1, type_name="alphazero"), 7 * 40),
 ("go9_1game", "env_game=go:env_board_size=9:actor_num_simulation=30:zero_num_parallel_games=1", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82), dict(vh=16, dv=1, type_name="alphazero"), 31 * 170),
 ("go9_n1", "env_game=go:env_board_size=9:actor_num_simulation=1:zero_num_parallel_games=4", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82), dict(vh=16, dv=1, type_name="alphazero"), 2 * 400),
 ("oth6", "env_game=othello:env_board_size=6:actor_num_simulation=12:zero_num_parallel_games=5", ("othello_6x6", 4, 6, 6, 8, 6, 6, 1, 1, 37), dict(vh=16, dv=1, type_name="alphazero"), 13 * 60),
 ("ttt_gumbel", "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=6:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_use_dirichlet_noise=false:actor_gumbel_sample_size=4", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9), dict(vh=256, dv=1, type_name="alphazero"), 17 * 40),
 ("go9_c64_small", "env_game=go:env_board_size=9:actor_num_simulation=20:zero_num_parallel_games=4", ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 2, 82), dict(vh=256, dv=1, type_name="alphazero"), 21 * 170),
]
bad = 0
for name, conf, da, kw, cyc in cases:
    try:
        a, sa = lines(conf + ":mz_device_env=false", da, cyc, kw)
        b, sb = lines(conf, da, cyc, kw)
        ok = a == b
        bad += not ok
        print(name, "records", len(a), "OK" if ok else "MISMATCH", flush=True)
    except Exception as e:
        bad += 1
        print(name, "ERROR", str(e)[:200], flush=True)
print("bad", bad)
