#!/usr/bin/env python3
"""Random search configurations (game, network type, PUCT / Gumbel, noise, simulations, games, seeds, chunking of run_cycles): the records of
the default mode (simulation kernels) must equal those of the lock-step mode with the host engines.  usage: fuzz_modes.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
GAMES = {
    "go": ("env_game=go:env_board_size=9", ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82), 170),
    "othello": ("env_game=othello:env_board_size=8", ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65), 62),
    "tictactoe": ("env_game=tictactoe", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9), 10),
}


def lines(conf, desc, cycles, chunks, wseed):
    wk = mz.Worker(conf, desc, mz.generate_weights(desc, wseed))
    wk.command("start")
    done, k = 0, 0
    while done < cycles:
        c = min(chunks[k % len(chunks)], cycles - done)
        assert wk.run_cycles(c) == c
        done += c
        k += 1
    return wk.pop_lines()


bad = 0
for case in range(cases):
    game = rng.choice(list(GAMES))
    base, dargs, glen = GAMES[game]
    typ = "alphazero" if game != "go" or rng.random() < 0.6 else "muzero"
    n = int(rng.choice([3, 8, 16, 25, 50, 120]))
    gumbel = rng.random() < 0.4
    m = int(rng.choice([2, 4, 8, 16]))
    games = int(rng.integers(1, 9))
    conf = (f"{base}:actor_num_simulation={n}:zero_num_parallel_games={games}:program_seed={int(rng.integers(1, 1000))}:nn_file_name=f.pt:zero_num_threads=2:"
            f"actor_use_gumbel={'true' if gumbel else 'false'}:actor_use_gumbel_noise={'true' if gumbel else 'false'}:actor_gumbel_sample_size={m}:"
            f"actor_use_dirichlet_noise={'false' if gumbel or rng.random() < 0.3 else 'true'}:"
            f"actor_select_action_by_count={'true' if rng.random() < 0.3 else 'false'}")
    if typ == "muzero":
        conf += ":nn_type_name=muzero"
    desc = mz.make_desc(*dargs, vh=16, dv=1, type_name=typ)
    cycles = (n + 1) * (glen + 6)  # whole games: the records only appear at their end
    chunks = [int(x) for x in rng.integers(1, 3 * (n + 1), 5)]
    wseed = int(rng.integers(0, 50))
    try:
        a = lines(conf + ":mz_device_env=false:mz_sim_kernel=false", desc, cycles, [cycles], wseed)
        b = lines(conf, desc, cycles, chunks, wseed)
        ok = a == b
    except Exception as e:  # noqa: BLE001
        ok = False
        print("ERROR", str(e)[:160])
    bad += not ok
    print(case, game, typ, "n", n, "gumbel", gumbel, m, "games", games, "records", len(a), "OK" if ok else "MISMATCH: " + conf, flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
