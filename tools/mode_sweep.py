#!/usr/bin/env python3
"""Small 9x9 Go searches (PUCT and Gumbel roots, 8 .. 100 simulations): records of the simulation kernel vs the lock-step kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz
def lines(conf, n):
    d = mz.make_desc("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, vh=16, dv=1, type_name="alphazero")
    wk = mz.Worker(conf + ":program_seed=3:nn_file_name=x.pt:zero_num_threads=2", d, mz.generate_weights(d, 2))
    wk.command("start")
    wk.run_cycles((n + 1) * 170)
    return wk.pop_lines()
bad = 0
for gum in ("false", "true"):
    for n in (8, 15, 16, 17, 31, 33, 64, 100):
        base = ("env_game=go:env_board_size=9:zero_num_parallel_games=7:actor_num_simulation=%d:actor_use_gumbel=%s:actor_use_gumbel_noise=%s:"
                "actor_use_dirichlet_noise=%s:actor_gumbel_sample_size=4") % (n, gum, gum, "false" if gum == "true" else "true")
        a = lines(base + ":mz_sim_kernel=false", n)
        b = lines(base + ":mz_sim_kernel=true", n)
        ok = a == b and len(a) > 0
        bad += not ok
        print(gum, n, len(a), "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
