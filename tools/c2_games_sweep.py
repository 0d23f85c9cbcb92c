#!/usr/bin/env python3
"""Experiment: does the per-simulation tower time of the C2 simulation kernel depend on how many games share an XCD's L2 (weight traffic)?
usage: MZ_SIM_PROF=1 python tools/c2_games_sweep.py 256 128 64 32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz
for games in [int(x) for x in sys.argv[1:]] or [256, 128]:
    d = mz.DESCS["c2"]()
    conf = mz.CONFIGS["c2"].replace("zero_num_parallel_games=256", f"zero_num_parallel_games={games}")
    conf += f":program_seed=1:nn_file_name=synthetic.pt:zero_num_threads={max(1, mz.usable_cpus() - 1)}:mz_cpu_base=0"
    wk = mz.Worker(conf, d, mz.generate_weights(d, 0))
    wk.command("start")
    wk.run_cycles(401)
    t0 = time.perf_counter()
    wk.run_cycles(401 * 2)
    dt = time.perf_counter() - t0
    print(f"games {games}: {dt / 2 * 1e3:.2f} ms per move", file=sys.stderr, flush=True)
    del wk
