#!/usr/bin/env python3
"""Throughput of all five BASELINE.json configs on one GPU (leaf-evals/s); C2 is what bench.py reports."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

PLAN = {"c1": (17 * 400, 17), "c2": (401 * 2, 40), "c3": (17 * 20, 17), "c4": (51 * 10, 51), "c5": (51 * 6, 51)}
out = {}
threads = min(32, os.cpu_count() or 1)
for key in sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]:
    d = mz.DESCS[key]()
    conf = f"{mz.CONFIGS[key]}:program_seed=1:nn_file_name=synthetic.pt:zero_num_threads={threads if key != 'c1' else 1}:mz_cpu_base=0"
    wk = mz.Worker(conf, d, mz.generate_weights(d, 0))
    wk.command("start")
    steps, warm = PLAN[key]
    wk.run_cycles(warm)
    s0 = wk.stats()
    t0 = time.perf_counter()
    wk.run_cycles(steps)
    dt = time.perf_counter() - t0
    s1 = wk.stats()
    out[key] = {"leaf_evals_per_sec": (s1["leaf_evals"] - s0["leaf_evals"]) / dt, "ms_per_cycle": dt / steps * 1e3,
                "moves_per_sec": (s1["moves"] - s0["moves"]) / dt, "games_per_sec": (s1["games"] - s0["games"]) / dt,
                "games_in_pool": s1["leaf_evals"] // s1["cycles"], "config": mz.CONFIGS[key]}
    print(key, json.dumps(out[key]), flush=True)
    del wk
json.dump(out, open(os.path.join("gpurun_out", "configs.json"), "w"), indent=1)
