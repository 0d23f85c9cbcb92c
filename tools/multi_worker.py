#!/usr/bin/env python3
"""W workers side by side on ONE GPU, each with games/W games of a BASELINE config (own host thread, own stream, own RNG streams program_seed + w * T + t — the
reference's slave-thread ids, each used once): the host part and the root representation of one worker's move overlap the other workers' search kernels.
usage: multi_worker.py c5 [--workers W] [--moves M] [--threads T]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402


def main():
    argv = sys.argv[1:]
    key, W, moves, threads, extra = "c5", 2, 40, None, ""
    i = 0
    while i < len(argv):
        if argv[i] == "--workers":
            W = int(argv[i + 1]); i += 2
        elif argv[i] == "--moves":
            moves = int(argv[i + 1]); i += 2
        elif argv[i] == "--threads":
            threads = int(argv[i + 1]); i += 2
        elif argv[i] == "--conf":
            extra = ":" + argv[i + 1]; i += 2
        else:
            key = argv[i]; i += 1
    d = mz.DESCS[key]()
    base = mz.CONFIGS[key]
    games = int(base.split("zero_num_parallel_games=")[1].split(":")[0])
    n = int(base.split("actor_num_simulation=")[1].split(":")[0])
    T = threads or max(1, (mz.usable_cpus() - 1) // W)
    w = mz.generate_weights(d, 0)
    workers = []
    for k in range(W):
        gk = len(range(k, games, W))
        conf = base.replace(f"zero_num_parallel_games={games}", f"zero_num_parallel_games={gk}")
        conf += f":program_seed={1 + k * T}:nn_file_name=synthetic.pt:zero_num_threads={T}:mz_rng_streams=0:mz_cpu_base={k * T}{extra}"
        wk = mz.Worker(conf, d, w)
        wk.command("start")
        workers.append(wk)
    warm = 14 if key == "c5" else 3

    def play(wk, m):
        for _ in range(m):
            wk.run_cycles(n + 1)
            wk.pop_lines(wait=False)

    def run_all(m):
        th = [threading.Thread(target=play, args=(wk, m)) for wk in workers]
        for t in th: t.start()
        for t in th: t.join()

    run_all(warm)
    s0 = [wk.stats() for wk in workers]
    t0 = time.perf_counter()
    run_all(moves)
    dt = time.perf_counter() - t0
    s1 = [wk.stats() for wk in workers]
    evals = sum(b["leaf_evals"] - a["leaf_evals"] for a, b in zip(s0, s1))
    print(json.dumps({"config": key, "workers": W, "games_per_worker": [len(range(k, games, W)) for k in range(W)], "host_threads_per_worker": T,
                      "leaf_evals_per_sec": evals / dt, "ms_per_move_of_a_worker": dt / moves * 1e3}))


if __name__ == "__main__":
    main()
