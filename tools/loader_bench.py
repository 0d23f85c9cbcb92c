#!/usr/bin/env python3
"""Throughput of the learner-side sampler: batches of learner_batch_size = 1024 samples from a replay buffer of 9x9 Go AlphaZero
games (BASELINE configs[1] shapes; the games are produced here by the self-play worker with a short search), on the product's DataLoader
(records -> flat arrays once, features replayed on the GPU per batch) and on the oracle's restatement of the reference's DataLoader (every sample
replays its game from the first move on the CPU, one thread — the reference runs `learner_num_thread` of those).
usage: loader_bench.py [games=512] [batches=20] [go|othello|atari]   (atari: the Atari-shaped env, batches of 256 samples of 32 x 96 x 96 planes)"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import minizero_amd as mz  # noqa: E402

games = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
game = sys.argv[3] if len(sys.argv) > 3 else "go"
batch = 1024
if game == "atari":
    batch = 256
    d = mz.make_desc("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, type_name="muzero_atari")
    conf = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
            "actor_gumbel_sample_size=4:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=0:"
            f"learner_n_step_return=5:learner_muzero_unrolling_step=5:env_atari_episode_length=120:actor_resign_threshold=-2:zero_num_parallel_games={min(games, 64)}")
else:
    d = mz.DESCS["c2" if game == "go" else "c3"]()
    conf = (f"env_game=go:env_board_size=9:actor_num_simulation=8:zero_num_parallel_games={min(games, 256)}" if game == "go" else
            f"env_game=othello:env_board_size=8:actor_num_simulation=8:zero_num_parallel_games={min(games, 256)}")
wk = mz.Worker(conf + f":program_seed=1:nn_file_name=weight_iter_0.pt:zero_num_threads={max(1, mz.usable_cpus() - 1)}", d, mz.generate_weights(d, 0))
wk.command("start")
lines = []
t0 = time.perf_counter()
while len(lines) < games:
    wk.run_cycles(9 * {"go": 170, "othello": 62, "atari": 121}[game])
    lines += wk.pop_lines()
print(f"{len(lines)} games ({sum(l.count(';B[') + l.count(';W[') for l in lines)} positions) from the worker in {time.perf_counter() - t0:.1f} s", flush=True)
del wk
lines = lines[:games]
path = os.path.join(tempfile.mkdtemp(), "0.sgf")
with open(path, "w") as f:
    f.write("\n".join(l.split(" ", 5)[5][:-2] for l in lines) + "\n")
lconf = conf + ("" if game == "atari" else ":nn_type_name=alphazero") + f":learner_batch_size={batch}:program_seed=13:zero_replay_buffer=20:zero_num_games_per_iteration=" + str(games)


SHAPES = None


def run(make, name):
    dl = make(lconf)
    if hasattr(dl, "initialize"):
        dl.initialize()
    t0 = time.perf_counter()
    dl.load_data_from_file(path)
    t_load = time.perf_counter() - t0
    global SHAPES
    if hasattr(dl, "shapes"):
        SHAPES = dl.shapes()
    B, nf, na, npol, nv, nr = SHAPES
    bufs = [np.zeros((B, max(n, 1)), np.float32) for n in (nf, na, npol, nv, nr)] + [np.zeros(B, np.float32), np.zeros((B, 2), np.int32)]
    dl.sample_data(*bufs)  # warm-up
    t0 = time.perf_counter()
    n = batches if name.startswith("gpu") else max(2, batches // 10)
    for _ in range(n):
        dl.sample_data(*bufs)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: load {t_load * 1e3:.0f} ms, {dt * 1e3:.2f} ms per batch of {B} -> {B / dt:,.0f} samples/s", flush=True)
    return bufs


a = run(lambda c: mz.DataLoader(c), "gpu, batch returned to host arrays")


def run_device():
    """the batch written to device buffers (what a GPU trainer passes: CUDA tensors)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    class DevArray:
        is_cuda = True

        def __init__(self, nbytes):
            self.p = C.c_void_p()
            assert hip.hipMalloc(C.byref(self.p), max(nbytes, 4)) == 0

        def is_contiguous(self): return True
        def data_ptr(self): return self.p.value

    dl = mz.DataLoader(lconf)
    dl.load_data_from_file(path)
    B, nf, na, npol, nv, nr = dl.shapes()
    bufs = [DevArray(B * nf * 4), DevArray(B * na * 4) if na else None, DevArray(B * npol * 4), DevArray(B * nv * 4), DevArray(B * nr * 4) if nr else None,
            DevArray(B * 4), DevArray(B * 8)]
    dl.sample_data(*bufs)
    t0 = time.perf_counter()
    for _ in range(batches):
        dl.sample_data(*bufs)
    dt = (time.perf_counter() - t0) / batches
    print(f"gpu, batch left on the device: {dt * 1e3:.2f} ms per batch of {B} -> {B / dt:,.0f} samples/s", flush=True)


run_device()
try:
    import oracle_lib
    oracle_lib.build()
    b = run(lambda c: oracle_lib.OracleLoader(c), "cpu oracle (1 thread)")
except Exception as e:  # the oracle is test infrastructure: the bench of the product does not depend on it
    print("oracle loader not available:", e)
