#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configs on one GPU (leaf-evals/s) with a roofline block per config; C2 is what bench.py reports.
usage: run_configs.py [c1 .. c5] [--moves M] [--threads T] [--conf k=v:..] [--out file.json]
The roofline numerator is the algorithmic work of the leaf evaluations (3x3 convolutions + linear layers of one forward, 2 x MAC),
the denominator the HIP-event time of the simulation-kernel launches on the worker's stream (worker stats: ms_forward)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3
MOVES = {"c1": 400, "c2": 4, "c3": 40, "c4": 20, "c5": 40}
WARM = {"c5": 14}  # moves before the timed region (default 3: the first synchronisation of the third move takes 8 ms once): the Atari-shaped worker's first moves pay one-off host allocations
KERNEL = {"c1": "sim_kernel<3,3,4,16,-1>", "c2": "sim_kernel<9,9,20,64,2>", "c3": "sim_kernel<8,8,4,64,0>", "c4": "sim_kernel_mz<9,9,20,68,64>",
          "c5": "the leaves of a Gumbel round evaluated side by side: pre_walk_kernel + pre_tower_kernel<6,6,84,64,4|2|1> + pre_fc_kernel x 2 + pre_tail_kernel "
                "(sim_rounds.hip: the rounds whose leaves outnumber the CUs), sim_pre_pair_kernel_mz<6,6,84,64> (two workgroups per leaf: the rounds that fit the chip twice and do not "
                "use their second expected leaves) or sim_pre_kernel_mz<6,6,84,64,2> (one workgroup per leaf), then sim_kernel_mz<6,6,64,84,64> (the simulations in order); "
                "launches_by_kernel has the counts; mz_sim_rounds=false: sim_kernel_mz_cluster<6,6,84,64>"}


def flops_per_leaf_eval(d):
    """2 x MAC of the network evaluated at a (non-root) leaf: AlphaZero = the whole forward; MuZero = dynamics + prediction."""
    P, C, A, nb = d.hidden_channel_height * d.hidden_channel_width, d.num_hidden_channels, d.action_size, d.num_blocks
    cin0 = d.num_input_channels if d.type == 0 else C + d.num_action_feature_channels
    conv = 2.0 * 9 * P * (cin0 * C + 2 * nb * C * C)
    pc = -(-A // P)
    heads = 2.0 * (P * C * pc + pc * P * A)  # policy conv1x1 + FC
    if d.discrete_value_size > 1:
        # DiscreteValueNetwork (ref network_unit.py:67-87): conv1x1 C -> hc = ceil(size / P), FC hc*P -> hidden, FC hidden -> size;
        # value head: hidden = num_value_hidden_channels, reward head: hidden = C (ref muzero_atari_network.py:48,65)
        sz = d.discrete_value_size
        hc = -(-sz // P)
        for hid in (d.num_value_hidden_channels, C):
            heads += 2.0 * (P * C * hc + hc * P * hid + hid * sz)
    else:
        heads += 2.0 * (P * C * 1 + P * d.num_value_hidden_channels + d.num_value_hidden_channels)
    return conv, heads


# beyond BASELINE.json: the shapes the reference's own worker script defaults to (scripts/zero-worker.sh:39: 64 games per GPU) with a Gumbel root — pools that leave most
# CUs idle with one workgroup per game, where evaluating a Gumbel round's leaves side by side pays (round 4: MuZero board games too)
EXTRA_CONFIGS = {
    "c4g64": ("c4", "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=50:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
                    "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:zero_num_parallel_games=64"),
    # round 5: shapes beyond BASELINE.json on the one-tile tower (sim_wide.inc sim_kernel_wide), BASELINE configs[1]'s search (n = 400, 256 games, one game per CU)
    "w9x128": ("w9x128", "env_game=go:env_board_size=9:actor_num_simulation=400:zero_num_parallel_games=256"),    # 9x9 Go, 6 blocks x 128 channels
    "w9x256": ("w9x256", "env_game=go:env_board_size=9:actor_num_simulation=400:zero_num_parallel_games=256"),    # 9x9 Go, the reference's default network: 1 block x 256 channels
    "w19x64": ("w19x64", "env_game=go:env_board_size=19:actor_num_simulation=400:zero_num_parallel_games=256"),   # 19x19 Go, 6 blocks x 64 channels
    # the reference's default network (1 block x 256 channels) on the other two games with a device leaf: BASELINE configs[2]'s search on Othello, configs[0]'s on TicTacToe with 256 games
    "w8x256oth": ("w8x256oth", mz.CONFIGS["c3"]),
    "w3x256ttt": ("w3x256ttt", "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=256"),
    # shapes WITHOUT a simulation-kernel instance: the lock-step worker on the per-layer kernels (net_wide.hip conv3x3_band; MZ_NO_CONV_BAND=1 in the environment: conv3x3_any);
    # no simulation-kernel launches to time, the block's roofline.wall_frac is the figure
    "l19x128": ("l19x128", "env_game=go:env_board_size=19:actor_num_simulation=400:zero_num_parallel_games=256"),  # 19x19 Go, 6 blocks x 128 channels (one tile = 237 KB)
    "l13x96": ("l13x96", "env_game=go:env_board_size=13:actor_num_simulation=400:zero_num_parallel_games=256"),    # 13x13 Go, 6 blocks x 96 channels (no 16 * 2^k)
}
# ... and BASELINE configs[4]'s whole node (512 games) on ONE GPU — NOT the BASELINE shard (64 games per GPU): what the same kernels reach when the pool fills the chip
EXTRA_CONFIGS["c5x512"] = ("c5", mz.CONFIGS["c5"].replace("zero_num_parallel_games=64", "zero_num_parallel_games=512"))
EXTRA_CONFIGS["w9x128mz"] = ("w9x128mz", "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=50:zero_num_parallel_games=256")  # BASELINE configs[3]'s search, 6 blocks x 128 channels
EXTRA_DESCS = {
    "w9x128mz": lambda: mz.make_desc("go_9x9", 18, 9, 9, 128, 9, 9, 1, 6, 82, type_name="muzero"),
    "w9x128": lambda: mz.make_desc("go_9x9", 18, 9, 9, 128, 9, 9, 1, 6, 82),
    "w9x256": lambda: mz.make_desc("go_9x9", 18, 9, 9, 256, 9, 9, 1, 1, 82),
    "w19x64": lambda: mz.make_desc("go_19x19", 18, 19, 19, 64, 19, 19, 1, 6, 362),
    "w8x256oth": lambda: mz.make_desc("othello_8x8", 4, 8, 8, 256, 8, 8, 1, 1, 65),
    "w3x256ttt": lambda: mz.make_desc("tictactoe", 4, 3, 3, 256, 3, 3, 1, 1, 9),
    "l19x128": lambda: mz.make_desc("go_19x19", 18, 19, 19, 128, 19, 19, 1, 6, 362),
    "l13x96": lambda: mz.make_desc("go_13x13", 18, 13, 13, 96, 13, 13, 1, 6, 170),
}
MOVES.update({"w9x128": 3, "w9x256": 3, "w19x64": 2, "c5x512": 30, "w9x128mz": 10, "l19x128": 1, "l13x96": 2, "w8x256oth": 30, "w3x256ttt": 100})
WARM.update({"w9x128": 1, "w9x256": 1, "w19x64": 1, "c5x512": 14, "l19x128": 1, "l13x96": 1, "w8x256oth": 3, "w3x256ttt": 20})
KERNEL.update({"w9x128mz": "sim_kernel_mz_wide<9,9,32,144,128>", "w9x128": "sim_kernel_wide<9,9,32,128,2>", "w9x256": "sim_kernel_wide<9,9,32,256,2>", "w19x64": "sim_kernel_wide<19,19,32,64,6>",
               "w8x256oth": "sim_kernel_wide<8,8,16,256,0>", "w3x256ttt": "sim_kernel_wide<3,3,16,256,-1>",
               "l19x128": "conv3x3_band (lock-step worker: per-layer kernels)", "l13x96": "conv3x3_band (lock-step worker: per-layer kernels)"})


def _by_kernel(s0, s1, launches):
    """Gumbel rounds (C5): how many launches of which kernel the `launches` rounds + simulation stretches were (worker stats)."""
    pre = s1.get("pre_launches", 0) - s0.get("pre_launches", 0)
    if pre <= 0:
        return None
    batch = s1.get("pre_batch_launches", 0) - s0.get("pre_batch_launches", 0)
    pairs = s1.get("pre_pair_launches", 0) - s0.get("pre_pair_launches", 0)
    out = {"sim_kernel_mz": launches - pre}
    if pre - batch - pairs:
        out["sim_pre_kernel_mz"] = pre - batch - pairs
    if pairs:
        out["sim_pre_pair_kernel_mz"] = pairs
    if batch:
        out.update({"pre_walk_kernel": batch, "pre_tower_kernel": batch, "pre_fc_kernel": 2 * batch, "pre_tail_kernel": batch})
    return out


RNG_STREAMS = 16  # fixed (bench.py --rng-streams): the records do not depend on the box's CPU quota, and tests/test_gpu_baseline_nets.py checks this very mode at full size


def run_config(key, moves=None, threads=None, extra_conf="", warm=None):
    """One config on device 0: a fresh worker, `warm` untimed moves, `moves` timed moves (one run_cycles call per move, like the `-mode sp`
    loop).  Returns the block bench.py (`other_configs`) and profiles/rNN_all_configs_n1.json both carry."""
    dkey, base = EXTRA_CONFIGS.get(key, (key, None))
    base = base or mz.CONFIGS[key]
    d = (EXTRA_DESCS.get(dkey) or mz.DESCS[dkey])()
    if threads is None:
        threads = max(1, mz.usable_cpus() - 1)
    conf = f"{base}:program_seed=1:nn_file_name=synthetic.pt:zero_num_threads={threads if key != 'c1' else 1}:mz_rng_streams={RNG_STREAMS if key != 'c1' else 1}:mz_cpu_base=0{extra_conf}"
    n = int(base.split("actor_num_simulation=")[1].split(":")[0])
    games = int(base.split("zero_num_parallel_games=")[1].split(":")[0])
    wk = mz.Worker(conf, d, mz.generate_weights(d, 0))
    wk.command("start")
    moves = moves or MOVES.get(key, 20)
    wk.run_cycles((WARM.get(key, 3) if warm is None else warm) * (n + 1))
    s0 = wk.stats()
    t0 = time.perf_counter()
    popped = 0
    for _ in range(moves):
        wk.run_cycles(n + 1)
        popped += len(wk.pop_lines(wait=False))  # like the `-mode sp` loop (include/minizero/actor_group.h): the records that are complete leave between two moves
    dt = time.perf_counter() - t0
    s1 = wk.stats()
    evals = s1["leaf_evals"] - s0["leaf_evals"]
    conv, heads = flops_per_leaf_eval(d)
    gpu_ms = s1["ms_forward"] - s0["ms_forward"]
    launches = s1["sim_launches"] - s0["sim_launches"]
    sim_evals = (s1["sim_cycles"] - s0["sim_cycles"]) * games
    ach = (conv + heads) * sim_evals / (gpu_ms * 1e-3) / 1e12 if launches else None
    res = {"leaf_evals_per_sec": evals / dt, "ms_per_move": dt / moves * 1e3, "moves_per_sec": (s1["moves"] - s0["moves"]) / dt,
           "games_per_sec": (s1["games"] - s0["games"]) / dt, "games_in_pool": games, "moves_timed": moves, "records_popped_between_moves": popped, "host_threads": threads if key != "c1" else 1, "host_rng_streams": RNG_STREAMS if key != "c1" else 1,
           "leaves_evaluated_ahead": s1.get("pre_evals", 0) - s0.get("pre_evals", 0), "simulations_that_found_their_leaf": s1.get("pre_hits", 0) - s0.get("pre_hits", 0),
           "config": base,
           "roofline": {"kernel": KERNEL.get(key, KERNEL.get(dkey)), "bound": "mfma", "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": (ach / F32_MFMA_PEAK_TFLOPS) if ach else None, "launches": launches,
                        "launches_by_kernel": _by_kernel(s0, s1, launches),
                        "avg_launch_ms": gpu_ms / launches if launches else None, "flops_per_leaf_eval": conv + heads,
                        "conv3x3_flops_per_leaf_eval": conv, "leaf_evals_in_launches": sim_evals,
                        "wall_frac": (conv + heads) * evals / dt / 1e12 / F32_MFMA_PEAK_TFLOPS,
                        "timing": "HIP events on the worker's stream around every simulation-kernel launch"}}
    wk.close()
    return res


def main():
    argv = sys.argv[1:]
    out_path = os.path.join("gpurun_out", "configs.json")
    moves_override = None
    extra_conf = ""
    threads = None
    keys = []
    i = 0
    while i < len(argv):
        if argv[i] == "--moves":
            moves_override = int(argv[i + 1]); i += 2
        elif argv[i] == "--conf":  # extra configuration keys for every worker, e.g. mz_sim_split=false
            extra_conf = ":" + argv[i + 1]; i += 2
        elif argv[i] == "--threads":  # zero_num_threads of every worker (1 = the host budget of one rank of eight on a 16-CPU quota)
            threads = int(argv[i + 1]); i += 2
        elif argv[i] == "--out":
            out_path = argv[i + 1]; i += 2
        else:
            keys.append(argv[i]); i += 1
    out = {}
    for key in keys or ["c1", "c2", "c3", "c4", "c5"]:
        out[key] = run_config(key, moves_override, threads, extra_conf)
        print(key, json.dumps(out[key]), flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
