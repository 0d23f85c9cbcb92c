#!/usr/bin/env python3
"""bench.py — self-play hot path on N MI355X of one node.

A "step" is one lock-step cycle of the worker: one MCTS simulation (select -> leaf environment + features ->
network forward -> expand + backup) for every game of the batch, i.e. `games` leaf evaluations per GPU.
Workload = BASELINE.json configs[1]: 9x9 Go AlphaZero, n=400, 6 blocks x 64 channels, 256 parallel games
per GPU, synthetic fixed-weight network (deterministic generator, seed 0), games from the empty board.
Games shard across GPUs with no data-path collective: weak scaling, rank r seeds program_seed + r.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak (== f32 vector peak)
GO_GAME_LENGTH_CAP = 163       # 9x9 Go: terminal once #actions > 2*81 (ref go.cpp:253-254); synthetic nets neither pass twice nor resign


def cpu_baseline(conf, seconds):
    """CPU restatement of the reference's actor (oracle/, kind "port") on the same workload, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # cpu_baseline leg only
    d = O.desc_c2()
    w = O.gen_weights(d, 0)
    import minizero_amd as mz
    cores = mz.usable_cpus()
    host_threads = max(1, min(64, cores))
    g = O.OracleGroup(conf + f":zero_num_threads=1:oracle_throughput_threads={host_threads}", d, w)
    g.cycles(1)  # warm-up (thread start, page-in)
    t0 = time.time()
    cycles = 0
    while time.time() - t0 < seconds or cycles < 3:
        g.cycles(1)
        cycles += 1
    dt = time.time() - t0
    evals = cycles * 256
    return {"value": evals / dt, "unit": "leaf-evals/s", "cores": cores, "kind": "port",
            "sample": f"{cycles} lock-step cycles x 256 games of the same workload ({dt:.1f} s): tree+env phase on "
                      f"{host_threads} threads, network forward (f32, same arithmetic as the GPU path) on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=802)   # 2 full moves of every game (401 cycles per move)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--threads", type=int, default=0, help="host threads for env/feature/candidate work (0 = cores / ranks)")
    ap.add_argument("--lanes", type=int, default=1, help="software-pipelined lanes the 256 games are split into")
    ap.add_argument("--zero-copy", type=int, default=3)
    ap.add_argument("--pin", type=int, default=1, help="pin the host threads of rank r to CPUs [r*threads, (r+1)*threads)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import minizero_amd as mz
    from minizero_amd.dist import Group, shard_seed
    if not torch.cuda.is_available() or mz.device_count() < 1:
        raise SystemExit("bench.py needs a GPU (libmzgpu has no CPU path)")
    torch.cuda.set_device(local_rank)
    grp = Group("nccl")  # RCCL over xGMI; only used for the weight broadcast, barriers and the final reductions

    cores = os.cpu_count() or 1
    usable = mz.usable_cpus()  # affinity mask capped by the cgroup CPU quota: spinning past the quota gets the container throttled
    threads = args.threads or max(1, min(32, usable // max(1, world) - 1))  # spin-wait pool incl. the calling thread
    base_conf = mz.CONFIGS["c2"].replace("zero_num_parallel_games=256", f"zero_num_parallel_games={args.games}")
    conf = (f"{base_conf}:zero_num_threads={threads}:mz_pipeline_lanes={args.lanes}:mz_zero_copy={args.zero_copy}:mz_cpu_base={local_rank * threads if args.pin else -1}:program_seed={shard_seed(1, rank)}:"
            "nn_file_name=synthetic_go_6bx64_seed0.pt")
    desc = mz.DESCS["c2"]()
    # the optional synchronous weight broadcast (load_model fan-out): rank 0's blob is the one every rank loads
    weights = grp.broadcast_weights(mz.generate_weights(desc, 0))
    worker = mz.Worker(conf, desc, weights, device=local_rank)
    worker.command("start")

    assert worker.run_cycles(args.warmup) == args.warmup
    s0 = worker.stats()
    grp.barrier()
    t0 = time.perf_counter()
    assert worker.run_cycles(args.steps) == args.steps
    grp.barrier()
    dt = time.perf_counter() - t0
    s1 = worker.stats()
    dt = float(grp.reduce([dt], "max")[0])
    moves, games_done = grp.reduce([s1["moves"] - s0["moves"], s1["games"] - s0["games"]], "sum")

    if rank == 0:
        evals = args.games * args.steps * world
        value = evals / dt
        # dominant kernel: sim_kernel (sim.hip) — every game's whole simulations (select, Go leaf, residual tower on the f32 MFMA pipe,
        # heads, candidates, expand + backup) in one launch per run of cycles.  Its GPU time inside the timed region comes from HIP
        # events recorded on the worker's own stream around every launch (worker stats: ms_forward); the algorithmic work per step is
        # 256 leaf evaluations x 73.3 MFLOP of 3x3 convolutions.
        net = worker.net()
        ms_fwd, ms_tower, fl_tower = net.time_forward(args.games, 100)          # the stand-alone tower + heads kernels, for reference
        ms_layer, fl_layer, bytes_layer = net.time_tower_conv(args.games, 200)  # the stand-alone per-layer kernel, for reference
        gpu_ms = s1["ms_forward"] - s0["ms_forward"]
        flops_per_step = fl_tower  # one step = one forward of args.games samples
        resident = gpu_ms > 0.5 * dt * 1e3  # the simulation kernel ran (otherwise ms_forward is host wait time of the lock-step path)
        achieved = (flops_per_step * args.steps) / (gpu_ms * 1e-3) / 1e12 if resident else fl_tower / (ms_tower * 1e-3) / 1e12
        # HBM traffic per step from the rocprofv3 PMC pass committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes; the x2 on
        # gfx950 was re-calibrated on a known-size copy kernel with the same 4-B/lane access, see profiles/README.md)
        traffic = None
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_sim.json")))["bytes_per_step"]
        except Exception:
            pass
        phase = {k: round((s1[k] - s0[k]) / args.steps, 4) for k in ("ms_select", "ms_env", "ms_forward", "ms_expand", "ms_move", "ms_total")}
        out = {
            "metric": "self-play leaf-evals/s (9x9 Go AlphaZero n=400)", "value": value, "unit": "leaf-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 9x9 Go AlphaZero, n=400, 6 blocks x 64 ch, 256 parallel games per GPU, "
                                   "synthetic fixed-weight net (seed 0), Dirichlet noise + random rotation + softmax-count moves (reference defaults)",
                       "games_per_gpu": args.games, "actor_num_simulation": 400, "pipeline_lanes": args.lanes, "host_threads_per_gpu": threads, "host_cores": cores, "host_cpus_usable": usable,
                       "sharding": f"{world} x independent actor pools, no data-path collective"},
            "moves_per_sec": moves / dt, "games_finished": games_done,
            "games_per_sec": (games_done / dt) if games_done else None,
            "games_per_sec_projected": moves / dt / GO_GAME_LENGTH_CAP,
            "games_per_sec_note": f"projected = moves/s / {GO_GAME_LENGTH_CAP} (9x9 games of the synthetic net run to the move cap)",
            "per_step_ms": phase,
            "forward": {"ms_per_forward": ms_fwd, "ms_tower_per_forward": ms_tower,
                        "per_layer_kernel": {"us_per_launch": ms_layer * 1e3, "tflops": fl_layer / (ms_layer * 1e-3) / 1e12}},
            "roofline": {"kernel": "sim_kernel<9,9,20,64,2> (per game: PUCT select, Go leaf position/planes/legal mask, stem + 12 x conv3x3 64->64 on "
                                   "v_mfma_f32_16x16x4_f32 with activations in LDS, heads, candidate sort, expand + backup)" if resident else
                                   "tower_fused<9,9,20,64>",
                         "bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_sim.json (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE pass of this bench, bytes per step)",
                         "flops_per_step": flops_per_step, "gpu_us_per_step": gpu_ms / args.steps * 1e3,
                         "timing": "HIP events on the worker's stream around every sim_kernel launch of the timed region",
                         # per step, without the weights (1.9 MB: they stay in the XCDs' L2s between steps): a children block written + one read per game
                         "compulsory_bytes_per_step": args.games * 2.0 * 2600,
                         "tower_alone": {"kernel": "tower_fused<9,9,20,64> (the same tower as a stand-alone launch of 256 samples)",
                                         "us_per_launch": ms_tower * 1e3, "achieved": fl_tower / (ms_tower * 1e-3) / 1e12,
                                         "frac": fl_tower / (ms_tower * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            del worker
            out["cpu_baseline"] = cpu_baseline(base_conf + ":program_seed=1:nn_file_name=synthetic_go_6bx64_seed0.pt", args.cpu_seconds)
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
