#!/usr/bin/env python3
"""bench.py — self-play hot path on N MI355X of one node.

A "step" is one MOVE of every game of the batch: actor_num_simulation + 1 = 401 lock-step cycles of the reference
(ref actor/actor_group.cpp:136-147: one MCTS simulation — select -> leaf environment + features -> network forward ->
expand + backup — of every game per cycle), followed by the per-move host work (move decision, Dirichlet noise of the next
root, record strings, root snapshot upload: ref actor_group.cpp:116-134, zero_actor.cpp:74-98).  One step = games x 401 leaf
evaluations per GPU, so a driver call with `--steps 20 --warmup 5` times 20 whole moves after 5 warm-up moves.
Workload = BASELINE.json configs[1]: 9x9 Go AlphaZero, n=400, 6 blocks x 64 channels, 256 parallel games per GPU, synthetic
fixed-weight network (deterministic generator, seed 0), games from the empty board.  Games shard across GPUs with no data-path
collective: weak scaling, rank r seeds program_seed + r.

After the timed region (not part of `value`) the same worker keeps playing for --game-moves more moves (default 164 = one full
game length: 9x9 games of the synthetic net run to the 163-move cap) and the line reports MEASURED games/s = finished
`SelfPlay true` records / wall-clock of that leg.

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
N_SIM = 400


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
    out = {"value": evals / dt, "unit": "leaf-evals/s", "cores": cores, "kind": "port",
           # the REAL reference (its own actor + LibTorch-CPU forward) was timed once, in the build container, on this workload: BASELINE.md "C2" row.
           # It cannot travel to the GPU box; the port's k-ordered fmaf forward is slower than LibTorch's, so read the GPU / port ratio with that in mind.
           "reference_in_build_container": {"value": 3300.0, "unit": "leaf-evals/s", "cores": 8, "source": "BASELINE.md (8 vCPU, 35 s sample of the same workload)"},
           "sample": f"{cycles} lock-step cycles x 256 games of the same workload ({dt:.1f} s): tree+env phase on "
                     f"{host_threads} threads, network forward (f32, same arithmetic as the GPU path) on {cores} threads",
           "moves_per_sec": evals / dt / (N_SIM + 1), "games_per_sec_at_163_moves": evals / dt / (N_SIM + 1) / 163.0}
    del g
    # BASELINE.json configs[0] as stated: TicTacToe AlphaZero n=16, 2 blocks x 16 channels, 8 parallel games, ONE actor thread
    # (the reference's deterministic contract), whole games
    d1 = O.desc_c1()
    g1 = O.OracleGroup(mz.CONFIGS["c1"] + ":program_seed=1:nn_file_name=synthetic_ttt_2bx16_seed0.pt", d1, O.gen_weights(d1, 0))
    g1.cycles(17)
    t0 = time.time()
    c1_cycles, e0, n0 = 0, g1.leaf_evals(), g1.games()
    while time.time() - t0 < 2.0:
        g1.cycles(17 * 8)
        c1_cycles += 17 * 8
    dt1 = time.time() - t0
    out["c1_tictactoe"] = {"leaf_evals_per_sec": (g1.leaf_evals() - e0) / dt1, "games_per_sec": (g1.games() - n0) / dt1, "cores": 1,
                           "sample": f"BASELINE configs[0]: TicTacToe AZ n=16, 2bx16, 8 games, {c1_cycles} cycles ({dt1:.1f} s), 1 actor thread"}
    return out


def kernel_fingerprint():
    """sha256 over the device sources of libmzgpu (every .hip / .inc and every header under minizero_amd/csrc except the host-only ones, names and contents, sorted):
    what a committed counter summary was taken on.  (`.git` does not travel to the GPU box, so a commit id cannot be read there.)"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "minizero_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".inc")) and name not in ("config.h", "env.h", "host_threads.h", "common_host.h"):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    if "--kernel-fingerprint" in sys.argv:
        print(kernel_fingerprint())
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed moves of every game (401 cycles each)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed warm-up moves")
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--game-moves", type=int, default=164, help="moves of the games/s leg after the timed region (0 = skip)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the per-move work (0 = cores / ranks); does NOT change the records (see --rng-streams)")
    ap.add_argument("--rng-streams", type=int, default=16,
                    help="host RNG generators per GPU (mz_rng_streams; ref actor_group.cpp:66-70: one per slave thread).  A FIXED number, so that the records behind `value` do not "
                         "depend on the box's CPU quota: tests/test_gpu_baseline_nets.py::test_*_full_size_*_bench_streams compares exactly this mode with the oracle at full size")
    ap.add_argument("--one-stream-moves", type=int, default=-1, help="timed moves of the extra leg with ONE RNG stream (the deterministic contract every parity test runs on; -1 = --steps, 0 = skip)")
    ap.add_argument("--lanes", type=int, default=1, help="software-pipelined lanes the games are split into")
    ap.add_argument("--zero-copy", type=int, default=3)
    ap.add_argument("--pin", type=int, default=1, help="pin the host threads of rank r to CPUs [r*threads, (r+1)*threads)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-moves", type=int, default=30, help="timed moves of the short C3 / C4 / C5 legs after the headline (0 = skip; N=1 only)")
    ap.add_argument("--extra-conf", default="", help="extra k=v:k=v keys (profiling variants only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the barriers / reductions / weight broadcast (nccl = RCCL; gloo: the world > 1 path on a box with fewer "
                         "GPUs than ranks, with --device-map)")
    ap.add_argument("--device-map", default=os.environ.get("MZ_BENCH_DEVICE_MAP", ""),
                    help="TEST HOOK: comma-separated physical device of every local rank, e.g. 0,0 = two ranks on GPU 0 (needs --backend gloo: RCCL refuses "
                         "two ranks on one device); default: local rank r drives device r")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"],
                    help="arithmetic of the residual tower: f32 (default, the headline: bit-exact against the oracle) or the opt-in bf16x3 "
                         "(split-bf16 operands on the 16-bit MFMA, outputs within 1e-3, records not bit-identical) — reported as its own line")
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
    device = local_rank
    if args.device_map:
        dm = [int(t) for t in args.device_map.split(",")]
        if len(dm) != world or any(o < 0 or o >= mz.device_count() for o in dm):
            raise SystemExit(f"--device-map {args.device_map}: need {world} ordinals below {mz.device_count()}")
        if args.backend == "nccl" and len(set(dm)) != len(dm):
            raise SystemExit("--device-map puts two ranks on one device: use --backend gloo (RCCL needs one device per rank)")
        device = dm[local_rank]
    torch.cuda.set_device(device)
    # RCCL over xGMI (or gloo, test hook above); only used for the weight broadcast, barriers and the final reductions
    grp = Group(args.backend, device=torch.device("cuda", device) if args.backend == "nccl" else None)

    cores = os.cpu_count() or 1
    usable = mz.usable_cpus()  # affinity mask capped by the cgroup CPU quota: spinning past the quota gets the container throttled
    threads = args.threads or max(1, min(32, usable // max(1, world) - 1))  # spin-wait pool incl. the calling thread
    base_conf = mz.CONFIGS["c2"].replace("zero_num_parallel_games=256", f"zero_num_parallel_games={args.games}")
    # mz_rng_streams=S: S generators as the reference has them with zero_num_threads = S slave threads (actor_group.cpp:66-70), the games statically partitioned over them,
    # so that the RNG-ordered host section of a move runs on the pool's threads whatever their number; rank r's generators are program_seed + r * S + t: no two ranks share one
    streams = max(1, args.rng_streams)

    def make_conf(nstreams):
        return (f"{base_conf}:zero_num_threads={threads}:mz_rng_streams={nstreams}:mz_pipeline_lanes={args.lanes}:mz_zero_copy={args.zero_copy}:mz_cpu_base={local_rank * threads if args.pin else -1}:"
                f"program_seed={shard_seed(1, rank, streams)}:nn_file_name=synthetic_go_6bx64_seed0.pt" + (":" + args.extra_conf if args.extra_conf else "") +
                (":mz_nn_precision=bf16x3" if args.precision == "bf16x3" else "") +
                # two RANKS on one device (test hook): kernels that assume an idle GPU to themselves stay off (another process cannot be seen from inside the worker)
                (":mz_sim_round_pairs=false" if args.device_map and len(set(args.device_map.split(","))) < world else ""))
    conf = make_conf(streams)
    desc = mz.DESCS["c2"]()
    # the optional synchronous weight broadcast (load_model fan-out): rank 0's blob is the one every rank loads
    weights = grp.broadcast_weights(mz.generate_weights(desc, 0))
    worker = mz.Worker(conf, desc, weights, device=device)
    worker.command("start")

    cpm = N_SIM + 1  # cycles per move
    if args.warmup > 0:
        assert worker.run_cycles(args.warmup * cpm) == args.warmup * cpm
    s0 = worker.stats()
    grp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):  # one call per move, like the `-mode sp` loop of the facade (commands are polled between moves)
        assert worker.run_cycles(cpm) == cpm
    grp.barrier()
    dt = time.perf_counter() - t0
    s1 = worker.stats()
    dt = float(grp.reduce([dt], "max")[0])
    moves, games_done = grp.reduce([s1["moves"] - s0["moves"], s1["games"] - s0["games"]], "sum")

    # ---- games/s leg: the same worker keeps playing (steady state: games at every stage of their 163 moves) ----
    g_dt, g_games, g_moves, g_evals = 0.0, 0.0, 0.0, 0.0
    if args.game_moves > 0:
        worker.pop_lines()
        grp.barrier()
        tg = time.perf_counter()
        lines = []
        for _ in range(args.game_moves):
            assert worker.run_cycles(cpm) == cpm
            lines += worker.pop_lines(wait=False)  # the `-mode sp` loop pops what is complete between two moves (include/minizero/actor_group.h)
        grp.barrier()
        g_dt = time.perf_counter() - tg
        s2 = worker.stats()
        lines += worker.pop_lines()
        finished = sum(1 for l in lines if l.startswith("SelfPlay true "))
        assert finished == s2["games"] - s1["games"]
        g_dt = float(grp.reduce([g_dt], "max")[0])
        g_games, g_moves, g_evals = grp.reduce([finished, s2["moves"] - s1["moves"], s2["leaf_evals"] - s1["leaf_evals"]], "sum")

    # what every rank ran with (device, seed, CPU range, first record): gathered as a sum of one-hot rows, printed by rank 0
    import zlib
    row = [0.0] * (5 * world)
    row[5 * rank:5 * rank + 5] = [float(device), float(shard_seed(1, rank, streams)), float(local_rank * threads if args.pin else -1), float(threads),
                                  float(zlib.crc32(worker.peek_records(1)[0].encode()))]
    per_rank = grp.reduce(row, "sum")

    # ---- one-stream leg (not `value`): the same workload with mz_rng_streams=1, the deterministic contract of the reference's zero_num_threads=1 that every
    # parity test runs on — a round-to-round comparable number next to the S-stream headline.  A second worker on the same device; the first one idles (its pool sleeps)
    one_stream = None
    osm = args.steps if args.one_stream_moves < 0 else args.one_stream_moves
    if osm > 0 and args.precision == "f32" and streams != 1:
        w1 = mz.Worker(make_conf(1), desc, weights, device=device)
        w1.command("start")
        assert w1.run_cycles(max(1, args.warmup) * cpm) == max(1, args.warmup) * cpm
        grp.barrier()
        t1 = time.perf_counter()
        for _ in range(osm):
            assert w1.run_cycles(cpm) == cpm
        grp.barrier()
        dt1 = float(grp.reduce([time.perf_counter() - t1], "max")[0])
        crc1 = zlib.crc32(w1.peek_records(1)[0].encode())
        w1.close()
        one_stream = {"value": args.games * cpm * osm * world / dt1, "unit": "leaf-evals/s", "ms_per_step": dt1 / osm * 1e3, "steps": osm, "warmup": max(1, args.warmup),
                      "host_rng_streams_per_gpu": 1, "first_record_crc32_rank0": crc1,
                      "note": "mz_rng_streams=1 (one generator, games in index order: the reference with zero_num_threads=1), same workload, timed after the headline on a fresh worker"}
    if rank == 0:
        evals = args.games * cpm * args.steps * world
        value = evals / dt
        # dominant kernel: sim_kernel (sim.hip) — every game's whole simulations (select, Go leaf, residual tower on the f32 MFMA pipe,
        # heads, candidates, expand + backup); a move's 401 simulations go out as three launches queued back to back (1 + 16 + 384 simulations:
        # the host's noise and rotation draws for the later parts overlap the earlier ones, worker.cpp runCyclesSim).  Its GPU time
        # inside the timed region comes from HIP events recorded on the worker's own stream around every launch (worker stats:
        # ms_forward); the algorithmic work per leaf evaluation is 73.3 MFLOP of 3x3 convolutions.
        net = worker.net()
        ms_fwd, ms_tower, fl_tower = net.time_forward(args.games, 100)          # the stand-alone tower + heads kernels, for reference
        ms_layer, fl_layer, bytes_layer = net.time_tower_conv(args.games, 200)  # the stand-alone per-layer kernel, for reference
        gpu_ms = s1["ms_forward"] - s0["ms_forward"]
        launches = s1["sim_launches"] - s0["sim_launches"]
        flops_per_eval = fl_tower / args.games
        flops_total = flops_per_eval * args.games * cpm * args.steps  # this rank's timed region
        resident = launches > 0
        if not resident:
            raise SystemExit("bench.py: the simulation kernel did not run (sim_launches == 0)")
        achieved = flops_total / (gpu_ms * 1e-3) / 1e12
        bf = args.precision == "bf16x3"
        # bf16x3: every product is three MFMAs (hi*hi, hi*lo, lo*hi) at the 16-bit rate: the roofline is the dense bf16 peak against the
        # MFMA work actually issued (3 x the f32-equivalent FLOPs)
        peak = 2500.0 if bf else F32_MFMA_PEAK_TFLOPS
        issued = achieved * (3.0 if bf else 1.0)
        # HBM traffic from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes; the x2 on
        # gfx950 per MI355X_MICROARCH.md, re-calibrated on a known-size copy kernel with the same 4-B/lane access, see profiles/README.md)
        # The counters are not collected in this run (a --pmc pass is a rocprofv3 run of its own: tools/refresh_c2_pmc.sh); the summary file records which kernel
        # SOURCES it was taken on (kernel_fingerprint below), and a file whose fingerprint is not the running build's is reported as stale, not as this build's.
        traffic, traffic_src, traffic_stale = None, None, None
        fp_now = kernel_fingerprint()
        for name in ("r06_pmc_sim.json", "r05_pmc_sim.json", "r04_pmc_sim.json", "r03_pmc_sim.json", "r02_pmc_sim.json", "r01_pmc_sim.json"):
            try:
                j = json.load(open(os.path.join(ROOT, "profiles", name)))
                per_cycle = j.get("bytes_per_cycle", j.get("bytes_per_step"))
                traffic = per_cycle * cpm * args.steps / launches
                traffic_stale = j.get("kernel_fingerprint") != fp_now
                traffic_src = (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this bench; bytes per lock-step cycle x cycles per average launch)" +
                               (f" — STALE: taken on kernel sources {j.get('kernel_fingerprint', 'unrecorded')}, this build is {fp_now}" if traffic_stale else
                                f"; taken on this build's kernel sources ({fp_now})"))
                break
            except Exception:
                pass
        phase = {k: round((s1[k] - s0[k]) / args.steps, 4) for k in ("ms_select", "ms_env", "ms_forward", "ms_expand", "ms_move", "ms_total")}
        out = {
            "metric": "self-play leaf-evals/s (9x9 Go AlphaZero n=400)" + (" — OPT-IN bf16x3 tower, not the headline" if bf else ""), "value": value, "unit": "leaf-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3 (split-bf16 MFMA inputs, f32 accumulate; outputs within 1e-3 of f32)" if bf else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 9x9 Go AlphaZero, n=400, 6 blocks x 64 ch, 256 parallel games per GPU, "
                                   "synthetic fixed-weight net (seed 0), Dirichlet noise + random rotation + softmax-count moves (reference defaults)",
                       "step": "one move of every game = 401 lock-step cycles = games x 401 leaf evaluations per GPU, per-move host work included",
                       "games_per_gpu": args.games, "actor_num_simulation": N_SIM, "leaf_evals_per_step": args.games * cpm * world,
                       "pipeline_lanes": args.lanes, "host_threads_per_gpu": threads, "host_rng_streams_per_gpu": streams, "host_cores": cores, "host_cpus_usable": usable,
                       "sharding": f"{world} x independent actor pools, no data-path collective", "backend": args.backend,
                       "ranks": [{"rank": r, "device": int(per_rank[5 * r]), "program_seed": int(per_rank[5 * r + 1]), "cpu_base": int(per_rank[5 * r + 2]),
                                  "host_threads": int(per_rank[5 * r + 3]), "first_record_crc32": int(per_rank[5 * r + 4])} for r in range(world)]},
            "moves_per_sec": moves / dt, "games_finished_in_timed_region": games_done,
            "one_rng_stream": one_stream,
            "games_per_sec": (g_games / g_dt) if g_dt > 0 else None,
            "games_leg": {"moves_per_game_slot": args.game_moves, "seconds": g_dt, "games_finished": g_games, "moves": g_moves,
                          "leaf_evals_per_sec": (g_evals / g_dt) if g_dt > 0 else None,
                          "note": "measured after the timed region on the same workers: finished `SelfPlay true` records / wall-clock "
                                  "(9x9 games of the synthetic net mostly run to the 163-move cap, ref go.cpp:253-254)"},
            "per_step_ms": phase,
            "forward": {"ms_per_forward": ms_fwd, "ms_tower_per_forward": ms_tower,
                        "per_layer_kernel": {"us_per_launch": ms_layer * 1e3, "tflops": fl_layer / (ms_layer * 1e-3) / 1e12}},
            "roofline": {"kernel": ("sim_kernel<9,9,20,64,2,BF=true> (per game: PUCT select, Go leaf position/planes/legal mask, stem + 12 x conv3x3 64->64 as three "
                                    "split-bf16 products on v_mfma_f32_16x16x32_bf16 with activations in LDS, heads, candidate sort, expand + backup)") if bf else
                                   ("sim_kernel<9,9,20,64,2> (per game: PUCT select, Go leaf position/planes/legal mask, stem + 12 x conv3x3 64->64 on "
                                    "v_mfma_f32_16x16x4_f32 with activations in LDS, heads, candidate sort, expand + backup)"),
                         "bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s",
                         "frac": issued / peak, "f32_equivalent_tflops": achieved, "traffic": None if bf else traffic, "traffic_source": None if bf else traffic_src, "traffic_stale": None if bf else traffic_stale,
                         "launches": launches, "avg_launch_ms": gpu_ms / launches, "flops_per_avg_launch": flops_total / launches,
                         "flops_per_leaf_eval": flops_per_eval, "gpu_ms_per_step": gpu_ms / args.steps,
                         "timing": "HIP events on the worker's stream around every sim_kernel launch of the timed region (a move = three launches of 1 + 16 + 384 cycles queued back to back, so that the host's noise / rotation draws overlap them)",
                         # per cycle, without the weights (1.9 MB: they stay in the XCDs' L2s): a children block written + one read per game
                         "compulsory_bytes_per_avg_launch": args.games * 2.0 * 2600 * cpm * args.steps / launches,
                         "tower_alone": {"kernel": "tower_fused<9,9,20,64> (the same tower as a stand-alone launch of 256 samples)",
                                         "us_per_launch": ms_tower * 1e3, "achieved": fl_tower / (ms_tower * 1e-3) / 1e12,
                                         "frac": fl_tower / (ms_tower * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}},
        }
        if world == 1 and args.other_moves > 0 and not bf:
            # BASELINE configs[2..4] on the same box, same worker code as tools/run_configs.py: a fresh worker per config, a few warm-up moves,
            # then `--other-moves` timed moves (one run_cycles call per move); leaf-evals/s by wall clock, roofline by HIP events around the launches
            worker.close()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import run_configs
            out["other_configs"] = {}
            for key in ("c3", "c4", "c5"):
                # (C5's moves are 3.5 ms each: 30 of them are 0.1 s, and one host hiccup on a loaded box showed as 0.64 M instead of 0.92 M in a round-5 line — its leg is
                # twenty times as many moves: 600 moves = 2.1 s of timed region since round 6)
                r = run_configs.run_config(key, moves=args.other_moves * (20 if key == "c5" else 1))
                out["other_configs"][key] = {"workload": r["config"], "leaf_evals_per_sec": r["leaf_evals_per_sec"], "ms_per_move": r["ms_per_move"],
                                             "games_in_pool": r["games_in_pool"], "moves_timed": r["moves_timed"], "host_threads": r["host_threads"],
                                             **({"note": "600 timed moves (2.1 s) cross three sequence boundaries of every game: the OBS tags of 64 x 200 moves (6 MB each) are gzip-compressed on the host "
                                                         "beside the search; a leg without a boundary (150 moves, what rounds 3-5 timed) is 2-3 % faster on the same box: profiles/r06_gpu_sanitize.txt"} if key == "c5" else {}),
                                             "roofline": {k: r["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "launches_by_kernel", "flops_per_leaf_eval", "wall_frac")}}
        if world == 1 and not args.no_cpu_baseline:
            del worker
            out["cpu_baseline"] = cpu_baseline(base_conf + ":program_seed=1:nn_file_name=synthetic_go_6bx64_seed0.pt", args.cpu_seconds)
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
