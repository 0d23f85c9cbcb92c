"""CPU test of the N > 1 path with world_size 2 on the gloo backend: the plumbing bench.py/the worker use for
multi-GPU runs (weight broadcast on load_model, barrier, max/sum reductions, shard seeds and game split).
Games themselves never communicate, so the data path has no collective to test."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    torch.set_num_threads(1)
    import minizero_amd as mz
    from minizero_amd.dist import Group, shard_seed, games_for_rank
    import oracle_lib as O
    grp = Group("gloo", device="cpu")
    assert (grp.rank, grp.world) == (rank, world)
    desc = mz.DESCS["c1"]()
    # rank 0 owns the "loaded model"; the others start from garbage and must end up with rank 0's blob
    w = mz.generate_weights(desc, 0) if rank == 0 else np.full(mz.param_count(desc), 7.0, np.float32)
    w = grp.broadcast_weights(w)
    assert np.array_equal(w, mz.generate_weights(desc, 0))
    grp.barrier()
    # one independent actor pool per rank (the CPU oracle stands in for the GPU worker here), rank-specific seed
    conf = f"env_game=tictactoe:actor_num_simulation=8:zero_num_parallel_games={games_for_rank(9, rank, world)}:program_seed={shard_seed(1, rank)}"
    g = O.OracleGroup(conf, O.desc_c1(), w)
    g.cycles(9 * 30)
    lines = g.lines()
    tot = grp.reduce([len(lines), g.leaf_evals()], "sum")
    tmax = grp.reduce([1.0 + rank], "max")
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.array([tot[0], tot[1], tmax[0], len(lines), g.leaf_evals()]))
    with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
        f.write("\n".join(lines))
    grp.close()


def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as tmp
    port = _free_port()
    tmp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    # reductions agree on both ranks and equal the per-rank sums / max
    assert r0[0] == r1[0] == r0[3] + r1[3] and r0[1] == r1[1] == r0[4] + r1[4] and r0[2] == r1[2] == 2.0
    # 9 games split 5 + 4, n+1 = 9 evaluations per move per game
    assert r0[4] == 5 * 9 * 30 and r1[4] == 4 * 9 * 30
    l0, l1 = (tmp_path / "r0.txt").read_text().splitlines(), (tmp_path / "r1.txt").read_text().splitlines()
    assert l0 and l1 and l0[0] != l1[0], "ranks must play different games (program_seed + rank)"


def test_shard_arithmetic():
    sys.path.insert(0, ROOT)
    from minizero_amd.dist import games_for_rank, shard_seed
    assert [games_for_rank(512, r, 8) for r in range(8)] == [64] * 8  # BASELINE configs[4]: 512 games over 8 GPUs
    assert sum(games_for_rank(10, r, 4) for r in range(4)) == 10 and [games_for_rank(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert [shard_seed(5, r) for r in range(3)] == [5, 6, 7]
