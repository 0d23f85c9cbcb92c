"""bench.py's world > 1 path executed on ONE GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with both ranks mapped onto GPU 0
(--device-map 0,0) and gloo standing in for RCCL (RCCL refuses two ranks on one device).  Everything else is the path the driver's 8-GPU run takes: rank / local
rank / world from the environment, `Group` rendezvous on 127.0.0.1, the weight broadcast, per-rank seeds (ref actor_group.cpp:66-70: program_seed + id) and CPU
pinning ranges, the barriers around the timed region, the max-over-ranks time and the sum-over-ranks counts, ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_world_size_two_on_one_gpu(mz):
    games, steps = 32, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--game-moves", "0", "--no-cpu-baseline", "--other-moves", "0",
           "--games", str(games), "--backend", "gloo", "--device-map", "0,0", "--threads", "2"]
    env = dict(os.environ)
    env.pop("MZ_DEVICE_MAP", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == steps and j["scaling"] == "weak" and j["config"]["backend"] == "gloo"
    assert j["config"]["leaf_evals_per_step"] == 2 * games * 401  # both ranks' games
    assert abs(j["value"] - j["config"]["leaf_evals_per_step"] * steps / (j["ms_per_step"] * steps * 1e-3)) < 1e-6 * j["value"]
    r0, r1 = j["config"]["ranks"]
    assert (r0["rank"], r1["rank"]) == (0, 1) and r0["device"] == r1["device"] == 0
    # ref actor_group.cpp:66-70: slave thread `id` seeds program_seed + id; a rank has S = --rng-streams generators (mz_rng_streams=S, whatever its thread count),
    # rank r the ids r * S .. r * S + S - 1
    S = j["config"]["host_rng_streams_per_gpu"]
    assert S == 16 and r1["program_seed"] == r0["program_seed"] + S and r0["host_threads"] == r1["host_threads"] == 2
    assert j["one_rng_stream"]["value"] > 0 and j["one_rng_stream"]["host_rng_streams_per_gpu"] == 1
    assert r0["first_record_crc32"] != r1["first_record_crc32"]              # different seeds, different games
    a0, a1 = (range(r["cpu_base"], r["cpu_base"] + r["host_threads"]) for r in (r0, r1))
    assert not set(a0) & set(a1), "the ranks' CPU pinning ranges overlap"
    assert j["moves_per_sec"] > 0 and j["roofline"]["launches"] in (2 * steps, 3 * steps)  # rank 0: two or three launches per move (worker.cpp runCyclesSim)


def test_bench_world_size_eight_on_one_gpu(mz):
    """The launch the driver makes on an 8-GPU node — `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` — rehearsed with eight ranks on GPU 0
    (gloo, --device-map 0,...,0), 8 games per rank: one JSON line, eight distinct seeds spaced by the RNG streams per rank, eight different first records,
    whole-job value = 8 ranks' leaf evaluations over the max-over-ranks time."""
    games, steps = 8, 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(steps), "--warmup", "1", "--game-moves", "0", "--no-cpu-baseline", "--other-moves", "0",
           "--games", str(games), "--backend", "gloo", "--device-map", ",".join(["0"] * 8), "--threads", "1", "--one-stream-moves", "0"]
    env = dict(os.environ)
    env.pop("MZ_DEVICE_MAP", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["leaf_evals_per_step"] == 8 * games * 401
    assert abs(j["value"] - j["config"]["leaf_evals_per_step"] * steps / (j["ms_per_step"] * steps * 1e-3)) < 1e-6 * j["value"]
    ranks = j["config"]["ranks"]
    S = j["config"]["host_rng_streams_per_gpu"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["device"] == 0 for r in ranks)
    assert [r["program_seed"] for r in ranks] == [1 + S * r for r in range(8)]
    assert len({r["first_record_crc32"] for r in ranks}) == 8
    assert len({r["cpu_base"] for r in ranks}) == 8
