"""GPU test of the process boundary: apps/mzgpu_sp speaks the reference's `-mode sp` stdin/stdout protocol
(ref scripts/zero-worker.sh:160-162, actor/actor_group.cpp:24-50,200-252) and the facade headers compile."""
import os
import subprocess
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sp_executable_protocol(mz, oracle, tmp_path):
    from minizero_amd.export_weights import write_mzw
    exe = os.path.join(ROOT, "apps", "mzgpu_sp")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    d = mz.DESCS["c1"]()
    w = mz.generate_weights(d, 0)
    pt = str(tmp_path / "weight_iter_0.pt")  # the server names .pt files; the worker opens the sibling .mzw
    write_mzw(pt[:-3] + ".mzw", d, w)
    cfg = tmp_path / "ttt.cfg"
    cfg.write_text("# a reference-style cfg with keys this path ignores\nzero_server_port=9999\nlearner_batch_size=1024\n"
                   "actor_num_simulation=16 # simulation number of MCTS\nzero_num_parallel_games=8\nzero_num_threads=2\n")
    conf_str = f"nn_file_name={pt}:program_auto_seed=false:program_seed=1:program_quiet=true"
    p = subprocess.Popen([exe, "-conf_file", str(cfg), "-conf_str", conf_str, "-mode", "sp", "-game", "tictactoe"],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    p.stdin.write("keep_alive\nstart\n")
    p.stdin.flush()
    lines = []
    t0 = time.time()
    while len(lines) < 40 and time.time() - t0 < 60:
        lines.append(p.stdout.readline().rstrip("\n"))
    p.stdin.write("stop\nquit\n")
    p.stdin.flush()
    p.wait(timeout=30)
    assert p.returncode == 0 and len(lines) == 40
    og = oracle.OracleGroup("env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=8:program_seed=1:nn_file_name=" + pt, oracle.desc_c1(), w)
    og.cycles(17 * 60)
    assert lines == og.lines()[:40]
    for l in lines:
        assert l.startswith("SelfPlay true ") and l.endswith(" #") and "EV[weight_iter_0.pt]" in l
    assert "[command] start" in p.stderr.read()


MULTI = [
    ("tictactoe", "tictactoe", "c1", "actor_num_simulation=16", 17, 30, 400),
    # the per-game simulation kernel + device Go rules from two host threads (per-device LDS attributes, argument blocks, streams)
    ("go", "go", None, "env_board_size=9:actor_num_simulation=8", 9, 3, 400),
]


@pytest.mark.parametrize("name,game,key,extra,cpm,lines_per_dev,moves", MULTI)
def test_one_process_drives_every_visible_gpu(mz, oracle, tmp_path, name, game, key, extra, cpm, lines_per_dev, moves):
    """ref actor_group.cpp:168-187 + scripts/zero-worker.sh:159-162: ONE `-mode sp` process, zero_num_parallel_games = batch x #GPUs, actor i on
    device i % G.  Here: worker g = games {i % G == g} on device g, seed program_seed + g, one host thread per device, one stdout mutex.
    On a one-GPU box the same logic runs with two logical devices mapped onto GPU 0 (MZ_DEVICE_MAP=0,0, include/minizero/actor_group.h)."""
    env = dict(os.environ)
    G = mz.device_count()
    if G < 2:
        G = 2
        env["MZ_DEVICE_MAP"] = "0,0"
    from minizero_amd.export_weights import write_mzw
    exe = os.path.join(ROOT, "apps", "mzgpu_sp")
    if key:
        d, od = mz.DESCS[key](), getattr(oracle, "desc_" + key)()
    else:
        args = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82)
        d, od = mz.make_desc(*args, vh=16, dv=1, type_name="alphazero"), oracle.make_desc(*args, 16, 1)
    w = mz.generate_weights(d, 0)
    pt = str(tmp_path / "weight_iter_0.pt")
    write_mzw(pt[:-3] + ".mzw", d, w)
    games = 4 * G + 1
    conf_str = f"nn_file_name={pt}:program_seed=5:{extra}:zero_num_parallel_games={games}:zero_num_threads={G}"
    p = subprocess.Popen([exe, "-conf_str", conf_str, "-mode", "sp", "-game", game], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env)
    p.stdin.write("start\n")
    p.stdin.flush()
    expected = {}
    for g in range(G):
        n_g = len(range(g, games, G))
        og = oracle.OracleGroup(f"env_game={game}:{extra}:zero_num_parallel_games={n_g}:program_seed={5 + g}:nn_file_name={pt}:zero_num_threads=1", od, w)
        og.cycles(cpm * moves)
        expected[g] = og.lines()
        assert len(expected[g]) >= lines_per_dev
    # every printed line is the next unseen line of exactly one device's stream (per-device order is kept, devices interleave freely)
    # (the devices run at their own pace: read until each has shown lines_per_dev records, or one of them runs out of expected ones)
    cursor = {g: 0 for g in range(G)}
    while min(cursor.values()) < lines_per_dev and all(cursor[g] < len(expected[g]) for g in range(G)):
        l = p.stdout.readline().rstrip("\n")
        assert l, "the worker stopped printing: " + p.stderr.read()[-2000:]
        owners = [g for g in range(G) if cursor[g] < len(expected[g]) and expected[g][cursor[g]] == l]
        assert owners, "a line that is not the next record of any device: " + l[:120]
        cursor[owners[0]] += 1
    p.stdin.write("quit\n")
    p.stdin.flush()
    try:  # drain: the device threads may be blocked on a full stdout pipe until they see the quit
        _, err = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert f"{games} games on {G} GPU(s)" in err, err[-2000:]
    assert min(cursor.values()) >= lines_per_dev, f"per-device records seen: {cursor}"


EIGHT = [
    # BASELINE configs[0] on eight logical devices
    ("tictactoe", "tictactoe", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"), "actor_num_simulation=16", 8 * 5 + 3, 17, 12, 200, 1),
    # BASELINE configs[4]'s node: 512 games = 64 per device (the Gumbel rounds' batched pipeline, pairs refused: eight workers share the GPU), on the small
    # muzero_atari test network so that eight CPU oracles finish in a minute; short episodes / sequences so that records leave early
    ("atari", "atari", ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari"),
     "nn_type_name=muzero:actor_num_simulation=50:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:"
     "actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=3:"
     "learner_n_step_return=1:learner_muzero_unrolling_step=1:env_atari_episode_length=5", 512, 51, 8, 24, 1),  # (the devices start one after the other: ample expected lines)
    # ... and with two RNG streams per worker: device g's generators are program_seed + 2 g, + 2 g + 1 (actor_group.cpp:66-70 across a node)
    ("tictactoe_two_streams", "tictactoe", ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"), "actor_num_simulation=16:mz_rng_streams=2", 8 * 6, 17, 10, 200, 2),
]


@pytest.mark.parametrize("name,game,args,extra,games,cpm,lines_per_dev,moves,S", EIGHT, ids=[e[0] for e in EIGHT])
def test_one_process_eight_logical_devices(mz, oracle, tmp_path, name, game, args, extra, games, cpm, lines_per_dev, moves, S):
    """The G = 8 rehearsal (ref actor_group.cpp:24-50,168-187; scripts/zero-worker.sh:159-162: ONE `-mode sp` process for the node's eight GPUs): eight logical
    devices on GPU 0 (MZ_DEVICE_MAP=0,0,0,0,0,0,0,0), eight host threads driving eight workers, one stdout.  Every printed line must be the next record of exactly
    one device's stream, device g = games {i % 8 == g} seeded program_seed + g * S (S RNG streams per worker) — eight OracleGroups say what those are."""
    from minizero_amd.export_weights import write_mzw
    env = dict(os.environ)
    G = 8
    env["MZ_DEVICE_MAP"] = ",".join(["0"] * G)
    exe = os.path.join(ROOT, "apps", "mzgpu_sp")
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    d, od = mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)
    w = mz.generate_weights(d, 0)
    pt = str(tmp_path / "weight_iter_0.pt")
    write_mzw(pt[:-3] + ".mzw", d, w)
    conf_str = f"nn_file_name={pt}:program_seed=5:{extra}:zero_num_parallel_games={games}:zero_num_threads={G}"
    p = subprocess.Popen([exe, "-conf_str", conf_str, "-mode", "sp", "-game", game], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    p.stdin.write("start\n")
    p.stdin.flush()
    oextra = extra.replace(":mz_rng_streams=2", "")
    expected = {}
    for g in range(G):
        n_g = len(range(g, games, G))
        og = oracle.OracleGroup(f"env_game={game}:{oextra}:zero_num_parallel_games={n_g}:program_seed={5 + g * S}:nn_file_name={pt}:zero_num_threads=1" +
                                (f":oracle_throughput_threads={S}" if S > 1 else ""), od, w)
        og.cycles(cpm * moves)
        expected[g] = og.lines()
        assert len(expected[g]) >= lines_per_dev, f"device {g}: the oracle finished {len(expected[g])} records"
    cursor = {g: 0 for g in range(G)}
    seen = 0
    while min(cursor.values()) < lines_per_dev and all(cursor[g] < len(expected[g]) for g in range(G)):
        l = p.stdout.readline().rstrip("\n")
        assert l, "the worker stopped printing: " + p.stderr.read()[-2000:]
        assert l.startswith("SelfPlay ") and l.endswith(" #") and l.count("SelfPlay ") == 1, "a torn line under eight writer threads: " + l[:200]
        owners = [g for g in range(G) if cursor[g] < len(expected[g]) and expected[g][cursor[g]] == l]
        assert owners, "a line that is not the next record of any device: " + l[:160]
        cursor[owners[0]] += 1
        seen += 1
    p.stdin.write("quit\n")
    p.stdin.flush()
    try:
        _, err = p.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert f"{games} games on {G} GPU(s)" in err, err[-2000:]
    assert min(cursor.values()) >= lines_per_dev, f"per-device records seen: {cursor}"
